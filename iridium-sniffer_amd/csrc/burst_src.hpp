// burst_src.hpp -- where a burst window's samples come from (shared by the stage-B kernels).
#pragma once
#include "common.hpp"
#include "types.hpp"
#include "kernels.hpp"

namespace irdm {

// ---- burst window sample access (ringbuf_extract, burst_detect.c:401-422) ----
// Samples at absolute index >= avail_end had not been written when the reference
// extracted the burst: it read whatever the ring slot held, i.e. the sample one
// ring length earlier (or the zero page before the ring first wrapped).
__device__ __forceinline__ float2 load_abs(const SampleSource &src, uint64_t a)
{
    if (a >= src.chunk_start) return load_iq(src.fmt, src.chunk, (size_t)(a - src.chunk_start));
    return load_iq(src.fmt, src.ring, (size_t)(a % src.ring_len));
}

__device__ __forceinline__ float2 burst_sample(const SampleSource &src, uint64_t start,
                                               uint64_t avail_end, int k)
{
    uint64_t a = start + (uint64_t)k;
    if (a >= avail_end) {
        if (a < src.ref_ring) return make_float2(0.0f, 0.0f);
        a -= src.ref_ring;
    }
    return load_abs(src, a);
}

}  // namespace irdm
