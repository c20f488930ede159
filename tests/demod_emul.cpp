// demod_emul.cpp -- TEST INFRASTRUCTURE: csrc/demod.hip (Gardner timing + PLL a lane per frame, slicer / unique word /
// DQPSK / LLR a wavefront per frame, which also writes the compact records) on the CPU emulation of tests/hip_emul/hip/hip_runtime.h.
// atan2f / sincosf / sqrtf resolve to the host libm here (on the device they are the device library's): what this checks
// is the kernels' logic and arithmetic order against the reference's own stage-C vectors (tests/golden/ref_stage_c.npz,
// produced by the reference's qpsk_demod.c), which were computed with the same host libm.
#include <hip/hip_runtime.h>
#include <vector>

#include "demod_emul.inc"

using namespace irdm;

extern "C" {

// samples: [n][2 * kMaxFrameSamples] floats (re, im); out: n DemodOut, packed: n DemodPacked
int demod_emul_run(const float *samples, const int *num_samples, const int *direction, int n, int use_gardner, float sps,
                   DemodOut *out, DemodPacked *packed)
{
    std::vector<BurstWork> work(n);
    memset(work.data(), 0, sizeof(BurstWork) * n);
    for (int i = 0; i < n; i++) {
        work[i].num_samples = num_samples[i];
        work[i].direction = direction[i];
        work[i].drop_reason = 0;
    }
    std::vector<float2> ws((size_t)n * 2 * kMaxSymbols);
    memset(out, 0, sizeof(DemodOut) * n);
    if (launch_demod(work.data(), n, reinterpret_cast<const float2 *>(samples), use_gardner, sps, ws.data(), out, nullptr) != 0)
        return -1;
    if (packed) {
        // the compact records as the chain's last kernel writes them (demod_par_kernel's export: what a packed_records context
        // polls), from a second run into a scratch DemodOut
        std::vector<DemodOut> out2(n);
        std::vector<BurstWork> work_back(n);
        memset(out2.data(), 0, sizeof(DemodOut) * n);
        if (launch_demod(work.data(), n, reinterpret_cast<const float2 *>(samples), use_gardner, sps, ws.data(), out2.data(), nullptr,
                         packed, work_back.data()) != 0)
            return -1;
        if (memcmp(work_back.data(), work.data(), sizeof(BurstWork) * n) != 0) return -2;      // (the work records travel with them)
    }
    return 0;
}

int demod_emul_sizes(int *out_bytes, int *packed_bytes, int *max_frame_samples, int *max_bits)
{
    *out_bytes = (int)sizeof(DemodOut);
    *packed_bytes = (int)sizeof(DemodPacked);
    *max_frame_samples = kMaxFrameSamples;
    *max_bits = kMaxBits;
    return 0;
}

}
