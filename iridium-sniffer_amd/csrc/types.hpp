// types.hpp -- device-resident state and work descriptors of the pipeline.
#pragma once
#include <stdint.h>

namespace irdm {

constexpr int kHistory = 512;            // iridium.h:46
constexpr int kMaxActive = 1024;         // max_bursts (<= 240) + bursts one frame can add before the squelch check
constexpr int kScanThreads = 1024;
constexpr int kFirTaps = 801;            // lpf_taps(.., 1e7, 1e5, 5e4), burst_downmix.c:251-261
constexpr int kRotSeg = 16;              // rotator checkpoint spacing (samples)
constexpr int kRotRun = 2048;            // checkpoints per block of the checkpoint arena: a centre bin's row is a list of
                                         // blocks, one per run of 2048 checkpoints (32768 samples) bursts on it have needed
constexpr int kFirTileOut = 128;         // decimator outputs per workgroup
constexpr int kMaxFrameSamples = 4440;   // IR_MAX_FRAME_LENGTH_SIMPLEX * 10
constexpr int kMaxBits = 896;
constexpr int kMaxSymbols = 448;         // 4440 samples / 10 sps + slack; matches kMaxBits / 2
constexpr int kFrameNeed = 5696;         // decimated samples after `start` that can reach the output frame
constexpr int kCfoN = 256, kCfoTotal = 4096, kCorrN = 2048, kSyncSearch = 840;

// active_burst_t (burst_detect.c:39-47) minus the dB fields, which the host
// derives with its own libm from peak_rel / base_sum (burst_detect.c:572, :583-586)
struct ActiveBurst {
    uint64_t id, start, last_active;
    int32_t center_bin;
    float peak_rel, base_sum;
    int32_t pad;
};

struct GoneBurst {
    uint64_t id, start, stop, last_active;
    int32_t center_bin;
    float peak_rel, base_sum;
    int32_t pad;
};

// detector parameters derived at create (burst_detect.c:180-226)
struct DetParams {
    int n, log_n, pre_len, post_len, width, max_bursts, max_len;
    float threshold;
};

// detector state carried between chunks (and, for multi-GPU time-chunk sharding,
// handed to the next rank together with baseline sum + history)
struct DetState {
    uint64_t index;          // absolute sample index of the next frame
    uint64_t burst_id;
    int32_t hist_idx, primed, squelch, n_act;
    uint32_t n_gone;         // bursts emitted by the current chunk
    uint32_t overflow;       // 1 if a fixed capacity was exceeded (results invalid)
    ActiveBurst act[kMaxActive];
};

#ifdef __HIPCC__
// The finished-burst records of a detector scan and its four header words (n_gone, overflow, hist_idx, primed), written
// by the GPU straight into pinned host memory behind the scan: the host needs ONE stream synchronise to have them.
// (The hipMemcpyAsync D2H round trips this replaces cost 0.43 ms per chunk for 32 KB next to the per-burst chains'
// kernels, 0.84 ms with three chains in flight -- more than the scan's own wait.)  Called by the first `n_blocks`
// workgroups of a kernel; `ctl` (optional): the band scan's control block.
__device__ __forceinline__ void gone_export_body(const DetState *__restrict__ st, const uint32_t *__restrict__ gone, int cap,
                                                 uint32_t *__restrict__ hp_gone, uint32_t *__restrict__ hp_hdr,
                                                 const uint32_t *__restrict__ ctl, uint32_t *__restrict__ hp_ctl, int ctl_words,
                                                 int n_blocks)
{
    if (blockIdx.x == 0 && ctl && (int)threadIdx.x < ctl_words)
        __hip_atomic_store(hp_ctl + threadIdx.x, ctl[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t n = st->n_gone;
    const size_t words = (size_t)(n < (uint32_t)cap ? n : (uint32_t)cap) * (sizeof(GoneBurst) / 4);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words; i += (size_t)n_blocks * blockDim.x)
        __hip_atomic_store(hp_gone + i, gone[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        __hip_atomic_store(hp_hdr + 0, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(hp_hdr + 1, st->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(hp_hdr + 2, (uint32_t)st->hist_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(hp_hdr + 3, (uint32_t)st->primed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
}
#endif

struct PeakCand {
    float rel;
    int32_t bin;
};

// prefilter list entry (scan_fast.hip): a bin whose |X|^2 may exceed the threshold
struct ListEntry {
    int32_t bin;
    float mag;
};
constexpr int kListCap = 1024;           // entries per frame of the sparse leader scan; more -> dense scan fallback
constexpr int kBandListCap = 4096;       // entries per frame of the band scan (at most fft_size); more -> sequential scan

// one decimator tile = kFirTileOut outputs of one burst
struct FirTile {
    int32_t burst;           // index into BurstWork[]
    int32_t first_out;       // first output index of the tile
};

// what the persistent decimator needs to know about one tile, written by fir_geom_kernel (downmix.hip) so that the
// decimator's workgroups read ONE record per tile instead of chasing tile -> burst -> rotator-table pointers
struct FirGeom {
    uint64_t a_tile;         // absolute index of the tile's first sample
    uint64_t avail_end;
    uint64_t ring_pos;       // a_tile % ring_len
    uint64_t burst_start;
    uint64_t ck_index;       // arena index of the rotator checkpoint of the tile's first segment
    float inc_re, inc_im;    // rotator increment of the burst's centre bin
    int32_t s0, span, n_seg, n_out;
    int64_t out_base;        // index of the tile's first output in dec[]
    uint64_t stale_pos;      // (a_tile - reference ring length) mod ring_len: where the tile's stale samples start
    uint64_t ck_index2;      // arena index of the next run's first checkpoint (a tile crosses at most one run boundary)
    int32_t ck_wrap;         // checkpoints of the tile in its first run: segment `seg` is ck_index + seg below it,
    int32_t pad;             // ck_index2 + (seg - ck_wrap) from there on (fir_ck)
};
static_assert(sizeof(FirGeom) == 96, "FirGeom is read with scalar loads");
#if defined(__HIPCC__) || defined(__host__)
__host__ __device__
#endif
inline uint64_t fir_ck(const FirGeom &g, int seg)
{
    return seg < g.ck_wrap ? g.ck_index + (uint64_t)seg : g.ck_index2 + (uint64_t)(seg - g.ck_wrap);
}
// device tile lists are allocated with room for the FirGeom records behind the FirTile array
constexpr size_t kFirTileUnits = 1 + sizeof(FirGeom) / sizeof(FirTile);

struct BurstWork {
    uint64_t start;          // absolute index of sample 0 of the burst window
    uint64_t avail_end;      // samples at/after this index read the stale ring slot (burst_detect.c:401-422)
    int32_t n;               // samples used (min(num_samples, 2 Mi)), burst_downmix.c:650-651
    int32_t dec_len;         // (n - 801 + 1) / M
    int32_t center_bin;
    int32_t simplex;         // center frequency after CFO > 1626 MHz (burst_downmix.c:764)
    // filled by the device
    int32_t start_idx;       // find_burst_start
    float center_offset;     // estimate_fine_cfo
    float incr_re, incr_im;  // cexpf(-2 pi center_offset i), host libm
    int32_t direction, uw_start, num_samples, drop_reason;
    float uw_corr, corr_re, corr_im;
    int32_t tile_base;       // index of the burst's first decimator tile (host; the persistent decimator's geometry pass)
    int32_t dec_off;         // the burst's row in the decimated / low-passed scratch (float2 units; host: rows by actual length)
    uint32_t box_max;        // bits of the start filter's largest output (post_tiles_kernel: atomic max over the burst's tiles; >= 0)
};
static_assert(sizeof(BurstWork) == 88, "BurstWork is mirrored word by word between device and pinned host memory");

struct DemodOut {
    int32_t ok, direction, confidence, n_symbols;
    float level, total_phase;
    uint8_t bits[kMaxBits];
    float llr[kMaxBits];
};

// the demodulator's result without the soft outputs, hard bits 8 per byte, MSB first (frame_output.c:168-197 needs no
// more of a frame): what the chain brings back in the packed_records mode, 136 bytes per burst instead of 4.5 KB
struct DemodPacked {
    int32_t ok, direction, confidence, n_symbols;
    float level, total_phase;
    uint8_t bits[kMaxBits / 8];
};
static_assert(sizeof(DemodPacked) == 136 && sizeof(DemodPacked) % 8 == 0, "copied 16 bytes at a time in pairs");

// frame_decode() result per demodulated frame (bitlayer.hip); lat / lon / alt are completed on the host
struct DecodedOut {
    int32_t type, sat_id, beam_id, pos_xyz[3], n_pages;
    uint32_t page_tmsi[12];
    int32_t page_msc[12];
    int32_t timeslot, sv_blocking, bc_type;
    uint32_t iri_time;
    int32_t bch_len;
};

// ida_decode() result per demodulated frame (bitlayer.hip); the LCW header text is formatted on the host
struct IdaOut {
    int32_t ok, ft, lcw_ft, lcw_code, ec_lcw;
    uint32_t lcw3_val;
    int32_t da_ctr, da_len, cont, crc_ok;
    uint32_t stored_crc, computed_crc;
    int32_t fixederrs, payload_len, bch_len;
    uint8_t payload[32];
    uint8_t bch_stream[256];
};

}  // namespace irdm
