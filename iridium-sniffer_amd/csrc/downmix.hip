// downmix.hip -- stage B on gfx950 (burst_downmix.c:643-797), batched over bursts.
//
//   rotator checkpoint table   rotator.h:36-46 recurrence, one lane per FFT bin (create time)
//   fir_decimate_kernel        step 1+2: coarse rotate fused into the 801-tap /M decimator
//   downmix_post1_kernel       step 2b noise LPF, step 3 find_burst_start, step 4 fine CFO (FFT 4096)
//   downmix_post2_kernel       step 5 fine rotate, step 6 RRC, step 7 sync correlation
//                              (FFT 2048 + 2 x IFFT 2048), step 8 phase align, step 9 frame cut
#include <algorithm>
#include <cstdio>
#include "common.hpp"
#include "types.hpp"
#include "kernels.hpp"
#include "burst_src.hpp"
#include "libm_port.hpp"

namespace irdm {

// 16 consecutive samples (one rotator segment, 16-sample aligned) with the widest loads the format allows, converted
// exactly as load_iq does
__device__ __forceinline__ void load_seg16(int fmt, const void *__restrict__ base, size_t idx, float2 (&x)[kRotSeg])
{
    if (fmt == 2) {
        const float4 *g = reinterpret_cast<const float4 *>(reinterpret_cast<const float2 *>(base) + idx);
#pragma unroll
        for (int u = 0; u < kRotSeg / 2; u++) {
            const float4 v = g[u];
            x[2 * u] = make_float2(v.x, v.y);
            x[2 * u + 1] = make_float2(v.z, v.w);
        }
    } else if (fmt == 1) {
        const int4 *g = reinterpret_cast<const int4 *>(reinterpret_cast<const short2 *>(base) + idx);   // 4 samples per 16 B
#pragma unroll
        for (int u = 0; u < kRotSeg / 4; u++) {
            const int4 v = g[u];
            const int w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const short re = (short)(w[k] & 0xffff), im = (short)(w[k] >> 16);
                x[4 * u + k] = make_float2((float)(re >> 8) / 128.0f, (float)(im >> 8) / 128.0f);
            }
        }
    } else {
        const int4 *g = reinterpret_cast<const int4 *>(reinterpret_cast<const char2 *>(base) + idx);    // 8 samples per 16 B
#pragma unroll
        for (int u = 0; u < kRotSeg / 8; u++) {
            const int4 v = g[u];
            const int w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const signed char r0 = (signed char)(w[k] & 0xff), i0 = (signed char)((w[k] >> 8) & 0xff);
                const signed char r1 = (signed char)((w[k] >> 16) & 0xff), i1 = (signed char)((w[k] >> 24) & 0xff);
                x[8 * u + 2 * k] = make_float2((float)r0 / 128.0f, (float)i0 / 128.0f);
                x[8 * u + 2 * k + 1] = make_float2((float)r1 / 128.0f, (float)i1 / 128.0f);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Rotator checkpoints: phase_k of the float recurrence phase *= incr
// (rotator.h:38-39) for k = 0, 16, 32, ...  The sequence depends only on the
// burst's centre bin (rotator_init per burst, burst_downmix.c:666-669), so it
// is computed once per bin -- as far as bursts have needed it -- and kept in HBM.
// ---------------------------------------------------------------------------
// The rows ON DEMAND, block by block.  The table is an ARENA of blocks of kRotRun checkpoints; a centre bin's row is the list
// of its blocks, one per run of kRotRun checkpoints, runs[bin * n_runs + run] (-1: not built) -- a row exists as far as the
// bursts on its bin have needed it so far (pipeline.cpp, rot_rows_prepare) and grows by blocks that need not be adjacent:
// nothing ever moves, chains in flight keep reading the blocks they were launched for.  news[i] = (bin, from, to, block):
// checkpoints from .. to - 1 (whole runs) into the blocks block, block + 1, ..; continued from checkpoint from - 1.  The lane
// that has written a run publishes its block (read by the decimator's geometry pass and the one-tile-per-workgroup
// decimators of the SAME stream, launched behind this kernel; later chains wait for the event recorded behind it).
__global__ void rotator_rows_kernel(const float2 *__restrict__ incr, float2 *__restrict__ table, int n_runs,
                                    const int4 *__restrict__ news, int n_new, int *__restrict__ runs_all)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_new) return;
    const int bin = news[i].x, from = news[i].y, to = news[i].z, block0 = news[i].w;
    const float2 inc = incr[bin];
    int *runs = runs_all + (size_t)bin * n_runs;
    float2 ph = make_float2(1.0f, 0.0f);
    if (from > 0) {
        ph = table[(size_t)runs[(from - 1) / kRotRun] * kRotRun + (from - 1) % kRotRun];
#pragma unroll
        for (int u = 0; u < kRotSeg; u++) ph = cmul(ph, inc);
    }
    for (int c0 = from; c0 < to; c0 += kRotRun) {
        const int block = block0 + (c0 - from) / kRotRun;
        float2 *row = table + (size_t)block * kRotRun;
        for (int c = 0; c < kRotRun; c++) {
            row[c] = ph;
#pragma unroll
            for (int u = 0; u < kRotSeg; u++) ph = cmul(ph, inc);
        }
        runs[c0 / kRotRun] = block;
    }
}

int launch_rotator_rows(const float2 *incr, float2 *table, int n_runs, const int4 *news, int n_new, int *runs, hipStream_t stream)
{
    if (n_new <= 0) return 0;
    hipLaunchKernelGGL(rotator_rows_kernel, dim3((n_new + 63) / 64), dim3(64), 0, stream, incr, table, n_runs, news, n_new, runs);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// the checkpoints of a tile as the one-tile-per-workgroup kernels read them: `runs` = blocks of the bin's row
// (rot_slot[bin * n_runs + run]), first checkpoint c0; a tile's checkpoints lie in one run or two
struct RotCk {
    const float2 *p0, *p1;
    int wrap;
    __device__ __forceinline__ RotCk(const float2 *table, const int *rot_slot, int n_runs, int bin, int c0)
    {
        const int r0 = c0 / kRotRun;
        const int *runs = rot_slot + (size_t)bin * n_runs;
        const int b0 = runs[r0];
        int b1 = r0 + 1 < n_runs ? runs[r0 + 1] : b0;
        if (b1 < 0) b1 = b0;
        p0 = table + (size_t)b0 * kRotRun + c0 % kRotRun;
        p1 = table + (size_t)b1 * kRotRun;
        wrap = kRotRun - c0 % kRotRun;
    }
    __device__ __forceinline__ float2 at(int seg) const { return seg < wrap ? p0[seg] : p1[seg - wrap]; }
};

// ---------------------------------------------------------------------------
// Rotate + decimate.  One workgroup = kFirTileOut outputs of one burst.
//   staging : samples [o0*M, o0*M + (nout-1)*M + 801) of the burst window are read
//             once (coalesced by 16-sample segments), rotated exactly as
//             rotator_rotate_n would (phase restored from the checkpoint table,
//             continued by the float recurrence), and written to LDS in polyphase
//             order  lds[(s % M) * ROW + s / M]  so that the tap loop reads
//             consecutive addresses across lanes (no bank conflicts).
//   taps    : out[i] = sum_{k<801} t[k] * x[i*M + k], k ascending, real x complex =
//             two independent mul+add chains (simd_generic.c:86-96).
// Algorithmic HBM bytes: 8 B (cf32) per burst-window sample in, 8 B per output.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kFirTileOut) void fir_decimate_kernel(
    SampleSource src, const BurstWork *__restrict__ work, const FirTile *__restrict__ tiles,
    int decim, int row, const float *__restrict__ taps, const int *__restrict__ tap_off,
    const float2 *__restrict__ rot_incr,
    const float2 *__restrict__ rot_table, int n_ckpt, float2 *__restrict__ dec, int order,
    const int *__restrict__ rot_slot)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    const int tid = threadIdx.x;
    const FirTile tile = tiles[blockIdx.x];
    const BurstWork w = work[tile.burst];
    const int o0 = tile.first_out;
    int n_out = w.dec_len - o0;
    if (n_out > kFirTileOut) n_out = kFirTileOut;
    const int span = (n_out - 1) * decim + kFirTaps;
    const int s0 = o0 * decim;                       // multiple of kRotSeg
    const float2 inc = rot_incr[w.center_bin];
    const RotCk ck(rot_table, rot_slot, n_ckpt, w.center_bin, s0 / kRotSeg);
    const int n_seg = (span + kRotSeg - 1) / kRotSeg;

    for (int seg = tid; seg < n_seg; seg += kFirTileOut) {
        float2 ph = ck.at(seg);
        const int k0 = seg * kRotSeg;
        int p = k0 % decim, q = k0 / decim;
        const uint64_t a0 = w.start + (uint64_t)(s0 + k0);
        float2 x[kRotSeg];
        if (src.fmt == 2 && a0 >= src.chunk_start && a0 + kRotSeg <= w.avail_end && k0 + kRotSeg <= span) {
            // fast path: the whole segment lies in the chunk being fed -> 8 x 16-byte loads in flight
            const float4 *g = reinterpret_cast<const float4 *>(
                reinterpret_cast<const float2 *>(src.chunk) + (a0 - src.chunk_start));
#pragma unroll
            for (int u = 0; u < kRotSeg / 2; u++) {
                const float4 v = g[u];
                x[2 * u] = make_float2(v.x, v.y);
                x[2 * u + 1] = make_float2(v.z, v.w);
            }
        } else if (src.fmt == 2 && a0 + kRotSeg <= src.chunk_start && a0 + kRotSeg <= w.avail_end &&
                   k0 + kRotSeg <= span) {
            // same from the history ring: ring_len is a multiple of 16 and a0 is too, so no wrap inside
            const float4 *g = reinterpret_cast<const float4 *>(
                reinterpret_cast<const float2 *>(src.ring) + (a0 % src.ring_len));
#pragma unroll
            for (int u = 0; u < kRotSeg / 2; u++) {
                const float4 v = g[u];
                x[2 * u] = make_float2(v.x, v.y);
                x[2 * u + 1] = make_float2(v.z, v.w);
            }
        } else {
#pragma unroll
            for (int u = 0; u < kRotSeg; u++)
                x[u] = (k0 + u < span) ? burst_sample(src, w.start, w.avail_end, s0 + k0 + u)
                                       : make_float2(0.0f, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < kRotSeg; u++) {
            if (k0 + u < span) {
                s[p * row + q] = cmul(x[u], ph);     // out[i] = in[i] * phase (rotator.h:38)
                ph = cmul(ph, inc);                  // phase *= incr          (rotator.h:39)
            }
            if (++p == decim) { p = 0; q++; }
        }
    }
    __syncthreads();

    if (tid < n_out && order == 1) {
        // the order of the reference's AVX2 kernel (simd_avx2.c:62-108, option fir_order 1; see fir_reg.hip,
        // fir_decimate_kernel_f): tap k into accumulator k % 4 with a fused multiply-add, horizontal sum, then the one
        // tap the vector loop leaves over with a separately rounded product and sum
        float ar[4] = { 0.0f, 0.0f, 0.0f, 0.0f }, ai[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
        const unsigned char *col = reinterpret_cast<const unsigned char *>(s + tid);
        static_assert((kFirTaps - 1) % 4 == 0, "the vector loop takes all taps but the last");
        for (int k0 = 0; k0 < kFirTaps - 1; k0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float2 v = *reinterpret_cast<const float2 *>(col + tap_off[k0 + u]);
                const float t = taps[k0 + u];
                ar[u] = __builtin_fmaf(t, v.x, ar[u]);
                ai[u] = __builtin_fmaf(t, v.y, ai[u]);
            }
        }
        float sr = (ar[0] + ar[2]) + (ar[1] + ar[3]), si = (ai[0] + ai[2]) + (ai[1] + ai[3]);
        {
            const float2 v = *reinterpret_cast<const float2 *>(col + tap_off[kFirTaps - 1]);
            const float t = taps[kFirTaps - 1];
            sr += t * v.x;
            si += t * v.y;
        }
        dec[(size_t)w.dec_off + o0 + tid] = make_float2(sr, si);
    } else if (tid < n_out) {
        // 801 taps, k ascending, two independent mul+add chains (simd_generic.c:86-96).  Eight LDS reads
        // are issued ahead of the eight dependent accumulations; (p, q) walk the polyphase tile in
        // scalar registers.
        float ar = 0.0f, ai = 0.0f;
        const unsigned char *col = reinterpret_cast<const unsigned char *>(s + tid);
        constexpr int U = 8;
        static_assert(kFirTaps % U == 1, "tail handles exactly one tap");
        // tap_off[k] = byte offset of polyphase slot (k % M, k / M): wavefront-uniform, fetched with
        // scalar loads together with the taps
        for (int k0 = 0; k0 < kFirTaps - 1; k0 += U) {
            float2 v[U];
            float t[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                v[u] = *reinterpret_cast<const float2 *>(col + tap_off[k0 + u]);
                t[u] = taps[k0 + u];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                ar += t[u] * v[u].x;
                ai += t[u] * v[u].y;
            }
        }
        {
            const float2 v = *reinterpret_cast<const float2 *>(col + tap_off[kFirTaps - 1]);
            const float t = taps[kFirTaps - 1];
            ar += t * v.x;
            ai += t * v.y;
        }
        dec[(size_t)w.dec_off + o0 + tid] = make_float2(ar, ai);
    }
}

// Same kernel with the decimation factor as a compile-time constant (M = 8 / 16 / 40 / 48 at 2 / 4 / 10 / 12 MHz):
//   * polyphase slot of tap k = (k % M, k / M) -> the LDS byte offset of tap r*M + p is  r*8 + p*ROW*8, the second
//     term an instruction immediate (< 64 KB), so the tap loop has no address arithmetic and no offset table;
//   * the 801 taps are staged into LDS once and read four at a time as one broadcast ds_read_b128: no scalar loads in
//     the loop (SMEM returns out of order and shares lgkmcnt with LDS, i.e. every tap fetch forced a full
//     lgkmcnt(0) drain of the sample reads in flight);
//   * one unrolled row of M taps per iteration: M sample reads in flight ahead of the dependent mul/add (hipcc packs
//     the re/im chains into v_pk_mul_f32 + v_pk_add_f32: two VALU instructions per tap, still one rounding each).
// Accumulation order is unchanged (k ascending, two independent chains).
// Measured (10 MHz, 167 M window samples per chunk): 0.91 ms vs 1.04 ms for the runtime-M kernel; staging alone (loads,
// rotation, polyphase scatter) is 0.47 ms of it, the tap loop 0.43 ms -- both instruction-issue bound on the same
// SIMDs (PMC: 48 % of wave cycles issuing, 45 % waiting with 1.5 waves per SIMD), so they add rather than overlap.
// Taps through scalar loads instead of LDS, or all staging loads issued up front, measured the same or worse.
constexpr int fir_tile_row_c(int decim)
{
    // odd row length (staging writes of neighbouring lanes land in different banks), no further padding: at M = 40
    // tile + taps = 51.5 KB, so three workgroups share a CU's 160 KB (161-sample rows would leave room for two)
    return (kFirTileOut + kFirTaps / decim + 2) | 1;
}

template <int M>
__global__ __launch_bounds__(kFirTileOut) void fir_decimate_kernel_m(
    SampleSource src, const BurstWork *__restrict__ work, const FirTile *__restrict__ tiles,
    const float *__restrict__ taps, const float2 *__restrict__ rot_incr,
    const float2 *__restrict__ rot_table, int n_ckpt, float2 *__restrict__ dec,
    const int *__restrict__ rot_slot)
{
    constexpr int ROW = fir_tile_row_c(M);
    constexpr int NR = kFirTaps / M;               // full rows of M taps
    constexpr int REM = kFirTaps - NR * M;         // taps of the last, partial row
    static_assert(M % 4 == 0, "taps are fetched four at a time");
    static_assert((size_t)(M - 1) * ROW * 8 + 8 < 65536, "polyphase offsets must fit the DS immediate");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    float *s_taps = reinterpret_cast<float *>(s + (size_t)ROW * M);      // 16-byte aligned: ROW*M*8 is a multiple of 16
    const int tid = threadIdx.x;
    const FirTile tile = tiles[blockIdx.x];
    const BurstWork w = work[tile.burst];
    const int o0 = tile.first_out;
    int n_out = w.dec_len - o0;
    if (n_out > kFirTileOut) n_out = kFirTileOut;
    const int span = (n_out - 1) * M + kFirTaps;
    const int s0 = o0 * M;                           // multiple of kRotSeg
    const float2 inc = rot_incr[w.center_bin];
    const RotCk ck(rot_table, rot_slot, n_ckpt, w.center_bin, s0 / kRotSeg);
    const int n_seg = (span + kRotSeg - 1) / kRotSeg;

    for (int i = tid; i < kFirTaps + 3; i += kFirTileOut) s_taps[i] = i < kFirTaps ? taps[i] : 0.0f;
    for (int seg = tid; seg < n_seg; seg += kFirTileOut) {
        float2 ph = ck.at(seg);
        const int k0 = seg * kRotSeg;
        int p = k0 % M, q = k0 / M;
        const uint64_t a0 = w.start + (uint64_t)(s0 + k0);
        float2 x[kRotSeg];
        // (a segment that runs past the tile's span is loaded whole as long as its samples exist: the slots past the
        // span are never read by a tap, and the tile has room for them)
        const bool whole = a0 + kRotSeg <= w.avail_end;
        if (whole && a0 >= src.chunk_start) {
            // the whole segment lies in the chunk being fed: 16-byte loads, all in flight (cf32 8, ci16 4, ci8 2 of them)
            load_seg16(src.fmt, src.chunk, (size_t)(a0 - src.chunk_start), x);
        } else if (whole && a0 + kRotSeg <= src.chunk_start) {
            // same from the history ring: ring_len is a multiple of 16 and a0 is too, so no wrap inside
            load_seg16(src.fmt, src.ring, (size_t)(a0 % src.ring_len), x);
        } else {
#pragma unroll
            for (int u = 0; u < kRotSeg; u++)
                x[u] = (k0 + u < span) ? burst_sample(src, w.start, w.avail_end, s0 + k0 + u)
                                       : make_float2(0.0f, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < kRotSeg; u++) {
            if (k0 + u < span) {
                s[p * ROW + q] = cmul(x[u], ph);     // out[i] = in[i] * phase (rotator.h:38)
                ph = cmul(ph, inc);                  // phase *= incr          (rotator.h:39)
            }
            if (++p == M) { p = 0; q++; }
        }
    }
    __syncthreads();

    if (tid < n_out) {
        float ar = 0.0f, ai = 0.0f;
        const float2 *col = s + tid;
#pragma unroll 1
        for (int r = 0; r < NR; r++) {
            const float2 *c = col + r;
            const float4 *t4 = reinterpret_cast<const float4 *>(s_taps + r * M);
            float2 v[M];
            float4 t[M / 4];
#pragma unroll
            for (int p = 0; p < M; p++) v[p] = c[p * ROW];
#pragma unroll
            for (int p = 0; p < M / 4; p++) t[p] = t4[p];
#pragma unroll
            for (int p = 0; p < M / 4; p++) {
                ar += t[p].x * v[4 * p].x;     ai += t[p].x * v[4 * p].y;
                ar += t[p].y * v[4 * p + 1].x; ai += t[p].y * v[4 * p + 1].y;
                ar += t[p].z * v[4 * p + 2].x; ai += t[p].z * v[4 * p + 2].y;
                ar += t[p].w * v[4 * p + 3].x; ai += t[p].w * v[4 * p + 3].y;
            }
        }
#pragma unroll
        for (int p = 0; p < REM; p++) {
            const float2 v = col[p * ROW + NR];
            const float t = s_taps[NR * M + p];
            ar += t * v.x;
            ai += t * v.y;
        }
        dec[(size_t)w.dec_off + o0 + tid] = make_float2(ar, ai);
    }
}

// ---------------------------------------------------------------------------
// The same decimator with the tile stored COLUMN-MAJOR: the M rotated samples that output column q needs for one row of
// taps (k = r*M .. r*M+M-1 -> samples (q+r)*M .. (q+r)*M+M-1) are contiguous, CS = M*8 + 16 bytes per column.
//   * tap loop: one ds_read_b128 brings the samples of TWO taps (M/2 reads per row instead of M ds_read_b64): per tap
//     and wavefront 0.5 + 0.25 LDS instructions + 2 VALU = 2.75 instead of 3.25 -- the loop is bound by instruction
//     issue (DESIGN.md "The decimator");
//   * the 16-byte pad makes the lane stride CS/4 = 2M+4 dwords: the lanes of a 16-lane ds_read_b128 group start in
//     different banks for every M used here (8, 16, 40, 48);
//   * staging: a lane's 16 consecutive rotated samples are contiguous in a column (or split once, at a multiple of 8
//     samples, between two columns): eight ds_write_b128 instead of sixteen ds_write_b64 with per-sample addressing.
// Accumulation order is unchanged (k ascending, two independent chains).
// ---------------------------------------------------------------------------
template <int M>
struct FirCol {
    static constexpr int NR = kFirTaps / M;                  // full rows of M taps
    static constexpr int REM = kFirTaps - NR * M;            // taps of the last, partial row
    static constexpr int COLS = kFirTileOut + NR + 1;        // columns a full tile touches
    static constexpr int CS = M * 8 + 16;                    // column stride, bytes
    static constexpr size_t LDS = (size_t)COLS * CS + sizeof(float) * (kFirTaps + 3);
    static_assert(M % 8 == 0, "a 16-sample segment splits between columns only at a multiple of 8 samples");
};

template <int M>
__global__ __launch_bounds__(kFirTileOut) void fir_decimate_kernel_c(
    SampleSource src, const BurstWork *__restrict__ work, const FirTile *__restrict__ tiles,
    const float *__restrict__ taps, const float2 *__restrict__ rot_incr,
    const float2 *__restrict__ rot_table, int n_ckpt, float2 *__restrict__ dec,
    const int *__restrict__ rot_slot)
{
    using C = FirCol<M>;
    constexpr int NR = C::NR, REM = C::REM, CS = C::CS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char *s = smem_raw;
    float *s_taps = reinterpret_cast<float *>(s + (size_t)C::COLS * CS);     // 16-byte aligned: CS is a multiple of 16
    const int tid = threadIdx.x;
    const FirTile tile = tiles[blockIdx.x];
    const BurstWork w = work[tile.burst];
    const int o0 = tile.first_out;
    int n_out = w.dec_len - o0;
    if (n_out > kFirTileOut) n_out = kFirTileOut;
    const int span = (n_out - 1) * M + kFirTaps;
    const int s0 = o0 * M;                           // multiple of kRotSeg
    const float2 inc = rot_incr[w.center_bin];
    const RotCk ck(rot_table, rot_slot, n_ckpt, w.center_bin, s0 / kRotSeg);
    const int n_seg = (span + kRotSeg - 1) / kRotSeg;

    for (int i = tid; i < kFirTaps + 3; i += kFirTileOut) s_taps[i] = i < kFirTaps ? taps[i] : 0.0f;
    for (int seg = tid; seg < n_seg; seg += kFirTileOut) {
        float2 ph = ck.at(seg);
        const int k0 = seg * kRotSeg;
        const uint64_t a0 = w.start + (uint64_t)(s0 + k0);
        float2 x[kRotSeg];
        // (a segment that runs past the tile's span is loaded whole as long as its samples exist: the slots past the
        // span are never read by a tap, and the tile has room for them)
        const bool whole = a0 + kRotSeg <= w.avail_end;
        if (whole && a0 >= src.chunk_start) {
            load_seg16(src.fmt, src.chunk, (size_t)(a0 - src.chunk_start), x);
        } else if (whole && a0 + kRotSeg <= src.chunk_start) {
            load_seg16(src.fmt, src.ring, (size_t)(a0 % src.ring_len), x);
        } else {
#pragma unroll
            for (int u = 0; u < kRotSeg; u++)
                x[u] = (k0 + u < span) ? burst_sample(src, w.start, w.avail_end, s0 + k0 + u)
                                       : make_float2(0.0f, 0.0f);
        }
        // rotate (rotator.h:38-39); samples past the tile's span are never read by a tap, their slots may hold anything
        float2 y[kRotSeg];
#pragma unroll
        for (int u = 0; u < kRotSeg; u++) {
            y[u] = cmul(x[u], ph);
            ph = cmul(ph, inc);
        }
        // samples k0 .. k0+7 and k0+8 .. k0+15: each half lies inside one column (k0 and M are multiples of 8)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int k = k0 + 8 * h;
            float4 *dst = reinterpret_cast<float4 *>(s + (size_t)(k / M) * CS + (size_t)(k % M) * 8);
#pragma unroll
            for (int u = 0; u < 4; u++)
                dst[u] = make_float4(y[8 * h + 2 * u].x, y[8 * h + 2 * u].y, y[8 * h + 2 * u + 1].x, y[8 * h + 2 * u + 1].y);
        }
    }
    __syncthreads();

    if (tid < n_out) {
        float ar = 0.0f, ai = 0.0f;
        const unsigned char *col = s + (size_t)tid * CS;
#pragma unroll 1
        for (int r = 0; r < NR; r++) {
            const float4 *c4 = reinterpret_cast<const float4 *>(col + (size_t)r * CS);
            const float4 *t4 = reinterpret_cast<const float4 *>(s_taps + r * M);
            float4 v[M / 2];
            float4 t[M / 4];
#pragma unroll
            for (int j = 0; j < M / 2; j++) v[j] = c4[j];
#pragma unroll
            for (int j = 0; j < M / 4; j++) t[j] = t4[j];
#pragma unroll
            for (int j = 0; j < M / 4; j++) {
                ar += t[j].x * v[2 * j].x;     ai += t[j].x * v[2 * j].y;
                ar += t[j].y * v[2 * j].z;     ai += t[j].y * v[2 * j].w;
                ar += t[j].z * v[2 * j + 1].x; ai += t[j].z * v[2 * j + 1].y;
                ar += t[j].w * v[2 * j + 1].z; ai += t[j].w * v[2 * j + 1].w;
            }
        }
        {
            const float2 *c2 = reinterpret_cast<const float2 *>(col + (size_t)NR * CS);
#pragma unroll
            for (int p = 0; p < REM; p++) {
                const float2 v = c2[p];
                const float t = s_taps[NR * M + p];
                ar += t * v.x;
                ai += t * v.y;
            }
        }
        dec[(size_t)w.dec_off + o0 + tid] = make_float2(ar, ai);
    }
}


// ---------------------------------------------------------------------------
// The column-major decimator as a PERSISTENT kernel: the workgroups that fit the chip's LDS walk tiles blockIdx.x,
// blockIdx.x + gridDim.x, ... of TO outputs each.  What that buys over one tile per workgroup (measured there at the
// bench's 667 bursts: staging alone 0.43 ms, tap loop alone 0.51 ms, together 0.80 ms):
//   * the raw samples of the NEXT tile are requested before the tap loop of the current one and wait in registers
//     (3 segments of 16 samples per lane at M = 40: 96 VGPRs), so the HBM latency of staging is covered by the tap loop;
//   * the tap loop is software-pipelined by hand: the LDS reads of step s+1 are issued before the multiplies of step s;
//   * the taps come through the scalar cache into SGPRs instead of as LDS broadcast reads -- a broadcast ds_read_b128
//     still costs its 4 LDS cycles, a third of the loop's LDS time (MI355X_MICROARCH.md, LDS table);
//   * one FirGeom record per tile (fir_geom_kernel) replaces the tile -> burst -> rotator-table pointer chase;
//   * larger tiles shrink the share of the 801-tap halo that is staged twice (21 of TO + 21 columns at M = 40).
// The loop is bound by the multiply-add chain, not by memory: v_pk_mul_f32 / v_pk_add_f32 issue once per ~7.4 cycles per
// SIMD (tools/ubench/valu_chain.hip: no faster with two wavefronts per SIMD, and plain v_mul/v_add at 4 cycles each
// cost the same per complex tap), a dependent v_pk_add_f32 can issue ~23 cycles after its producer, and the reference
// rounds the product before the add (no FMA): 16 SIMD-cycles per complex tap and wavefront = 0.36 ms for the bench's
// 4.4 M outputs, before staging.  Two wavefronts per SIMD (or two outputs per lane) are needed to cover the chain, and
// the tile's LDS footprint allows six per CU.  Forming the product of tap q + 4 ahead of the add of tap q halved the
// tap loop's cycle count (s_memtime) but cost 8 % wall time (register copies at the loop edge, SGPR spills); tiles of
// 256 / 448 outputs measured 0.84 / 0.70 ms against 0.66 ms.
// Values and accumulation order are those of fir_decimate_kernel_c.
// ---------------------------------------------------------------------------
template <int M, int TO>
struct FirW {
    static constexpr int NR = kFirTaps / M;
    static constexpr int REM = kFirTaps - NR * M;
    static constexpr int COLS = TO + NR + 1;
    static constexpr int CS = M * 8 + 16;
    static constexpr size_t LDS = (size_t)COLS * CS + 16;     // + the word the claimed tile index is broadcast through
    static constexpr int SPAN_MAX = (TO - 1) * M + kFirTaps;
    static constexpr int SEG_MAX = (SPAN_MAX + kRotSeg - 1) / kRotSeg;
    static constexpr int SLOTS = (SEG_MAX + TO - 1) / TO;
    // taps per pipeline step: a divisor of M, multiple of 4, at most 24
    static constexpr int T = (M % 24 == 0) ? 24 : (M % 20 == 0) ? 20 : (M % 16 == 0) ? 16 : 8;
    static constexpr int SPR = M / T;                 // steps per row
    static constexpr int NS = NR * SPR;               // pipeline steps
    static_assert(M % 8 == 0 && M % T == 0 && T % 4 == 0 && NS % 2 == 0, "step size");
    static_assert(LDS <= 160 * 1024, "tile does not fit the LDS");
};

// one lane per tile: everything the decimator's workgroups need, in one record
__global__ void fir_geom_kernel(const BurstWork *__restrict__ work, int n_bursts, int n_tiles, int M,
                                int tile_out, uint64_t ring_len, uint64_t ref_ring, const float2 *__restrict__ rot_incr, int n_ckpt,
                                FirGeom *__restrict__ geom, unsigned *__restrict__ next_tile,
                                const int *__restrict__ rot_slot)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) *next_tile = 0;
    if (t >= n_tiles) return;
    // the burst whose tiles include t: the last one with tile_base <= t among those that have tiles (bursts without --
    // dropped before the decimator -- carry the tile_base of the next one, so the LAST burst with tile_base <= t is it)
    int lo = 0, hi = n_bursts - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (work[mid].tile_base <= t) lo = mid; else hi = mid - 1;
    }
    FirTile tile;
    tile.burst = lo;
    tile.first_out = (t - work[lo].tile_base) * tile_out;
    const BurstWork *w = work + tile.burst;
    const int o0 = tile.first_out;
    int n_out = w->dec_len - o0;
    if (n_out > tile_out) n_out = tile_out;
    FirGeom g;
    g.n_out = n_out;
    g.span = (n_out - 1) * M + kFirTaps;
    g.s0 = o0 * M;
    g.n_seg = (g.span + kRotSeg - 1) / kRotSeg;
    g.burst_start = w->start;
    g.a_tile = w->start + (uint64_t)g.s0;
    g.avail_end = w->avail_end;
    g.ring_pos = g.a_tile % ring_len;
    const int cb = w->center_bin;
    const float2 inc = rot_incr[cb];
    g.inc_re = inc.x;
    g.inc_im = inc.y;
    {
        // the bin's row of the checkpoint arena: a block per run of kRotRun checkpoints (rot_slot[cb * n_ckpt + run], n_ckpt =
        // runs per bin here); a tile's checkpoints lie in one run or two
        const int c0 = g.s0 / kRotSeg, r0 = c0 / kRotRun;
        const int *runs = rot_slot + (size_t)cb * n_ckpt;
        const int b0 = runs[r0];
        int b1 = r0 + 1 < n_ckpt ? runs[r0 + 1] : b0;
        if (b1 < 0) b1 = b0;                                  // (not built: not needed by this burst)
        g.ck_index = (uint64_t)b0 * kRotRun + (uint64_t)(c0 % kRotRun);
        g.ck_wrap = kRotRun - c0 % kRotRun;
        g.ck_index2 = (uint64_t)b1 * kRotRun;
    }
    g.out_base = (int64_t)w->dec_off + o0;
    g.stale_pos = (g.ring_pos + ring_len - ref_ring % ring_len) % ring_len;
    g.pad = 0;
    geom[t] = g;
}

__device__ unsigned long long g_fir_prof_dev[8];
#define FIR_PROF_MARK(i)                                                         \
    do {                                                                         \
        if (PROF) {                                                              \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();        \
            prof[i] += now_ - tlast;                                             \
            tlast = now_;                                                        \
        }                                                                        \
    } while (0)

#include "fir_mac.inc"

template <int M, int FMT, int TO, bool PROF = false>
__global__ __launch_bounds__(TO) __attribute__((amdgpu_waves_per_eu(2, 2))) void fir_decimate_kernel_w(
    SampleSource src, const FirGeom *__restrict__ geom, unsigned *__restrict__ next_tile, int n_tiles, int budget,
    const float *__restrict__ taps, const float2 *__restrict__ rot_table, float2 *__restrict__ dec)
{
    using W = FirW<M, TO>;
    constexpr int NR = W::NR, REM = W::REM, CS = W::CS, SLOTS = W::SLOTS, T = W::T, SPR = W::SPR, NS = W::NS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char *s = smem_raw;
    const int tid = threadIdx.x;

    float2 x[SLOTS][kRotSeg];
    float2 ph0[SLOTS];
    // request the raw samples of tile `g`: whole segments inside the chunk or the ring; the others are read in consume()
    auto prefetch = [&](const FirGeom &g) {
#pragma unroll
        for (int j = 0; j < SLOTS; j++) {
            const int seg = tid + TO * j;
            const int k0 = seg * kRotSeg;
            const uint64_t a0 = g.a_tile + (uint64_t)k0;
            const bool whole = seg < g.n_seg && a0 + kRotSeg <= g.avail_end;      // (may run past the span: never read)
            const bool in_chunk = whole && a0 >= src.chunk_start;
            const bool in_ring = whole && a0 + kRotSeg <= src.chunk_start;
            uint64_t rp = g.ring_pos + (uint64_t)k0;
            if (rp >= src.ring_len) rp -= src.ring_len;
            // (segments that are not loaded here read the start of the ring: always mapped, never used)
            const void *base = in_chunk ? src.chunk : src.ring;
            const size_t idx = in_chunk ? (size_t)(a0 - src.chunk_start) : in_ring ? (size_t)rp : 0;
            load_seg16(FMT, base, idx, x[j]);
            ph0[j] = rot_table[fir_ck(g, seg < g.n_seg ? seg : 0)];
        }
    };
    // rotate the samples of tile `g` (rotator.h:38-39) and store them in the tile
    auto consume = [&](const FirGeom &g) {
#pragma unroll
        for (int j = 0; j < SLOTS; j++) {
            const int seg = tid + TO * j;
            if (seg >= g.n_seg) continue;
            const int k0 = seg * kRotSeg;
            const uint64_t a0 = g.a_tile + (uint64_t)k0;
            const bool whole = a0 + kRotSeg <= g.avail_end;
            const bool fast = whole && (a0 >= src.chunk_start || a0 + kRotSeg <= src.chunk_start);
            float2 ph = ph0[j];
            const float2 inc = make_float2(g.inc_re, g.inc_im);
            if (!fast) {
                // a segment across the chunk / ring / availability boundary: sample by sample
#pragma unroll 1
                for (int u = 0; u < kRotSeg; u++) {
                    const int k = k0 + u;
                    const float2 xs = (k < g.span) ? burst_sample(src, g.burst_start, g.avail_end, g.s0 + k)
                                                   : make_float2(0.0f, 0.0f);
                    *reinterpret_cast<float2 *>(s + (size_t)(k / M) * CS + (size_t)(k % M) * 8) = cmul(xs, ph);
                    ph = cmul(ph, inc);
                }
                continue;
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int k = k0 + 8 * h;
                float4 *dst = reinterpret_cast<float4 *>(s + (size_t)(k / M) * CS + (size_t)(k % M) * 8);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const float2 y0 = cmul(x[j][8 * h + 2 * u], ph);
                    ph = cmul(ph, inc);
                    const float2 y1 = cmul(x[j][8 * h + 2 * u + 1], ph);
                    ph = cmul(ph, inc);
                    dst[u] = make_float4(y0.x, y0.y, y1.x, y1.y);
                }
            }
        }
    };

    unsigned long long prof[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    unsigned long long tlast = PROF ? __builtin_amdgcn_s_memtime() : 0;
    // Tiles are claimed dynamically (the workgroups do not run at the same speed: a wavefront alone on its SIMD is
    // faster than two that share one, and the last tile of a burst is short): the first two tiles of a workgroup are
    // blockIdx.x and blockIdx.x + gridDim.x, every further one comes from the counter fir_geom_kernel zeroed.  A
    // workgroup retires after `budget` tiles: its LDS and registers go back to the dispatcher, which hands them to the
    // high-priority detector stream first -- a grid that stayed resident for the whole launch kept the other streams'
    // kernels out of every CU (measured: the band scan of the next chunk took 1.15 ms instead of 0.86 ms).
    unsigned *s_claim = reinterpret_cast<unsigned *>(s + (size_t)W::COLS * CS);
    int t = blockIdx.x;
    int t1 = t + (int)gridDim.x;
    int n_claimed = 2;
    FirGeom g = geom[t];
    prefetch(g);
    FIR_PROF_MARK(0);
    for (;;) {
        const bool more = t1 < n_tiles;
        // the last round requests its own tile again (no branch around the loads: a join would wait for them)
        const FirGeom gn = geom[more ? t1 : t];
        unsigned claimed = 0x7fffffffu;
        if (tid == 0 && n_claimed < budget) claimed = atomicAdd(next_tile, 1u) + 2u * gridDim.x;
        n_claimed++;
        consume(g);
        if (tid == 0) *s_claim = claimed;
        FIR_PROF_MARK(1);
        __syncthreads();
        FIR_PROF_MARK(2);
        prefetch(gn);
        const int t2 = (int)*s_claim;
        FIR_PROF_MARK(3);

        if (tid < g.n_out) {
            const unsigned char *col = s + (size_t)tid * CS;
            // the multiply-add groups are inline asm (fir_mac.inc): taps straight from SGPR pairs, products one tap ahead
            v2f acc = { 0.0f, 0.0f };
            v4f va[T / 2], vb[T / 2];
            uint64_t ta[T / 2], tb[T / 2];
            const uint64_t *taps64 = reinterpret_cast<const uint64_t *>(taps);      // (T is even: pairs never straddle a step)
            auto issue = [&](int st, v4f (&v)[T / 2], uint64_t (&tt)[T / 2]) {
                const v4f *c4 = reinterpret_cast<const v4f *>(col + (size_t)(st / SPR) * CS + (size_t)(st % SPR) * (T * 8));
#pragma unroll
                for (int q = 0; q < T / 2; q++) v[q] = c4[q];
#pragma unroll
                for (int q = 0; q < T / 2; q++) tt[q] = taps64[(st * T) / 2 + q];
            };
            issue(0, va, ta);
#pragma unroll 1
            for (int st = 0; st < NS; st += 2) {
                // the first tap pair of a step comes before the next step's requests: whatever the step waits for was
                // requested a whole step ago (scalar loads return out of order, so the wait is for everything)
                fir_mac<1>(acc, va, ta);
                __builtin_amdgcn_sched_barrier(0);
                issue(st + 1, vb, tb);
                __builtin_amdgcn_sched_barrier(0);
                fir_mac<T / 2 - 1>(acc, va + 1, ta + 1);
                fir_mac<1>(acc, vb, tb);
                __builtin_amdgcn_sched_barrier(0);
                issue(st + 2 < NS ? st + 2 : 0, va, ta);       // (the last round's request is not used)
                __builtin_amdgcn_sched_barrier(0);
                fir_mac<T / 2 - 1>(acc, vb + 1, tb + 1);
            }
            float ar = acc.x, ai = acc.y;
            {
                const float2 *c2 = reinterpret_cast<const float2 *>(col + (size_t)NR * CS);
#pragma unroll
                for (int q = 0; q < REM; q++) {
                    const float2 v = c2[q];
                    const float tq = taps[NR * M + q];
                    ar += tq * v.x;
                    ai += tq * v.y;
                }
            }
            dec[g.out_base + tid] = make_float2(ar, ai);
        }
        FIR_PROF_MARK(4);
        __syncthreads();
        FIR_PROF_MARK(5);
        if (PROF) prof[6] += 1;
        if (!more) break;
        t = t1;
        t1 = t2;
        g = gn;
    }
    if (PROF && (tid & 63) == 0) {
        for (int i = 0; i < 7; i++) atomicAdd(&g_fir_prof_dev[i], prof[i]);
        atomicAdd(&g_fir_prof_dev[7], 1ull);
    }
}

int g_fir_force_generic = 0;   // test hook: 1 = always use the runtime-M kernel
int g_fir_layout = 3;          // 3: register-resident columns, travelling accumulators (fir_reg.hip; M = 40 / 48, else 2),
                               // 2: persistent column-major LDS kernel (fir_decimate_kernel_w), 1: column-major, one tile per
                               // workgroup (fir_decimate_kernel_c), 0: polyphase rows (fir_decimate_kernel_m)
int g_fir_prof = 0;

int fir_tile_row(int decim)
{
    int r = kFirTileOut + kFirTaps / decim + 2;
    while ((r & 15) != 1) r++;
    return r;
}

int g_fir_reserve_cus = 0;     // persistent kernel: CUs left to the other streams' kernels
int g_fir_budget = 4;          // persistent kernel: tiles per workgroup before it retires (0: one resident grid)

// Which of the reference's two forms of its dispatched kernels (simd_kernels.h) the kernels follow (DESIGN.md "Arithmetic
// contract"; option "fir_order", alias "simd_order"; defined in detect.hip):
//   1 (default)  simd_avx2.c -- what the reference runs on x86 unless --no-simd is given.  avx2_fir_ccf_dec (:62-108): four
//                accumulators, fused multiply-adds (fir_decimate_kernel_f at M = 40 / 48; the runtime-M kernel otherwise);
//                avx2_fir_ccf (:28-55) / avx2_fir_fff (:115-138): a fused multiply-add per tap for the outputs of the vector
//                body, the generic form for the last n % 4 / n % 8 outputs (post1's noise filter and start filter, post2's RRC
//                filter); avx2_mag_squared (:304-323) / avx2_fftshift_mag (:177-221): fma(re, re, im*im) (post1, K1)
//   0            simd_generic.c (--no-simd, every non-x86 host): one accumulator, every product and sum rounded
//                (fir_decimate_kernel_r / _w / _c / _m)
// (the other dispatched kernels are the same operations in both forms: tests/test_oracle_vs_ref.py)

// `aligned`: the pipeline's ring lengths are multiples of 8 samples (fir_reg.hip fetches columns in pieces of 8)
// fir_layout 4 (fir_order 1 only): the AVX2 order on the matrix cores
static bool fir_mfma_ok(int decim, int aligned)
{
    return !g_fir_force_generic && g_fir_order == 1 && g_fir_layout == 4 && aligned && fir_reg_supported(decim);
}

static bool fir_fma_ok(int decim, int aligned)
{
    return !g_fir_force_generic && g_fir_order == 1 && g_fir_layout == 3 && aligned && fir_reg_supported(decim);
}

static bool fir_reg_ok(int decim, int aligned)
{
    return !g_fir_force_generic && g_fir_order == 0 && g_fir_layout == 3 && aligned && fir_reg_supported(decim);
}

static bool fir_wide_ok(int decim, int aligned)
{
    return !g_fir_force_generic && g_fir_order == 0 && (g_fir_layout == 2 || (g_fir_layout == 3 && !fir_reg_ok(decim, aligned))) &&
           (decim == 8 || decim == 16 || decim == 40 || decim == 48);
}

static int fir_wide_tile(int)
{
    // 256 outputs (one workgroup per CU, a wavefront per SIMD) and 448 (seven wavefronts, the whole LDS) measured
    // 0.84 and 0.70 ms against 0.66 ms for three workgroups of 128 per CU
    return 128;
}

// 1: launch_fir_decimate() reads the FirTile list (the one-tile-per-workgroup kernels); 0: only BurstWork::tile_base
int fir_needs_tile_list(int decim, int aligned)
{
    return fir_wide_ok(decim, aligned) || fir_reg_ok(decim, aligned) || fir_fma_ok(decim, aligned) || fir_mfma_ok(decim, aligned) ? 0 : 1;
}

// outputs per FirTile for the kernel launch_fir_decimate() will pick
int fir_tile_out(int decim, int aligned)
{
    if (fir_mfma_ok(decim, aligned)) return fir_mfma_tile_out(decim);
    if (fir_fma_ok(decim, aligned)) return fir_fma_tile_out(decim);
    if (fir_reg_ok(decim, aligned)) return fir_reg_tile_out(decim);
    return fir_wide_ok(decim, aligned) ? fir_wide_tile(decim) : kFirTileOut;
}

template <int M, int FMT, int TO>
static int launch_fir_w(const SampleSource &src, const FirGeom *geom, unsigned *next_tile, int n_tiles, const float *taps,
                         const float2 *rot_table, float2 *dec, int n_cu, hipStream_t stream)
{
    using W = FirW<M, TO>;
    int slots = (n_cu - g_fir_reserve_cus) * (int)((160 * 1024) / W::LDS);
    if (slots < 1) slots = 1;
    // budget tiles per workgroup (at least 2: the two static ones); a budget of 0 keeps one resident grid
    int budget = g_fir_budget >= 2 ? g_fir_budget : 0x3fffffff;
    int grid = n_tiles < slots ? n_tiles : slots;
    if (g_fir_budget >= 2 && (n_tiles + budget - 1) / budget > grid) grid = (n_tiles + budget - 1) / budget;
    if constexpr (M == 40 && FMT == 2) if (g_fir_prof) {
        unsigned long long z[8] = { 0 };
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fir_prof_dev), z, sizeof(z));
        (void)hipFuncSetAttribute((const void *)fir_decimate_kernel_w<M, FMT, TO, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W::LDS);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipStreamSynchronize(stream);
        (void)hipEventRecord(e0, stream);
        hipLaunchKernelGGL((fir_decimate_kernel_w<M, FMT, TO, true>), dim3(grid), dim3(TO), W::LDS, stream, src, geom, next_tile, n_tiles, budget, taps, rot_table, dec);
        (void)hipEventRecord(e1, stream);
        (void)hipStreamSynchronize(stream);
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        fprintf(stderr, "fir prof kernel %.4f ms; ", ms);
        (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_fir_prof_dev), sizeof(z));
        const double w = z[7] ? (double)z[7] : 1.0;
        fprintf(stderr, "fir prof TO=%d (cycles per wave, %llu waves, %.1f tiles/wave): prologue %.0f consume %.0f barrierA %.0f prefetch %.0f taps %.0f barrierB %.0f\n",
                TO, z[7], z[6] / w, z[0] / w, z[1] / w, z[2] / w, z[3] / w, z[4] / w, z[5] / w);
        return 0;
    }
    (void)hipFuncSetAttribute((const void *)fir_decimate_kernel_w<M, FMT, TO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W::LDS);
    hipLaunchKernelGGL((fir_decimate_kernel_w<M, FMT, TO>), dim3(grid), dim3(TO), W::LDS, stream, src, geom, next_tile, n_tiles, budget, taps, rot_table, dec);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int M, int TO>
static int launch_fir_w_fmt(const SampleSource &src, const FirGeom *geom, unsigned *next_tile, int n_tiles, const float *taps,
                            const float2 *rot_table, float2 *dec, int n_cu, hipStream_t stream)
{
    if (src.fmt == 2) return launch_fir_w<M, 2, TO>(src, geom, next_tile, n_tiles, taps, rot_table, dec, n_cu, stream);
    if (src.fmt == 1) return launch_fir_w<M, 1, TO>(src, geom, next_tile, n_tiles, taps, rot_table, dec, n_cu, stream);
    return launch_fir_w<M, 0, TO>(src, geom, next_tile, n_tiles, taps, rot_table, dec, n_cu, stream);
}

int launch_fir_decimate(const SampleSource &src, const BurstWork *work, int n_bursts, FirTile *tiles, size_t tiles_cap,
                        int n_tiles, int decim, const float *taps, const int *tap_off, const float2 *rot_incr,
                        const float2 *rot_table, int n_ckpt, float2 *dec,
                        hipStream_t stream, unsigned long long *kclk, const int *rot_slot)
{
    if (n_tiles <= 0) return 0;
#define IRDM_LAUNCH_FIR_M(MM)                                                                                  \
    do {                                                                                                       \
        const size_t lds_m = sizeof(float2) * (size_t)fir_tile_row_c(MM) * MM + sizeof(float) * (kFirTaps + 3);  \
        (void)hipFuncSetAttribute((const void *)fir_decimate_kernel_m<MM>,                                     \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m);                     \
        hipLaunchKernelGGL(fir_decimate_kernel_m<MM>, dim3(n_tiles), dim3(kFirTileOut), lds_m, stream, src,    \
                           work, tiles, taps, rot_incr, rot_table, n_ckpt, dec, rot_slot);                   \
        return hipGetLastError() == hipSuccess ? 0 : -1;                                                       \
    } while (0)
#define IRDM_LAUNCH_FIR_C(MM)                                                                                  \
    do {                                                                                                       \
        (void)hipFuncSetAttribute((const void *)fir_decimate_kernel_c<MM>,                                     \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)FirCol<MM>::LDS);           \
        hipLaunchKernelGGL(fir_decimate_kernel_c<MM>, dim3(n_tiles), dim3(kFirTileOut), FirCol<MM>::LDS, stream, src, \
                           work, tiles, taps, rot_incr, rot_table, n_ckpt, dec, rot_slot);                   \
        return hipGetLastError() == hipSuccess ? 0 : -1;                                                       \
    } while (0)
    const int aligned = src.ring_len % 8 == 0 && src.ref_ring % 8 == 0 && (src.chunk_start == ~0ull || src.chunk_start % 8 == 0);
    if (fir_mfma_ok(decim, aligned) && tiles_cap >= (size_t)n_tiles) {
        FirGeom *geom = reinterpret_cast<FirGeom *>(tiles + tiles_cap);
        unsigned *next_tile = reinterpret_cast<unsigned *>(geom + tiles_cap);
        hipLaunchKernelGGL(fir_geom_kernel, dim3((n_tiles + 255) / 256), dim3(256), 0, stream, work, n_bursts, n_tiles, decim,
                           fir_mfma_tile_out(decim), src.ring_len, src.ref_ring, rot_incr, n_ckpt, geom, next_tile, rot_slot);
        return launch_fir_mfma(src, geom, n_tiles, decim, taps, rot_table, dec, stream, kclk) == 0 ? 0 : -1;
    }
    if (fir_fma_ok(decim, aligned) && tiles_cap >= (size_t)n_tiles) {
        FirGeom *geom = reinterpret_cast<FirGeom *>(tiles + tiles_cap);
        unsigned *next_tile = reinterpret_cast<unsigned *>(geom + tiles_cap);
        hipLaunchKernelGGL(fir_geom_kernel, dim3((n_tiles + 255) / 256), dim3(256), 0, stream, work, n_bursts, n_tiles, decim,
                           fir_fma_tile_out(decim), src.ring_len, src.ref_ring, rot_incr, n_ckpt, geom, next_tile, rot_slot);
        return launch_fir_fma(src, geom, n_tiles, decim, taps, rot_table, dec, stream, kclk, next_tile) == 0 ? 0 : -1;
    }
    if (fir_reg_ok(decim, aligned) && tiles_cap >= (size_t)n_tiles) {
        FirGeom *geom = reinterpret_cast<FirGeom *>(tiles + tiles_cap);
        unsigned *next_tile = reinterpret_cast<unsigned *>(geom + tiles_cap);
        hipLaunchKernelGGL(fir_geom_kernel, dim3((n_tiles + 255) / 256), dim3(256), 0, stream, work, n_bursts, n_tiles, decim,
                           fir_reg_tile_out(decim), src.ring_len, src.ref_ring, rot_incr, n_ckpt, geom, next_tile, rot_slot);
        return launch_fir_reg(src, geom, n_tiles, decim, taps, rot_table, dec, stream, kclk) == 0 ? 0 : -1;
    }
    if (fir_wide_ok(decim, aligned) && tiles_cap >= (size_t)n_tiles) {
        FirGeom *geom = reinterpret_cast<FirGeom *>(tiles + tiles_cap);
        unsigned *next_tile = reinterpret_cast<unsigned *>(geom + tiles_cap);      // (one spare record behind the last)
        const int to = fir_wide_tile(decim);
        hipLaunchKernelGGL(fir_geom_kernel, dim3((n_tiles + 255) / 256), dim3(256), 0, stream, work, n_bursts, n_tiles, decim,
                           to, src.ring_len, src.ref_ring, rot_incr, n_ckpt, geom, next_tile, rot_slot);
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
            if (n_cu <= 0) n_cu = 256;
        }
        switch (decim) {
        case 8: return launch_fir_w_fmt<8, 128>(src, geom, next_tile, n_tiles, taps, rot_table, dec, n_cu, stream);
        case 16: return launch_fir_w_fmt<16, 128>(src, geom, next_tile, n_tiles, taps, rot_table, dec, n_cu, stream);
        case 48: return launch_fir_w_fmt<48, 128>(src, geom, next_tile, n_tiles, taps, rot_table, dec, n_cu, stream);
        case 40: return launch_fir_w_fmt<40, 128>(src, geom, next_tile, n_tiles, taps, rot_table, dec, n_cu, stream);
        default: break;
        }
    }
    if (!g_fir_force_generic && g_fir_order == 0 && g_fir_layout == 1) {
        switch (decim) {
        case 8: IRDM_LAUNCH_FIR_C(8);
        case 16: IRDM_LAUNCH_FIR_C(16);
        case 40: IRDM_LAUNCH_FIR_C(40);
        case 48: IRDM_LAUNCH_FIR_C(48);
        default: break;
        }
    }
#undef IRDM_LAUNCH_FIR_C
    if (!g_fir_force_generic && g_fir_order == 0) {
        switch (decim) {
        case 8: IRDM_LAUNCH_FIR_M(8);
        case 16: IRDM_LAUNCH_FIR_M(16);
        case 40: IRDM_LAUNCH_FIR_M(40);
        case 48: IRDM_LAUNCH_FIR_M(48);
        default: break;
        }
    }
#undef IRDM_LAUNCH_FIR_M
    const int row = fir_tile_row(decim);
    const size_t lds = sizeof(float2) * (size_t)row * decim;
    if (lds > 160 * 1024) return -1;
    (void)hipFuncSetAttribute((const void *)fir_decimate_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(fir_decimate_kernel, dim3(n_tiles), dim3(kFirTileOut), lds, stream, src, work,
                       tiles, decim, row, taps, tap_off, rot_incr, rot_table, n_ckpt, dec, g_fir_order, rot_slot);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Word copy executed BY THE GPU when the stream reaches it, with system-scope loads and stores.  Used around the host
// step of the per-burst chain, between device memory and a mapped pinned buffer the host writes in between:
//   * a small hipMemcpyAsync may read its (pinned) source when it is enqueued, i.e. before the host step has run;
//   * a kernel that follows other kernels is dispatched with an agent-scope acquire, so plain loads of host memory
//     can be served from L2 lines an earlier kernel left there (measured: the second pipeline of a process, and
//     every step after the first, read the work records WITHOUT the host's update).  sc0 sc1 accesses go to memory.
__global__ void copy_words_kernel(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        __hip_atomic_store(dst + i, __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
}

// the scan's records and header words into pinned host memory (types.hpp, gone_export_body): the stand-alone form, for
// the sequential scans and the fallback paths; the band scan does this in its last kernel
__global__ void gone_export_kernel(const DetState *__restrict__ st, const uint32_t *__restrict__ gone, int cap,
                                   uint32_t *__restrict__ hp_gone, uint32_t *__restrict__ hp_hdr,
                                   const uint32_t *__restrict__ ctl, uint32_t *__restrict__ hp_ctl, int ctl_words)
{
    gone_export_body(st, gone, cap, hp_gone, hp_hdr, ctl, hp_ctl, ctl_words, (int)gridDim.x);
}

int launch_gone_export(const DetState *st, const GoneBurst *gone, int cap, GoneBurst *hp_gone, uint32_t *hp_hdr,
                       const void *ctl, void *hp_ctl, int ctl_bytes, hipStream_t stream)
{
    hipLaunchKernelGGL(gone_export_kernel, dim3(32), dim3(256), 0, stream, st, reinterpret_cast<const uint32_t *>(gone), cap,
                       reinterpret_cast<uint32_t *>(hp_gone), hp_hdr, static_cast<const uint32_t *>(ctl),
                       static_cast<uint32_t *>(hp_ctl), ctl_bytes / 4);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_copy_words(void *dst, const void *src, size_t bytes, hipStream_t stream);

// kernel clock (common.hpp): fold the per-slot entry / exit stamps of the launch that has just ended into the record's
// sums and re-arm the slots
__device__ __forceinline__ void kclk_fold_wave(unsigned long long *__restrict__ k, int l)
{
    unsigned long long lo = k[l], hi = k[64 + l];
    k[l] = ~0ull;
    k[64 + l] = 0ull;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long lo2 = ((unsigned long long)__shfl_xor((unsigned)(lo >> 32), d) << 32) | __shfl_xor((unsigned)lo, d);
        const unsigned long long hi2 = ((unsigned long long)__shfl_xor((unsigned)(hi >> 32), d) << 32) | __shfl_xor((unsigned)hi, d);
        lo = lo2 < lo ? lo2 : lo;
        hi = hi2 > hi ? hi2 : hi;
    }
    if (l == 0 && lo != ~0ull && hi >= lo) {
        k[128] += hi - lo;
        k[129] += 1ull;
        k[130] = hi - lo;
    }
}

__global__ __launch_bounds__(64) void kclk_fold_kernel(unsigned long long *__restrict__ k) { kclk_fold_wave(k, (int)threadIdx.x); }

int launch_kclk_fold(unsigned long long *kclk, hipStream_t stream)
{
    if (!kclk) return 0;
    hipLaunchKernelGGL(kclk_fold_kernel, dim3(1), dim3(64), 0, stream, kclk);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
int launch_copy_to_host(void *dst, const void *src, size_t bytes, hipStream_t stream);

// Device memory -> pinned host memory, 16 bytes per lane, when the stream gets there.  (Results of the per-burst chain:
// a hipMemcpyAsync D2H is carried out by the runtime's copy path, which next to the chains' kernels answered late --
// see gone_export_kernel.)  Plain stores: nothing on the device reads this memory back; the fence and the end of the
// kernel make it visible to the host.
__global__ void copy_to_host_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n16,
                                    uint32_t *__restrict__ dst_tail, const uint32_t *__restrict__ src_tail, int n_tail)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
    if (blockIdx.x == 0 && (int)threadIdx.x < n_tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
    __threadfence_system();
}

__global__ void copy2_to_host_kernel(uint4 *__restrict__ dst_a, const uint4 *__restrict__ src_a, size_t n_a,
                                     uint4 *__restrict__ dst_b, const uint4 *__restrict__ src_b, size_t n_b)
{
    const size_t n = n_a + n_b;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (i < n_a) dst_a[i] = src_a[i];
        else dst_b[i - n_a] = src_b[i - n_a];
    }
    __threadfence_system();
}

// two ranges with one launch (both 16-byte aligned, lengths multiples of 16 bytes)
int launch_copy2_to_host(void *dst_a, const void *src_a, size_t bytes_a, void *dst_b, const void *src_b, size_t bytes_b,
                         hipStream_t stream)
{
    if (((reinterpret_cast<uintptr_t>(dst_a) | reinterpret_cast<uintptr_t>(src_a) | reinterpret_cast<uintptr_t>(dst_b) |
          reinterpret_cast<uintptr_t>(src_b) | bytes_a | bytes_b) & 15) != 0) {
        if (launch_copy_to_host(dst_a, src_a, bytes_a, stream) != 0) return -1;
        return launch_copy_to_host(dst_b, src_b, bytes_b, stream);
    }
    const size_t n = (bytes_a + bytes_b) / 16;
    if (!n) return 0;
    const size_t wg = (size_t)g_small_wg;
    const int grid = (int)std::min<size_t>((n + wg - 1) / wg, 512 * 256 / wg);
    hipLaunchKernelGGL(copy2_to_host_kernel, dim3(grid), dim3((unsigned)wg), 0, stream, static_cast<uint4 *>(dst_a),
                       static_cast<const uint4 *>(src_a), bytes_a / 16, static_cast<uint4 *>(dst_b),
                       static_cast<const uint4 *>(src_b), bytes_b / 16);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_copy_to_host(void *dst, const void *src, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15 || (bytes & 3))
        return launch_copy_words(dst, src, bytes, stream);
    const size_t n16 = bytes / 16;
    const int n_tail = (int)((bytes - n16 * 16) / 4);
    const size_t wg = (size_t)g_small_wg;
    const int grid = (int)std::min<size_t>((n16 + wg - 1) / wg + 1, 512 * 256 / wg);
    hipLaunchKernelGGL(copy_to_host_kernel, dim3(grid), dim3((unsigned)wg), 0, stream, static_cast<uint4 *>(dst),
                       static_cast<const uint4 *>(src), n16, reinterpret_cast<uint32_t *>(static_cast<char *>(dst) + n16 * 16),
                       reinterpret_cast<const uint32_t *>(static_cast<const char *>(src) + n16 * 16), n_tail);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// A device-to-device copy of a chunk's size by kernel: 16 bytes per lane over the whole chip -- option copy_wide 1.  The
// default (0) stays hipMemcpyAsync: it goes through the DMA engines (a 512 MB chunk in ~5 ms, ~100 GB/s) BESIDE the
// kernels, and in a pipelined run that is the better place for a copy nobody waits for: chunks not fed in place
// (bench.py --ingest 0) 64.7-65.4 Gsamples/s with it against 60.7-61.0 with this kernel taking the CUs.
__global__ __launch_bounds__(256) void copy_wide_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n16)
{
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

int g_copy_wide = 0;        // 1: the history-ring copies by kernel instead of hipMemcpyAsync (A/B: slower in run)
int launch_copy_wide(void *dst, const void *src, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return 0;
    if (!g_copy_wide || bytes % 16 != 0 || (reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) % 16 != 0 ||
        bytes < (1u << 20))
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream) == hipSuccess ? 0 : -1;
    const size_t n16 = bytes / 16;
    const int grid = (int)std::min<size_t>((n16 + 255) / 256, 8192);
    hipLaunchKernelGGL(copy_wide_kernel, dim3(grid), dim3(256), 0, stream, static_cast<uint4 *>(dst), static_cast<const uint4 *>(src), n16);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_copy_words(void *dst, const void *src, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return 0;
    const size_t n = bytes / 4;
    const size_t wg = (size_t)g_small_wg;
    const int grid = (int)std::min<size_t>((n + wg - 1) / wg, 256 * 256 / wg);
    hipLaunchKernelGGL(copy_words_kernel, dim3(grid), dim3((unsigned)wg), 0, stream, static_cast<uint32_t *>(dst),
                       static_cast<const uint32_t *>(src), n);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// The stream waits here until the host has published `seq` in a mapped pinned word (the per-burst chain's host step:
// hipLaunchHostFunc did not hold later work back on this ROCm, measured -- so the hand-shake is explicit).  One lane,
// system-scope loads, bounded: ~2 s without an answer sets *err and lets the stream go (the results are then wrong and
// bursts_finish reports the error).
__global__ void wait_host_flag_kernel(const uint32_t *flag, uint32_t seq, uint32_t *err)
{
    for (long long n = 0; n < 20000000ll; n++) {
        if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == seq) return;
        __builtin_amdgcn_s_sleep(8);
    }
    *err = 1;
}

int launch_wait_host_flag(const uint32_t *flag, uint32_t seq, uint32_t *err, hipStream_t stream)
{
    hipLaunchKernelGGL(wait_host_flag_kernel, dim3(1), dim3(1), 0, stream, flag, seq, err);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// burst_data_t.samples re-gathered (stage probe for parity tests)
__global__ void gather_burst_kernel(SampleSource src, uint64_t start, uint64_t avail_end, int n,
                                    float2 *__restrict__ out)
{
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
        out[k] = burst_sample(src, start, avail_end, k);
}

int launch_gather_burst(const SampleSource &src, uint64_t start, uint64_t avail_end, int n,
                        float2 *out, hipStream_t stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(gather_burst_kernel, dim3(256), dim3(256), 0, stream, src, start, avail_end, n, out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- block reductions (256 threads) ----
__device__ __forceinline__ float block_max(float v, float *red)
{
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_down(v, off);
        v = o > v ? o : v;
    }
    const int tid = threadIdx.x;
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); i++) r = red[i] > r ? red[i] : r;
    __syncthreads();
    return r;
}

__device__ __forceinline__ int block_min_int(int v, int *red)
{
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_down(v, off);
        v = o < v ? o : v;
    }
    const int tid = threadIdx.x;
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    int r = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); i++) r = red[i] < r ? red[i] : r;
    __syncthreads();
    return r;
}

// first-strict-max semantics of `if (m > max) { max = m; idx = i; }` (burst_downmix.c:497-505,
// :565-586): larger value wins, ties keep the lower index, start value (0, 0)
__device__ __forceinline__ void argmax_combine(float &m, int &i, float om, int oi)
{
    if (om > m || (om == m && oi < i)) { m = om; i = oi; }
}

__device__ __forceinline__ void block_argmax(float &m, int &i, float *redf, int *redi)
{
    for (int off = 32; off > 0; off >>= 1) {
        const float om = __shfl_down(m, off);
        const int oi = __shfl_down(i, off);
        argmax_combine(m, i, om, oi);
    }
    const int tid = threadIdx.x;
    if ((tid & 63) == 0) { redf[tid >> 6] = m; redi[tid >> 6] = i; }
    __syncthreads();
    m = redf[0]; i = redi[0];
    for (int k = 1; k < (int)(blockDim.x >> 6); k++) argmax_combine(m, i, redf[k], redi[k]);
    __syncthreads();
}

__device__ __forceinline__ float parabolic(float alpha, float beta, float gamma)
{
    const float denom = alpha - 2.0f * beta + gamma;
    if (fabsf(denom) > 1e-10f) return 0.5f * (alpha - gamma) / denom;
    return 0.0f;
}

constexpr int kPostThreads = 256;

// ---------------------------------------------------------------------------
// post1: one workgroup per burst.  Steps 2b and 3 walk the burst in tiles of kPostTile outputs staged in LDS (four per
// thread: a burst of 6000 decimated samples is six rounds of load - barrier - filter - barrier - box filter - barrier
// instead of twenty-four; each round is mostly the latency of its loads): the
// noise filter reads its 25 neighbours and the start filter its 20 from LDS with unrolled loops (the first version read
// them from HBM/L2 one dependent load per tap: 0.5 ms for 667 bursts).  The start filter's outputs are kept in the
// burst's row of `dec` (as floats, behind the part of the row the tiles still read) for the threshold pass.
// NT / SN: compile-time tap counts (25 / 20 at every supported rate), 0 = the runtime values.
// ---------------------------------------------------------------------------
constexpr int kPostMaxTaps = 64;
constexpr int kPostTile = 4 * kPostThreads;
static_assert(sizeof(float2) * (kPostTile + 2 * kPostMaxTaps) + sizeof(float) * (kPostTile + kPostMaxTaps) <= sizeof(float2) * kCfoTotal,
              "the tile buffers live in the CFO transform's LDS");

// what the host's fine-CFO step reads, stored straight into the burst's record in mapped pinned memory (system scope):
// the helper thread is released by an event behind this kernel, no copy pass in between
__device__ __forceinline__ void post1_publish(BurstWork &hp, const BurstWork &w)
{
    __hip_atomic_store(&hp.start_idx, w.start_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(reinterpret_cast<uint32_t *>(&hp.center_offset), __float_as_uint(w.center_offset), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&hp.drop_reason, w.drop_reason, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
}

template <int NT, int SN>
__global__ __launch_bounds__(kPostThreads) void downmix_post1_kernel(
    BurstWork *__restrict__ work, float2 *__restrict__ dec,
    float2 *__restrict__ lpf, const float *__restrict__ noise_taps, int noise_ntaps_rt,
    const float *__restrict__ start_taps, int start_ntaps_rt, int search_depth, int pre_start,
    const float *__restrict__ cfo_window, const float2 *__restrict__ tw4096, BurstWork *__restrict__ hp_work, int order)
{
    __shared__ __attribute__((aligned(16))) float2 s[kCfoTotal];
    __shared__ float redf[4];
    __shared__ int redi[4];
    __builtin_amdgcn_s_setprio(2);      // latency-bound, next to the decimator of the next chunk
    const int tid = threadIdx.x;
    BurstWork &w = work[blockIdx.x];
    if (w.drop_reason != 0) return;
    const int noise_ntaps = NT ? NT : noise_ntaps_rt;
    const int start_ntaps = SN ? SN : start_ntaps_rt;
    const int dec_len = w.dec_len;
    float2 *x = dec + (size_t)w.dec_off;
    float *fscr = reinterpret_cast<float *>(x);         // start-filter outputs: float i aliases x[i / 2], read long before
    float2 *y = lpf + (size_t)w.dec_off;

    // step 3 geometry (burst_downmix.c:441-478)
    int search = search_depth < dec_len ? search_depth : dec_len;
    int mag_len = search + start_ntaps - 1;
    if (mag_len > dec_len) mag_len = dec_len;
    int flen = mag_len - start_ntaps + 1;
    if (flen > search) flen = search;

    // the tile buffers live in the FFT's LDS (not in use yet)
    float2 *xs = s;                                          // kPostTile + 2 * kPostMaxTaps samples
    float *m2 = reinterpret_cast<float *>(s + kPostTile + 2 * kPostMaxTaps);         // kPostTile + kPostMaxTaps
    const bool do_lpf = dec_len - noise_ntaps + 1 > 0;       // burst_downmix.c:683-698
    const int half = (noise_ntaps - 1) / 2;
    const int span_y = kPostTile + start_ntaps - 1;          // LPF outputs a tile's start filter needs
    const int span_x = span_y + noise_ntaps - 1;
    // option "fir_order" 1: the reference's AVX2 forms (simd_avx2.c) -- the outputs of a kernel's vector body take a fused
    // multiply-add per tap, its last n % 4 (fir_ccf, mag_squared) / n % 8 (fir_fff) outputs the generic form
    const int lpf_vec = order ? (dec_len & ~3) : 0;          // avx2_fir_ccf over dec_len outputs (burst_downmix.c:693)
    const int mag_vec = order ? (mag_len & ~3) : 0;          // avx2_mag_squared over mag_len (burst_downmix.c:450)
    const int box_vec = order ? (flen > 0 ? (flen & ~7) : 0) : 0;   // avx2_fir_fff over flen outputs (burst_downmix.c:458)
    float mx = -1e30f;
    for (int B = 0; B < dec_len; B += kPostTile) {
        for (int q = tid; q < span_x; q += kPostThreads) {
            const int j = B - half + q;
            xs[q] = (j >= 0 && j < dec_len) ? x[j] : make_float2(0.0f, 0.0f);
        }
        __syncthreads();
        // step 2b: centred LPF over the zero-padded burst, outputs B .. B + span_y
        for (int p = tid; p < span_y; p += kPostThreads) {
            if (B + p >= dec_len) break;
            float2 v;
            if (do_lpf) {
                float ar = 0.0f, ai = 0.0f;
                if (B + p < lpf_vec) {
#pragma unroll
                    for (int k = 0; k < (NT ? NT : 1); k++) {
                        if (NT) {
                            const float2 u = xs[p + k];
                            const float t = noise_taps[k];
                            ar = __builtin_fmaf(t, u.x, ar);
                            ai = __builtin_fmaf(t, u.y, ai);
                        }
                    }
                    if (!NT) {
                        for (int k = 0; k < noise_ntaps; k++) {
                            const float2 u = xs[p + k];
                            const float t = noise_taps[k];
                            ar = __builtin_fmaf(t, u.x, ar);
                            ai = __builtin_fmaf(t, u.y, ai);
                        }
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < (NT ? NT : 1); k++) {
                        if (NT) {
                            const float2 u = xs[p + k];
                            const float t = noise_taps[k];
                            ar += t * u.x;
                            ai += t * u.y;
                        }
                    }
                    if (!NT) {
                        for (int k = 0; k < noise_ntaps; k++) {
                            const float2 u = xs[p + k];
                            const float t = noise_taps[k];
                            ar += t * u.x;
                            ai += t * u.y;
                        }
                    }
                }
                v = make_float2(ar, ai);
            } else {
                v = xs[p + half];
            }
            if (p < kPostTile) y[B + p] = v;
            m2[p] = B + p < mag_vec ? mag2_fma(v) : mag2(v);
        }
        __syncthreads();
        // step 3, first half: the box filter over |y|^2
        for (int o = tid; o < kPostTile && B + o < flen; o += kPostThreads) {
            float acc = 0.0f;
            if (B + o < box_vec) {
#pragma unroll
                for (int k = 0; k < (SN ? SN : 1); k++)
                    if (SN) acc = __builtin_fmaf(start_taps[k], m2[o + k], acc);
                if (!SN)
                    for (int k = 0; k < start_ntaps; k++) acc = __builtin_fmaf(start_taps[k], m2[o + k], acc);
            } else {
#pragma unroll
                for (int k = 0; k < (SN ? SN : 1); k++)
                    if (SN) acc += start_taps[k] * m2[o + k];
                if (!SN)
                    for (int k = 0; k < start_ntaps; k++) acc += start_taps[k] * m2[o + k];
            }
            fscr[B + o] = acc;
            mx = acc > mx ? acc : mx;
        }
        __syncthreads();
    }

    int start = 0;
    if (flen > 0) {
        mx = block_max(mx, redf);
        const float thr = 0.45f * mx;                       // START_THRESHOLD
        int first = flen;
        for (int i = tid; i < flen; i += kPostThreads) {
            if (fscr[i] >= thr) { first = i; break; }
        }
        start = block_min_int(first, redi);
        if (start > 0) {
            start = start + (start_ntaps - 1) / 2 - pre_start;
            if (start < 0) start = 0;
        }
    }
    if (start >= dec_len - 100) {                           // burst_downmix.c:702-705
        if (tid == 0) {
            w.start_idx = start;
            w.drop_reason = 3;
            if (hp_work) post1_publish(hp_work[blockIdx.x], w);
        }
        return;
    }
    const int frame_len = dec_len - start;

    // step 4 (burst_downmix.c:482-535): x^2 * blackman(256), zero-padded 4096-pt FFT
    int n = kCfoN < frame_len ? kCfoN : frame_len;
    for (int i = tid; i < kCfoTotal; i += kPostThreads) {
        float2 v = make_float2(0.0f, 0.0f);
        if (i < n) {
            const float2 sv = y[start + i];
            const float2 sq = cmul(sv, sv);                 // simd_csquare_window: (s*s)*w
            const float wv = cfo_window[i];
            v = make_float2(sq.x * wv, sq.y * wv);
        }
        s[bitrev((unsigned)i, 12)] = v;
    }
    __syncthreads();
    fft_lds_radix2<12, kPostThreads, -1>(s, tw4096);
    float bm = 0.0f;
    int bi = 0;
    for (int i = tid; i < kCfoTotal; i += kPostThreads) {
        const float m = mag2(s[i]);
        if (m > bm) { bm = m; bi = i; }
    }
    block_argmax(bm, bi, redf, redi);
    if (tid == 0) {
        const int idx = bi >= kCfoTotal / 2 ? bi - kCfoTotal : bi;
        float corr = 0.0f;
        if (bi > 0 && bi < kCfoTotal - 1) {
            const int im1 = idx - 1 < 0 ? idx - 1 + kCfoTotal : idx - 1;
            const int ip1 = idx + 1 < 0 ? idx + 1 + kCfoTotal : idx + 1;
            corr = parabolic(mag2(s[im1]), bm, mag2(s[ip1]));
        }
        w.start_idx = start;
        w.center_offset = ((float)idx + corr) / (float)kCfoTotal / 2.0f;
        if (hp_work) post1_publish(hp_work[blockIdx.x], w);
    }
}

// ---------------------------------------------------------------------------
// post1 as TWO launches (round 6).  The one-workgroup-per-burst kernel above walks a burst tile by tile -- seven rounds of
// load - barrier - filter - barrier - box filter - barrier for a 6700-sample burst, each round mostly the latency of its
// loads -- and then runs a 12-stage radix-2 transform with a barrier per stage: 87 us alone for 36 MB of data, every
// workgroup's latency end to end, and it holds the chip that long.
//   post_tiles_kernel  grid (tiles of the longest burst, bursts): ONE tile of kPostTile outputs per workgroup -- steps 2b
//                      and 3's box filter for all tiles of all bursts at once (the same loops on the same operands as
//                      above); the start filter's outputs go to a row of floats of their own (`box`: the tiles of a burst
//                      run side by side, the alias into `dec` would be read while it is written), the burst's maximum is
//                      an atomic max over its tiles (the outputs are sums of products of non-negative numbers: their
//                      bits order like unsigned integers).
//   post_cfo_kernel    one workgroup per burst: threshold search, burst start, drop rule, fine CFO.  The 4096-point
//                      transform of 256 samples: the first four stages of the pinned radix-2 transform combine x[i] with
//                      x[i + 2048 .. 256] = 0 -- a + W * 0 = a, a - W * 0 = a, exactly -- so they are a copy: position
//                      16 q + r holds x[bitrev8(q)] after them; stages 5..12 run two per barrier (fft_lds_radix2x2).
// ---------------------------------------------------------------------------
template <int NT, int SN>
__global__ __launch_bounds__(kPostThreads) void post_tiles_kernel(
    BurstWork *__restrict__ work, const float2 *__restrict__ dec, float2 *__restrict__ lpf, float *__restrict__ box,
    const float *__restrict__ noise_taps, int noise_ntaps_rt, const float *__restrict__ start_taps, int start_ntaps_rt,
    int search_depth, int order, unsigned long long *__restrict__ kclk)
{
    __shared__ __attribute__((aligned(16))) float2 xs[kPostTile + 2 * kPostMaxTaps];
    __shared__ float m2[kPostTile + kPostMaxTaps];
    __shared__ float redf[4];
    __builtin_amdgcn_s_setprio(2);
    const int tid = threadIdx.x;
    // (kernel clock: the decimator in front of this launch has ended -- its stamps are folded here instead of by a launch
    // of their own at the chain's end)
    if (kclk && blockIdx.x == 0 && blockIdx.y == 0 && tid < 64) kclk_fold_wave(kclk, tid);
    BurstWork &w = work[blockIdx.y];
    if (w.drop_reason != 0) return;
    const int dec_len = w.dec_len;
    const int B = (int)blockIdx.x * kPostTile;
    if (B >= dec_len) return;
    const int noise_ntaps = NT ? NT : noise_ntaps_rt;
    const int start_ntaps = SN ? SN : start_ntaps_rt;
    const float2 *x = dec + (size_t)w.dec_off;
    float2 *y = lpf + (size_t)w.dec_off;
    float *fscr = box + (size_t)w.dec_off;

    // step 3 geometry (burst_downmix.c:441-478)
    int search = search_depth < dec_len ? search_depth : dec_len;
    int mag_len = search + start_ntaps - 1;
    if (mag_len > dec_len) mag_len = dec_len;
    int flen = mag_len - start_ntaps + 1;
    if (flen > search) flen = search;

    const bool do_lpf = dec_len - noise_ntaps + 1 > 0;       // burst_downmix.c:683-698
    const int half = (noise_ntaps - 1) / 2;
    const int span_y = kPostTile + start_ntaps - 1;          // LPF outputs a tile's start filter needs
    const int span_x = span_y + noise_ntaps - 1;
    const int lpf_vec = order ? (dec_len & ~3) : 0;          // avx2_fir_ccf over dec_len outputs (burst_downmix.c:693)
    const int mag_vec = order ? (mag_len & ~3) : 0;          // avx2_mag_squared over mag_len (burst_downmix.c:450)
    const int box_vec = order ? (flen > 0 ? (flen & ~7) : 0) : 0;   // avx2_fir_fff over flen outputs (burst_downmix.c:458)
    for (int q = tid; q < span_x; q += kPostThreads) {
        const int j = B - half + q;
        xs[q] = (j >= 0 && j < dec_len) ? x[j] : make_float2(0.0f, 0.0f);
    }
    __syncthreads();
    // step 2b: centred LPF over the zero-padded burst, outputs B .. B + span_y
    for (int p = tid; p < span_y; p += kPostThreads) {
        if (B + p >= dec_len) break;
        float2 v;
        if (do_lpf) {
            float ar = 0.0f, ai = 0.0f;
            if (B + p < lpf_vec) {
#pragma unroll
                for (int k = 0; k < (NT ? NT : 1); k++) {
                    if (NT) {
                        const float2 u = xs[p + k];
                        const float t = noise_taps[k];
                        ar = __builtin_fmaf(t, u.x, ar);
                        ai = __builtin_fmaf(t, u.y, ai);
                    }
                }
                if (!NT) {
                    for (int k = 0; k < noise_ntaps; k++) {
                        const float2 u = xs[p + k];
                        const float t = noise_taps[k];
                        ar = __builtin_fmaf(t, u.x, ar);
                        ai = __builtin_fmaf(t, u.y, ai);
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < (NT ? NT : 1); k++) {
                    if (NT) {
                        const float2 u = xs[p + k];
                        const float t = noise_taps[k];
                        ar += t * u.x;
                        ai += t * u.y;
                    }
                }
                if (!NT) {
                    for (int k = 0; k < noise_ntaps; k++) {
                        const float2 u = xs[p + k];
                        const float t = noise_taps[k];
                        ar += t * u.x;
                        ai += t * u.y;
                    }
                }
            }
            v = make_float2(ar, ai);
        } else {
            v = xs[p + half];
        }
        if (p < kPostTile) y[B + p] = v;
        m2[p] = B + p < mag_vec ? mag2_fma(v) : mag2(v);
    }
    __syncthreads();
    // step 3, first half: the box filter over |y|^2
    float mx = -1.0f;                                        // (no output of this thread yet; the outputs are >= 0)
    for (int o = tid; o < kPostTile && B + o < flen; o += kPostThreads) {
        float acc = 0.0f;
        if (B + o < box_vec) {
#pragma unroll
            for (int k = 0; k < (SN ? SN : 1); k++)
                if (SN) acc = __builtin_fmaf(start_taps[k], m2[o + k], acc);
            if (!SN)
                for (int k = 0; k < start_ntaps; k++) acc = __builtin_fmaf(start_taps[k], m2[o + k], acc);
        } else {
#pragma unroll
            for (int k = 0; k < (SN ? SN : 1); k++)
                if (SN) acc += start_taps[k] * m2[o + k];
            if (!SN)
                for (int k = 0; k < start_ntaps; k++) acc += start_taps[k] * m2[o + k];
        }
        fscr[B + o] = acc;
        mx = acc > mx ? acc : mx;
    }
    mx = block_max(mx, redf);
    // (`if (v > max) max = v` from -1e30 over outputs that are never negative: the largest output, +0 at least)
    if (tid == 0 && mx >= 0.0f) atomicMax(&w.box_max, __float_as_uint(mx));
}

template <int SN>
__global__ __launch_bounds__(kPostThreads) void post_cfo_kernel(
    BurstWork *__restrict__ work, const float2 *__restrict__ lpf, const float *__restrict__ box, int start_ntaps_rt,
    int search_depth, int pre_start, const float *__restrict__ cfo_window, const float2 *__restrict__ tw4096,
    BurstWork *__restrict__ hp_work)
{
    __shared__ __attribute__((aligned(16))) float2 s[kCfoTotal];
    __shared__ float redf[4];
    __shared__ int redi[4];
    __builtin_amdgcn_s_setprio(2);
    const int tid = threadIdx.x;
    BurstWork &w = work[blockIdx.x];
    if (w.drop_reason != 0) return;
    const int start_ntaps = SN ? SN : start_ntaps_rt;
    const int dec_len = w.dec_len;
    const float2 *y = lpf + (size_t)w.dec_off;
    const float *fscr = box + (size_t)w.dec_off;
    int search = search_depth < dec_len ? search_depth : dec_len;
    int mag_len = search + start_ntaps - 1;
    if (mag_len > dec_len) mag_len = dec_len;
    int flen = mag_len - start_ntaps + 1;
    if (flen > search) flen = search;

    int start = 0;
    if (flen > 0) {
        const float thr = 0.45f * __uint_as_float(w.box_max);          // START_THRESHOLD
        int first = flen;
        for (int i = tid; i < flen; i += kPostThreads) {
            if (fscr[i] >= thr) { first = i; break; }
        }
        start = block_min_int(first, redi);
        if (start > 0) {
            start = start + (start_ntaps - 1) / 2 - pre_start;
            if (start < 0) start = 0;
        }
    }
    if (start >= dec_len - 100) {                           // burst_downmix.c:702-705
        if (tid == 0) {
            w.start_idx = start;
            w.drop_reason = 3;
            if (hp_work) post1_publish(hp_work[blockIdx.x], w);
        }
        return;
    }
    const int frame_len = dec_len - start;

    // step 4 (burst_downmix.c:482-535): x^2 * blackman(256), zero-padded 4096-pt FFT -- entered behind its first four
    // stages: sixteen copies of x[bitrev8(q)] at positions 16 q .. 16 q + 15
    static_assert(kCfoTotal == 4096 && kCfoN == 256 && kPostThreads == 256, "the pruned transform's geometry");
    const int n = kCfoN < frame_len ? kCfoN : frame_len;
    {
        const int i = (int)bitrev((unsigned)tid, 8);
        float2 v = make_float2(0.0f, 0.0f);
        if (i < n) {
            const float2 sv = y[start + i];
            const float2 sq = cmul(sv, sv);                 // simd_csquare_window: (s*s)*w
            const float wv = cfo_window[i];
            v = make_float2(sq.x * wv, sq.y * wv);
        }
        float4 *d4 = reinterpret_cast<float4 *>(s + 16 * tid);
        const float4 vv = make_float4(v.x, v.y, v.x, v.y);
#pragma unroll
        for (int r = 0; r < 8; r++) d4[r] = vv;
    }
    __syncthreads();
    fft_lds_radix2x2<12, kPostThreads, -1, 5>(s, tw4096);
    float bm = 0.0f;
    int bi = 0;
    for (int i = tid; i < kCfoTotal; i += kPostThreads) {
        const float m = mag2(s[i]);
        if (m > bm) { bm = m; bi = i; }
    }
    block_argmax(bm, bi, redf, redi);
    if (tid == 0) {
        const int idx = bi >= kCfoTotal / 2 ? bi - kCfoTotal : bi;
        float corr = 0.0f;
        if (bi > 0 && bi < kCfoTotal - 1) {
            const int im1 = idx - 1 < 0 ? idx - 1 + kCfoTotal : idx - 1;
            const int ip1 = idx + 1 < 0 ? idx + 1 + kCfoTotal : idx + 1;
            corr = parabolic(mag2(s[im1]), bm, mag2(s[ip1]));
        }
        w.start_idx = start;
        w.center_offset = ((float)idx + corr) / (float)kCfoTotal / 2.0f;
        if (hp_work) post1_publish(hp_work[blockIdx.x], w);
    }
}

int g_small_wg = 256;       // threads per workgroup of the chain's little copy / threshold kernels.  Option small_wg 64: a single
                            // wavefront finds a slot beside the decimator's resident grid where a 256-thread workgroup needs one
                            // on all four SIMDs of a CU -- measured: nothing in run (71.1 / 71.9 against 72.5 / 71.1 Gsamples/s,
                            // 12 MHz dense 25.5 / 25.5 against 26.0 / 25.5), every stage serial 1.82 against 1.88 ms
int g_post_generic = 0;     // test hook: 1 = the runtime-tap-count instance of post1 / post2
int g_rot_store = 1;        // rot_phase: 1 = the phases leave as rows through LDS (rot_phase_rows_kernel), 0 = a row per lane

int g_post_split = 1;       // post1: 1 = post_tiles_kernel + post_cfo_kernel (a tile per workgroup, the pruned transform), 0 = one
                            // workgroup per burst (downmix_post1_kernel)

int launch_downmix_post1(BurstWork *work, int n_bursts, int max_dec_len, float2 *dec,
                         float2 *lpf, float *box, const float *noise_taps, int noise_ntaps,
                         const float *start_taps, int start_ntaps, int search_depth, int pre_start,
                         const float *cfo_window, const float2 *tw4096, BurstWork *hp_work, hipStream_t stream,
                         unsigned long long *kclk)
{
    if (n_bursts <= 0) return 0;
    if (noise_ntaps > kPostMaxTaps || start_ntaps > kPostMaxTaps) return -1;
    const bool fixed = noise_ntaps == 25 && start_ntaps == 20 && !g_post_generic;
    if (g_post_split) {
        const dim3 grid((unsigned)((max_dec_len + kPostTile - 1) / kPostTile), (unsigned)n_bursts);
        if (grid.x > 0) {
            if (fixed)
                hipLaunchKernelGGL((post_tiles_kernel<25, 20>), grid, dim3(kPostThreads), 0, stream, work, dec, lpf, box, noise_taps,
                                   noise_ntaps, start_taps, start_ntaps, search_depth, g_fir_order, kclk);
            else
                hipLaunchKernelGGL((post_tiles_kernel<0, 0>), grid, dim3(kPostThreads), 0, stream, work, dec, lpf, box, noise_taps,
                                   noise_ntaps, start_taps, start_ntaps, search_depth, g_fir_order, kclk);
        } else if (launch_kclk_fold(kclk, stream) != 0) {
            return -1;
        }
        if (fixed)
            hipLaunchKernelGGL((post_cfo_kernel<20>), dim3(n_bursts), dim3(kPostThreads), 0, stream, work, lpf, box, start_ntaps,
                               search_depth, pre_start, cfo_window, tw4096, hp_work);
        else
            hipLaunchKernelGGL((post_cfo_kernel<0>), dim3(n_bursts), dim3(kPostThreads), 0, stream, work, lpf, box, start_ntaps,
                               search_depth, pre_start, cfo_window, tw4096, hp_work);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    if (launch_kclk_fold(kclk, stream) != 0) return -1;
    if (fixed)
        hipLaunchKernelGGL((downmix_post1_kernel<25, 20>), dim3(n_bursts), dim3(kPostThreads), 0, stream, work, dec,
                           lpf, noise_taps, noise_ntaps, start_taps, start_ntaps, search_depth,
                           pre_start, cfo_window, tw4096, hp_work, g_fir_order);
    else
        hipLaunchKernelGGL((downmix_post1_kernel<0, 0>), dim3(n_bursts), dim3(kPostThreads), 0, stream, work, dec,
                           lpf, noise_taps, noise_ntaps, start_taps, start_ntaps, search_depth,
                           pre_start, cfo_window, tw4096, hp_work, g_fir_order);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------------------
// Step 5's phase sequence (burst_downmix.c:713-720): phase_0 = 1, phase_{k+1} = phase_k * incr in float -- a chain of
// up to kFrameNeed dependent complex multiplies per burst (~45 cycles each: ~0.1 ms).  One LANE per burst: eleven
// wavefronts walk 667 chains, instead of 667 workgroups each waiting for its lane 0 with 48 KB of LDS in hand.
// The phases go to the burst's row of rrc_ws (post2 overwrites the row with the RRC output afterwards).
// ---------------------------------------------------------------------------
// cfo.on_device: the libm step between post1 and post2 -- cexpf(-2 pi offset i) and the centre frequency that decides
// between the normal and the simplex frame limits (burst_downmix.c:663-671, :716-719, :764) -- is taken HERE, with
// glibc's sincosf restated in libm_port.hpp (bit-identical to the host's over every float of the step's range:
// tools/check_sincosf.cpp, tools/check_sincosf_gpu.hip) and the reference's own mixed float / double expression for
// the frequency.  The chain then never leaves the GPU (before: event -> helper thread -> sequence number -> a kernel
// spinning on pinned memory, 0.4 ms per chunk in run).
typedef unsigned rot_u32x4 __attribute__((vector_size(16)));

// what the chain of a burst starts from: the libm step (or the host's results), then the burst's record fields
struct RotStart {
    bool live;
    int drop, dec_len, start, simplex;
    float inc_re, inc_im;
};
// Decimated samples behind `start` that can reach the output frame of a burst: the unique word starts at most 889 samples in
// (the correlation peak lies in the first 840 samples, burst_downmix.c:565-586, :633-636: 839 - 271 + 1 + 320), the frame
// is at most 1910 / 4440 samples long (normal / simplex, iridium.h:23-27), and the 51-tap RRC filter reads 25 samples past
// its output: 2824 / 5354, rounded up.  The class is known behind the fine CFO (BurstWork::simplex).
constexpr int kFrameNeedNormal = 2832;
static_assert(kFrameNeed >= 889 + kMaxFrameSamples + 25 && kFrameNeedNormal >= 889 + 1910 + 25, "frame need");
__device__ __forceinline__ int frame_need(int simplex) { return simplex ? kFrameNeed : kFrameNeedNormal; }
__device__ __forceinline__ RotStart rot_phase_start(BurstWork *__restrict__ work, int n_bursts, const BurstWork *__restrict__ hp_work,
                                                    const CfoStep &cfo)
{
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.x * 64 + threadIdx.x;
    const bool live = b < n_bursts;
    int w_drop = 1, w_dec_len = 0, w_start = 0, w_simplex = 0;
    float inc_re = 1.0f, inc_im = 0.0f;
    if (live) {
        BurstWork &w = work[b];
        if (cfo.on_device) {
            const float rel = (w.center_bin - cfo.n_fft / 2) / (float)cfo.n_fft;
            double cf = cfo.center_frequency;
            cf += rel * cfo.sample_rate;                                        // burst_downmix.c:663-671 (float product)
            if (!w.drop_reason) {
                const float phase_inc = -2.0f * 3.14159274101257324f * w.center_offset;       // -2.0f * (float)M_PI * offset
                float re, im;
                if (libm_cexpf_i<true>(phase_inc, &re, &im) != 0) {
                    // (outside the restated domain: cannot happen, |offset| <= 1/4; drop rather than be wrong)
                    w.drop_reason = 9;
                    re = 1.0f;
                    im = 0.0f;
                }
                w.incr_re = re;
                w.incr_im = im;
                cf += w.center_offset * cfo.out_rate;
            }
            w.simplex = cf > 1626000000 ? 1 : 0;                                // iridium.h:18
        } else if (hp_work) {
            // what the host's fine-CFO step left in the mapped pinned record (system-scope loads: the wait kernel in front
            // of this one has seen the helper thread's sequence number)
            w.incr_re = __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t *>(&hp_work[b].incr_re), __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_SYSTEM));
            w.incr_im = __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t *>(&hp_work[b].incr_im), __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_SYSTEM));
            w.simplex = __hip_atomic_load(&hp_work[b].simplex, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        w_drop = w.drop_reason;
        w_dec_len = w.dec_len;
        w_start = w.start_idx;
        w_simplex = w.simplex;
        inc_re = w.incr_re;
        inc_im = w.incr_im;
    }
    return RotStart{ live, w_drop, w_dec_len, w_start, w_simplex, inc_re, inc_im };
}

// rot_store 0: every lane stores into its own row, two phases per 16-byte store (one phase per store: the number of cache
// lines a store instruction touches, 64, set the pace: 0.24 ms against 0.11)
__global__ __launch_bounds__(64) void rot_phase_kernel(BurstWork *__restrict__ work, int n_bursts,
                                                       float2 *__restrict__ rrc_ws, const BurstWork *__restrict__ hp_work,
                                                       CfoStep cfo)
{
    const RotStart s0 = rot_phase_start(work, n_bursts, hp_work, cfo);
    if (!s0.live || s0.drop != 0) return;
    const int b = blockIdx.x * 64 + threadIdx.x;
    const int frame_len = s0.dec_len - s0.start;
    const int L = frame_len < frame_need(s0.simplex) ? frame_len : frame_need(s0.simplex);
    const float2 inc = make_float2(s0.inc_re, s0.inc_im);
    float2 ph = make_float2(1.0f, 0.0f);
    float2 *r = rrc_ws + (size_t)b * kFrameNeed;
    static_assert(kFrameNeed % 2 == 0, "rows start 16-byte aligned");
    int k = 0;
    for (; k + 2 <= L; k += 2) {
        const float2 p0 = ph;
        ph = cmul(ph, inc);
        const float2 p1 = ph;
        ph = cmul(ph, inc);
        *reinterpret_cast<float4 *>(r + k) = make_float4(p0.x, p0.y, p1.x, p1.y);
    }
    if (k < L) r[k] = ph;
}

// rot_store 1 (default): the phases leave as rows; 3: the same with plain stores under a branch instead of buffer stores
// (what told the store hazard below from an LDS problem)
template <int PITCH, bool BUF>
__global__ __launch_bounds__(64) void rot_phase_rows_kernel(BurstWork *__restrict__ work, int n_bursts,
                                                            float2 *__restrict__ rrc_ws, const BurstWork *__restrict__ hp_work,
                                                            CfoStep cfo)
{
    const RotStart s0 = rot_phase_start(work, n_bursts, hp_work, cfo);
    const bool live = s0.live;
    const int w_drop = s0.drop, w_dec_len = s0.dec_len, w_start = s0.start;
    const float inc_re = s0.inc_re, inc_im = s0.inc_im;
    // The chain: phase_{k+1} = phase_k * incr, one lane per burst.  With every lane storing into its own row a store
    // instruction touched 64 cache lines, and their number, not the multiply chain, set the pace (27 ns per step against
    // the chain's 14).  Here a tile of kRotTile steps goes through LDS -- lane b writes its phases to row b of the tile
    // -- and leaves as rows: eight lanes per burst, 16 bytes each, one store instruction = eight bursts x 128 contiguous
    // bytes.  A first version of this (two workgroup barriers and eight read-then-store passes per tile, in line with the
    // chain) took 0.44 ms instead of 0.16: the passes' LDS round trips were added to the chain.  Now the loop is skewed by
    // one tile and has no barrier: an iteration issues the sixteen LDS reads of tile t - 1, runs the chain of tile t into
    // the OTHER tile buffer, and stores tile t - 1 behind it -- a wavefront's LDS instructions complete in order, the
    // wavefront is the workgroup, so a wave barrier (an ordering point for the compiler, no instruction) is all the reads
    // need behind the writes.  Every lane runs to the wavefront's longest frame (the extra products are never stored); a
    // burst's length gates the stores of ITS row, whichever lanes make them; a pair of phases is stored whole where its
    // first element is in the frame (the second lands inside the row, behind the frame: kFrameNeed is even).
    constexpr int kRotTile = 16;                         // steps per tile: 128 bytes of a row
    constexpr int kRotPitch = PITCH;                     // float2 per LDS row (17: rows start on different banks)
    __shared__ float2 s_tile[2][64 * kRotPitch];
    __shared__ int s_len[64];
    static_assert(kFrameNeed % 2 == 0 && kRotTile % 2 == 0, "rows start 16-byte aligned, two phases per store");
    const int lane = (int)threadIdx.x;
    int L = 0;
    if (live && w_drop == 0) {
        const int frame_len = w_dec_len - w_start;
        const int need = frame_need(s0.simplex);
        L = frame_len < need ? frame_len : need;
        if (L < 0) L = 0;
    }
    s_len[lane] = L;
    int Lmax = L;
    for (int d = 32; d >= 1; d >>= 1) {
        const int o = __shfl_xor(Lmax, d);
        Lmax = o > Lmax ? o : Lmax;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // lane -> (burst of pass p, pair of phases): 8 lanes x 2 phases = one 128-byte piece of a row.  The rows leave as
    // buffer stores over this workgroup's 64 rows (fewer in the last workgroup: rows behind the batch are out of range):
    // a store whose pair lies behind its burst's frame gets an out-of-range offset and is dropped by the address
    // check -- no branch, so the stores sit in the chain's basic block.  The tile's offset is part of the VECTOR offset,
    // the scalar offset is the constant 0: with the tile's offset in an SGPR the compiler scheduled the chain's next
    // packed multiply -- which overwrites two of the store's four data registers -- directly behind the 128-bit store
    // (LLVM takes a buffer store with a register in the scalar-offset field to be free of the "store wider than 64 bits,
    // then a VALU write of its data registers" hazard), and on gfx950 the rows of every other burst arrived with the
    // NEXT step's products in them (tools/rot_store_debug.py; lanes 8-15, 24-31, ... of the store).  Without a register
    // there the hazard recognizer separates the two.
    const int part = lane & 7;
    int len_of[8], off_of[8];
    const int wg0 = (int)blockIdx.x * 64;
    const int rows = n_bursts - wg0 < 64 ? n_bursts - wg0 : 64;
    const __amdgpu_buffer_rsrc_t r_rows = __builtin_amdgcn_make_buffer_rsrc(rrc_ws + (size_t)wg0 * kFrameNeed, 0,
                                                                            rows * kFrameNeed * (int)sizeof(float2), 0x00020000);
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const int lb = p * 8 + (lane >> 3);
        len_of[p] = s_len[lb];
        off_of[p] = (lb * kFrameNeed + 2 * part) * (int)sizeof(float2);
    }
    const float2 inc = make_float2(inc_re, inc_im);
    float2 ph = make_float2(1.0f, 0.0f);
    const int n_tiles = (Lmax + kRotTile - 1) / kRotTile;
    float2 q0[8], q1[8];
    // tile u out of its buffer: sixteen LDS reads, in flight while whatever follows runs
    auto fetch = [&](int u) {
        const float2 *src = s_tile[u & 1];
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const int lb = p * 8 + (lane >> 3);
            q0[p] = src[lb * kRotPitch + 2 * part];
            q1[p] = src[lb * kRotPitch + 2 * part + 1];
        }
    };
    // the chain of tile u into its buffer
    auto chain = [&](int u) {
        float2 *dst = s_tile[u & 1];
#pragma unroll
        for (int j = 0; j < kRotTile; j++) {
            dst[lane * kRotPitch + j] = ph;
            ph = cmul(ph, inc);
        }
    };
    // tile u (fetched) to its rows
    auto store = [&](int u) {
        const int k = u * kRotTile + 2 * part;
#pragma unroll
        for (int p = 0; p < 8; p++)
            if (BUF)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(rot_u32x4, make_float4(q0[p].x, q0[p].y, q1[p].x, q1[p].y)),
                                                       r_rows, k < len_of[p] ? off_of[p] + u * kRotTile * (int)sizeof(float2) : 0x7ffffff0,
                                                       0, 0);
            else if (k < len_of[p])
                *reinterpret_cast<float4 *>(reinterpret_cast<char *>(rrc_ws + (size_t)wg0 * kFrameNeed) + off_of[p] +
                                            u * kRotTile * (int)sizeof(float2)) = make_float4(q0[p].x, q0[p].y, q1[p].x, q1[p].y);
    };
    auto order = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    if (n_tiles > 0) {
        chain(0);
        order();
        for (int t = 1; t < n_tiles; t++) {     // one basic block: the reads, the chain and the stores of an iteration
            fetch(t - 1);
            // (chain(t) and store(t - 1) written into each other: a store pass in the gap behind every other step)
            float2 *dst = s_tile[t & 1];
            const int k = (t - 1) * kRotTile + 2 * part;
#pragma unroll
            for (int j = 0; j < kRotTile; j++) {
                dst[lane * kRotPitch + j] = ph;
                ph = cmul(ph, inc);
                if (j & 1) {
                    const int p = j >> 1;
                    if (BUF)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(rot_u32x4, make_float4(q0[p].x, q0[p].y, q1[p].x, q1[p].y)),
                                                               r_rows, k < len_of[p] ? off_of[p] + (t - 1) * kRotTile * (int)sizeof(float2) : 0x7ffffff0,
                                                               0, 0);
                    else if (k < len_of[p])
                        *reinterpret_cast<float4 *>(reinterpret_cast<char *>(rrc_ws + (size_t)wg0 * kFrameNeed) + off_of[p] +
                                                    (t - 1) * kRotTile * (int)sizeof(float2)) = make_float4(q0[p].x, q0[p].y, q1[p].x, q1[p].y);
                }
            }
            order();
        }
        fetch(n_tiles - 1);
        store(n_tiles - 1);
    }
}

// ---------------------------------------------------------------------------
// post2: one workgroup per burst.  RT: compile-time RRC tap count (51), 0 = the runtime value.
// ---------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(kPostThreads) void downmix_post2_kernel(
    BurstWork *__restrict__ work, const float2 *__restrict__ lpf,
    const float *__restrict__ rrc_taps, int rrc_ntaps_rt, const float2 *__restrict__ tw2048,
    const float2 *__restrict__ dl_fft, const float2 *__restrict__ ul_fft, int dl_len, int ul_len,
    float sps, float2 *__restrict__ rrc_ws, float2 *__restrict__ frames, int order)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // the rotated frame (steps 5-6) and the three correlation buffers (step 7) are never live together: they share the
    // same 48 KB, so three workgroups fit a CU and all bursts of a chunk are resident at once
    float2 *rot = reinterpret_cast<float2 *>(smem_raw);          // kFrameNeed (<= 3 * kCorrN)
    float2 *fa = rot;                                            // kCorrN
    float2 *fd = fa + kCorrN;                                    // kCorrN
    float2 *fu = fd + kCorrN;                                    // kCorrN
    static_assert(kFrameNeed <= 3 * kCorrN, "the frame buffer aliases the correlation buffers");
    float *redf = reinterpret_cast<float *>(fu + kCorrN);
    int *redi = reinterpret_cast<int *>(redf + 4);
    __builtin_amdgcn_s_setprio(2);      // latency-bound, next to the decimator of the next chunk
    const int tid = threadIdx.x;
    BurstWork &w = work[blockIdx.x];
    if (w.drop_reason != 0) return;
    const int start = w.start_idx;
    const int frame_len = w.dec_len - start;
    const int L = frame_len < frame_need(w.simplex) ? frame_len : frame_need(w.simplex);
    const float2 *x = lpf + (size_t)w.dec_off + start;
    float2 *r = rrc_ws + (size_t)blockIdx.x * kFrameNeed;

    // step 5 (burst_downmix.c:713-720): the phase sequence comes from rot_phase_kernel (in r)
    const int rrc_ntaps = RT ? RT : rrc_ntaps_rt;
    for (int k = tid; k < L; k += kPostThreads) rot[k] = cmul(x[k], r[k]);
    __syncthreads();

    // step 6 (burst_downmix.c:723-734): centred 51-tap RRC over the zero-padded frame
    const int half = (rrc_ntaps - 1) / 2;
    // option "fir_order" 1: avx2_fir_ccf over frame_len outputs (burst_downmix.c:733) -- a fused multiply-add per tap but
    // for the last frame_len % 4 outputs
    const int rrc_vec = order ? (frame_len & ~3) : 0;
    for (int i = tid; i < L; i += kPostThreads) {
        float ar = 0.0f, ai = 0.0f;
        if (i < rrc_vec) {
            if (RT && i >= half && i + (RT - 1 - half) < L) {
#pragma unroll
                for (int k = 0; k < (RT ? RT : 1); k++) {
                    const float2 v = rot[i + k - half];
                    const float t = rrc_taps[k];
                    ar = __builtin_fmaf(t, v.x, ar);
                    ai = __builtin_fmaf(t, v.y, ai);
                }
            } else {
                for (int k = 0; k < rrc_ntaps; k++) {
                    const int j = i + k - half;
                    const float2 v = (j >= 0 && j < L) ? rot[j] : make_float2(0.0f, 0.0f);
                    const float t = rrc_taps[k];
                    ar = __builtin_fmaf(t, v.x, ar);
                    ai = __builtin_fmaf(t, v.y, ai);
                }
            }
        } else if (RT && i >= half && i + (RT - 1 - half) < L) {
            // every tap inside the frame: no bounds checks, unrolled
#pragma unroll
            for (int k = 0; k < (RT ? RT : 1); k++) {
                const float2 v = rot[i + k - half];
                const float t = rrc_taps[k];
                ar += t * v.x;
                ai += t * v.y;
            }
        } else {
            for (int k = 0; k < rrc_ntaps; k++) {
                const int j = i + k - half;
                // j >= L only for outputs that can never reach the frame (kFrameNeed margin)
                const float2 v = (j >= 0 && j < L) ? rot[j] : make_float2(0.0f, 0.0f);
                const float t = rrc_taps[k];
                ar += t * v.x;
                ai += t * v.y;
            }
        }
        r[i] = make_float2(ar, ai);
    }
    __syncthreads();

    // step 7 (burst_downmix.c:539-639)
    const int sl = kSyncSearch < frame_len ? kSyncSearch : frame_len;
    // (the forward transform behind its first stage: the inputs 1024 .. 2047 are zero, so the stage's butterflies a + 0,
    // a - 0 are copies -- positions 2 m and 2 m + 1 hold x[bitrev10(m)]; stages 2..11 two per barrier)
    static_assert(kSyncSearch <= kCorrN / 2 && kCorrN == 2048, "the pruned forward transform");
    for (int m = tid; m < kCorrN / 2; m += kPostThreads) {
        const int i = (int)bitrev((unsigned)m, 10);
        const float2 v = i < sl ? r[i] : make_float2(0.0f, 0.0f);
        *reinterpret_cast<float4 *>(fa + 2 * m) = make_float4(v.x, v.y, v.x, v.y);
    }
    __syncthreads();
    fft_lds_radix2x2<11, kPostThreads, -1, 2>(fa, tw2048);
    for (int i = tid; i < kCorrN; i += kPostThreads) {
        const float2 v = fa[i];
        const unsigned br = bitrev((unsigned)i, 11);
        fd[br] = cmul(v, dl_fft[i]);
        fu[br] = cmul(v, ul_fft[i]);
    }
    __syncthreads();
    // (both backward transforms side by side: they share their barriers)
    fft_lds_radix2x2<11, kPostThreads, +1, 1, 2>(fd, tw2048, kCorrN);
    float mdl = 0.0f, mul = 0.0f;
    int odl = 0, oul = 0;
    for (int i = tid; i < sl; i += kPostThreads) {
        const float a = mag2(fd[i]);
        if (a > mdl) { mdl = a; odl = i; }
        const float b = mag2(fu[i]);
        if (b > mul) { mul = b; oul = i; }
    }
    block_argmax(mdl, odl, redf, redi);
    block_argmax(mul, oul, redf, redi);

    int direction, off, slen;
    const float2 *res;
    if (mdl >= mul) { direction = 1; off = odl; res = fd; slen = dl_len; }
    else            { direction = 2; off = oul; res = fu; slen = ul_len; }
    const float2 peak = res[off];
    float corr = 0.0f;
    if (off > 0 && off < sl - 1)
        corr = parabolic(mag2(res[off - 1]), mag2(res[off]), mag2(res[off + 1]));
    const int pre_syms = direction == 1 ? 16 : 32;                 // burst_downmix.c:633-634
    const int uw_start = off - slen + 1 + (int)(pre_syms * sps);

    int drop = 0, ext = 0;
    if (uw_start < 0 || uw_start >= frame_len) {
        drop = 4;                                                  // burst_downmix.c:744-747
    } else {
        const int max_len = (int)((w.simplex ? 444 : 191) * sps);  // iridium.h:23-27
        const int min_len = (int)((w.simplex ? 80 : 131) * sps);
        const int avail = frame_len - uw_start;
        if (avail < min_len) drop = 5;                             // burst_downmix.c:773-776
        else ext = avail < max_len ? avail : max_len;
    }
    if (!drop) {
        // step 8 (burst_downmix.c:750-760): constant phase conj(peak/|peak|), incr = 1
        const float mag = cabs_f(peak);
        const float2 pc = mag > 0 ? make_float2(peak.x / mag, -(peak.y / mag)) : make_float2(1.0f, 0.0f);
        float2 *out = frames + (size_t)blockIdx.x * kMaxFrameSamples;
        for (int k = tid; k < ext; k += kPostThreads) out[k] = cmul(r[uw_start + k], pc);
    }
    if (tid == 0) {
        w.direction = direction;
        w.uw_start = uw_start;
        w.uw_corr = corr;
        w.corr_re = peak.x;
        w.corr_im = peak.y;
        w.num_samples = ext;
        w.drop_reason = drop;
    }
}

// libm_port.hpp on the device for arbitrary arguments (tests/test_gpu_libm.py, tools/check_sincosf_gpu.hip)
__global__ void sincosf_probe_kernel(const float *__restrict__ x, size_t n, float *__restrict__ re, float *__restrict__ im)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float r = 0.0f, q = 0.0f;
        if (libm_cexpf_i<true>(x[i], &r, &q) != 0) r = q = __uint_as_float(0x7fc00000u);
        re[i] = r;
        im[i] = q;
    }
}

int launch_sincosf_probe(const float *x, size_t n, float *re, float *im, hipStream_t stream)
{
    if (!n) return 0;
    hipLaunchKernelGGL(sincosf_probe_kernel, dim3(2048), dim3(256), 0, stream, x, n, re, im);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_downmix_post2(BurstWork *work, int n_bursts, const float2 *lpf,
                         const float *rrc_taps, int rrc_ntaps, const float2 *tw2048,
                         const float2 *dl_fft, const float2 *ul_fft, int dl_len, int ul_len,
                         float sps, float2 *rrc_ws, float2 *frames, const BurstWork *hp_work, const CfoStep &cfo,
                         hipStream_t stream)
{
    if (n_bursts <= 0) return 0;
    const size_t lds = sizeof(float2) * (3 * kCorrN) + 64;
    if (g_rot_store == 1)
        hipLaunchKernelGGL((rot_phase_rows_kernel<17, true>), dim3((n_bursts + 63) / 64), dim3(64), 0, stream, work, n_bursts, rrc_ws, hp_work, cfo);
    else if (g_rot_store == 3)
        hipLaunchKernelGGL((rot_phase_rows_kernel<17, false>), dim3((n_bursts + 63) / 64), dim3(64), 0, stream, work, n_bursts, rrc_ws, hp_work, cfo);
    else
        hipLaunchKernelGGL(rot_phase_kernel, dim3((n_bursts + 63) / 64), dim3(64), 0, stream, work, n_bursts, rrc_ws, hp_work, cfo);
    if (rrc_ntaps == 51 && !g_post_generic) {
        (void)hipFuncSetAttribute((const void *)downmix_post2_kernel<51>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((downmix_post2_kernel<51>), dim3(n_bursts), dim3(kPostThreads), lds, stream, work, lpf,
                           rrc_taps, rrc_ntaps, tw2048, dl_fft, ul_fft, dl_len, ul_len, sps,
                           rrc_ws, frames, g_fir_order);
    } else {
        (void)hipFuncSetAttribute((const void *)downmix_post2_kernel<0>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((downmix_post2_kernel<0>), dim3(n_bursts), dim3(kPostThreads), lds, stream, work, lpf,
                           rrc_taps, rrc_ntaps, tw2048, dl_fft, ul_fft, dl_len, ul_len, sps,
                           rrc_ws, frames, g_fir_order);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace irdm
