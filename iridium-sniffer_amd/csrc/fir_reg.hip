// fir_reg.hip -- step 1+2 of stage B (burst_downmix.c:663-672, :417-437; rotator.h:36-46; simd_generic.c:86-96):
// coarse rotate fused into the 801-tap /M decimator, with the rotated samples held in REGISTERS.
//
//   out[q] = sum_{k<801} t[k] * y[q*M + k],  k ascending, product and sum rounded separately (no FMA),
//   y[s]   = x[s] * phase_s,  phase_{s+1} = phase_s * incr  (float recurrence, checkpoint every 16 samples)
//
// Write k = r*M + p: output q meets "column" c = q + r (the M samples c*M .. c*M + M-1) in "row" r of the taps.
// A lane keeps ONE column per chain in VGPRs for its whole life; what travels is the accumulator:
//
//   lane l of chain j holds column  c = q0 + 128*d + 64*j + l  (d = double block of the strip)
//   in row r it works on output     q = c - r
//   between rows every accumulator moves one lane to the right (v_mov_b32_dpp wave_shr:1); chain A's lane 63 feeds
//   chain B's lane 0, chain B's lane 63 enters a delay line (`carry`, shifted along) and comes out NR + 1 rows later
//   as lane 0 of chain A in the next double block.
//
// Every lane of a wavefront is in the same row at the same time, so the taps are wavefront-uniform: SGPR operands
// straight from the scalar cache.  After row NR (the last, partial row of taps) lane l holds the finished output
// c - NR.  The order of the 801 multiply-adds of an output is exactly k ascending, on one accumulator.
//
// Why registers: the column-major LDS tile of the previous kernel (fir_decimate_kernel_w) stores M*8 bytes per output in
// flight, so the LDS holds at most two output chains per SIMD -- too few independent chains to cover the latency of the
// dependent add (tools/ubench/valu_issue.hip, profiles/r3_valu_issue.txt: one chain per lane at 1.5 wavefronts per SIMD
// issues a tap in 8.3-9.5 ns; two chains per lane x two wavefronts 6.7 ns).  The register file is 3.2x the LDS: two
// chains per lane and two wavefronts per SIMD (4 chains per SIMD) fit in 2 x 256 VGPRs.  No LDS, no barrier, no
// staging pass (the rotation happens where the samples land), every sample is read once per strip (strips of a burst
// overlap by NR columns only), and the LDS is left to the detector's FFT (K1), which runs beside this kernel.
#include <cstdio>
#include "common.hpp"
#include "types.hpp"
#include "kernels.hpp"
#include "burst_src.hpp"

// (A/B builds only: -DIRDM_FIR_KCLK=0 compiles the kernel-clock stamps out of the decimator)
#ifndef IRDM_FIR_KCLK
#define IRDM_FIR_KCLK 1
#endif

namespace irdm {

#include "fir_mac.inc"

int g_fir_strip = 3;           // double blocks (128 columns) per strip: a strip yields 128*g_fir_strip - NR outputs
int g_chain_cus = 0;           // CUs the per-burst chains' streams may use (0: all of them)
int g_fir_grid = -1;           // > 0: at most this many single-wavefront workgroups in flight, each walking strips; 0: one per
                               // strip; -1 (default): fir_decimate_kernel_f seven per CU, fir_decimate_kernel_r one per strip
int g_fir_slice = 0;           // > 0: strips per launch (the chunk's strips as several launches); 0: one launch

template <int M>
struct FirR {
    static constexpr int NR = kFirTaps / M;                  // full rows of M taps
    static constexpr int REM = kFirTaps - NR * M;            // taps of the last, partial row (>= 1)
    static_assert(M % 8 == 0 && NR < 63 && REM >= 1, "geometry");
};

int fir_reg_supported(int decim) { return decim == 40 || decim == 48; }

// outputs per strip (= FirTile unit of the host's tile count)
int fir_reg_tile_out(int decim)
{
    int s = g_fir_strip < 1 ? 1 : (g_fir_strip > 64 ? 64 : g_fir_strip);
    return 128 * s - kFirTaps / decim;
}

// 8 consecutive samples (16-byte aligned in every format), converted exactly as load_iq does
template <int FMT>
__device__ __forceinline__ void load_piece(const void *__restrict__ base, size_t idx, v2f *x)
{
    if (FMT == 2) {
        // (ordinary loads: a lane's eight 16-byte loads share one 128-byte line, and the line has to survive in the
        // cache between them -- with non-temporal loads every piece fetched its line again: 0.45 -> 0.97 ms in run)
        const v4f *g = reinterpret_cast<const v4f *>(reinterpret_cast<const float2 *>(base) + idx);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const v4f v = g[u];
            x[2 * u] = v2f{ v.x, v.y };
            x[2 * u + 1] = v2f{ v.z, v.w };
        }
    } else if (FMT == 1) {
        const int4 *g = reinterpret_cast<const int4 *>(reinterpret_cast<const short2 *>(base) + idx);   // 4 samples per 16 B
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int4 v = g[u];
            const int w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const short re = (short)(w[k] & 0xffff), im = (short)(w[k] >> 16);
                x[4 * u + k] = v2f{ (float)(re >> 8) / 128.0f, (float)(im >> 8) / 128.0f };
            }
        }
    } else {
        const int4 v = *reinterpret_cast<const int4 *>(reinterpret_cast<const char2 *>(base) + idx);    // 8 samples per 16 B
        const int w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const signed char r0 = (signed char)(w[k] & 0xff), i0 = (signed char)((w[k] >> 8) & 0xff);
            const signed char r1 = (signed char)((w[k] >> 16) & 0xff), i1 = (signed char)((w[k] >> 24) & 0xff);
            x[2 * k] = v2f{ (float)r0 / 128.0f, (float)i0 / 128.0f };
            x[2 * k + 1] = v2f{ (float)r1 / 128.0f, (float)i1 / 128.0f };
        }
    }
}

__device__ __forceinline__ v2f cmul2(v2f x, v2f y)
{
    const float2 r = cmul(make_float2(x.x, x.y), make_float2(y.x, y.y));
    return v2f{ r.x, r.y };
}

__device__ __forceinline__ float lane_get(float v, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// v_writelane_b32: lane 0 of v <- the wavefront-uniform s
__device__ __forceinline__ float lane_set0(float v, float s)
{
    float r;
    asm("v_writelane_b32 %0, %1, 0" : "=v"(r) : "s"(s), "0"(v));
    return r;
}

// lane i <- lane i-1 (lane 0 keeps its value: it is overwritten by the caller)
__device__ __forceinline__ float lane_shr1(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}

template <int M, int FMT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void fir_decimate_kernel_r(
    SampleSource src, const FirGeom *__restrict__ geom, const float *__restrict__ taps,
    const float2 *__restrict__ rot_table, float2 *__restrict__ dec, int n_tiles, unsigned long long *__restrict__ kclk)
{
    using R = FirR<M>;
    constexpr int NR = R::NR, REM = R::REM;
    const int lane = threadIdx.x;
    if (IRDM_FIR_KCLK) kclk_enter(kclk);
    // (a grid smaller than the strip count -- option fir_grid -- walks the strips with the grid's stride)
#pragma unroll 1
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const FirGeom g = geom[tile];
    const int n_cols = g.n_out + NR;                 // columns that feed a stored output
    const int n_blk = (n_cols + 127) >> 7;
    const v2f inc = { g.inc_re, g.inc_im };
    const uint64_t *taps64 = reinterpret_cast<const uint64_t *>(taps);
    v2f carry = { 0.0f, 0.0f };

#pragma unroll 1
    for (int d = 0; d < n_blk; d++) {
        v2f y[2][M];
        // ---- fetch: column -> registers (ringbuf_extract semantics, burst_detect.c:401-422: a sample at or past
        // avail_end reads the ring slot as the reference found it: one reference ring length earlier, or zero) ----
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int cr = 128 * d + 64 * j + lane;
            const bool needed = cr < n_cols;
            const int k0 = needed ? cr * M : 0;                  // sample offset from the strip's first sample
            const uint64_t a0 = g.a_tile + (uint64_t)k0;
            uint64_t rp = g.ring_pos + (uint64_t)k0;             // a0 mod ring_len
            if (rp >= src.ring_len) rp -= src.ring_len;
            const bool in_chunk = a0 >= src.chunk_start;
            // simple column: all M samples written, on one side of the chunk start, no wrap of the ring inside
            const bool simple = a0 + M <= g.avail_end && (in_chunk || (a0 + M <= src.chunk_start && rp + M <= src.ring_len));
            if (__builtin_amdgcn_ballot_w64(needed && !simple) == 0) {
                const void *base = in_chunk ? src.chunk : src.ring;
                const size_t idx = !needed ? 0 : in_chunk ? (size_t)(a0 - src.chunk_start) : (size_t)rp;
#pragma unroll
                for (int i = 0; i < M / 8; i++) load_piece<FMT>(base, idx + 8 * i, &y[j][8 * i]);
            } else {
                // pieces of 8 samples: every boundary (avail_end up to a ragged stream end, chunk start, ring wrap,
                // reference ring length) is a multiple of 8 samples except a ragged avail_end, so a piece has ONE
                // written source and ONE stale source, selected per sample
                uint64_t rs = g.stale_pos + (uint64_t)k0;        // (a0 - ref_ring) mod ring_len
                if (rs >= src.ring_len) rs -= src.ring_len;
#pragma unroll
                for (int i = 0; i < M / 8; i++) {
                    const uint64_t a = a0 + 8 * i;
                    v2f xn[8], xs[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) xn[u] = xs[u] = v2f{ 0.0f, 0.0f };
                    if (needed && a < g.avail_end) {
                        uint64_t p = rp + 8 * i;
                        if (p >= src.ring_len) p -= src.ring_len;
                        if (a >= src.chunk_start) load_piece<FMT>(src.chunk, (size_t)(a - src.chunk_start), xn);
                        else load_piece<FMT>(src.ring, (size_t)p, xn);
                    }
                    if (needed && a + 8 > g.avail_end && a >= src.ref_ring) {
                        const uint64_t as = a - src.ref_ring;
                        uint64_t p = rs + 8 * i;
                        if (p >= src.ring_len) p -= src.ring_len;
                        if (as >= src.chunk_start) load_piece<FMT>(src.chunk, (size_t)(as - src.chunk_start), xs);
                        else load_piece<FMT>(src.ring, (size_t)p, xs);
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) y[j][8 * i + u] = (a + u < g.avail_end) ? xn[u] : xs[u];
                }
            }
        }
        // ---- rotate in place (rotator.h:38-39): phase restored from the checkpoint table (every kRotSeg samples),
        // a column that starts 8 samples behind a checkpoint runs the recurrence 8 steps first ----
        {
            v2f ph[2];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int cr = 128 * d + 64 * j + lane;
                const int k0 = cr < n_cols ? cr * M : 0;
                const float2 c = rot_table[fir_ck(g, k0 / kRotSeg)];
                ph[j] = v2f{ c.x, c.y };
            }
            if (M % kRotSeg != 0) {
                v2f pw[2] = { ph[0], ph[1] };
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    pw[0] = cmul2(pw[0], inc);
                    pw[1] = cmul2(pw[1], inc);
                }
                // (64*j*M and 128*d*M are multiples of 16: the parity depends on the lane only)
                if ((lane * M) & 8) {
                    ph[0] = pw[0];
                    ph[1] = pw[1];
                }
            }
#pragma unroll
            for (int p = 0; p < M; p++) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    y[j][p] = cmul2(y[j][p], ph[j]);        // out[i] = in[i] * phase (rotator.h:38)
                    ph[j] = cmul2(ph[j], inc);              // phase *= incr          (rotator.h:39)
                }
            }
        }
        // ---- the taps ----
        v2f acc[2] = { v2f{ 0.0f, 0.0f }, v2f{ 0.0f, 0.0f } };
        auto shift = [&]() {
            const float ax = lane_get(acc[0].x, 63), ay = lane_get(acc[0].y, 63);
            const float bx = lane_get(acc[1].x, 63), by = lane_get(acc[1].y, 63);
            // `carry` is a delay line: it moves with the accumulators, so what chain B's lane 63 left in lane 0 one double
            // block ago (NR + 1 shifts ago: rows r+1 .. NR there, rows 0 .. r here) is in lane NR + 1 now
            carry.x = lane_shr1(carry.x);
            carry.y = lane_shr1(carry.y);
            const float cx = lane_get(carry.x, NR + 1), cy = lane_get(carry.y, NR + 1);
            acc[0].x = lane_set0(lane_shr1(acc[0].x), cx);
            acc[0].y = lane_set0(lane_shr1(acc[0].y), cy);
            acc[1].x = lane_set0(lane_shr1(acc[1].x), ax);
            acc[1].y = lane_set0(lane_shr1(acc[1].y), ay);
            carry.x = lane_set0(carry.x, bx);
            carry.y = lane_set0(carry.y, by);
        };
        // The taps of a row reach the SGPRs half a row ahead (scalar loads return out of order, so a wait is for
        // everything outstanding: each half's request is issued right behind the first multiply-add group of the
        // half before it, and is waited for a whole half -- 80 to 96 packed instructions -- later).
        constexpr int H = M / 2;                             // taps per half row (20 / 24)
        uint64_t ta[H / 2], tb[H / 2];
        auto request = [&](uint64_t (&t)[H / 2], int tap0) {
#pragma unroll
            for (int i = 0; i < H / 2; i++) t[i] = taps64[tap0 / 2 + i];
        };
        auto mac_head = [&](const uint64_t *t, int p0) { fir_mac2x8(acc[0], acc[1], &y[0][p0], &y[1][p0], t); };
        auto mac_rest = [&](const uint64_t *t, int p0) {
            fir_mac2x8(acc[0], acc[1], &y[0][p0 + 8], &y[1][p0 + 8], t + 4);
            if constexpr (H == 24) fir_mac2x8(acc[0], acc[1], &y[0][p0 + 16], &y[1][p0 + 16], t + 8);
            else fir_mac2x4(acc[0], acc[1], &y[0][p0 + 16], &y[1][p0 + 16], t + 8);
        };
        static_assert(H == 20 || H == 24, "half rows of 8 + 8 + 4 or 8 + 8 + 8 taps");
        request(ta, 0);
#pragma unroll 1
        for (int r = 0; r < NR; r++) {
            shift();                                         // (row 0: zeros move)
            mac_head(ta, 0);
            __builtin_amdgcn_sched_barrier(0);
            request(tb, r * M + H);
            __builtin_amdgcn_sched_barrier(0);
            mac_rest(ta, 0);
            mac_head(tb, H);
            __builtin_amdgcn_sched_barrier(0);
            // the next row's first half; behind the last full row: the partial row's, if it has that many taps
            request(ta, (r + 1 < NR || REM >= H) ? (r + 1) * M : 0);
            __builtin_amdgcn_sched_barrier(0);
            mac_rest(tb, H);
        }
        shift();
        {
            constexpr int P0 = REM >= H ? H : 0;             // taps of the partial row already in `ta`
            if constexpr (REM >= H) {
                mac_head(ta, 0);
                mac_rest(ta, 0);
            }
            const uint64_t *t = taps64 + (NR * M + P0) / 2;
#pragma unroll
            for (int gq = 0; gq < (REM - P0) / 8; gq++)
                fir_mac2x8(acc[0], acc[1], &y[0][P0 + 8 * gq], &y[1][P0 + 8 * gq], t + 4 * gq);
#pragma unroll
            for (int p = P0 + (REM - P0) / 8 * 8; p < REM; p++) {
                const float tq = taps[NR * M + p];
                acc[0] = acc[0] + y[0][p] * tq;
                acc[1] = acc[1] + y[1][p] * tq;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int q = 128 * d + 64 * j + lane - NR;
            if (q >= 0 && q < g.n_out) dec[g.out_base + q] = make_float2(acc[j].x, acc[j].y);
        }
    }
    }
    if (IRDM_FIR_KCLK) kclk_leave(kclk);
}

// ---------------------------------------------------------------------------------------------------------------
// The decimator in the order of the reference's AVX2 kernel (option fir_order 1): simd_avx2.c:62-108, avx2_fir_ccf_dec
// -- what simd_init() selects on every x86 host with AVX2 + FMA unless --no-simd is given (simd_generic.c:33-57).
//
//   acc_j = fma(t[4m + j], y[qM + 4m + j], acc_j)   j = 0..3, m = 0..199      (four accumulators, FUSED multiply-adds)
//   out   = ((acc_0 + acc_2) + (acc_1 + acc_3)) + t[800] * y[qM + 800]        (horizontal sum; the one tap left over
//                                                                               by the vector loop: product and sum
//                                                                               rounded on their own, -std=c99)
//
// One v_pk_fma_f32 per tap and output instead of v_pk_mul_f32 + v_pk_add_f32, and four independent chains per output
// instead of one: half the arithmetic instructions of fir_decimate_kernel_r, whose 801 separately rounded multiply-adds
// on ONE accumulator are what kept it at the packed-fp32 issue limit.  Same idea as above -- a lane keeps
// its columns of M rotated samples in VGPRs, the accumulators travel -- laid out for four accumulators per output:
//
//   lane l holds columns 2l (A) and 2l + 1 (B) of the strip: 2M consecutive samples, rotated by ONE run of the phase
//   recurrence;  in row r column c works on output q = c - r;  between rows an output's accumulators move from column c
//   to c + 1:  A -> B is the SAME lane (no instruction: the two accumulator sets swap names, rows are unrolled in
//   pairs), B -> A of the next lane is one DPP shift per register -- 8 shifts per row for the 2 x 4 accumulators of a
//   lane, no lane-63 hand-offs and no delay line: a strip is ONE block of 128 columns (128 - NR outputs) and the strips
//   of a burst overlap by NR columns (16 - 19 % of the samples read twice; the kernel it replaces carried accumulators
//   from block to block through a delay line, 18 instructions per row).
//   Per row and lane: 2M fused multiply-adds + 8 shifts (M = 40: 88 instructions; fir_decimate_kernel_r: 338).
// The taps are wavefront-uniform SGPR operands as above (csrc/fir_fma.inc, generated).
// ---------------------------------------------------------------------------------------------------------------
#include "fir_fma.inc"

int fir_fma_tile_out(int decim) { return 128 - kFirTaps / decim; }

// column `cr` of the strip -> y[0..M-1] (ringbuf_extract semantics as in fir_decimate_kernel_r: a sample at or past
// avail_end reads the ring slot as the reference found it, one reference ring length earlier, or zero)
template <int M, int FMT>
__device__ __forceinline__ void fir_fetch_column(const SampleSource &src, const FirGeom &g, int cr, int n_cols, v2f *y)
{
    const bool needed = cr < n_cols;
    const int k0 = needed ? cr * M : 0;                  // sample offset from the strip's first sample
    const uint64_t a0 = g.a_tile + (uint64_t)k0;
    uint64_t rp = g.ring_pos + (uint64_t)k0;             // a0 mod ring_len
    if (rp >= src.ring_len) rp -= src.ring_len;
    const bool in_chunk = a0 >= src.chunk_start;
    // simple column: all M samples written, on one side of the chunk start, no wrap of the ring inside
    const bool simple = a0 + M <= g.avail_end && (in_chunk || (a0 + M <= src.chunk_start && rp + M <= src.ring_len));
    if (__builtin_amdgcn_ballot_w64(needed && !simple) == 0) {
        const void *base = in_chunk ? src.chunk : src.ring;
        const size_t idx = !needed ? 0 : in_chunk ? (size_t)(a0 - src.chunk_start) : (size_t)rp;
#pragma unroll
        for (int i = 0; i < M / 8; i++) load_piece<FMT>(base, idx + 8 * i, &y[8 * i]);
    } else {
        uint64_t rs = g.stale_pos + (uint64_t)k0;        // (a0 - ref_ring) mod ring_len
        if (rs >= src.ring_len) rs -= src.ring_len;
#pragma unroll
        for (int i = 0; i < M / 8; i++) {
            const uint64_t a = a0 + 8 * i;
            v2f xn[8], xs[8];
#pragma unroll
            for (int u = 0; u < 8; u++) xn[u] = xs[u] = v2f{ 0.0f, 0.0f };
            if (needed && a < g.avail_end) {
                uint64_t p = rp + 8 * i;
                if (p >= src.ring_len) p -= src.ring_len;
                if (a >= src.chunk_start) load_piece<FMT>(src.chunk, (size_t)(a - src.chunk_start), xn);
                else load_piece<FMT>(src.ring, (size_t)p, xn);
            }
            if (needed && a + 8 > g.avail_end && a >= src.ref_ring) {
                const uint64_t as = a - src.ref_ring;
                uint64_t p = rs + 8 * i;
                if (p >= src.ring_len) p -= src.ring_len;
                if (as >= src.chunk_start) load_piece<FMT>(src.chunk, (size_t)(as - src.chunk_start), xs);
                else load_piece<FMT>(src.ring, (size_t)p, xs);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) y[8 * i + u] = (a + u < g.avail_end) ? xn[u] : xs[u];
        }
    }
}

template <int M, int FMT, bool CLAIM = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void fir_decimate_kernel_f(
    SampleSource src, const FirGeom *__restrict__ geom, const float *__restrict__ taps,
    const float2 *__restrict__ rot_table, float2 *__restrict__ dec, int n_tiles, unsigned long long *__restrict__ kclk,
    unsigned *__restrict__ next_tile)
{
    using R = FirR<M>;
    constexpr int NR = R::NR, REM = R::REM;
    constexpr int VT = REM - 1;                              // taps of the partial row the vector loop takes
    static_assert(NR % 2 == 0, "rows are unrolled in pairs (the accumulator sets swap names)");
    static_assert(M % 4 == 0 && VT % 8 == 0 && (kFirTaps - 1) % 4 == 0, "tap k -> accumulator k % 4 = (k % M) % 4; one tap left over");
    const int lane = threadIdx.x;
    kclk_enter(kclk);
    // Strips: fixed shares (tile += gridDim.x), or -- option fir_claim, off: measured slower -- the workgroup's first strip
    // its index and every further one CLAIMED from a counter (next_tile, zeroed by fir_geom_kernel): a resident grid with
    // fixed shares ends when its slowest wavefront does, and in run the wavefronts that share a SIMD with the lane-per-burst
    // kernels of the other chains fall behind (0.46 ms in run against 0.33 alone at four contexts); the claim for the strip
    // after this one is made before this one's work.
    // (a template parameter: the claim's few registers cost the fixed-share kernel a 20-byte spill when both lived in one body)
    const bool claim = CLAIM && next_tile != nullptr && (int)gridDim.x < n_tiles;
#pragma unroll 1
    for (int tile = blockIdx.x; tile < n_tiles;) {
        int tile_next = tile + (int)gridDim.x;
        if (CLAIM && claim) {
            unsigned c = 0;
            if (lane == 0) c = atomicAdd(next_tile, 1u);
            tile_next = (int)gridDim.x + (int)__builtin_amdgcn_readfirstlane((int)c);
        }
        const FirGeom g = geom[tile];
        tile = tile_next;
        const int n_cols = g.n_out + NR;                     // columns that feed a stored output (<= 128)
        const v2f inc = { g.inc_re, g.inc_im };
        const uint64_t *taps64 = reinterpret_cast<const uint64_t *>(taps);
        v2f y[2][M];
#pragma unroll
        for (int j = 0; j < 2; j++) fir_fetch_column<M, FMT>(src, g, 2 * lane + j, n_cols, y[j]);
        // ---- rotate in place (rotator.h:38-39): the phase restored from the checkpoint table (every kRotSeg samples);
        // column 2l starts on a checkpoint, column 2l + 1 on one too (M % 16 == 0) or 8 samples behind one ----
        {
            v2f ph[2];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int cr = 2 * lane + j;
                const int k0 = cr < n_cols ? cr * M : 0;
                const float2 c = rot_table[fir_ck(g, k0 / kRotSeg)];
                ph[j] = v2f{ c.x, c.y };
            }
            if (M % kRotSeg != 0) {
                static_assert(M % kRotSeg == 0 || M % kRotSeg == 8, "a column starts on a checkpoint or 8 samples behind one");
                // (2l M is a multiple of 16; (2l + 1) M = 8 mod 16 -- unless the column is not needed: then k0 = 0 and the
                // phase is not used)
                v2f pw = ph[1];
#pragma unroll
                for (int u = 0; u < 8; u++) pw = cmul_pk(pw, inc);
                if (2 * lane + 1 < n_cols) ph[1] = pw;
            }
#pragma unroll
            for (int p = 0; p < M; p++) {
                v2f r0, r1, n0, n1;
                cmul_pk2(r0, y[0][p], ph[0], r1, y[1][p], ph[1]);       // out[i] = in[i] * phase (rotator.h:38)
                cmul_pk2(n0, ph[0], inc, n1, ph[1], inc);               // phase *= incr          (rotator.h:39)
                y[0][p] = r0;
                y[1][p] = r1;
                ph[0] = n0;
                ph[1] = n1;
            }
        }
        // ---- the taps ----
        const v2f zero = { 0.0f, 0.0f };
        v2f X[4] = { zero, zero, zero, zero }, Y[4] = { zero, zero, zero, zero };
        constexpr int H = M / 2;                             // taps per half row (20 / 24)
        static_assert(H == 20 || H == 24, "half rows of 8 + 8 + 4 or 8 + 8 + 8 taps");
        uint64_t ta[H / 2], tb[H / 2];
        auto request = [&](uint64_t (&t)[H / 2], int tap0) {
#pragma unroll
            for (int i = 0; i < H / 2; i++) t[i] = taps64[tap0 / 2 + i];
        };
        // one row: xa = the accumulators that are with column A in this row, xb = with column B; the next half row's taps
        // are requested behind the first group of the half before it (scalar loads return out of order)
        auto half_rest = [&](v2f (&xa)[4], v2f (&xb)[4], const uint64_t *t, int p0) {
            fir_fma2x8(xa, xb, &y[0][p0 + 8], &y[1][p0 + 8], t + 4);
            if constexpr (H == 24) fir_fma2x8(xa, xb, &y[0][p0 + 16], &y[1][p0 + 16], t + 8);
            else fir_fma2x4(xa, xb, &y[0][p0 + 16], &y[1][p0 + 16], t + 8);
        };
        auto row = [&](v2f (&xa)[4], v2f (&xb)[4], int r, int next_tap0) {
            fir_fma2x8(xa, xb, &y[0][0], &y[1][0], ta);
            __builtin_amdgcn_sched_barrier(0);
            request(tb, r * M + H);
            __builtin_amdgcn_sched_barrier(0);
            half_rest(xa, xb, ta, 0);
            fir_fma2x8(xa, xb, &y[0][H], &y[1][H], tb);
            __builtin_amdgcn_sched_barrier(0);
            request(ta, next_tap0);
            __builtin_amdgcn_sched_barrier(0);
            half_rest(xa, xb, tb, H);
        };
        auto shift = [&](v2f (&x)[4]) {                      // the B accumulators of lane l - 1 become the A accumulators of lane l
#pragma unroll
            for (int j = 0; j < 4; j++) {
                x[j].x = lane_shr1(x[j].x);
                x[j].y = lane_shr1(x[j].y);
            }
        };
        request(ta, 0);
#pragma unroll 1
        for (int rp = 0; rp < NR / 2; rp++) {
            row(X, Y, 2 * rp, (2 * rp + 1) * M);             // X with column A, Y with column B
            shift(Y);                                        // -> Y with column A, X with column B
            // (behind the last full row: the partial row's first taps, if the vector loop takes any of it)
            row(Y, X, 2 * rp + 1, (2 * rp + 2 < NR || VT >= H) ? (2 * rp + 2) * M : 0);
            shift(X);                                        // -> X with column A, Y with column B
        }
        if constexpr (VT > 0) {
            // the partial row's taps NR M .. NR M + VT - 1 (M = 48: 32 of its 33)
            static_assert(VT < H || VT == H + 8, "VT = 32 at M = 48");
            if constexpr (VT >= H) {
                fir_fma2x8(X, Y, &y[0][0], &y[1][0], ta);
                half_rest(X, Y, ta, 0);
                const uint64_t *t = taps64 + (NR * M + H) / 2;
                fir_fma2x8(X, Y, &y[0][H], &y[1][H], t);
            } else {
                const uint64_t *t = taps64 + (NR * M) / 2;
#pragma unroll
                for (int gq = 0; gq < VT / 8; gq++) fir_fma2x8(X, Y, &y[0][8 * gq], &y[1][8 * gq], t + 4 * gq);
            }
        }
        // ---- horizontal sum (simd_avx2.c:89-96) and the tap the vector loop leaves over (:102-105) ----
        {
            const float tq = taps[kFirTaps - 1];
            const v2f sa = (X[0] + X[2]) + (X[1] + X[3]), sb = (Y[0] + Y[2]) + (Y[1] + Y[3]);
            const v2f oa = sa + y[0][VT] * tq, ob = sb + y[1][VT] * tq;
            const int qa = 2 * lane - NR, qb = qa + 1;
            if (qa >= 0 && qa < g.n_out) dec[g.out_base + qa] = make_float2(oa.x, oa.y);
            if (qb >= 0 && qb < g.n_out) dec[g.out_base + qb] = make_float2(ob.x, ob.y);
        }
    }
    kclk_leave(kclk);
}

int g_fir_claim = 0;           // 1: the resident grid of fir_decimate_kernel_f claims its strips from a counter; 0 (default): fixed shares.
                               // Measured (profiles/r5_fir_claim.json): with the claim the kernel takes 0.54 ms ALONE against 0.33 -- the strip
                               // index reaches the geometry record's scalar loads through an atomic and a v_readfirstlane at every strip --
                               // and 58-59 Gsamples/s against 68.5-70.2 in run: fixed shares stay.

template <int M>
static int launch_fir_f_fmt(const SampleSource &src, const FirGeom *geom, int n_tiles, const float *taps,
                            const float2 *rot_table, float2 *dec, hipStream_t stream, unsigned long long *kclk, unsigned *next_tile)
{
    if (!g_fir_claim) next_tile = nullptr;
    // Workgroups: by default a resident grid of seven single-wavefront workgroups per CU that walk the strips, i.e. seven
    // of a CU's eight 256-register slots (two per SIMD) -- the eighth is where the waves of K1, the scan's passes and
    // the per-burst filters of the other streams live while this kernel runs.  Measured (10 MHz, in run, six / seven per
    // CU): the decimator's own span 0.55 -> 0.41 / 0.40 ms, the step 1.09 -> 1.06-1.09 / 1.08 ms against one workgroup
    // per strip, whose short-lived waves (40 000 per chunk) lose every freed slot to the higher-priority streams (12 MHz
    // dense, six: 2.7 -> 2.2 ms, step 3.13 -> 2.96 ms).  Option fir_grid: n > 0 workgroups, 0 one per strip.
    int grid = n_tiles;
    if (g_fir_grid > 0) grid = g_fir_grid < n_tiles ? g_fir_grid : n_tiles;
    else if (g_fir_grid < 0) {
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
            if (n_cu <= 0) n_cu = 256;
        }
        // (the per-burst chains' streams may be masked off some CUs -- IRDM_CHAIN_CU_RESERVE --: the resident grid is
        // seven per CU THEY may use; sized for the whole device, the workgroups that found no slot ran as a second round
        // and doubled the kernel's time: what round 5's first CU-mask measurement, 0.61 ms, had measured)
        const int cus = g_chain_cus > 0 && g_chain_cus < n_cu ? g_chain_cus : n_cu;
        if (7 * cus < n_tiles) grid = 7 * cus;
    }
    if (next_tile != nullptr) {
        if (src.fmt == 2) hipLaunchKernelGGL((fir_decimate_kernel_f<M, 2, true>), dim3(grid), dim3(64), 0, stream, src, geom, taps, rot_table, dec, n_tiles, kclk, next_tile);
        else if (src.fmt == 1) hipLaunchKernelGGL((fir_decimate_kernel_f<M, 1, true>), dim3(grid), dim3(64), 0, stream, src, geom, taps, rot_table, dec, n_tiles, kclk, next_tile);
        else hipLaunchKernelGGL((fir_decimate_kernel_f<M, 0, true>), dim3(grid), dim3(64), 0, stream, src, geom, taps, rot_table, dec, n_tiles, kclk, next_tile);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    if (src.fmt == 2) hipLaunchKernelGGL((fir_decimate_kernel_f<M, 2>), dim3(grid), dim3(64), 0, stream, src, geom, taps, rot_table, dec, n_tiles, kclk, next_tile);
    else if (src.fmt == 1) hipLaunchKernelGGL((fir_decimate_kernel_f<M, 1>), dim3(grid), dim3(64), 0, stream, src, geom, taps, rot_table, dec, n_tiles, kclk, next_tile);
    else hipLaunchKernelGGL((fir_decimate_kernel_f<M, 0>), dim3(grid), dim3(64), 0, stream, src, geom, taps, rot_table, dec, n_tiles, kclk, next_tile);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_fir_fma(const SampleSource &src, const FirGeom *geom, int n_tiles, int decim, const float *taps,
                   const float2 *rot_table, float2 *dec, hipStream_t stream, unsigned long long *kclk, unsigned *next_tile)
{
    if (src.ring_len % 8 != 0 || src.ref_ring % 8 != 0 || (src.chunk_start != ~0ull && src.chunk_start % 8 != 0)) return 1;
    switch (decim) {
    case 40: return launch_fir_f_fmt<40>(src, geom, n_tiles, taps, rot_table, dec, stream, kclk, next_tile);
    case 48: return launch_fir_f_fmt<48>(src, geom, n_tiles, taps, rot_table, dec, stream, kclk, next_tile);
    default: return 1;
    }
}

template <int M>
static int launch_fir_r_fmt(const SampleSource &src, const FirGeom *geom, int n_tiles, const float *taps,
                            const float2 *rot_table, float2 *dec, hipStream_t stream, unsigned long long *kclk)
{
    // Slices (option fir_slice): the strips of a chunk as several launches of `g_fir_slice` strips each.  A wavefront of
    // this kernel owns half a SIMD's registers for its whole life and the dispatcher refills a freed slot from the SAME
    // launch: next to one launch of 12 000 strips a pass of the detector scan (another stream, higher priority) that
    // arrives after the first generation of wavefronts waits until the launch has drained (measured: passes of 60-90 us
    // stretched to 350-390 us).  The strips of a generation end together anyway (equal lengths), so a launch boundary per
    // generation costs a dispatch (~2 us) and lets everything that is waiting in.
    const int slice = g_fir_slice > 0 && g_fir_grid <= 0 ? g_fir_slice : n_tiles;
    for (int t0 = 0; t0 < n_tiles; t0 += slice) {
        const int cnt = n_tiles - t0 < slice ? n_tiles - t0 : slice;
        const int grid = g_fir_grid > 0 && g_fir_grid < cnt ? g_fir_grid : cnt;
        if (src.fmt == 2) hipLaunchKernelGGL((fir_decimate_kernel_r<M, 2>), dim3(grid), dim3(64), 0, stream, src, geom + t0, taps, rot_table, dec, cnt, kclk);
        else if (src.fmt == 1) hipLaunchKernelGGL((fir_decimate_kernel_r<M, 1>), dim3(grid), dim3(64), 0, stream, src, geom + t0, taps, rot_table, dec, cnt, kclk);
        else hipLaunchKernelGGL((fir_decimate_kernel_r<M, 0>), dim3(grid), dim3(64), 0, stream, src, geom + t0, taps, rot_table, dec, cnt, kclk);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_fir_reg(const SampleSource &src, const FirGeom *geom, int n_tiles, int decim, const float *taps,
                   const float2 *rot_table, float2 *dec, hipStream_t stream, unsigned long long *kclk)
{
    // every boundary a column can meet must be a multiple of 8 samples (see the fetch); chunks start at multiples of the
    // feed block, which is a multiple of the FFT size
    // (chunk_start = ~0: everything is read from the ring)
    if (src.ring_len % 8 != 0 || src.ref_ring % 8 != 0 || (src.chunk_start != ~0ull && src.chunk_start % 8 != 0)) {
        fprintf(stderr, "irdm_hip: fir_reg: ring_len %llu / ref_ring %llu / chunk_start %llu not multiples of 8 samples\n",
                (unsigned long long)src.ring_len, (unsigned long long)src.ref_ring, (unsigned long long)src.chunk_start);
        return 1;
    }
    switch (decim) {
    case 40: return launch_fir_r_fmt<40>(src, geom, n_tiles, taps, rot_table, dec, stream, kclk);
    case 48: return launch_fir_r_fmt<48>(src, geom, n_tiles, taps, rot_table, dec, stream, kclk);
    default: return 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The decimator on the matrix cores (option fir_layout 4; fir_order 1 only): avx2_fir_ccf_dec's arithmetic as a
// Toeplitz-taps x samples product on v_mfma_f32_16x16x4_f32.
//
// fir_decimate_kernel_f above issues one v_pk_fma_f32 per tap and output pair and is bound by that instruction's issue
// rate (42 TFLOP/s useful at 53 % of the HBM roofline, beside a K1 that wants the same pipe).  The f32-input MFMA is an
// fmaf chain in k order, bit for bit (tools/ubench/mfma_fir.hip -> profiles/r5_mfma_fir.txt: the instruction against
// fma(a[k], b[k], acc), k = 0..3, on 512 000 values with zeros, -0, large and small operands; the whole filter against the
// host's loop on 262 144 outputs, M = 40 and 48: no differing bit), and fma(0, y, acc) == acc: for ONE polyphase j
//
//   D[i][n] += sum_k A[i][k] * B[k][n]     i   16 consecutive outputs q0 + i of a strip (the row group)
//                                          n   16 real streams: (re, im) of 8 strips filtered side by side
//                                          k   4 consecutive positions u = 4 s + kk of the sequence y_j[u] = y[q0 M + 4 u + j]
//   A[i][k] = t[4 (u - (M/4) i) + j], zero outside taps 0..799 (the Toeplitz band);  B[k][n] = y_j[u] of stream n
//
// with s ascending over the 88 (M = 40) / 96 (48) steps that a row group's positions take, every accumulator sees its taps
// in ascending order: acc_j = fma(t[4m + j], y[qM + 4m + j], acc_j), the AVX2 kernel's four accumulators = the four
// wavefronts of a workgroup, summed (a0 + a2) + (a1 + a3) + t[800] * y[qM + 800] by the VALU afterwards.  57 % of the
// multiply-adds the instruction performs are the filter's (16 rows share a window of 350 positions of which 200 are a
// row's own); the core sustains 64 TFLOP/s useful at two chains per SIMD, 50 at one (the ubench).
//
// A workgroup (4 wavefronts, wavefront j = polyphase j) walks an OCTET of strips (8 FirGeom tiles of kFirMfmaTile
// outputs) 16 outputs at a time.  The rotated samples of the 8 strips live in LDS, de-interleaved by polyphase and
// stream, S[j][n][position mod CP], a circular window: a step's chain reads WSEG segments (16 samples = 4 positions each)
// while the M segments per strip the NEXT step adds are loaded (fir_fetch_column<16>: the ring, the stale tail, every
// format, exactly as above), rotated from their checkpoint (a segment starts on one) and written behind the window --
// WSEG + M segments = the buffer (128 at M = 40: 132 KB; 144 at 48: 148 KB).  Every sample is fetched and rotated ONCE
// per tile (tiles overlap by the window's 88 segments per 256 outputs: 1.07 x the algorithmic bytes).  A operands: 88 / 96
// VGPRs per lane, the same for every row group; B: one ds_read_b32 per MFMA, lane (kk, n) -> bank 4 n + kk.
//
// What the form costs in exactness: nothing for finite samples.  A NaN or infinity among the samples poisons the (up to
// 15) outputs of its row group whose taps there are ZERO (0 x inf), where the reference's outputs further than 800 samples
// away stay finite: cf32 input with non-finite samples is outside this kernel's contract (ci8 / ci16 cannot produce them).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kFirMfmaTile = 256;          // outputs per FirTile (16 row groups)
int fir_mfma_tile_out(int) { return kFirMfmaTile; }

typedef float f32x4_t __attribute__((vector_size(16)));

template <int M>
struct FirX {
    static constexpr int Q = M / 4;                                     // positions of one polyphase per output
    static constexpr int STEPS = ((15 * Q + 200 + 1) + 3) / 4;          // MFMA steps per row group: positions 0 .. 15 Q + 200 (the tail tap's)
    static constexpr int WSEG = STEPS;                                  // segments (4 positions) a row group's window holds
    static constexpr int ADV = M;                                       // segments a step adds: 16 outputs x M samples / 16
    static constexpr int CSEG = WSEG + ADV;                             // the buffer, in segments
    static constexpr int CP = 4 * CSEG;                                 // ... in positions
    static constexpr int PITCH = (CP + 63) / 64 * 64 + 4;               // words per (polyphase, stream): == 4 (mod 64)
    static constexpr size_t LDS = ((size_t)4 * 16 * PITCH + 4 * 16 * 16) * 4 + 8 * sizeof(FirGeom);
    static_assert(M % 16 == 0 || M % 16 == 8, "geometry");
    static_assert(LDS <= 160 * 1024, "the window does not fit the LDS");
};

template <int M, int FMT>
__global__ __launch_bounds__(256) void fir_decimate_kernel_x(SampleSource src, const FirGeom *__restrict__ geom,
                                                             const float *__restrict__ taps, const float2 *__restrict__ rot_table,
                                                             float2 *__restrict__ dec, int n_tiles, unsigned long long *__restrict__ kclk)
{
    using X = FirX<M>;
    constexpr int Q = X::Q, STEPS = X::STEPS, ADV = X::ADV, CSEG = X::CSEG, CP = X::CP, PITCH = X::PITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char fir_x_lds[];
    float *S = reinterpret_cast<float *>(fir_x_lds);                            // [4 j][16 n][PITCH]
    float *Cx = S + (size_t)4 * 16 * PITCH;                                     // [4 j][16 rows][16 cols]
    FirGeom *G = reinterpret_cast<FirGeom *>(Cx + 4 * 16 * 16);                 // the octet's 8 tiles
    const int tid = threadIdx.x, lane = tid & 63, j = tid >> 6;
    const int kk = lane >> 4, col = lane & 15;
    kclk_enter(kclk);

    // this wavefront's A operands: lane (kk, i) of step s holds t[4 (4 s + kk - Q i) + j], zero outside taps 0..799
    float a_reg[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; s++) {
        const int u = 4 * s + kk - Q * col;                 // (col doubles as the A operand's row index i = lane & 15)
        a_reg[s] = (u >= 0 && u < 200) ? taps[4 * u + j] : 0.0f;
    }
    const float t_last = taps[kFirTaps - 1];

    // one segment task: segment `seg` of strip `st` -> 16 raw samples (fetch), then rotate + write (finish)
    auto fetch = [&](int st, int seg, v2f *y) {
        const FirGeom g = G[st];
        fir_fetch_column<16, FMT>(src, g, seg, g.n_seg, y);
    };
    auto finish = [&](int st, int seg, v2f *y) {
        const FirGeom &g = G[st];
        const int sg = seg < g.n_seg ? seg : 0;             // (a segment beyond the strip was fetched as segment 0: finite filler)
        const float2 c = rot_table[fir_ck(g, sg)];
        v2f ph = { c.x, c.y };
        const v2f inc = { g.inc_re, g.inc_im };
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const v2f r = cmul_pk(y[u], ph);                // out[i] = in[i] * phase (rotator.h:38)
            ph = cmul_pk(ph, inc);                          // phase *= incr          (rotator.h:39)
            y[u] = r;
        }
        // sample 16 seg + u: polyphase u & 3, position 4 seg + (u >> 2); four positions of a polyphase with one 16-byte store
        const int p = 4 * (seg % CSEG);
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            float *re = S + ((size_t)(jj * 16 + 2 * st)) * PITCH + p;
            *reinterpret_cast<float4 *>(re) = make_float4(y[jj].x, y[jj + 4].x, y[jj + 8].x, y[jj + 12].x);
            *reinterpret_cast<float4 *>(re + PITCH) = make_float4(y[jj].y, y[jj + 4].y, y[jj + 8].y, y[jj + 12].y);
        }
    };

    const int n_octets = (n_tiles + 7) / 8;
#pragma unroll 1
    for (int oc = blockIdx.x; oc < n_octets; oc += gridDim.x) {
        __syncthreads();                                    // (the previous octet's last reads of G and S)
        if (tid < 8) {
            const int t = oc * 8 + tid;
            FirGeom g = geom[t < n_tiles ? t : n_tiles - 1];
            if (t >= n_tiles) {                             // (an octet's missing strips: nothing stored, filler samples)
                g.n_out = 0;
                g.n_seg = 0;
            }
            G[tid] = g;
        }
        __syncthreads();
        int max_out = 0;
#pragma unroll
        for (int st = 0; st < 8; st++) max_out = max(max_out, G[st].n_out);
        const int n_groups = (max_out + 15) / 16;
        // ---- the first window: segments 0 .. WSEG - 1 of the 8 strips ----
#pragma unroll 1
        for (int t0 = 0; t0 < 8 * X::WSEG; t0 += 256) {
            // (every lane fetches -- the fetch votes across the wavefront --, the lanes beyond the last task a task again)
            const bool valid = t0 + tid < 8 * X::WSEG;
            const int task = valid ? t0 + tid : 8 * X::WSEG - 1;
            v2f y[16];
            const int st = task / X::WSEG, seg = task % X::WSEG;
            fetch(st, seg, y);
            if (valid) finish(st, seg, y);
        }
        __syncthreads();
#pragma unroll 1
        for (int g = 0; g < n_groups; g++) {
            // ---- the segments the NEXT step adds: requested now, rotated and written behind this step's chain ----
            const bool more = g + 1 < n_groups;
            v2f ya[16], yb[16];
            const int ta = tid, tb = 256 + tid;             // (8 ADV = 320 / 384 tasks: two rounds)
            const int sta = ta / ADV, sega = X::WSEG + ADV * g + ta % ADV;
            const int stb = tb / ADV, segb = X::WSEG + ADV * g + tb % ADV;
            const bool has_b = tb < 8 * ADV;
            if (more) {
                fetch(sta, sega, ya);
                if (has_b) fetch(stb, segb, yb);
            }
            // ---- the chain: STEPS dependent MFMAs, B from the circular window ----
            f32x4_t acc = { 0.0f, 0.0f, 0.0f, 0.0f };
            {
                const float *bp = S + ((size_t)(j * 16 + col)) * PITCH;
                int p = (4 * ADV * g + kk) % CP;
#pragma unroll
                for (int s = 0; s < STEPS; s++) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_reg[s], bp[p], acc, 0, 0, 0);
                    p += 4;
                    p = p >= CP ? p - CP : p;
                }
            }
            if (more) {
                finish(sta, sega, ya);
                if (has_b) finish(stb, segb, yb);
            }
            // ---- combine the four accumulators, add the tail tap, store ----
            // D: lane l, register r holds row 4 (l >> 4) + r, column l & 15
#pragma unroll
            for (int r = 0; r < 4; r++) Cx[((size_t)j * 16 + 4 * kk + r) * 16 + col] = acc[r];
            __syncthreads();
            {
                // thread -> (strip, row, re | im): a strip's 16 x 2 values are 128 contiguous bytes of `dec`
                const int st = tid >> 5, row = (tid >> 1) & 15, comp = tid & 1, n = 2 * st + comp;
                const float *c = Cx + (size_t)row * 16 + n;
                const float a0 = c[0], a1 = c[256], a2 = c[512], a3 = c[768];
                int pt = (4 * ADV * g + Q * row + 200) % CP;             // sample (q0 + row) M + 800: polyphase 0
                const float yt = S[(size_t)n * PITCH + pt];
                const float v = ((a0 + a2) + (a1 + a3)) + t_last * yt;
                const int q = 16 * g + row;
                if (q < G[st].n_out) reinterpret_cast<float *>(dec + G[st].out_base + q)[comp] = v;
            }
            __syncthreads();
        }
    }
    kclk_leave(kclk);
}

template <int M>
static int launch_fir_x_fmt(const SampleSource &src, const FirGeom *geom, int n_tiles, const float *taps,
                            const float2 *rot_table, float2 *dec, hipStream_t stream, unsigned long long *kclk)
{
    using X = FirX<M>;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    // one workgroup per CU (the window takes most of the LDS), each walking octets of strips
    const int n_octets = (n_tiles + 7) / 8;
    int grid = g_fir_grid > 0 ? g_fir_grid : n_cu;
    if (grid > n_octets) grid = n_octets;
#define IRDM_LAUNCH_FIR_X(FMTv)                                                                                          \
    {                                                                                                                  \
        (void)hipFuncSetAttribute((const void *)fir_decimate_kernel_x<M, FMTv>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)X::LDS); \
        hipLaunchKernelGGL((fir_decimate_kernel_x<M, FMTv>), dim3(grid), dim3(256), X::LDS, stream, src, geom, taps, rot_table, dec, n_tiles, kclk); \
    }
    if (src.fmt == 2) IRDM_LAUNCH_FIR_X(2)
    else if (src.fmt == 1) IRDM_LAUNCH_FIR_X(1)
    else IRDM_LAUNCH_FIR_X(0)
#undef IRDM_LAUNCH_FIR_X
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_fir_mfma(const SampleSource &src, const FirGeom *geom, int n_tiles, int decim, const float *taps,
                    const float2 *rot_table, float2 *dec, hipStream_t stream, unsigned long long *kclk)
{
    if (src.ring_len % 8 != 0 || src.ref_ring % 8 != 0 || (src.chunk_start != ~0ull && src.chunk_start % 8 != 0)) return 1;
    switch (decim) {
    case 40: return launch_fir_x_fmt<40>(src, geom, n_tiles, taps, rot_table, dec, stream, kclk);
    case 48: return launch_fir_x_fmt<48>(src, geom, n_tiles, taps, rot_table, dec, stream, kclk);
    default: return 1;
    }
}

}  // namespace irdm
