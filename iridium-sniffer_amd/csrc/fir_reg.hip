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

constexpr int kFirStrip = 3;   // double blocks (128 columns) per strip of fir_decimate_kernel_r: a strip yields 128 * kFirStrip - NR outputs

template <int M>
struct FirR {
    static constexpr int NR = kFirTaps / M;                  // full rows of M taps
    static constexpr int REM = kFirTaps - NR * M;            // taps of the last, partial row (>= 1)
    static_assert(M % 8 == 0 && NR < 63 && REM >= 1, "geometry");
};

int fir_reg_supported(int decim) { return decim == 40 || decim == 48; }

// outputs per strip (= FirTile unit of the host's tile count)
int fir_reg_tile_out(int decim) { return 128 * kFirStrip - kFirTaps / decim; }

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

template <int M, int FMT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void fir_decimate_kernel_f(
    SampleSource src, const FirGeom *__restrict__ geom, const float *__restrict__ taps,
    const float2 *__restrict__ rot_table, float2 *__restrict__ dec, int n_tiles, unsigned long long *__restrict__ kclk)
{
    using R = FirR<M>;
    constexpr int NR = R::NR, REM = R::REM;
    constexpr int VT = REM - 1;                              // taps of the partial row the vector loop takes
    static_assert(NR % 2 == 0, "rows are unrolled in pairs (the accumulator sets swap names)");
    static_assert(M % 4 == 0 && VT % 8 == 0 && (kFirTaps - 1) % 4 == 0, "tap k -> accumulator k % 4 = (k % M) % 4; one tap left over");
    const int lane = threadIdx.x;
    kclk_enter(kclk);
    // Strips in fixed shares (tile += gridDim.x) of a resident grid.  (Strips claimed from a counter -- a resident grid with
    // fixed shares ends when its slowest wavefront does -- measured slower, 0.54 against 0.33 ms alone: the strip index
    // reached the geometry record's scalar loads through an atomic and a v_readfirstlane at every strip,
    // profiles/r5_fir_claim.json.)
#pragma unroll 1
    for (int tile = blockIdx.x; tile < n_tiles;) {
        const int tile_next = tile + (int)gridDim.x;
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

template <int M>
static int launch_fir_f_fmt(const SampleSource &src, const FirGeom *geom, int n_tiles, const float *taps,
                            const float2 *rot_table, float2 *dec, hipStream_t stream, unsigned long long *kclk)
{
    // A resident grid of seven single-wavefront workgroups per CU that walk the strips, i.e. seven of a CU's eight
    // 256-register slots (two per SIMD) -- the eighth is where the waves of K1, the scan's passes and the per-burst filters of
    // the other streams live while this kernel runs.  Measured (10 MHz, in run, six / seven per CU): the decimator's own span
    // 0.55 -> 0.41 / 0.40 ms, the step 1.09 -> 1.06-1.09 / 1.08 ms against one workgroup per strip, whose short-lived waves
    // (40 000 per chunk) lose every freed slot to the higher-priority streams; five and six per CU again in round 6: 70.2 /
    // 72.7 against 74.1-74.4 Gsamples/s (profiles/r6_option_ab.json).
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    const int grid = 7 * n_cu < n_tiles ? 7 * n_cu : n_tiles;
    if (src.fmt == 2) hipLaunchKernelGGL((fir_decimate_kernel_f<M, 2>), dim3(grid), dim3(64), 0, stream, src, geom, taps, rot_table, dec, n_tiles, kclk);
    else if (src.fmt == 1) hipLaunchKernelGGL((fir_decimate_kernel_f<M, 1>), dim3(grid), dim3(64), 0, stream, src, geom, taps, rot_table, dec, n_tiles, kclk);
    else hipLaunchKernelGGL((fir_decimate_kernel_f<M, 0>), dim3(grid), dim3(64), 0, stream, src, geom, taps, rot_table, dec, n_tiles, kclk);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_fir_fma(const SampleSource &src, const FirGeom *geom, int n_tiles, int decim, const float *taps,
                   const float2 *rot_table, float2 *dec, hipStream_t stream, unsigned long long *kclk)
{
    if (src.ring_len % 8 != 0 || src.ref_ring % 8 != 0 || (src.chunk_start != ~0ull && src.chunk_start % 8 != 0)) return 1;
    switch (decim) {
    case 40: return launch_fir_f_fmt<40>(src, geom, n_tiles, taps, rot_table, dec, stream, kclk);
    case 48: return launch_fir_f_fmt<48>(src, geom, n_tiles, taps, rot_table, dec, stream, kclk);
    default: return 1;
    }
}

template <int M>
static int launch_fir_r_fmt(const SampleSource &src, const FirGeom *geom, int n_tiles, const float *taps,
                            const float2 *rot_table, float2 *dec, hipStream_t stream, unsigned long long *kclk)
{
    // one single-wavefront workgroup per strip
    if (src.fmt == 2) hipLaunchKernelGGL((fir_decimate_kernel_r<M, 2>), dim3(n_tiles), dim3(64), 0, stream, src, geom, taps, rot_table, dec, n_tiles, kclk);
    else if (src.fmt == 1) hipLaunchKernelGGL((fir_decimate_kernel_r<M, 1>), dim3(n_tiles), dim3(64), 0, stream, src, geom, taps, rot_table, dec, n_tiles, kclk);
    else hipLaunchKernelGGL((fir_decimate_kernel_r<M, 0>), dim3(n_tiles), dim3(64), 0, stream, src, geom, taps, rot_table, dec, n_tiles, kclk);
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

}  // namespace irdm
