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

#include "fir_fma.inc"      // (cmul_pk: the four-product complex multiply as three packed instructions)

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

// one lane per tile: everything the decimator's workgroups need, in one record
__global__ void fir_geom_kernel(const BurstWork *__restrict__ work, int n_bursts, int n_tiles, int M,
                                int tile_out, uint64_t ring_len, uint64_t ref_ring, const float2 *__restrict__ rot_incr, int n_ckpt,
                                FirGeom *__restrict__ geom, const int *__restrict__ rot_slot)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
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

int fir_tile_row(int decim)
{
    int r = kFirTileOut + kFirTaps / decim + 2;
    while ((r & 15) != 1) r++;
    return r;
}

// Which decimator a batch takes (DESIGN.md "The decimator"):
//   M = 40 / 48 (10 / 12 MHz), the source's ring lengths multiples of 8 samples (`aligned`: fir_reg.hip fetches columns in
//   pieces of 8): the register-resident kernels of fir_reg.hip -- fir_decimate_kernel_f in the reference's AVX2 order
//   (simd_avx2.c:62-108: four accumulators, fused multiply-adds; order 1), fir_decimate_kernel_r in its scalar order
//   (simd_generic.c:86-96: one accumulator, every product and sum rounded; order 0, --no-simd);
//   anything else (2 / 4 MHz, a caller's burst window presented as a chunk, the test hook `generic`): fir_decimate_kernel
//   above, any M, either order, one tile of kFirTileOut outputs per workgroup from the FirTile list.
// `order`: which of the reference's two forms of its dispatched kernels the product follows (simd_kernels.h; option
// "fir_order", alias "simd_order"; per pipeline).  The other dispatched kernels with two forms (fir_ccf, fir_fff,
// fftshift_mag, mag_squared) follow the same switch in post_tiles / post2 / K1.
static bool fir_reg_path(int decim, int aligned, int generic) { return !generic && aligned && fir_reg_supported(decim); }

// 1: launch_fir_decimate() reads the FirTile list (the one-tile-per-workgroup kernel); 0: only BurstWork::tile_base
int fir_needs_tile_list(int decim, int aligned, int generic) { return fir_reg_path(decim, aligned, generic) ? 0 : 1; }

// outputs per FirTile for the kernel launch_fir_decimate() will pick
int fir_tile_out(int decim, int aligned, int generic, int order)
{
    if (!fir_reg_path(decim, aligned, generic)) return kFirTileOut;
    return order ? fir_fma_tile_out(decim) : fir_reg_tile_out(decim);
}

int launch_fir_decimate(const SampleSource &src, const BurstWork *work, int n_bursts, FirTile *tiles, size_t tiles_cap,
                        int n_tiles, int decim, const float *taps, const int *tap_off, const float2 *rot_incr,
                        const float2 *rot_table, int n_ckpt, float2 *dec,
                        hipStream_t stream, unsigned long long *kclk, const int *rot_slot, int order, int generic)
{
    if (n_tiles <= 0) return 0;
    const int aligned = src.ring_len % 8 == 0 && src.ref_ring % 8 == 0 && (src.chunk_start == ~0ull || src.chunk_start % 8 == 0);
    if (fir_reg_path(decim, aligned, generic) && tiles_cap >= (size_t)n_tiles) {
        // strip geometry (one FirGeom record per strip, behind the FirTile array), then the register-resident decimator
        FirGeom *geom = reinterpret_cast<FirGeom *>(tiles + tiles_cap);
        hipLaunchKernelGGL(fir_geom_kernel, dim3((n_tiles + 255) / 256), dim3(256), 0, stream, work, n_bursts, n_tiles, decim,
                           order ? fir_fma_tile_out(decim) : fir_reg_tile_out(decim), src.ring_len, src.ref_ring, rot_incr, n_ckpt,
                           geom, rot_slot);
        return (order ? launch_fir_fma(src, geom, n_tiles, decim, taps, rot_table, dec, stream, kclk)
                      : launch_fir_reg(src, geom, n_tiles, decim, taps, rot_table, dec, stream, kclk)) == 0 ? 0 : -1;
    }
    const int row = fir_tile_row(decim);
    const size_t lds = sizeof(float2) * (size_t)row * decim;
    if (lds > 160 * 1024) return -1;
    (void)hipFuncSetAttribute((const void *)fir_decimate_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(fir_decimate_kernel, dim3(n_tiles), dim3(kFirTileOut), lds, stream, src, work,
                       tiles, decim, row, taps, tap_off, rot_incr, rot_table, n_ckpt, dec, order, rot_slot);
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
    const size_t wg = 256;
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
    const size_t wg = 256;
    const int grid = (int)std::min<size_t>((n16 + wg - 1) / wg + 1, 512 * 256 / wg);
    hipLaunchKernelGGL(copy_to_host_kernel, dim3(grid), dim3((unsigned)wg), 0, stream, static_cast<uint4 *>(dst),
                       static_cast<const uint4 *>(src), n16, reinterpret_cast<uint32_t *>(static_cast<char *>(dst) + n16 * 16),
                       reinterpret_cast<const uint32_t *>(static_cast<const char *>(src) + n16 * 16), n_tail);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_copy_words(void *dst, const void *src, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return 0;
    const size_t n = bytes / 4;
    const size_t wg = 256;
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
// post1 (steps 2b, 3, 4): post_tiles_kernel + post_cfo_kernel below.  NT / SN: compile-time tap counts (25 / 20 at every
// supported rate), 0 = the runtime values.
// ---------------------------------------------------------------------------
constexpr int kPostMaxTaps = 64;
constexpr int kPostTile = 4 * kPostThreads;

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

// ---------------------------------------------------------------------------
// post1 as TWO launches (round 6).  One workgroup per burst (rounds 1-5) walked a burst tile by tile -- seven rounds of
// load - barrier - filter - barrier - box filter - barrier for a 6700-sample burst, each round mostly the latency of its
// loads -- and then ran a 12-stage radix-2 transform with a barrier per stage: 87 us alone for 36 MB of data, every
// workgroup's latency end to end, holding the chip that long.  Now 38 + 30 us:
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

int launch_downmix_post1(BurstWork *work, int n_bursts, int max_dec_len, float2 *dec,
                         float2 *lpf, float *box, const float *noise_taps, int noise_ntaps,
                         const float *start_taps, int start_ntaps, int search_depth, int pre_start,
                         const float *cfo_window, const float2 *tw4096, BurstWork *hp_work, hipStream_t stream,
                         unsigned long long *kclk, int order, int generic)
{
    if (n_bursts <= 0) return 0;
    if (noise_ntaps > kPostMaxTaps || start_ntaps > kPostMaxTaps) return -1;
    const bool fixed = noise_ntaps == 25 && start_ntaps == 20 && !generic;
    const dim3 grid((unsigned)((max_dec_len + kPostTile - 1) / kPostTile), (unsigned)n_bursts);
    if (grid.x > 0) {
        if (fixed)
            hipLaunchKernelGGL((post_tiles_kernel<25, 20>), grid, dim3(kPostThreads), 0, stream, work, dec, lpf, box, noise_taps,
                               noise_ntaps, start_taps, start_ntaps, search_depth, order, kclk);
        else
            hipLaunchKernelGGL((post_tiles_kernel<0, 0>), grid, dim3(kPostThreads), 0, stream, work, dec, lpf, box, noise_taps,
                               noise_ntaps, start_taps, start_ntaps, search_depth, order, kclk);
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
    const int b = blockIdx.x * 64 + ((int)threadIdx.x & 63);
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

// the phases leave as rows (a row per lane -- rounds 2-4 -- touched 64 cache lines per store instruction: 0.24 ms against 0.11;
// the same rows with plain stores under a branch instead of buffer stores told the store hazard below from an LDS problem)
template <int PITCH>
__global__ __launch_bounds__(128) void rot_phase_rows_kernel(BurstWork *__restrict__ work, int n_bursts,
                                                             float2 *__restrict__ rrc_ws, const BurstWork *__restrict__ hp_work,
                                                             CfoStep cfo)
{
    // The chain: phase_{k+1} = phase_k * incr, one lane per burst, 64 bursts per workgroup of TWO wavefronts.  With every
    // lane storing into its own row a store instruction touched 64 cache lines, and their number, not the multiply chain,
    // set the pace (27 ns per step against the chain's 14).  So a tile of kRotTile steps goes through LDS -- lane b writes
    // its phases to row b of the tile -- and leaves as rows: eight lanes per burst, 16 bytes each, one store instruction
    // = eight bursts x 128 contiguous bytes.  A lone wavefront issues an instruction every ~7 cycles whatever it depends on
    // (profiles/r3_valu_issue.txt), and with the chain, the LDS reads and the row stores in ONE wavefront (round 5: one
    // basic block per tile, skewed by a tile) a step cost ten instructions of which four are the chain's.  Here wavefront
    // 0 runs nothing but the chain (tile t into buffer t & 1) and wavefront 1 nothing but the rows (tile t - 1 out of the
    // other buffer), a workgroup barrier per tile between them.  Every lane runs to the workgroup's longest frame (the
    // extra products are never stored); a burst's length gates the stores of ITS row, whichever lanes make them; a pair
    // of phases is stored whole where its first element is in the frame (the second lands inside the row, behind the
    // frame: kFrameNeed is even).
    constexpr int kRotTile = 16;                         // steps per tile: 128 bytes of a row
    constexpr int kRotPitch = PITCH;                     // float2 per LDS row (17: rows start on different banks)
    __shared__ float2 s_tile[2][64 * kRotPitch];
    __shared__ int s_len[64];
    static_assert(kFrameNeed % 2 == 0 && kRotTile % 2 == 0, "rows start 16-byte aligned, two phases per store");
    const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;   // (wavefront-uniform: scalar branches)
    v2f inc = { 1.0f, 0.0f };
    if (role == 0) {
        const RotStart s0 = rot_phase_start(work, n_bursts, hp_work, cfo);
        int L = 0;
        if (s0.live && s0.drop == 0) {
            const int frame_len = s0.dec_len - s0.start;
            const int need = frame_need(s0.simplex);
            L = frame_len < need ? frame_len : need;
            if (L < 0) L = 0;
        }
        s_len[lane] = L;
        inc = v2f{ s0.inc_re, s0.inc_im };
    } else {
        __builtin_amdgcn_s_setprio(3);
    }
    __syncthreads();
    int Lmax = s_len[lane];
    for (int d = 32; d >= 1; d >>= 1) {
        const int o = __shfl_xor(Lmax, d);
        Lmax = o > Lmax ? o : Lmax;
    }
    const int n_tiles = (Lmax + kRotTile - 1) / kRotTile;
    // wavefront 1: lane -> (burst of pass p, pair of phases): 8 lanes x 2 phases = one 128-byte piece of a row.  The rows
    // leave as buffer stores over this workgroup's 64 rows (fewer in the last workgroup: rows behind the batch are out of
    // range): a store whose pair lies behind its burst's frame gets an out-of-range offset and is dropped by the address
    // check -- no branch.  The tile's offset is part of the VECTOR offset, the scalar offset is the constant 0: with the
    // tile's offset in an SGPR the compiler scheduled a write of two of the store's four data registers directly behind
    // the 128-bit store (LLVM takes a buffer store with a register in the scalar-offset field to be free of the "store
    // wider than 64 bits, then a VALU write of its data registers" hazard), and on gfx950 the rows of every other burst
    // arrived with the NEXT values in them (docs/rounds/tools/rot_store_debug.py; lanes 8-15, 24-31, ... of the store;
    // tests/test_isa_store_hazard.py).  Without a register there the hazard recognizer separates the two.
    const int part = lane & 7;
    int len_of[8], off_of[8];
    const int wg0 = (int)blockIdx.x * 64;
    const int rows = n_bursts - wg0 < 64 ? n_bursts - wg0 : 64;
    const __amdgpu_buffer_rsrc_t r_rows = __builtin_amdgcn_make_buffer_rsrc(rrc_ws + (size_t)wg0 * kFrameNeed, 0,
                                                                            rows * kFrameNeed * (int)sizeof(float2), 0x00020000);
    if (role == 1) {
#pragma unroll
        for (int p = 0; p < 8; p++) {
            const int lb = p * 8 + (lane >> 3);
            len_of[p] = s_len[lb];
            off_of[p] = (lb * kFrameNeed + 2 * part) * (int)sizeof(float2);
        }
    }
    v2f ph = { 1.0f, 0.0f };
    // tile t is computed in phase t and stored in phase t + 1
    for (int t = 0; t <= n_tiles; t++) {
        if (role == 0) {
            if (t < n_tiles) {
                float2 *dst = s_tile[t & 1];
#pragma unroll
                for (int j = 0; j < kRotTile; j++) {
                    dst[lane * kRotPitch + j] = make_float2(ph.x, ph.y);
                    ph = cmul_pk(ph, inc);          // (cmul()'s four products and two sums, three packed instructions)
                }
            }
        } else if (t > 0) {
            const int u = t - 1;
            const float2 *src = s_tile[u & 1];
            const int k = u * kRotTile + 2 * part;
            float2 q0[8], q1[8];
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const int lb = p * 8 + (lane >> 3);
                q0[p] = src[lb * kRotPitch + 2 * part];
                q1[p] = src[lb * kRotPitch + 2 * part + 1];
            }
#pragma unroll
            for (int p = 0; p < 8; p++)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(rot_u32x4, make_float4(q0[p].x, q0[p].y, q1[p].x, q1[p].y)),
                                                       r_rows, k < len_of[p] ? off_of[p] + u * kRotTile * (int)sizeof(float2) : 0x7ffffff0,
                                                       0, 0);
        }
        __syncthreads();
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
                         hipStream_t stream, int order, int generic)
{
    if (n_bursts <= 0) return 0;
    const size_t lds = sizeof(float2) * (3 * kCorrN) + 64;
    hipLaunchKernelGGL((rot_phase_rows_kernel<17>), dim3((n_bursts + 63) / 64), dim3(128), 0, stream, work, n_bursts, rrc_ws, hp_work, cfo);
    if (rrc_ntaps == 51 && !generic) {
        (void)hipFuncSetAttribute((const void *)downmix_post2_kernel<51>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((downmix_post2_kernel<51>), dim3(n_bursts), dim3(kPostThreads), lds, stream, work, lpf,
                           rrc_taps, rrc_ntaps, tw2048, dl_fft, ul_fft, dl_len, ul_len, sps,
                           rrc_ws, frames, order);
    } else {
        (void)hipFuncSetAttribute((const void *)downmix_post2_kernel<0>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((downmix_post2_kernel<0>), dim3(n_bursts), dim3(kPostThreads), lds, stream, work, lpf,
                           rrc_taps, rrc_ntaps, tw2048, dl_fft, ul_fft, dl_len, ul_len, sps,
                           rrc_ws, frames, order);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace irdm
