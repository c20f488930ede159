// detect.hip -- stage A on gfx950: fused window/FFT/|.|^2 kernel and the
// burst-detector scan (burst_detect.c:426-632, :679-699).
#include "common.hpp"
#include "types.hpp"
#include "kernels.hpp"

namespace irdm {

// ---------------------------------------------------------------------------
// K1: load (ci8 | cf32) -> Blackman/0.42 window -> N-point pinned FFT in LDS ->
//     fftshift -> |.|^2  (simd_window_cf + fftwf_execute + simd_fftshift_mag,
//     burst_detect.c:679-687; opencl/burst_fft.c:52-80 window_multiply /
//     fftshift_magnitude).  One workgroup per frame, grid-stride.
//     HBM: 8 B (cf32) or 2 B (ci8) read + 4 B written per sample.
// ---------------------------------------------------------------------------
template <int LOGN, int NT, int FMT>
__global__ __launch_bounds__(NT) void fft_mag_kernel(const void *__restrict__ iq,
                                                      const float *__restrict__ window,
                                                      const float2 *__restrict__ tw,
                                                      float *__restrict__ mag, int n_frames, int order)
{
    constexpr int N = 1 << LOGN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *s = reinterpret_cast<float2 *>(smem_raw);
    const int tid = threadIdx.x;

    for (int frame = blockIdx.x; frame < n_frames; frame += gridDim.x) {
        const size_t base = (size_t)frame * N;
        for (int i = tid; i < N; i += NT) {
            const float2 x = load_iq<FMT>(iq, base + i);
            const float w = window[i];
            s[bitrev((unsigned)i, LOGN)] = make_float2(x.x * w, x.y * w);
        }
        __syncthreads();
        fft_lds_radix2<LOGN, NT, -1>(s, tw);
        for (int i = tid; i < N; i += NT) {
            const float2 v = s[(i + N / 2) & (N - 1)];
            mag[base + i] = mag2_simd(v, order);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// K1 (fast form, N = 4096 / 8192 / 16384): the same pinned radix-2 DIT arithmetic, evaluated as
// three radix-16 register passes (+ a final 1- or 2-stage pass for N > 4096) instead of log2 N
// LDS round trips.  N/16 threads, 16 points per thread.
//
//   pass A  stages 1-4   thread t loads x[t + T*r] (coalesced), i.e. the bit-reversed positions
//                        16*rev(t) + q; twiddles are compile-time indices
//   pass B  stages 5-8   positions hi2*256 + q*16 + lo2
//   pass C  stages 9-12  positions h3*4096 + q*256 + lo3
//   pass D  stages 13..  positions p + q*4096, fused with fftshift + |.|^2
// Every butterfly is a' = a + W.b, b' = a - W.b with W.b in four-product form and the table
// twiddle of the radix-2 flow graph, so the result is bit-identical to fft_lds_radix2 (products by
// the exact twiddles (1,0) and (0,-1) differ from the copy/swap only in the sign of zeros, which
// |.|^2 cannot see).  LDS exchange layouts are chosen so that both the writes and the reads of each
// exchange touch consecutive (or odd-stride) addresses: no bank conflicts.
// ---------------------------------------------------------------------------
__device__ __forceinline__ constexpr int rev4c(int q) { return ((q & 1) << 3) | ((q & 2) << 1) | ((q & 4) >> 1) | ((q & 8) >> 3); }

template <int LOGN, int K>
__device__ __forceinline__ void dit_stages(float2 (&v)[1 << K], int c, int s0, const float2 *__restrict__ tw)
{
#pragma unroll
    for (int i = 1; i <= K; i++) {
        const int half = 1 << (i - 1);
#pragma unroll
        for (int q0 = 0; q0 < (1 << K); q0++) {
            if (q0 & half) continue;
            const int jq = q0 & (half - 1);
            const int tix = (c + (jq << s0)) << (LOGN - s0 - i);
            const float2 t = cmul(tw[tix], v[q0 + half]);
            const float2 a = v[q0];
            v[q0] = make_float2(a.x + t.x, a.y + t.y);
            v[q0 + half] = make_float2(a.x - t.x, a.y - t.y);
        }
        // keep each stage's twiddle loads inside the stage (otherwise all of them are hoisted and the
        // kernel needs ~250 VGPRs = one workgroup per CU)
        __builtin_amdgcn_sched_barrier(0);
    }
}

// stages 1..4 on bit-reversed-loaded points: twiddle indices are compile-time; the exact
// twiddles are applied as copy / swap
template <int LOGN>
__device__ __forceinline__ void dit_first16(float2 (&v)[16], const float2 *__restrict__ tw)
{
    constexpr int N = 1 << LOGN;
#pragma unroll
    for (int i = 1; i <= 4; i++) {
        const int half = 1 << (i - 1);
#pragma unroll
        for (int q0 = 0; q0 < 16; q0++) {
            if (q0 & half) continue;
            const int tix = (q0 & (half - 1)) << (LOGN - i);
            const float2 b = v[q0 + half];
            float2 t;
            if (tix == 0) t = b;
            else if (tix == N / 4) t = make_float2(b.y, -b.x);
            else t = cmul(tw[tix], b);
            const float2 a = v[q0];
            v[q0] = make_float2(a.x + t.x, a.y + t.y);
            v[q0 + half] = make_float2(a.x - t.x, a.y - t.y);
        }
    }
}

// LISTS: the band scan's prefilter fused into the store stage -- the bins with |X|^2 > pre[bin] of this frame go to
// the frame's candidate list (scan_fast.hip, prefilter_kernel: same entries, same unordered layout, count may exceed
// cap) while the values are still in registers, instead of being read back from HBM by a kernel of their own
// (n * 4 B per frame and 60-120 us on the detector's critical path).
template <int LOGN, int FMT, bool LISTS>
__global__ __launch_bounds__((1 << LOGN) / 16) void fft_mag_r16_kernel(const void *__restrict__ iq,
                                                                       const float *__restrict__ window,
                                                                       const float2 *__restrict__ tw,
                                                                       float *__restrict__ mag, int n_frames,
                                                                       const float *__restrict__ pre,
                                                                       unsigned *__restrict__ counts,
                                                                       ListEntry *__restrict__ entries, int cap,
                                                                       int order)
{
    __shared__ int s_cnt;
    constexpr int N = 1 << LOGN, T = N / 16, LB = LOGN - 8, NB = 1 << LB, ROW = NB + 1, RD = LOGN - 12;
    __builtin_amdgcn_s_setprio(2);      // the scan of this chunk waits for K1; it shares SIMDs with the per-burst chains
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *X = reinterpret_cast<float2 *>(smem_raw);
    const int t = threadIdx.x;
    const int lo2 = t >> LB, hb = t & (NB - 1);
    const int lo3 = t & 255, h3 = t >> 8;
    const int h3r = (int)(bitrev((unsigned)h3, LB - 4 > 0 ? LB - 4 : 1)) & ((1 << (LB - 4)) - 1);

    // one frame per workgroup (no grid-stride loop: a persistent loop makes the compiler hoist every
    // frame-invariant address into registers, ~250 VGPRs)
    {
        const int frame = blockIdx.x;
        if (frame >= n_frames) return;
        const size_t base = (size_t)frame * N;
        if (LISTS && t == 0) s_cnt = 0;            // (several barriers before the first candidate)
        ListEntry *const list = LISTS ? entries + (size_t)frame * cap : nullptr;
        auto put = [&](int k, float m) {
            __builtin_nontemporal_store(m, &mag[base + k]);
            if (LISTS && m > pre[k]) {
                const int slot = atomicAdd(&s_cnt, 1);
                if (slot < cap) {
                    list[slot].bin = k;
                    list[slot].mag = m;
                }
            }
        };
        float2 v[16];
        // ---- pass A ----
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int idx = t + T * r;
            // streamed once: non-temporal load (and stores below), the chunk and its magnitudes are far larger than L2 / MALL
            float2 x;
            if (FMT == 2) {
                const double raw = __builtin_nontemporal_load(reinterpret_cast<const double *>(iq) + base + idx);
                x = *reinterpret_cast<const float2 *>(&raw);
            } else {
                x = load_iq<FMT>(iq, base + idx);
            }
            const float w = window[idx];
            v[rev4c(r)] = make_float2(x.x * w, x.y * w);
        }
        dit_first16<LOGN>(v, tw);
#pragma unroll
        for (int q = 0; q < 16; q++) X[q * T + t] = v[q];
        __syncthreads();
        // ---- pass B ----
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = X[lo2 * T + rev4c(q) * NB + hb];
        __syncthreads();
        dit_stages<LOGN, 4>(v, lo2, 4, tw);
#pragma unroll
        for (int q = 0; q < 16; q++) X[(q * 16 + lo2) * ROW + hb] = v[q];
        __syncthreads();
        // ---- pass C ----
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = X[lo3 * ROW + rev4c(q) * (NB / 16) + h3r];
        dit_stages<LOGN, 4>(v, lo3, 8, tw);
        if (RD == 0) {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int p = q * 256 + lo3;
                put((p + N / 2) & (N - 1), mag2_simd(v[q], order));
            }
            __syncthreads();
        } else {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 16; q++) X[h3 * 4096 + q * 256 + lo3] = v[q];
            __syncthreads();
            // ---- pass D: stages 13.. on points p + q*4096, fused with fftshift + |.|^2 ----
            constexpr int RQ = 1 << RD;                 // points per group
            constexpr int GP = 16 / RQ;                 // groups per thread
#pragma unroll
            for (int i = 0; i < GP; i++) {
                const int pp = t + T * i;               // < 4096
                float2 d[RQ];
#pragma unroll
                for (int q = 0; q < RQ; q++) d[q] = X[pp + q * 4096];
                dit_stages<LOGN, RD>(d, pp, 12, tw);
#pragma unroll
                for (int q = 0; q < RQ; q++) {
                    const int k = pp + q * 4096;
                    put((k + N / 2) & (N - 1), mag2_simd(d[q], order));
                }
            }
            __syncthreads();
        }
        if (LISTS && t == 0) counts[frame] = (unsigned)s_cnt;
    }
}

// ---------------------------------------------------------------------------
// K1, streaming form (N = 8192 / 16384): "p32" -- 32 points per lane, N/32 lanes per frame (256 / 512).
//
// A lane loads SIXTEEN BYTES at a time: the adjacent samples n = 2 pi and 2 pi + 1 (pi = t + T r, r < 16; cf32; ci16 /
// ci8 pairs are 8 / 4 bytes), so a wavefront's load covers 1 KB of the stream.  Bit reversal sends the lowest index bit
// to the highest position bit: the even samples are the upper operand E and the odd samples the lower operand O of the
// LAST stage (X[k] = E[k] + W.O[k]), each of them a 4096-point transform of its own -- the same pinned radix-2 DIT
// flow graph, regrouped.  A lane therefore carries two (N = 16384: two of four) independent 4096-point sub-transforms
// side by side through three radix-16 register passes:
//
//   pass A  stages 1-4    sub-position p = 16 rev8(t') + q     twiddles are compile-time indices -> SGPR operands
//   pass B  stages 5-8    p = hi 256 + q 16 + lo               per-lane twiddles, SHARED by the lane's two sub-transforms
//   pass C  stages 9-12   p = q 256 + lo3
//   last    stage 13 [and 14] join the sub-transforms of a lane [and, N = 16384, of the lane 32 further on: the two
//           halves of a wavefront swap registers with v_permlane32_swap, no LDS], then fftshift + |.|^2
//
// Two exchanges through the LDS instead of three, and ONE sub-transform at a time: 34 KB per 256 lanes (68 KB per 512 at
// N = 16384) instead of 66 / 133 KB, so four workgroups of four wavefronts (two of eight) share a CU, and a workgroup
// needs ONE free wavefront slot per SIMD to start -- it fits beside the register-filling decimator (fir_reg.hip), where
// the radix-16 kernel above (8 / 16 wavefronts and 66 / 133 KB per frame) waits for a whole CU to drain.
// Exchange layouts (8-byte slots; conflict-free for ds_write_b64's 16-lane and ds_read_b64's 32-lane groups, all
// addresses base + immediate):
//   A -> B   slot = (alpha & 15) 256 + e 16 + (alpha >> 4),  alpha = p >> 4, e = p & 15       (32 KB, no padding)
//   B -> C   slot = mid 272 + low 17 + top,  p = top 256 + mid 16 + low                       (34 KB)
// The magnitudes leave through the same buffer, transposed so that a lane stores (and tests against `pre`) four
// consecutive bins with one 16-byte access.
// The butterflies are csrc/fft_bfly.inc: five packed instructions each, no moves.  Arithmetic: every butterfly is
// a' = a + W.b, b' = a - W.b with the table twiddle of the radix-2 flow graph and W.b in four-product form -- bit for
// bit fft_lds_radix2 (in passes B.. the exact twiddles (1,0), (0,-1) are multiplied rather than copied / swapped, which
// changes the sign of zeros only; |.|^2 cannot see it).
// ---------------------------------------------------------------------------
#include "fft_bfly.inc"

// Buffer accesses: the address of every global access of the kernel is  resource (SGPRs) + lane offset (one VGPR that never
// changes) + scalar / immediate offset (the compile-time part), so that no vector instruction is spent on addresses.
// aux 2 = non-temporal: the chunk and its magnitudes are streamed once and are far larger than L2 / MALL.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t k1_rsrc(const void *p, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)(bytes > 0x7fffffffu ? 0x7fffffffu : bytes), 0x00020000);
}
__device__ __forceinline__ v2f k1_load2(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}

// sample pair (2 lane + {0, 1}) of the 2 T samples behind soff: converted exactly as load_iq does
template <int FMT>
__device__ __forceinline__ void load_pair(__amdgpu_buffer_rsrc_t r, int lane, int soff, v2f &x0, v2f &x1)
{
    if (FMT == 2) {
        const v4f v = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, soff, 2));
        x0 = v2f{ v.x, v.y };
        x1 = v2f{ v.z, v.w };
    } else if (FMT == 1) {
        const int2 v = __builtin_bit_cast(int2, __builtin_amdgcn_raw_buffer_load_b64(r, lane * 8, soff, 2));
        const short r0 = (short)(v.x & 0xffff), i0 = (short)(v.x >> 16), r1 = (short)(v.y & 0xffff), i1 = (short)(v.y >> 16);
        x0 = v2f{ (float)(r0 >> 8) / 128.0f, (float)(i0 >> 8) / 128.0f };     // load_iq<1>
        x1 = v2f{ (float)(r1 >> 8) / 128.0f, (float)(i1 >> 8) / 128.0f };
    } else {
        const int v = __builtin_amdgcn_raw_buffer_load_b32(r, lane * 4, soff, 2);
        const signed char r0 = (signed char)(v & 0xff), i0 = (signed char)((v >> 8) & 0xff);
        const signed char r1 = (signed char)((v >> 16) & 0xff), i1 = (signed char)((v >> 24) & 0xff);
        x0 = v2f{ (float)r0 / 128.0f, (float)i0 / 128.0f };                   // load_iq<0>
        x1 = v2f{ (float)r1 / 128.0f, (float)i1 / 128.0f };
    }
}

template <int LOGN>
struct P32 {
    static constexpr int N = 1 << LOGN, T = N / 32;
    static constexpr int G = N / 8192;                       // groups of two sub-transforms (lanes of one group exchange)
    static constexpr int REGION = 16 * 272 + (G > 1 ? 1 : 0);   // slots per group (+1: pass A's two groups write one
                                                             // instruction's slots on different banks)
    static constexpr size_t LDS = sizeof(float2) * (size_t)G * REGION;
    static_assert(LOGN == 13 || LOGN == 14, "two or four 4096-point sub-transforms");
    static_assert(LDS >= sizeof(float) * (size_t)N, "the magnitudes are staged in the exchange buffer");
};

// The fifteen twiddles of stages s0+1 .. s0+4 (1 + 2 + 4 + 8) for the lane whose position below 2^s0 is c.  Requested
// BEFORE the exchange that precedes the pass: they do not depend on the data, and the exchange's barriers cover their
// latency (fetched inside the pass, every stage began with a wait for its own loads).
template <int LOGN>
__device__ __forceinline__ void dit16_twiddles(v2f (&w)[15], int c, int s0, __amdgpu_buffer_rsrc_t r_tw)
{
#pragma unroll
    for (int i = 1; i <= 4; i++) {
        const int half = 1 << (i - 1);
        // twiddle index (c + (jq << s0)) << (LOGN - s0 - i): the lane's part in the vector offset, jq's in the scalar offset
        const int voff = c << (LOGN - s0 - i + 3);
#pragma unroll
        for (int jq = 0; jq < 8; jq++) {
            if (jq >= half) continue;
            w[half - 1 + jq] = k1_load2(r_tw, voff, jq << (LOGN - i + 3));
        }
    }
}

// stages s0+1 .. s0+4 of both sub-transforms of a lane
__device__ __forceinline__ void dit16_x2(v2f (&a)[16], v2f (&b)[16], const v2f (&w)[15])
{
#pragma unroll
    for (int i = 1; i <= 4; i++) {
        const int half = 1 << (i - 1);
#pragma unroll
        for (int q0 = 0; q0 < 16; q0++) {
            if (q0 & half) continue;
            bfly2_v(a[q0], a[q0 + half], b[q0], b[q0 + half], w[half - 1 + (q0 & (half - 1))]);
        }
    }
}

template <int LOGN, int FMT, bool LISTS>
__global__ __launch_bounds__((1 << LOGN) / 32, 4) void fft_mag_p32_kernel(const void *__restrict__ iq,
                                                                          const float *__restrict__ window,
                                                                          const float2 *__restrict__ tw,
                                                                          float *__restrict__ mag, int n_frames,
                                                                          const float *__restrict__ pre,
                                                                          unsigned *__restrict__ counts,
                                                                          ListEntry *__restrict__ entries, int cap,
                                                                          unsigned long long *__restrict__ kclk, int order)
{
    using G_ = P32<LOGN>;
    constexpr int N = G_::N, T = G_::T, G = G_::G, REGION = G_::REGION;
    constexpr int BPS = FMT == 2 ? 8 : (FMT == 1 ? 4 : 2);
    __shared__ int s_cnt;
    __builtin_amdgcn_s_setprio(2);      // the scan of this chunk waits for K1; it shares SIMDs with the per-burst chains
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v2f *const X = reinterpret_cast<v2f *>(smem_raw);
    const int t = threadIdx.x;
    const int frame = blockIdx.x;
    if (frame >= n_frames) return;
    kclk_enter(kclk);
    if (LISTS && t == 0) s_cnt = 0;                 // (several barriers before the first candidate)
    const __amdgpu_buffer_rsrc_t r_tw = k1_rsrc(tw, sizeof(float2) * (N / 2));

    // ---- load, window, pass A ----
    v2f a[16], b[16];                               // the lane's two sub-transforms: even samples (E), odd samples (O)
    {
        const __amdgpu_buffer_rsrc_t r_in = k1_rsrc(reinterpret_cast<const char *>(iq) + (size_t)frame * N * BPS, (size_t)N * BPS);
        const __amdgpu_buffer_rsrc_t r_win = k1_rsrc(window, sizeof(float) * N);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            v2f x0, x1;
            load_pair<FMT>(r_in, t, r * T * 2 * BPS, x0, x1);        // samples 2 (t + T r) + {0, 1}
            const v2f w = k1_load2(r_win, t * 8, r * T * 8);
            a[rev4c(r)] = x0 * w.x;                 // simd_window_cf: re * w, im * w
            b[rev4c(r)] = x1 * w.y;
        }
        const uint64_t *const tw64 = reinterpret_cast<const uint64_t *>(tw);
#pragma unroll
        for (int i = 1; i <= 4; i++) {
            const int half = 1 << (i - 1);
#pragma unroll
            for (int q0 = 0; q0 < 16; q0++) {
                if (q0 & half) continue;
                const int tix = (q0 & (half - 1)) << (LOGN - i);
                if (tix == 0) bfly2_one(a[q0], a[q0 + half], b[q0], b[q0 + half]);
                else if (tix == N / 4) bfly2_mi(a[q0], a[q0 + half], b[q0], b[q0 + half]);
                else bfly2_s(a[q0], a[q0 + half], b[q0], b[q0 + half], tw64[tix]);
            }
        }
    }
    // ---- A -> B ----
    const int hi = t & 15, lo = (t >> 4) & 15;      // pass B: lane = (hi, lo), p = hi 256 + q 16 + lo
    v2f w[15];
    dit16_twiddles<LOGN>(w, lo, 4, r_tw);
    __builtin_amdgcn_sched_barrier(0);
    {
        const int ga = G > 1 ? (t & 1) : 0, tp = G > 1 ? (t >> 1) : t;
        const int al_lo = (int)bitrev((unsigned)(tp >> 4), 4), al_hi = (int)bitrev((unsigned)(tp & 15), 4);   // alpha = rev8(t')
        v2f *const Xw = X + ga * REGION + al_lo * 256 + al_hi;
        const int gb = G > 1 ? (t >> 8) : 0;
        const v2f *const Xr = X + gb * REGION + (t & 255);
#pragma unroll
        for (int e = 0; e < 16; e++) Xw[e * 16] = a[e];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; q++) a[q] = Xr[q * 256];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; e++) Xw[e * 16] = b[e];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; q++) b[q] = Xr[q * 256];
    }
    // ---- pass B ----
    dit16_x2(a, b, w);
    __syncthreads();
    // ---- B -> C ----
    const int gc = G > 1 ? ((t >> 5) & 1) : 0;
    const int lo3 = G > 1 ? (((t >> 6) << 5) | (t & 31)) : t;
    dit16_twiddles<LOGN>(w, lo3, 8, r_tw);          // pass C: p = q 256 + lo3
    __builtin_amdgcn_sched_barrier(0);
    {
        const int gb = G > 1 ? (t >> 8) : 0;
        v2f *const Xw = X + gb * REGION + lo * 17 + hi;
        const v2f *const Xr = X + gc * REGION + (lo3 >> 4) * 272 + (lo3 & 15) * 17;
#pragma unroll
        for (int q = 0; q < 16; q++) Xw[q * 272] = a[q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; q++) a[q] = Xr[q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; q++) Xw[q * 272] = b[q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; q++) b[q] = Xr[q];
    }
    // ---- pass C ----
    dit16_x2(a, b, w);
    __syncthreads();                                // every lane has read its points: the buffer takes the magnitudes
    // ---- last stage(s), fftshift, |.|^2 -> LDS ----
    float *const M = reinterpret_cast<float *>(smem_raw);
    if (G == 1) {
        // stage 13: X[p] = E[p] + W^p O[p], X[p + 4096] = E[p] - W^p O[p],  p = q 256 + lo3
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
            const v2f w0 = k1_load2(r_tw, lo3 * 8, q * 2048), w1 = k1_load2(r_tw, lo3 * 8, (q + 1) * 2048);
            bfly2_vv(a[q], b[q], w0, a[q + 1], b[q + 1], w1);
        }
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int p = q * 256 + lo3;
            M[p + N / 2] = mag2_simd(make_float2(a[q].x, a[q].y), order);       // bin k = p       -> (k + N/2) mod N
            M[p] = mag2_simd(make_float2(b[q].x, b[q].y), order);               // bin k = p + N/2 -> p
        }
    } else {
        // the sub-transforms of a lane are (b0 = 0 | 1, b1 = gc); stage 13 joins b1 = 0 with b1 = 1 (twiddle W^2p), stage 14
        // b0 = 0 with b0 = 1.  The wavefront's lower half (b1 = 0) takes q < 8, the upper half q >= 8: after the swap
        // a[j] / b[j] hold the b1 = 0 operands and a[j + 8] / b[j + 8] the b1 = 1 operands of p = (j + 8 gc) 256 + lo3.
#pragma unroll
        for (int j = 0; j < 8; j++) {
            swap32(a[j], a[j + 8]);
            swap32(b[j], b[j + 8]);
        }
        const int pl = gc * 2048 + lo3;             // p = j 256 + pl
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const v2f w = k1_load2(r_tw, pl * 16, j * 4096);                                    // W^(2p)
            const v2f w0 = k1_load2(r_tw, pl * 8, j * 2048), w1 = k1_load2(r_tw, pl * 8, j * 2048 + 32768);   // W^p, W^(p + 4096)
            bfly2_v(a[j], a[j + 8], b[j], b[j + 8], w);               // -> U0[p], U0[p + 4096], U1[p], U1[p + 4096]
            bfly2_vv(a[j], b[j], w0, a[j + 8], b[j + 8], w1);
            // a[j] = X[p], b[j] = X[p + 8192], a[j + 8] = X[p + 4096], b[j + 8] = X[p + 12288]
            const int p = j * 256 + pl;
            M[p + 8192] = mag2_simd(make_float2(a[j].x, a[j].y), order);
            M[p] = mag2_simd(make_float2(b[j].x, b[j].y), order);
            M[p + 12288] = mag2_simd(make_float2(a[j + 8].x, a[j + 8].y), order);
            M[p + 4096] = mag2_simd(make_float2(b[j + 8].x, b[j + 8].y), order);
        }
    }
    __syncthreads();
    // ---- store: four consecutive bins per lane and access; the band scan's candidate test on the way ----
    {
        const __amdgpu_buffer_rsrc_t r_out = k1_rsrc(mag + (size_t)frame * N, sizeof(float) * N);
        const __amdgpu_buffer_rsrc_t r_pre = k1_rsrc(pre, sizeof(float) * N);
        ListEntry *const list = LISTS ? entries + (size_t)frame * cap : nullptr;
        // The candidate tests leave wavefront masks in scalar registers (a v_cmp each, nothing else on the vector unit);
        // a wavefront with candidates in a half of its accesses takes ONE slot range for all of them from the frame's
        // counter (one LDS atomic) and every candidate's lane writes its entry at  base + the candidates of the tests
        // before + its rank among the test's lanes.
#pragma unroll
        for (int h = 0; h < 2; h++) {
            v4f m[4];
            unsigned long long hit[16];
#pragma unroll
            for (int i4 = 0; i4 < 4; i4++) {
                const int i = 4 * h + i4;
                m[i4] = *reinterpret_cast<const v4f *>(M + 4 * (t + T * i));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(k1_u32x4, m[i4]), r_out, t * 16, i * T * 16, 2);
                if (LISTS) {
                    const v4f pr = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r_pre, t * 16, i * T * 16, 0));
                    hit[4 * i4 + 0] = __builtin_amdgcn_ballot_w64(m[i4].x > pr.x);
                    hit[4 * i4 + 1] = __builtin_amdgcn_ballot_w64(m[i4].y > pr.y);
                    hit[4 * i4 + 2] = __builtin_amdgcn_ballot_w64(m[i4].z > pr.z);
                    hit[4 * i4 + 3] = __builtin_amdgcn_ballot_w64(m[i4].w > pr.w);
                }
            }
            if (LISTS) {
                unsigned long long any = 0;
#pragma unroll
                for (int j = 0; j < 16; j++) any |= hit[j];
                if (any != 0) {                                              // (wavefront-uniform)
                    int total = 0;
#pragma unroll
                    for (int j = 0; j < 16; j++) total += __popcll(hit[j]);
                    int base = 0;
                    if ((t & 63) == 0) base = atomicAdd(&s_cnt, total);
                    base = __builtin_amdgcn_readfirstlane(base);
                    const unsigned long long below = (1ull << (t & 63)) - 1ull, me = 1ull << (t & 63);
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        if (hit[j] != 0) {                                   // (wavefront-uniform)
                            const int slot = base + __popcll(hit[j] & below);
                            if ((hit[j] & me) && slot < cap) {
                                const int i4 = j >> 2, c = j & 3;
                                list[slot].bin = 4 * (t + T * (4 * h + i4)) + c;
                                list[slot].mag = c == 0 ? m[i4].x : (c == 1 ? m[i4].y : (c == 2 ? m[i4].z : m[i4].w));
                            }
                            base += __popcll(hit[j]);
                        }
                    }
                }
            }
        }
    }
    if (LISTS) {
        __syncthreads();
        if (t == 0) counts[frame] = (unsigned)s_cnt;
    }
    kclk_leave(kclk);
}

template <int LOGN, bool LISTS>
static int launch_p32(int fmt, const void *iq, const float *window, const float2 *tw, float *mag, int n_frames,
                      const float *pre, unsigned *counts, ListEntry *entries, int cap, unsigned long long *kclk,
                      hipStream_t stream, int order)
{
    const size_t lds = P32<LOGN>::LDS;
    const dim3 grid(n_frames), block(P32<LOGN>::T);
#define IRDM_LAUNCH_P32(F)                                                                               \
    do {                                                                                                 \
        (void)hipFuncSetAttribute((const void *)fft_mag_p32_kernel<LOGN, F, LISTS>,                      \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
        hipLaunchKernelGGL((fft_mag_p32_kernel<LOGN, F, LISTS>), grid, block, lds, stream, iq, window, tw, mag, \
                           n_frames, pre, counts, entries, cap, kclk, order);                            \
    } while (0)
    if (fmt == 2) IRDM_LAUNCH_P32(2);
    else if (fmt == 1) IRDM_LAUNCH_P32(1);
    else IRDM_LAUNCH_P32(0);
#undef IRDM_LAUNCH_P32
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Which K1 a frame size takes: 8192 / 16384 points (10 / 12 MHz) fft_mag_p32_kernel, 4096 fft_mag_r16_kernel, 256 .. 2048
// (2 MHz, the plug point's small sizes) fft_mag_kernel -- the same butterflies on the same operands in all three.
// `order`: fftshift_mag in the reference's AVX2 form (1: fma(re, re, im * im)) or its generic form (0), see mag2_simd.
template <int LOGN, bool LISTS>
static int launch_r16(int fmt, const void *iq, const float *window, const float2 *tw, float *mag, int n_frames,
                      const float *pre, unsigned *counts, ListEntry *entries, int cap, hipStream_t stream, int order)
{
    constexpr int NB_ = 1 << (LOGN - 8);
    const size_t lds = sizeof(float2) * ((size_t)256 * (NB_ + 1) > ((size_t)1 << LOGN) ? (size_t)256 * (NB_ + 1) : ((size_t)1 << LOGN));
#define IRDM_LAUNCH_R16(F)                                                                               \
    do {                                                                                                 \
        (void)hipFuncSetAttribute((const void *)fft_mag_r16_kernel<LOGN, F, LISTS>,                      \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
        hipLaunchKernelGGL((fft_mag_r16_kernel<LOGN, F, LISTS>), dim3(n_frames), dim3((1 << LOGN) / 16), lds, \
                           stream, iq, window, tw, mag, n_frames, pre, counts, entries, cap, order);     \
    } while (0)
    if (fmt == 2) IRDM_LAUNCH_R16(2);
    else if (fmt == 1) IRDM_LAUNCH_R16(1);
    else IRDM_LAUNCH_R16(0);
#undef IRDM_LAUNCH_R16
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// K1 with the candidate lists of the band scan (see fft_mag_r16_kernel); 1 if this FFT size has no such kernel
int launch_fft_mag_lists(int log_n, int fmt, const void *iq, const float *window, const float2 *tw, float *mag,
                         int n_frames, const float *pre, unsigned *counts, ListEntry *entries, int cap,
                         hipStream_t stream, unsigned long long *kclk, int order)
{
    if (n_frames <= 0) return 0;
    if (fmt < 0 || fmt > 2 || log_n < 12 || log_n > 14) return 1;
    if (log_n == 13) return launch_p32<13, true>(fmt, iq, window, tw, mag, n_frames, pre, counts, entries, cap, kclk, stream, order);
    if (log_n == 14) return launch_p32<14, true>(fmt, iq, window, tw, mag, n_frames, pre, counts, entries, cap, kclk, stream, order);
    return launch_r16<12, true>(fmt, iq, window, tw, mag, n_frames, pre, counts, entries, cap, stream, order);
}

int launch_fft_mag(int log_n, int fmt, const void *iq, const float *window, const float2 *tw,
                   float *mag, int n_frames, hipStream_t stream, unsigned long long *kclk, int order)
{
    if (n_frames <= 0) return 0;
    const int grid = n_frames < 4096 ? n_frames : 4096;
    const int f = fmt;
    if (f < 0 || f > 2) return -1;
    if (log_n == 13) return launch_p32<13, false>(fmt, iq, window, tw, mag, n_frames, nullptr, nullptr, nullptr, 0, kclk, stream, order);
    if (log_n == 14) return launch_p32<14, false>(fmt, iq, window, tw, mag, n_frames, nullptr, nullptr, nullptr, 0, kclk, stream, order);
    if (log_n == 12) return launch_r16<12, false>(fmt, iq, window, tw, mag, n_frames, nullptr, nullptr, nullptr, 0, stream, order);
#define IRDM_LAUNCH_FFT_F(LOGN, NT, F)                                                         \
    do {                                                                                       \
        size_t lds = sizeof(float2) << LOGN;                                                   \
        (void)hipFuncSetAttribute((const void *)fft_mag_kernel<LOGN, NT, F>,                   \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
        hipLaunchKernelGGL((fft_mag_kernel<LOGN, NT, F>), dim3(grid), dim3(NT), lds,           \
                           stream, iq, window, tw, mag, n_frames, order);                      \
    } while (0)
#define IRDM_LAUNCH_FFT(LOGN, NT)                                                              \
    do {                                                                                       \
        if (f == 2) IRDM_LAUNCH_FFT_F(LOGN, NT, 2);                                            \
        else if (f == 1) IRDM_LAUNCH_FFT_F(LOGN, NT, 1);                                       \
        else IRDM_LAUNCH_FFT_F(LOGN, NT, 0);                                                   \
    } while (0)
    switch (log_n) {
    case 8:  IRDM_LAUNCH_FFT(8, 64); break;
    case 9:  IRDM_LAUNCH_FFT(9, 128); break;
    case 10: IRDM_LAUNCH_FFT(10, 256); break;
    case 11: IRDM_LAUNCH_FFT(11, 256); break;
    default: return -1;
    }
#undef IRDM_LAUNCH_FFT
#undef IRDM_LAUNCH_FFT_F
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---------------------------------------------------------------------------
// K2: detector scan.  One persistent workgroup walks the chunk's magnitude
// frames in order; each thread owns J = N/1024 contiguous bins.
//
// Per frame (burst_detect.c:689-698):
//   relative magnitude vs the 512-frame baseline (update_filters_pre, :426-434,
//   simd_relative_mag) -> per-bin threshold flags -> update_bursts (:458-469) ->
//   peak candidates = flags & old mask & range & DC notch (:522-552) ->
//   delete_gone_bursts (:490-518, forced baseline update) -> mask rebuild ->
//   create_new_bursts in descending-magnitude order (+ squelch, :556-632) ->
//   update_filters_post (:438-454): sum = (sum - old) + new; hist row <- mag.
//
// The dense per-bin work (division, flags, baseline recurrence) runs on all
// 1024 threads; the sparse list work on wave 0 / lane 0.
// ---------------------------------------------------------------------------
struct ScanShared {
    int n_cand[3];
    int n_act;
    int squelch;
    int flag_force, flag_deleted, flag_reset, flag_complex;
    unsigned n_gone, overflow;
    unsigned long long burst_id;
};

template <int J>
__global__ __launch_bounds__(kScanThreads) void detect_scan_kernel(
    DetParams P, DetState *__restrict__ st, float *__restrict__ sum_g, float *__restrict__ hist,
    const float *__restrict__ mag, int n_frames, GoneBurst *__restrict__ gone, int gone_cap,
    PeakCand *__restrict__ cand_a, PeakCand *__restrict__ cand_b)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int N = P.n;
    float *s_sum = reinterpret_cast<float *>(smem_raw);                       // N floats
    unsigned char *s_mask = reinterpret_cast<unsigned char *>(s_sum + N);      // N bytes
    unsigned short *s_cross = reinterpret_cast<unsigned short *>(s_mask + N);  // 1024
    ActiveBurst *s_act = reinterpret_cast<ActiveBurst *>(s_cross + kScanThreads);
    ScanShared &sh = *reinterpret_cast<ScanShared *>(s_act + kMaxActive);

    const int tid = threadIdx.x;
    const int b0 = tid * J;
    const float thr = P.threshold;
    const int half_bw = P.width / 2;
    const int dc = N / 2;

    // ---- load carried state ----
    for (int j = 0; j < J; j++) s_sum[b0 + j] = sum_g[b0 + j];
    int hist_idx = st->hist_idx, primed = st->primed;       // tracked redundantly by every thread
    uint64_t index = st->index;
    if (tid == 0) {
        sh.n_cand[0] = sh.n_cand[1] = sh.n_cand[2] = 0;
        sh.n_act = st->n_act;
        sh.squelch = st->squelch;
        sh.n_gone = st->n_gone;
        sh.overflow = st->overflow;
        sh.burst_id = st->burst_id;
        sh.flag_force = sh.flag_deleted = sh.flag_reset = sh.flag_complex = 0;
    }
    for (int i = tid; i < st->n_act; i += kScanThreads) s_act[i] = st->act[i];
    __syncthreads();
    // rebuild mask from carried bursts (update_burst_mask, :482-486)
    for (int j = 0; j < J; j++) s_mask[b0 + j] = 1;
    __syncthreads();
    for (int i = tid; i < sh.n_act; i += kScanThreads) {
        int lo = s_act[i].center_bin - half_bw, hi = s_act[i].center_bin + half_bw;
        if (lo < 0) lo = 0;
        if (hi >= N) hi = N - 1;
        for (int b = lo; b <= hi; b++) s_mask[b] = 0;
    }
    __syncthreads();

    float m[J], m_next[J];
    if (n_frames > 0)
        for (int j = 0; j < J; j++) m_next[j] = mag[b0 + j];

    // dense baseline update for this thread's bins (simd_baseline_update + memcpy, :441-452)
    auto baseline_update = [&](const float(&mm)[J]) {
        float *hrow = hist + (size_t)hist_idx * N + b0;
        float oldv[J];
        for (int j = 0; j < J; j++) oldv[j] = hrow[j];   // unconditional loads (no per-element branch)
        for (int j = 0; j < J; j++) {
            const float old = primed ? oldv[j] : 0.0f;   // rows not yet rewritten since a reset read 0 (:623-624)
            const float d = s_sum[b0 + j] - old;
            s_sum[b0 + j] = d + mm[j];
            hrow[j] = mm[j];
        }
        if (++hist_idx == kHistory) {
            primed = 1;
            hist_idx = 0;
        }
    };

    for (int f = 0; f < n_frames; f++, index += (uint64_t)N) {
        for (int j = 0; j < J; j++) m[j] = m_next[j];
        if (f + 1 < n_frames) {
            const float *nx = mag + (size_t)(f + 1) * N + b0;
            for (int j = 0; j < J; j++) m_next[j] = nx[j];
        }
        const int par = f % 3;

        if (!primed) {                       // update_filters_pre returns 0 (:427-428)
            if (sh.n_act == 0) baseline_update(m);      // n_act is 0 whenever !primed
            continue;
        }

        // ---- A: dense flags + peak candidates (old mask) ----
        unsigned cross = 0;
        for (int j = 0; j < J; j++) {
            const float base = s_sum[b0 + j];
            const float rel = base > 0 ? m[j] / base : 0.0f;
            if (rel > thr) {
                cross |= 1u << j;
                const int bin = b0 + j;
                if (s_mask[bin] && bin >= half_bw && bin < N - half_bw &&
                    !(bin >= dc - 3 && bin <= dc + 3)) {
                    const int slot = atomicAdd(&sh.n_cand[par], 1);
                    cand_a[slot].rel = rel;
                    cand_a[slot].bin = bin;
                }
            }
        }
        s_cross[tid] = (unsigned short)cross;
        __syncthreads();                                                    // S1

        const int n_cand = sh.n_cand[par];
        const int n_act0 = sh.n_act;
        if (tid == 0) sh.n_cand[(f + 2) % 3] = 0;   // re-armed two frames ahead: ordered by S1 of frame f+1

        if (n_cand == 0 && n_act0 == 0) {
            // quiet frame: only the squelch decay (:629-630) and the baseline update
            if (tid == 0 && sh.squelch > 0) sh.squelch--;
            baseline_update(m);
            continue;
        }

        // ---- B: update_bursts + expiry detection (wave 0) ----
        if (tid < 64) {
            int any_del = 0;
            for (int i = tid; i < n_act0; i += 64) {
                ActiveBurst &b = s_act[i];
                const int cb = b.center_bin;
                bool hit = false;
                for (int d = -1; d <= 1; d++) {
                    const int x = cb + d;
                    if (x < 0 || x > N - 1) continue;
                    hit |= (s_cross[x / J] >> (x % J)) & 1;
                }
                if (hit) b.last_active = index;
                const bool too_long = P.max_len > 0 && (b.last_active - b.start > (uint64_t)P.max_len);
                if (b.last_active + (uint64_t)P.post_len <= index || too_long) any_del = 1;
            }
            any_del = __any(any_del);
            if (tid == 0) sh.flag_complex = (any_del || n_cand > 0) ? 1 : 0;
        }
        __syncthreads();                                                    // S2

        if (!sh.flag_complex) {
            if (tid == 0 && sh.squelch > 0) sh.squelch--;
            // bursts are active -> no baseline update
            continue;
        }

        // ---- complex path: a burst starts or ends in this frame ----
        // P1: rank-sort the candidates, descending rel, ties by ascending bin
        //     (glibc qsort is a stable merge sort, burst_detect.c:551)
        for (int i = tid; i < n_cand; i += kScanThreads) {
            const PeakCand c = cand_a[i];
            int rank = 0;
            for (int k = 0; k < n_cand; k++) {
                const PeakCand o = cand_a[k];
                rank += (o.rel > c.rel || (o.rel == c.rel && o.bin < c.bin)) ? 1 : 0;
            }
            cand_b[rank] = c;
        }
        // P2: delete_gone_bursts (thread 0, order preserving)
        if (tid == 0) {
            int force = 0, w = 0, deleted = 0;
            for (int i = 0; i < n_act0; i++) {
                const ActiveBurst b = s_act[i];
                const bool too_long = P.max_len > 0 && (b.last_active - b.start > (uint64_t)P.max_len);
                if (too_long) force = 1;
                if (b.last_active + (uint64_t)P.post_len <= index || too_long) {
                    if ((int)sh.n_gone < gone_cap) {
                        GoneBurst g;
                        g.id = b.id; g.start = b.start; g.stop = index; g.last_active = b.last_active;
                        g.center_bin = b.center_bin; g.peak_rel = b.peak_rel; g.base_sum = b.base_sum;
                        g.pad = 0;
                        gone[sh.n_gone] = g;
                    } else {
                        sh.overflow = 1;
                    }
                    sh.n_gone++;
                    deleted = 1;
                } else {
                    if (w != i) s_act[w] = b;
                    w++;
                }
            }
            sh.n_act = w;
            sh.flag_force = force;
            sh.flag_deleted = deleted;
        }
        __syncthreads();                                                    // S3
        if (sh.flag_force) baseline_update(m);            // update_filters_post(d, 1), :516-517
        if (sh.flag_deleted) {
            for (int j = 0; j < J; j++) s_mask[b0 + j] = 1;
        }
        __syncthreads();                                                    // S4
        if (sh.flag_deleted) {
            for (int i = tid; i < sh.n_act; i += kScanThreads) {
                int lo = s_act[i].center_bin - half_bw, hi = s_act[i].center_bin + half_bw;
                if (lo < 0) lo = 0;
                if (hi >= N) hi = N - 1;
                for (int b = lo; b <= hi; b++) s_mask[b] = 0;
            }
        }
        __syncthreads();                                                    // S5
        // P4: create_new_bursts + squelch (thread 0)
        if (tid == 0) {
            unsigned char *vmask = s_mask;   // single thread: program order suffices (never volatile: FLAT sc0 sc1 loads)
            int na = sh.n_act;
            for (int i = 0; i < n_cand; i++) {
                const PeakCand c = cand_b[i];
                if (vmask[c.bin] == 0) continue;
                if (na < kMaxActive) {
                    ActiveBurst b;
                    b.id = sh.burst_id;
                    b.center_bin = c.bin;
                    b.peak_rel = c.rel;
                    b.start = index - (uint64_t)P.pre_len;
                    b.last_active = b.start;
                    b.base_sum = s_sum[c.bin];
                    b.pad = 0;
                    s_act[na] = b;
                } else {
                    sh.overflow = 1;
                }
                na++;
                sh.burst_id += 10;
                int lo = c.bin - half_bw, hi = c.bin + half_bw;
                if (lo < 0) lo = 0;
                if (hi >= N) hi = N - 1;
                for (int b = lo; b <= hi; b++) vmask[b] = 0;
            }
            int reset = 0;
            if (P.max_bursts > 0 && na > P.max_bursts) {                     // squelch (:594-627)
                const int lim = na < kMaxActive ? na : kMaxActive;
                for (int i = 0; i < lim; i++) {
                    const ActiveBurst b = s_act[i];
                    if (b.start != index - (uint64_t)P.pre_len) {
                        if ((int)sh.n_gone < gone_cap) {
                            GoneBurst g;
                            g.id = b.id; g.start = b.start; g.stop = index; g.last_active = b.last_active;
                            g.center_bin = b.center_bin; g.peak_rel = b.peak_rel; g.base_sum = b.base_sum;
                            g.pad = 0;
                            gone[sh.n_gone] = g;
                        } else {
                            sh.overflow = 1;
                        }
                        sh.n_gone++;
                    }
                }
                na = 0;
                sh.squelch += 3;
                if (sh.squelch >= 10) {
                    reset = 1;
                    sh.squelch = 0;
                    // the un-primed frames that follow skip the candidate pass and with it the re-arming of the
                    // rotating counters: clear them here, or this frame's count would be taken for real
                    // candidates when detection resumes 512 frames later
                    sh.n_cand[0] = sh.n_cand[1] = sh.n_cand[2] = 0;
                }
                sh.flag_deleted = 2;      // mask must be cleared to all-ones
            } else if (sh.squelch > 0) {
                sh.squelch--;
            }
            sh.n_act = na < kMaxActive ? na : kMaxActive;
            sh.flag_reset = reset;
        }
        __syncthreads();                                                    // S6
        if (sh.flag_deleted == 2) {
            for (int j = 0; j < J; j++) s_mask[b0 + j] = 1;
        }
        if (sh.flag_reset) {                                                // :621-626
            hist_idx = 0;
            primed = 0;
            for (int j = 0; j < J; j++) s_sum[b0 + j] = 0.0f;
        }
        const int n_act_end = sh.n_act;
        __syncthreads();                                                    // S7 (flags are re-armed below)
        if (tid == 0) sh.flag_force = sh.flag_deleted = sh.flag_reset = sh.flag_complex = 0;
        if (n_act_end == 0) baseline_update(m);           // update_filters_post(d, 0), :698
    }

    // ---- store carried state ----
    __syncthreads();
    for (int j = 0; j < J; j++) sum_g[b0 + j] = s_sum[b0 + j];
    for (int i = tid; i < sh.n_act; i += kScanThreads) st->act[i] = s_act[i];
    if (tid == 0) {
        st->index = index;
        st->burst_id = sh.burst_id;
        st->hist_idx = hist_idx;
        st->primed = primed;
        st->squelch = sh.squelch;
        st->n_act = sh.n_act;
        st->n_gone = sh.n_gone;
        st->overflow = sh.overflow;
    }
}

int launch_detect_scan(const DetParams &P, DetState *st, float *sum, float *hist, const float *mag,
                       int n_frames, GoneBurst *gone, int gone_cap, PeakCand *cand_a,
                       PeakCand *cand_b, hipStream_t stream)
{
    const int J = P.n / kScanThreads;
    const size_t lds = (size_t)P.n * 4 + (size_t)P.n + kScanThreads * 2 + sizeof(ActiveBurst) * kMaxActive + sizeof(ScanShared);
#define IRDM_LAUNCH_SCAN(JJ)                                                                   \
    do {                                                                                       \
        (void)hipFuncSetAttribute((const void *)detect_scan_kernel<JJ>,                              \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
        hipLaunchKernelGGL((detect_scan_kernel<JJ>), dim3(1), dim3(kScanThreads), lds, stream, \
                           P, st, sum, hist, mag, n_frames, gone, gone_cap, cand_a, cand_b);   \
    } while (0)
    switch (J) {
    case 1: IRDM_LAUNCH_SCAN(1); break;
    case 2: IRDM_LAUNCH_SCAN(2); break;
    case 4: IRDM_LAUNCH_SCAN(4); break;
    case 8: IRDM_LAUNCH_SCAN(8); break;
    case 16: IRDM_LAUNCH_SCAN(16); break;
    default: return -1;
    }
#undef IRDM_LAUNCH_SCAN
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace irdm
