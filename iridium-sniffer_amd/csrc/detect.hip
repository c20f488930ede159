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
                                                      float *__restrict__ mag, int n_frames)
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
            mag[base + i] = mag2(v);
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
                                                                       ListEntry *__restrict__ entries, int cap)
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
                put((p + N / 2) & (N - 1), mag2(v[q]));
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
                    put((k + N / 2) & (N - 1), mag2(d[q]));
                }
            }
            __syncthreads();
        }
        if (LISTS && t == 0) counts[frame] = (unsigned)s_cnt;
    }
}

int g_fft_force_radix2 = 0;   // test hook: 1 = always use the radix-2 LDS kernel

// K1 with the candidate lists of the band scan (see fft_mag_r16_kernel); 1 if this FFT size has no such kernel
int launch_fft_mag_lists(int log_n, int fmt, const void *iq, const float *window, const float2 *tw, float *mag,
                         int n_frames, const float *pre, unsigned *counts, ListEntry *entries, int cap,
                         hipStream_t stream)
{
    if (n_frames <= 0) return 0;
    if (fmt < 0 || fmt > 2 || log_n < 12 || log_n > 14 || g_fft_force_radix2) return 1;
#define IRDM_LAUNCH_R16L_F(LOGN, F)                                                            \
    do {                                                                                       \
        constexpr int NB_ = 1 << (LOGN - 8);                                                   \
        size_t lds = sizeof(float2) * ((size_t)256 * (NB_ + 1) > ((size_t)1 << LOGN)           \
                                           ? (size_t)256 * (NB_ + 1) : ((size_t)1 << LOGN));   \
        (void)hipFuncSetAttribute((const void *)fft_mag_r16_kernel<LOGN, F, true>,             \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
        hipLaunchKernelGGL((fft_mag_r16_kernel<LOGN, F, true>), dim3(n_frames), dim3((1 << LOGN) / 16), lds, \
                           stream, iq, window, tw, mag, n_frames, pre, counts, entries, cap);  \
    } while (0)
#define IRDM_LAUNCH_R16L(LOGN)                                                                 \
    do {                                                                                       \
        if (fmt == 2) IRDM_LAUNCH_R16L_F(LOGN, 2);                                             \
        else if (fmt == 1) IRDM_LAUNCH_R16L_F(LOGN, 1);                                        \
        else IRDM_LAUNCH_R16L_F(LOGN, 0);                                                      \
    } while (0)
    switch (log_n) {
    case 12: IRDM_LAUNCH_R16L(12); break;
    case 13: IRDM_LAUNCH_R16L(13); break;
    default: IRDM_LAUNCH_R16L(14); break;
    }
#undef IRDM_LAUNCH_R16L
#undef IRDM_LAUNCH_R16L_F
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_fft_mag(int log_n, int fmt, const void *iq, const float *window, const float2 *tw,
                   float *mag, int n_frames, hipStream_t stream)
{
    if (n_frames <= 0) return 0;
    int grid = n_frames < 4096 ? n_frames : 4096;
    const int f = fmt;
    if (f < 0 || f > 2) return -1;
#define IRDM_LAUNCH_FFT_F(LOGN, NT, F)                                                         \
    do {                                                                                       \
        size_t lds = sizeof(float2) << LOGN;                                                   \
        (void)hipFuncSetAttribute((const void *)fft_mag_kernel<LOGN, NT, F>,                   \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
        hipLaunchKernelGGL((fft_mag_kernel<LOGN, NT, F>), dim3(grid), dim3(NT), lds,           \
                           stream, iq, window, tw, mag, n_frames);                             \
    } while (0)
#define IRDM_LAUNCH_FFT(LOGN, NT)                                                              \
    do {                                                                                       \
        if (f == 2) IRDM_LAUNCH_FFT_F(LOGN, NT, 2);                                            \
        else if (f == 1) IRDM_LAUNCH_FFT_F(LOGN, NT, 1);                                       \
        else IRDM_LAUNCH_FFT_F(LOGN, NT, 0);                                                   \
    } while (0)
#define IRDM_LAUNCH_R16_F(LOGN, F)                                                             \
    do {                                                                                       \
        constexpr int NB_ = 1 << (LOGN - 8);                                                   \
        size_t lds = sizeof(float2) * ((size_t)256 * (NB_ + 1) > ((size_t)1 << LOGN)           \
                                           ? (size_t)256 * (NB_ + 1) : ((size_t)1 << LOGN));   \
        (void)hipFuncSetAttribute((const void *)fft_mag_r16_kernel<LOGN, F, false>,            \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
        hipLaunchKernelGGL((fft_mag_r16_kernel<LOGN, F, false>), dim3(n_frames), dim3((1 << LOGN) / 16), lds, \
                           stream, iq, window, tw, mag, n_frames, (const float *)nullptr,      \
                           (unsigned *)nullptr, (ListEntry *)nullptr, 0);                      \
    } while (0)
#define IRDM_LAUNCH_R16(LOGN)                                                                  \
    do {                                                                                       \
        if (f == 2) IRDM_LAUNCH_R16_F(LOGN, 2);                                                \
        else if (f == 1) IRDM_LAUNCH_R16_F(LOGN, 1);                                           \
        else IRDM_LAUNCH_R16_F(LOGN, 0);                                                       \
    } while (0)
    if (!g_fft_force_radix2) {
        switch (log_n) {
        case 12: IRDM_LAUNCH_R16(12); return hipGetLastError() == hipSuccess ? 0 : -1;
        case 13: IRDM_LAUNCH_R16(13); return hipGetLastError() == hipSuccess ? 0 : -1;
        case 14: IRDM_LAUNCH_R16(14); return hipGetLastError() == hipSuccess ? 0 : -1;
        default: break;
        }
    }
#undef IRDM_LAUNCH_R16
#undef IRDM_LAUNCH_R16_F
    switch (log_n) {
    case 8:  IRDM_LAUNCH_FFT(8, 64); break;
    case 9:  IRDM_LAUNCH_FFT(9, 128); break;
    case 10: IRDM_LAUNCH_FFT(10, 256); break;
    case 11: IRDM_LAUNCH_FFT(11, 256); break;
    case 12: IRDM_LAUNCH_FFT(12, 512); break;
    case 13: IRDM_LAUNCH_FFT(13, 512); break;
    case 14: IRDM_LAUNCH_FFT(14, 1024); break;
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
