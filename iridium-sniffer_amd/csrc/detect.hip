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
            float2 x;
            if (FMT == 2) {
                x = reinterpret_cast<const float2 *>(iq)[base + i];
            } else {
                // simd_convert_i8_cf (simd_generic.c:147-153): int8 / 128.0f
                char2 v = reinterpret_cast<const char2 *>(iq)[base + i];
                x = make_float2((float)v.x / 128.0f, (float)v.y / 128.0f);
            }
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

int launch_fft_mag(int log_n, int fmt, const void *iq, const float *window, const float2 *tw,
                   float *mag, int n_frames, hipStream_t stream)
{
    if (n_frames <= 0) return 0;
    int grid = n_frames < 4096 ? n_frames : 4096;
    const int f = fmt == 2 ? 2 : 0;
#define IRDM_LAUNCH_FFT(LOGN, NT)                                                              \
    do {                                                                                       \
        size_t lds = sizeof(float2) << LOGN;                                                   \
        if (f == 2) {                                                                          \
            (void)hipFuncSetAttribute((const void *)fft_mag_kernel<LOGN, NT, 2>,                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
            hipLaunchKernelGGL((fft_mag_kernel<LOGN, NT, 2>), dim3(grid), dim3(NT), lds,       \
                               stream, iq, window, tw, mag, n_frames);                         \
        } else {                                                                               \
            (void)hipFuncSetAttribute((const void *)fft_mag_kernel<LOGN, NT, 0>,                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
            hipLaunchKernelGGL((fft_mag_kernel<LOGN, NT, 0>), dim3(grid), dim3(NT), lds,       \
                               stream, iq, window, tw, mag, n_frames);                         \
        }                                                                                      \
    } while (0)
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
