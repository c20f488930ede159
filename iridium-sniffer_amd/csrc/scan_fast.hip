// scan_fast.hip -- detector scan, sparse form (burst_detect.c:426-632, :689-698).
//
// The state machine is sequential across frames, but almost none of its work is:
//   * while any burst is active the baseline is frozen (update_filters_post,
//     burst_detect.c:438-440), so "rel > threshold" only has to be evaluated
//     for the few bins that can possibly cross.  prefilter_kernel (all CUs)
//     lists, per frame, the bins with mag > 0.5*thr*baseline_ref; the scan
//     re-evaluates exactly those with the live baseline (mag / sum > thr,
//     correctly rounded division) -- one wavefront ("leader"), no barriers.
//   * while no burst is active every bin's running sum is updated each frame
//     (simd_baseline_update) -- dense, all 1024 threads, in runs of frames with
//     an empty prefilter list, each thread also re-checking its own bins
//     exactly (safety net: a crossing the list did not announce aborts the
//     kernel, the host restores the pre-chunk state and runs the dense scan).
//
// Exactness: the lists are only hints.  Every decision uses the same float
// operations as the reference on the live state; a stale or overflowing list
// is detected (CMD_VALIDATE / safety net / count > cap) and never trusted.
#include "common.hpp"
#include "types.hpp"
#include "kernels.hpp"

namespace irdm {

__global__ void prefilter_threshold_kernel(const float *__restrict__ sum, float thr,
                                           float *__restrict__ pre, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pre[i] = 0.5f * thr * sum[i];
}

// one workgroup per frame: bins with mag > pre[bin] -> (bin, mag), unordered
__global__ __launch_bounds__(256) void prefilter_kernel(const float *__restrict__ mag,
                                                        const float *__restrict__ pre, int n,
                                                        unsigned *__restrict__ counts,
                                                        ListEntry *__restrict__ entries, int n_frames)
{
    __shared__ int cnt;
    const int tid = threadIdx.x;
    for (int frame = blockIdx.x; frame < n_frames; frame += gridDim.x) {
        if (tid == 0) cnt = 0;
        __syncthreads();
        const float4 *m4 = reinterpret_cast<const float4 *>(mag + (size_t)frame * n);
        const float4 *p4 = reinterpret_cast<const float4 *>(pre);
        ListEntry *out = entries + (size_t)frame * kListCap;
        for (int q = tid; q < n / 4; q += 256) {
            const float4 m = m4[q], p = p4[q];
            const float mv[4] = { m.x, m.y, m.z, m.w }, pv[4] = { p.x, p.y, p.z, p.w };
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (mv[u] > pv[u]) {
                    const int slot = atomicAdd(&cnt, 1);
                    if (slot < kListCap) {
                        out[slot].bin = 4 * q + u;
                        out[slot].mag = mv[u];
                    }
                }
            }
        }
        __syncthreads();
        if (tid == 0) counts[frame] = (unsigned)cnt;
        __syncthreads();
    }
}

int launch_prefilter(const float *sum, float thr, float *pre, const float *mag, int n,
                     unsigned *counts, ListEntry *entries, int n_frames, hipStream_t stream)
{
    if (n_frames <= 0) return 0;
    hipLaunchKernelGGL(prefilter_threshold_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, sum, thr, pre, n);
    const int grid = n_frames < 8192 ? n_frames : 8192;
    hipLaunchKernelGGL(prefilter_kernel, dim3(grid), dim3(256), 0, stream, mag, pre, n, counts, entries, n_frames);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

enum { CMD_EXIT = 0, CMD_BULK = 1, CMD_VALIDATE = 2, CMD_ZERO = 3 };
constexpr int kFastMaxActive = 512;
constexpr int kStageFrames = 32;
constexpr int kStageEntries = 64;

struct FastShared {
    int cmd, f0, run, detect, hist_idx, primed;
    int abort;
};

template <int J>
__global__ __launch_bounds__(kScanThreads) void detect_scan_fast_kernel(
    DetParams P, DetState *__restrict__ st, float *__restrict__ sum_g, float *__restrict__ hist,
    const float *__restrict__ mag, int n_frames, const unsigned *__restrict__ counts,
    const ListEntry *__restrict__ entries, const float *__restrict__ pre,
    GoneBurst *__restrict__ gone, int gone_cap, int *__restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int N = P.n;
    float *s_sum = reinterpret_cast<float *>(smem_raw);                                   // N
    unsigned char *s_mask = reinterpret_cast<unsigned char *>(s_sum + N);                 // N bytes
    unsigned *s_bits = reinterpret_cast<unsigned *>(s_mask + N);                          // N/32
    ActiveBurst *s_act = reinterpret_cast<ActiveBurst *>(s_bits + N / 32);                // kFastMaxActive
    PeakCand *s_cand = reinterpret_cast<PeakCand *>(s_act + kFastMaxActive);              // kListCap
    ListEntry *s_stage = reinterpret_cast<ListEntry *>(s_cand + kListCap);                // 32 x 64
    unsigned *s_stage_cnt = reinterpret_cast<unsigned *>(s_stage + kStageFrames * kStageEntries);
    FastShared &sh = *reinterpret_cast<FastShared *>(s_stage_cnt + kStageFrames);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int b0 = tid * J;
    const float thr = P.threshold;
    const int half_bw = P.width / 2;
    const int dc = N / 2;
    const uint64_t index0 = st->index;

    auto valid_bin = [&](int bin) {
        return bin >= half_bw && bin < N - half_bw && !(bin >= dc - 3 && bin <= dc + 3);
    };

    // ---- load carried state ----
    for (int j = 0; j < J; j++) {
        s_sum[b0 + j] = sum_g[b0 + j];
        s_mask[b0 + j] = 1;
    }
    for (int i = tid; i < N / 32; i += kScanThreads) s_bits[i] = 0;
    const int n_act_in = st->n_act;
    if (tid == 0) {
        sh.cmd = CMD_EXIT;
        sh.abort = n_act_in > kFastMaxActive ? 4 : 0;
    }
    for (int i = tid; i < n_act_in && i < kFastMaxActive; i += kScanThreads) s_act[i] = st->act[i];
    __syncthreads();
    for (int i = tid; i < n_act_in && i < kFastMaxActive; i += kScanThreads) {
        int lo = s_act[i].center_bin - half_bw, hi = s_act[i].center_bin + half_bw;
        if (lo < 0) lo = 0;
        if (hi >= N) hi = N - 1;
        for (int b = lo; b <= hi; b++) s_mask[b] = 0;
    }
    __syncthreads();

    // ---- commands every thread executes ----
    auto exec = [&](int cmd) {
        if (cmd == CMD_BULK) {
            // run consecutive baseline updates (simd_baseline_update + memcpy, burst_detect.c:441-452)
            const int f0 = sh.f0, run = sh.run, detect = sh.detect;
            int hidx = sh.hist_idx, prm = sh.primed;
            float s[J];
            for (int j = 0; j < J; j++) s[j] = s_sum[b0 + j];
            bool bad = false;
            for (int k = 0; k < run; k++) {
                const float *mrow = mag + (size_t)(f0 + k) * N + b0;
                float *hrow = hist + (size_t)hidx * N + b0;
                float m[J], old[J];
                for (int j = 0; j < J; j++) m[j] = mrow[j];
                for (int j = 0; j < J; j++) old[j] = prm ? hrow[j] : 0.0f;
                for (int j = 0; j < J; j++) {
                    if (detect) {
                        const float rel = s[j] > 0 ? m[j] / s[j] : 0.0f;
                        if (rel > thr && valid_bin(b0 + j)) bad = true;      // safety net
                    }
                    const float d = s[j] - old[j];
                    s[j] = d + m[j];
                    hrow[j] = m[j];
                }
                if (++hidx == kHistory) { prm = 1; hidx = 0; }
            }
            for (int j = 0; j < J; j++) s_sum[b0 + j] = s[j];
            if (bad) atomicOr(&sh.abort, 1);
        } else if (cmd == CMD_VALIDATE) {
            // the prefilter lists are complete only while pre[b] <= 0.9*thr*sum[b]
            const float *pr = pre + b0;
            bool bad = false;
            for (int j = 0; j < J; j++) {
                const float sv = s_sum[b0 + j];
                if (sv > 0 && pr[j] > 0.9f * thr * sv) bad = true;
            }
            if (bad) atomicOr(&sh.abort, 2);
        } else if (cmd == CMD_ZERO) {
            for (int j = 0; j < J; j++) s_sum[b0 + j] = 0.0f;
        }
    };

    if (tid >= 64) {
        // ---- workers ----
        while (true) {
            __syncthreads();
            const int cmd = sh.cmd;
            if (cmd == CMD_EXIT) break;
            exec(cmd);
            __syncthreads();
        }
    } else {
        // ---- leader wavefront: the sequential state machine ----
        int hist_idx = st->hist_idx, primed = st->primed, squelch = st->squelch;
        int n_act = n_act_in < kFastMaxActive ? n_act_in : kFastMaxActive;
        unsigned n_gone = st->n_gone;
        unsigned long long burst_id = st->burst_id;
        int abort_code = sh.abort;
        int stage_base = -1;
        volatile unsigned char *vmask = s_mask;
        const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));

        auto issue = [&](int cmd, int f0, int run, int detect) {
            if (lane == 0) {
                sh.cmd = cmd; sh.f0 = f0; sh.run = run; sh.detect = detect;
                sh.hist_idx = hist_idx; sh.primed = primed;
            }
            __syncthreads();
            exec(cmd);
            __syncthreads();
            if (cmd == CMD_BULK) {
                const int tot = hist_idx + run;
                if (tot >= kHistory) primed = 1;
                hist_idx = tot % kHistory;
            }
            abort_code |= sh.abort;
        };

        auto mask_range = [&](int cb) {
            int lo = cb - half_bw, hi = cb + half_bw;
            if (lo < 0) lo = 0;
            if (hi >= N) hi = N - 1;
            for (int b = lo + lane; b <= hi; b += 64) vmask[b] = 0;
        };
        auto mask_all_ones = [&]() {
            unsigned *m32 = reinterpret_cast<unsigned *>(s_mask);
            for (int i = lane; i < N / 4; i += 64) m32[i] = 0x01010101u;
        };
        auto push_gone = [&](const ActiveBurst &b, uint64_t stop, unsigned slot) {
            if ((int)slot < gone_cap) {
                GoneBurst g;
                g.id = b.id; g.start = b.start; g.stop = stop; g.last_active = b.last_active;
                g.center_bin = b.center_bin; g.peak_rel = b.peak_rel; g.base_sum = b.base_sum; g.pad = 0;
                gone[slot] = g;
            }
        };

        // stage the lists of kStageFrames frames (first kStageEntries entries each) into LDS
        auto ensure_staged = [&](int f) {
            if (stage_base >= 0 && f >= stage_base && f < stage_base + kStageFrames) return;
            stage_base = f;
            const int nf = n_frames - f < kStageFrames ? n_frames - f : kStageFrames;
            unsigned c = 0;
            if (lane < nf) c = counts[f + lane];
            if (lane < kStageFrames) s_stage_cnt[lane] = c;
            ListEntry tmp[kStageFrames];
#pragma unroll
            for (int k = 0; k < kStageFrames; k++) {
                const unsigned ck = __shfl(c, k);
                tmp[k].bin = 0; tmp[k].mag = 0.0f;
                if (k < nf && (unsigned)lane < ck) tmp[k] = entries[(size_t)(f + k) * kListCap + lane];
            }
#pragma unroll
            for (int k = 0; k < kStageFrames; k++) s_stage[k * kStageEntries + lane] = tmp[k];
            __builtin_amdgcn_wave_barrier();
        };

        // one frame of the state machine; returns with n_act etc. updated
        auto process_frame = [&](int f) {
            const uint64_t index = index0 + (uint64_t)f * N;
            ensure_staged(f);
            const int sf = f - stage_base;
            const int cnt = (int)s_stage_cnt[sf];
            if (cnt > kListCap) { abort_code |= 8; return; }

            // phase 1: exact re-evaluation of the listed bins
            int n_cand = 0;
            for (int base = 0; base < cnt; base += 64) {
                const int i = base + lane;
                ListEntry e;
                e.bin = 0; e.mag = 0.0f;
                if (i < cnt) e = base == 0 ? s_stage[sf * kStageEntries + lane]
                                           : entries[(size_t)f * kListCap + i];
                const float sv = s_sum[e.bin];
                const float rel = (i < cnt && sv > 0) ? e.mag / sv : 0.0f;       // simd_relative_mag
                const bool cross = i < cnt && rel > thr;
                if (cross) atomicOr(&s_bits[e.bin >> 5], 1u << (e.bin & 31));
                const bool cand = cross && vmask[e.bin] && valid_bin(e.bin);     // :522-548
                const unsigned long long cm = __ballot(cand);
                if (cand) {
                    PeakCand c;
                    c.rel = rel; c.bin = e.bin;
                    s_cand[n_cand + __popcll(cm & lt_mask)] = c;
                }
                n_cand += __popcll(cm);
            }
            __builtin_amdgcn_wave_barrier();

            // phase 2: update_bursts (:458-469) + expiry test (:498-505)
            bool any_del = false;
            for (int base = 0; base < n_act; base += 64) {
                const int i = base + lane;
                bool del = false;
                if (i < n_act) {
                    const int cb = s_act[i].center_bin;
                    bool hit = false;
                    for (int d = -1; d <= 1; d++) {
                        const int x = cb + d;
                        if (x >= 0 && x <= N - 1) hit |= (s_bits[x >> 5] >> (x & 31)) & 1u;
                    }
                    uint64_t la = s_act[i].last_active;
                    if (hit) { la = index; s_act[i].last_active = index; }
                    const bool too_long = P.max_len > 0 && (la - s_act[i].start > (uint64_t)P.max_len);
                    del = (la + (uint64_t)P.post_len <= index) || too_long;
                }
                any_del |= __any(del) != 0;
            }
            // clear the crossing bitmap (every listed bin's word)
            for (int base = 0; base < cnt; base += 64) {
                const int i = base + lane;
                if (i < cnt) {
                    const int bin = base == 0 ? s_stage[sf * kStageEntries + lane].bin
                                              : entries[(size_t)f * kListCap + i].bin;
                    s_bits[bin >> 5] = 0;
                }
            }
            __builtin_amdgcn_wave_barrier();

            if (n_cand == 0 && !any_del) {
                if (squelch > 0) squelch--;                                       // :629-630
                return;
            }

            // ---- a burst starts or ends in this frame ----
            // delete_gone_bursts (:490-518), order preserving
            bool force = false;
            int w = 0;
            for (int base = 0; base < n_act; base += 64) {
                const int i = base + lane;
                ActiveBurst b;
                bool valid = i < n_act, del = false;
                if (valid) {
                    b = s_act[i];
                    const bool too_long = P.max_len > 0 && (b.last_active - b.start > (uint64_t)P.max_len);
                    if (too_long) force = true;
                    del = (b.last_active + (uint64_t)P.post_len <= index) || too_long;
                }
                const unsigned long long dm = __ballot(del), km = __ballot(valid && !del);
                if (del) push_gone(b, index, n_gone + __popcll(dm & lt_mask));
                n_gone += __popcll(dm);
                __builtin_amdgcn_wave_barrier();
                if (valid && !del) s_act[w + __popcll(km & lt_mask)] = b;
                w += __popcll(km);
            }
            force = __any(force) != 0;
            const bool deleted = w != n_act;
            n_act = w;
            if (force) issue(CMD_BULK, f, 1, 0);                                  // update_filters_post(d, 1)
            if (deleted) {                                                        // update_burst_mask (:482-486)
                mask_all_ones();
                __builtin_amdgcn_wave_barrier();
                for (int i = 0; i < n_act; i++) mask_range(s_act[i].center_bin);
                __builtin_amdgcn_wave_barrier();
            }
            // create_new_bursts (:556-591): descending magnitude, ties by ascending bin, skipping
            // bins masked by bursts created earlier in the same frame == repeated arg-max
            while (true) {
                float br = -1.0f;
                int bb = 0x7fffffff;
                for (int k = lane; k < n_cand; k += 64) {
                    const PeakCand c = s_cand[k];
                    if (vmask[c.bin] && (c.rel > br || (c.rel == br && c.bin < bb))) { br = c.rel; bb = c.bin; }
                }
                for (int off = 32; off > 0; off >>= 1) {
                    const float orr = __shfl_xor(br, off);
                    const int ob = __shfl_xor(bb, off);
                    if (orr > br || (orr == br && ob < bb)) { br = orr; bb = ob; }
                }
                if (bb == 0x7fffffff) break;
                if (n_act < kFastMaxActive) {
                    if (lane == 0) {
                        ActiveBurst b;
                        b.id = burst_id;
                        b.center_bin = bb;
                        b.peak_rel = br;
                        b.start = index - (uint64_t)P.pre_len;
                        b.last_active = b.start;
                        b.base_sum = s_sum[bb];
                        b.pad = 0;
                        s_act[n_act] = b;
                    }
                } else {
                    abort_code |= 4;
                }
                n_act++;
                burst_id += 10;
                mask_range(bb);
                __builtin_amdgcn_wave_barrier();
                if (abort_code) break;
            }
            if (abort_code) return;
            // squelch (:594-631)
            if (P.max_bursts > 0 && n_act > P.max_bursts) {
                for (int base = 0; base < n_act; base += 64) {
                    const int i = base + lane;
                    ActiveBurst b;
                    bool out = false;
                    if (i < n_act) {
                        b = s_act[i];
                        out = b.start != index - (uint64_t)P.pre_len;
                    }
                    const unsigned long long om = __ballot(out);
                    if (out) push_gone(b, index, n_gone + __popcll(om & lt_mask));
                    n_gone += __popcll(om);
                }
                n_act = 0;
                mask_all_ones();
                squelch += 3;
                if (squelch >= 10) {
                    hist_idx = 0;
                    primed = 0;
                    squelch = 0;
                    issue(CMD_ZERO, 0, 0, 0);
                }
            } else if (squelch > 0) {
                squelch--;
            }
        };

        int f = 0;
        while (f < n_frames && !abort_code) {
            if (!primed) {
                // update_filters_pre returns 0 (:427-428): updates only, up to the priming frame
                int run = kHistory - hist_idx;
                if (run > n_frames - f) run = n_frames - f;
                issue(CMD_BULK, f, run, 0);
                f += run;
                continue;
            }
            if (n_act == 0) {
                // quiet: frames with an empty list are updated in bulk (safety net on)
                ensure_staged(f);
                const int sf = f - stage_base;
                unsigned c = 1;
                if (lane >= sf && lane < kStageFrames && f + (lane - sf) < n_frames) c = s_stage_cnt[lane];
                else if (lane < sf) c = 0;
                const unsigned long long nz = __ballot(c != 0) >> sf;
                int run = nz ? __builtin_ctzll(nz) : 64 - sf;
                if (run > 0) {
                    if (run > n_frames - f) run = n_frames - f;
                    issue(CMD_BULK, f, run, 1);
                    squelch = squelch > run ? squelch - run : 0;
                    f += run;
                    continue;
                }
                process_frame(f);
                if (abort_code) break;
                if (n_act == 0 && primed) issue(CMD_BULK, f, 1, 0);               // update_filters_post(d, 0)
                else if (!primed) issue(CMD_BULK, f, 1, 0);                       // after a squelch reset
                else issue(CMD_VALIDATE, 0, 0, 0);                                // quiet -> busy: lists must be complete
                f++;
                continue;
            }
            // busy: leader only
            const int hist_before = hist_idx;
            process_frame(f);
            if (abort_code) break;
            if (n_act == 0 || !primed) issue(CMD_BULK, f, 1, 0);
            else if (hist_idx != hist_before) issue(CMD_VALIDATE, 0, 0, 0);       // a forced update moved the baseline
            f++;
        }
        if (lane == 0) sh.cmd = CMD_EXIT;
        __syncthreads();

        if (lane == 0) {
            status[0] = abort_code | sh.abort;
            if (n_gone > (unsigned)gone_cap) st->overflow = 1;
            st->index = index0 + (uint64_t)n_frames * N;
            st->burst_id = burst_id;
            st->hist_idx = hist_idx;
            st->primed = primed;
            st->squelch = squelch;
            st->n_act = n_act;
            st->n_gone = n_gone;
        }
        for (int i = lane; i < n_act; i += 64) st->act[i] = s_act[i];
    }
    __syncthreads();
    for (int j = 0; j < J; j++) sum_g[b0 + j] = s_sum[b0 + j];
}

size_t scan_fast_lds_bytes(int n)
{
    return (size_t)n * 4 + (size_t)n + (size_t)n / 8 + sizeof(ActiveBurst) * kFastMaxActive +
           sizeof(PeakCand) * kListCap + sizeof(ListEntry) * kStageFrames * kStageEntries +
           sizeof(unsigned) * kStageFrames + sizeof(FastShared) + 16;
}

int launch_detect_scan_fast(const DetParams &P, DetState *st, float *sum, float *hist, const float *mag,
                            int n_frames, const unsigned *counts, const ListEntry *entries,
                            const float *pre, GoneBurst *gone, int gone_cap, int *status,
                            hipStream_t stream)
{
    const int J = P.n / kScanThreads;
    const size_t lds = scan_fast_lds_bytes(P.n);
#define IRDM_LAUNCH_FAST(JJ)                                                                     \
    do {                                                                                         \
        (void)hipFuncSetAttribute((const void *)detect_scan_fast_kernel<JJ>,                     \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
        hipLaunchKernelGGL((detect_scan_fast_kernel<JJ>), dim3(1), dim3(kScanThreads), lds,      \
                           stream, P, st, sum, hist, mag, n_frames, counts, entries, pre, gone,  \
                           gone_cap, status);                                                    \
    } while (0)
    switch (J) {
    case 1: IRDM_LAUNCH_FAST(1); break;
    case 2: IRDM_LAUNCH_FAST(2); break;
    case 4: IRDM_LAUNCH_FAST(4); break;
    case 8: IRDM_LAUNCH_FAST(8); break;
    case 16: IRDM_LAUNCH_FAST(16); break;
    default: return -1;
    }
#undef IRDM_LAUNCH_FAST
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace irdm
