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

#ifdef IRDM_SCAN_PROFILE
#define IRDM_TICK() wall_clock64()
#else
#define IRDM_TICK() 0ll
#endif

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

// exclusive scan of min(count, cap) over frames -> offsets into the compact entry stream (one workgroup)
__global__ __launch_bounds__(1024) void list_offsets_kernel(const unsigned *__restrict__ counts,
                                                            unsigned *__restrict__ goff, int n_frames)
{
    __shared__ unsigned part[1024];
    const int tid = threadIdx.x;
    const int per = (n_frames + 1023) / 1024;
    const int lo = tid * per, hi = lo + per < n_frames ? lo + per : n_frames;
    unsigned s = 0;
    for (int i = lo; i < hi; i++) s += counts[i] < (unsigned)kListCap ? counts[i] : (unsigned)kListCap;
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        unsigned run = 0;
        for (int i = 0; i < 1024; i++) { const unsigned v = part[i]; part[i] = run; run += v; }
        goff[n_frames] = run;
    }
    __syncthreads();
    unsigned run = part[tid];
    for (int i = lo; i < hi; i++) {
        goff[i] = run;
        run += counts[i] < (unsigned)kListCap ? counts[i] : (unsigned)kListCap;
    }
}

__global__ __launch_bounds__(256) void list_compact_kernel(const unsigned *__restrict__ counts,
                                                           const unsigned *__restrict__ goff,
                                                           const ListEntry *__restrict__ entries,
                                                           ListEntry *__restrict__ compact, int n_frames)
{
    for (int frame = blockIdx.x; frame < n_frames; frame += gridDim.x) {
        const unsigned c = counts[frame] < (unsigned)kListCap ? counts[frame] : (unsigned)kListCap;
        const ListEntry *src = entries + (size_t)frame * kListCap;
        ListEntry *dst = compact + goff[frame];
        for (unsigned i = threadIdx.x; i < c; i += 256) dst[i] = src[i];
    }
}

int launch_prefilter(const float *sum, float thr, float *pre, const float *mag, int n,
                     unsigned *counts, ListEntry *entries, unsigned *goff, ListEntry *compact,
                     int n_frames, hipStream_t stream)
{
    if (n_frames <= 0) return 0;
    hipLaunchKernelGGL(prefilter_threshold_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, sum, thr, pre, n);
    const int grid = n_frames < 8192 ? n_frames : 8192;
    hipLaunchKernelGGL(prefilter_kernel, dim3(grid), dim3(256), 0, stream, mag, pre, n, counts, entries, n_frames);
    hipLaunchKernelGGL(list_offsets_kernel, dim3(1), dim3(1024), 0, stream, counts, goff, n_frames);
    hipLaunchKernelGGL(list_compact_kernel, dim3(grid), dim3(256), 0, stream, counts, goff, entries, compact, n_frames);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

enum { CMD_EXIT = 0, CMD_BULK = 1, CMD_VALIDATE = 2, CMD_ZERO = 3 };
constexpr int kFastMaxActive = 64;      // active bursts live in the leader's lanes; more -> dense fallback
constexpr int kStageCap = 4096;         // list entries staged in LDS per batch
constexpr int kStageFrames = 63;
constexpr int kGoneLds = 256;           // gone records buffered in LDS between flushes        // frames per batch (lane k <-> frame k, lane k+1 holds its end offset)

struct FastShared {
    int cmd, f0, run, detect, hist_idx, primed;
    int abort;
};

constexpr int kFastThreads = 256;   // 4 wavefronts: one per SIMD, the full VGPR budget each (no spills)

enum { S_TOP = 0, S_CPLX_B = 1, S_FRAME_END = 2 };

// Q = float4 groups per thread; thread t owns bins (q*256 + t)*4 .. +3, so every global
// access of a dense command is a fully coalesced 1 KiB row segment per wavefront.
//
// Control structure: wavefront 0 ("leader") runs the sequential state machine and, whenever
// dense per-bin work is needed, publishes ONE command; all four wavefronts execute it between
// two barriers.  There is exactly one call site of the command body and one of the leader
// step, so everything inlines and the leader's state lives in registers.
template <int Q>
__global__ __launch_bounds__(kFastThreads) void detect_scan_fast_kernel(
    DetParams P, DetState *__restrict__ st, float *__restrict__ sum_g, float *__restrict__ hist,
    const float *__restrict__ mag, int n_frames, const unsigned *__restrict__ counts,
    const unsigned *__restrict__ goff, const ListEntry *__restrict__ compact,
    const float *__restrict__ pre, GoneBurst *__restrict__ gone, int gone_cap, int *__restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int J = 4 * Q;
    const int N = P.n;
    float *s_sum = reinterpret_cast<float *>(smem_raw);                                   // N
    unsigned char *s_mask = reinterpret_cast<unsigned char *>(s_sum + N);                 // N bytes
    ActiveBurst *s_act = reinterpret_cast<ActiveBurst *>(s_mask + N);                     // kFastMaxActive
    PeakCand *s_cand = reinterpret_cast<PeakCand *>(s_act + kFastMaxActive);              // kListCap
    ListEntry *s_ent = reinterpret_cast<ListEntry *>(s_cand + kListCap);                  // kStageCap
    unsigned char *s_flag = reinterpret_cast<unsigned char *>(s_ent + kStageCap);         // kStageCap
    GoneBurst *s_gone = reinterpret_cast<GoneBurst *>(s_flag + kStageCap);                // kGoneLds
    FastShared &sh = *reinterpret_cast<FastShared *>(s_gone + kGoneLds);
    // NOTE: never `volatile` here -- a volatile access through a generic pointer compiles to a
    // system-coherent FLAT load (microseconds); LDS ops of one wavefront execute in order, so a
    // compiler barrier between the write and read phases is all the ordering that is needed.
    unsigned char *vmask = s_mask;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const float thr = P.threshold;
    const int half_bw = P.width / 2;
    const int dc = N / 2;
    const uint64_t index0 = st->index;
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));

#define WAVE_SYNC() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)
#define BIN_OF(j) ((((j) >> 2) * kFastThreads + tid) * 4 + ((j) & 3))
#define VALID_BIN(b) ((b) >= half_bw && (b) < N - half_bw && !((b) >= dc - 3 && (b) <= dc + 3))
#define MASK_RANGE(cb)                                                   \
    do {                                                                 \
        int lo_ = (cb) - half_bw, hi_ = (cb) + half_bw;                  \
        if (lo_ < 0) lo_ = 0;                                            \
        if (hi_ >= N) hi_ = N - 1;                                       \
        for (int b_ = lo_ + lane; b_ <= hi_; b_ += 64) vmask[b_] = 0;    \
    } while (0)
#define MASK_ALL_ONES()                                                  \
    do {                                                                 \
        unsigned *m32_ = reinterpret_cast<unsigned *>(s_mask);           \
        for (int i_ = lane; i_ < N / 4; i_ += 64) m32_[i_] = 0x01010101u;\
    } while (0)
#define PUSH_GONE(b, stopv, slot)                                                                  \
    do {                                                                                           \
        const unsigned ls_ = (slot) - gone_base;                                                   \
        if (ls_ < (unsigned)kGoneLds) {                                                            \
            GoneBurst g_;                                                                          \
            g_.id = (b).id; g_.start = (b).start; g_.stop = (stopv); g_.last_active = (b).last_active; \
            g_.center_bin = (b).center_bin; g_.peak_rel = (b).peak_rel; g_.base_sum = (b).base_sum;    \
            g_.pad = 0;                                                                            \
            s_gone[ls_] = g_;                                                                      \
        } else {                                                                                   \
            abort_code |= 16;                                                                      \
        }                                                                                          \
    } while (0)
    // gone records collect in LDS (no global store, hence no vmcnt wait, inside the frame loop)
#define FLUSH_GONE()                                                                               \
    do {                                                                                           \
        const unsigned cnt_ = n_gone - gone_base;                                                  \
        for (unsigned i_ = lane; i_ < cnt_ && i_ < (unsigned)kGoneLds; i_ += 64)                   \
            if ((int)(gone_base + i_) < gone_cap) gone[gone_base + i_] = s_gone[i_];               \
        gone_base = n_gone;                                                                        \
    } while (0)

    // ---- load carried state ----
#pragma unroll
    for (int j = 0; j < J; j++) {
        s_sum[BIN_OF(j)] = sum_g[BIN_OF(j)];
        s_mask[BIN_OF(j)] = 1;
    }
    const int n_act_in = st->n_act;
    if (tid == 0) {
        sh.cmd = CMD_EXIT;
        sh.abort = n_act_in > kFastMaxActive ? 4 : 0;
    }
    for (int i = tid; i < n_act_in && i < kFastMaxActive; i += kFastThreads) s_act[i] = st->act[i];
    __syncthreads();
    for (int i = tid; i < n_act_in && i < kFastMaxActive; i += kFastThreads) {
        int lo = s_act[i].center_bin - half_bw, hi = s_act[i].center_bin + half_bw;
        if (lo < 0) lo = 0;
        if (hi >= N) hi = N - 1;
        for (int b = lo; b <= hi; b++) s_mask[b] = 0;
    }
    __syncthreads();

    // ---- leader state (registers of wavefront 0) ----
    int hist_idx = st->hist_idx, primed = st->primed, squelch = st->squelch;
    int n_act = n_act_in < kFastMaxActive ? n_act_in : kFastMaxActive;
    unsigned n_gone = st->n_gone;
    unsigned gone_base = n_gone;
    unsigned long long burst_id = st->burst_id;
    int abort_code = sh.abort;
    int r_cb = 0;                          // lane i mirrors active burst i
    uint64_t r_la = 0, r_start = 0;
    if (lane < n_act) {
        r_cb = s_act[lane].center_bin;
        r_la = s_act[lane].last_active;
        r_start = s_act[lane].start;
    }
    int sb = 0, snf = 0;                   // staged batch: frames [sb, sb+snf)
    unsigned r_off = 0;                    // lane k: offset of frame sb+k in s_ent (lane snf: total)
    bool flags_valid = false;
    int f = 0, state = S_TOP;
    long long t_cmd[4] = {0, 0, 0, 0}, t_lead = 0, t_s1 = 0, t_s2 = 0, t_s3 = 0, t_s4 = 0, t_cal = 0, t_cA = 0, t_cB = 0, t_fe = 0;
    int n_cmd[4] = {0, 0, 0, 0}, n_cplx = 0, n_sparse = 0;
    const long long t_begin = IRDM_TICK();
    int e0 = 0, e1 = 0, n_cand = 0, hist_before = 0;
    bool any_cand = false, was_quiet = false;

    for (;;) {
        const long long t_l0 = IRDM_TICK();
        if (tid < 64) {
            // ================= leader step: run until a dense command is needed =================
            int cmd = -1, c_f0 = 0, c_run = 0, c_detect = 0;
            while (cmd < 0) {
                if (abort_code || (state == S_TOP && f >= n_frames)) { cmd = CMD_EXIT; break; }
                if (state == S_TOP) {
                    const long long ta = IRDM_TICK();
                    if (!primed) {
                        // update_filters_pre returns 0 (:427-428): updates only, up to the priming frame
                        int run = kHistory - hist_idx;
                        if (run > n_frames - f) run = n_frames - f;
                        cmd = CMD_BULK; c_f0 = f; c_run = run; c_detect = 0;
                        f += run;
                        break;
                    }
                    if (f < sb || f >= sb + snf) {
                        // ---- stage the compact lists of up to kStageFrames frames starting at f ----
                        const int nf = n_frames - f < kStageFrames ? n_frames - f : kStageFrames;
                        unsigned g = 0, c = 0;
                        if (lane <= nf) g = goff[f + lane];
                        if (lane < nf) c = counts[f + lane];
                        if (__any(c > (unsigned)kListCap)) { abort_code |= 8; continue; }
                        const unsigned g0 = (unsigned)__builtin_amdgcn_readfirstlane((int)g);
                        r_off = g - g0;
                        const unsigned long long fit =
                            __ballot(lane >= 1 && lane <= nf && r_off <= (unsigned)kStageCap);
                        snf = __popcll(fit);
                        sb = f;
                        const int total = __builtin_amdgcn_readlane((int)r_off, snf);
                        const ListEntry *src = compact + g0;
                        for (int i = lane; i < total; i += 64 * 8) {
                            ListEntry t[8];
#pragma unroll
                            for (int u = 0; u < 8; u++)
                                if (i + 64 * u < total) t[u] = src[i + 64 * u];
#pragma unroll
                            for (int u = 0; u < 8; u++)
                                if (i + 64 * u < total) s_ent[i + 64 * u] = t[u];
                        }
                        WAVE_SYNC();
                        flags_valid = false;
                    }
                    const int k0 = f - sb;
                    if (n_act == 0) {
                        // quiet: frames with an empty list are updated in bulk, every thread re-checking
                        // its own bins exactly (safety net)
                        const unsigned nxt = __shfl_down(r_off, 1);
                        const bool nonempty = lane >= k0 && lane < snf && nxt != r_off;
                        const unsigned long long nz = __ballot(nonempty) >> k0;
                        const int run = nz ? __builtin_ctzll(nz) : snf - k0;
                        if (run > 0) {
                            cmd = CMD_BULK; c_f0 = f; c_run = run; c_detect = 1;
                            squelch = squelch > run ? squelch - run : 0;
                            f += run;
                            break;
                        }
                    }
                    const long long tb = IRDM_TICK();
                    t_s1 += tb - ta;
                    e0 = __builtin_amdgcn_readlane((int)r_off, k0);
                    e1 = __builtin_amdgcn_readlane((int)r_off, k0 + 1);
                    if (!flags_valid) {
                        // exact threshold test of the staged entries from this frame on
                        // (simd_relative_mag + `> threshold`)
                        const int total = __builtin_amdgcn_readlane((int)r_off, snf);
                        for (int i = e0 + lane; i < total; i += 64) {
                            const ListEntry e = s_ent[i];
                            const float sv = s_sum[e.bin];
                            const float rel = sv > 0 ? e.mag / sv : 0.0f;
                            s_flag[i] = rel > thr ? 1 : 0;
                        }
                        WAVE_SYNC();
                        flags_valid = true;
                    }
                    // ---- one frame, sparse ----
                    const long long tc = IRDM_TICK();
                    t_s2 += tc - tb;
                    n_sparse++;
                    const uint64_t index = index0 + (uint64_t)f * N;
                    was_quiet = n_act == 0;
                    hist_before = hist_idx;
                    any_cand = false;
                    for (int base = e0; base < e1; base += 64) {
                        const int i = base + lane;
                        bool cross = false;
                        int bin = -100;
                        if (i < e1) {
                            cross = s_flag[i] != 0;
                            bin = s_ent[i].bin;
                        }
                        // update_bursts (:458-469): a crossing bin within +-1 of an active burst's centre
                        for (int j = 0; j < n_act; j++) {
                            const int cb = __builtin_amdgcn_readlane(r_cb, j);
                            const bool near = cross && bin >= cb - 1 && bin <= cb + 1;
                            if (__any(near) && lane == j) r_la = index;
                        }
                        // peak candidates (:522-548): crossing, not under the previous frame's mask, in range
                        const bool cand = cross && vmask[bin < 0 ? 0 : bin] && VALID_BIN(bin);
                        any_cand |= __any(cand) != 0;
                    }
                    bool del = false;
                    if (lane < n_act) {
                        const bool too_long = P.max_len > 0 && (r_la - r_start > (uint64_t)P.max_len);
                        del = (r_la + (uint64_t)P.post_len <= index) || too_long;
                    }
                    const long long td = IRDM_TICK();
                    t_s3 += td - tc;
                    { const long long te = IRDM_TICK(); t_cal += te - td; }
                    if (!any_cand && !__any(del)) {
                        if (squelch > 0) squelch--;                               // :629-630
                        state = S_FRAME_END;
                        continue;
                    }
                    n_cplx++;
                    const long long tA0 = IRDM_TICK();
                    // ---- a burst ends or may start in this frame (burst_detect.c:490-632), part A ----
                    if (lane < n_act) s_act[lane].last_active = r_la;
                    WAVE_SYNC();
                    n_cand = 0;
                    if (any_cand) {
                        for (int base = e0; base < e1; base += 64) {
                            const int i = base + lane;
                            bool cand = false;
                            PeakCand c;
                            c.rel = 0.0f; c.bin = 0;
                            if (i < e1 && s_flag[i]) {
                                const ListEntry e = s_ent[i];
                                cand = vmask[e.bin] && VALID_BIN(e.bin);
                                c.bin = e.bin;
                                c.rel = e.mag / s_sum[e.bin];
                            }
                            const unsigned long long cm = __ballot(cand);
                            if (cand) s_cand[n_cand + __popcll(cm & lt_mask)] = c;
                            n_cand += __popcll(cm);
                        }
                        WAVE_SYNC();
                    }
                    // delete_gone_bursts (:490-518), order preserving (n_act <= 64: one pass)
                    bool force = false;
                    {
                        ActiveBurst b;
                        const bool valid = lane < n_act;
                        bool dl = false;
                        if (valid) {
                            b = s_act[lane];
                            const bool too_long = P.max_len > 0 && (b.last_active - b.start > (uint64_t)P.max_len);
                            if (too_long) force = true;
                            dl = (b.last_active + (uint64_t)P.post_len <= index) || too_long;
                        }
                        const unsigned long long dm = __ballot(dl), km = __ballot(valid && !dl);
                        if (dl) PUSH_GONE(b, index, n_gone + __popcll(dm & lt_mask));
                        n_gone += __popcll(dm);
                        WAVE_SYNC();
                        if (valid && !dl) s_act[__popcll(km & lt_mask)] = b;
                        WAVE_SYNC();
                        force = __any(force) != 0;
                        const int w = __popcll(km);
                        if (w != n_act) {
                            // update_burst_mask (:482-486): only the deleted bursts' ranges can change:
                            // set them to 1, then re-apply every surviving burst's range
                            unsigned long long d2 = dm;
                            while (d2) {
                                const int src = __builtin_ctzll(d2);
                                d2 &= d2 - 1;
                                const int cbd = __builtin_amdgcn_readlane(b.center_bin, src);
                                int lo_ = cbd - half_bw, hi_ = cbd + half_bw;
                                if (lo_ < 0) lo_ = 0;
                                if (hi_ >= N) hi_ = N - 1;
                                for (int b_ = lo_ + lane; b_ <= hi_; b_ += 64) vmask[b_] = 1;
                            }
                            n_act = w;
                            WAVE_SYNC();
                            for (int i = 0; i < n_act; i++) MASK_RANGE(s_act[i].center_bin);
                            WAVE_SYNC();
                        }
                    }
                    t_cA += IRDM_TICK() - tA0;
                    state = S_CPLX_B;
                    if (force) {                                                  // update_filters_post(d, 1)
                        cmd = CMD_BULK; c_f0 = f; c_run = 1; c_detect = 0;
                        break;
                    }
                    continue;
                }
                if (state == S_CPLX_B) {
                    const long long tB0 = IRDM_TICK();
                    const uint64_t index = index0 + (uint64_t)f * N;
                    // create_new_bursts (:556-591): descending magnitude, ties by ascending bin, skipping
                    // bins masked by bursts created earlier in the same frame == repeated arg-max
                    while (n_cand > 0) {
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
                        if (n_act >= kFastMaxActive) { abort_code |= 4; break; }
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
                        n_act++;
                        burst_id += 10;
                        MASK_RANGE(bb);
                        WAVE_SYNC();
                    }
                    if (abort_code) continue;
                    bool reset = false;
                    if (P.max_bursts > 0 && n_act > P.max_bursts) {               // squelch (:594-631)
                        ActiveBurst b;
                        bool out = false;
                        if (lane < n_act) {
                            b = s_act[lane];
                            out = b.start != index - (uint64_t)P.pre_len;
                        }
                        const unsigned long long om = __ballot(out);
                        if (out) PUSH_GONE(b, index, n_gone + __popcll(om & lt_mask));
                        n_gone += __popcll(om);
                        n_act = 0;
                        MASK_ALL_ONES();
                        squelch += 3;
                        if (squelch >= 10) {
                            hist_idx = 0;
                            primed = 0;
                            squelch = 0;
                            reset = true;
                        }
                    } else if (squelch > 0) {
                        squelch--;
                    }
                    WAVE_SYNC();
                    if (lane < n_act) {
                        r_cb = s_act[lane].center_bin;
                        r_la = s_act[lane].last_active;
                        r_start = s_act[lane].start;
                    }
                    state = S_FRAME_END;
                    t_cB += IRDM_TICK() - tB0;
                    if (reset) { cmd = CMD_ZERO; break; }
                    continue;
                }
                // S_FRAME_END: update_filters_post(d, 0) (:698)
                state = S_TOP;
                if (n_gone - gone_base > (unsigned)(kGoneLds - 80)) { WAVE_SYNC(); FLUSH_GONE(); }
                if (n_act == 0 || !primed) { cmd = CMD_BULK; c_f0 = f; c_run = 1; c_detect = 0; }
                else if (was_quiet || hist_idx != hist_before) { cmd = CMD_VALIDATE; }
                f++;
            }
            if (lane == 0) {
                sh.cmd = cmd; sh.f0 = c_f0; sh.run = c_run; sh.detect = c_detect;
                sh.hist_idx = hist_idx; sh.primed = primed;
            }
            if (cmd == CMD_BULK) {
                const int tot = hist_idx + c_run;
                if (tot >= kHistory) primed = 1;
                hist_idx = tot % kHistory;
                flags_valid = false;                 // the baseline moves
            } else if (cmd == CMD_ZERO) {
                flags_valid = false;
            }
        }
        const long long t_l1 = IRDM_TICK();
        t_lead += t_l1 - t_l0;
        __syncthreads();
        const int cmd = sh.cmd;
        if (cmd == CMD_EXIT) break;

        // ================= dense command, all threads =================
        if (cmd == CMD_BULK) {
            // consecutive baseline updates (simd_baseline_update + memcpy, burst_detect.c:441-452)
            const int f0 = sh.f0, run = sh.run, detect = sh.detect;
            int hidx = sh.hist_idx, prm = sh.primed;
            float s[J];
#pragma unroll
            for (int j = 0; j < J; j++) s[j] = s_sum[BIN_OF(j)];
            bool bad = false;
            // groups of G frames with every load of the group in flight at once; the history rows of a
            // group are distinct (G <= 512), so reads never alias the group's writes
            constexpr int G = (64 / J) > 0 ? (64 / J) : 1;
            for (int k0 = 0; k0 < run; k0 += G) {
                float4 m[G][Q], old[G][Q];
                int hrow_idx[G];
                int hx = hidx, px = prm;
                // all loads are issued unconditionally (a register-or-load select makes hipcc branch
                // around every load and serialise them); frames past the run re-read the last frame
                int prm_g[G];
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const int kk = k0 + g < run ? k0 + g : run - 1;
                    hrow_idx[g] = hx;
                    prm_g[g] = px;
                    const float4 *mrow = reinterpret_cast<const float4 *>(mag + (size_t)(f0 + kk) * N);
                    const float4 *hrow = reinterpret_cast<const float4 *>(hist + (size_t)hx * N);
#pragma unroll
                    for (int q = 0; q < Q; q++) m[g][q] = mrow[q * kFastThreads + tid];
#pragma unroll
                    for (int q = 0; q < Q; q++) old[g][q] = hrow[q * kFastThreads + tid];
                    if (k0 + g < run && ++hx == kHistory) { px = 1; hx = 0; }
                }
#pragma unroll
                for (int g = 0; g < G; g++) {
                    if (!prm_g[g]) {      // rows not yet rewritten since a reset read as zero (:623-624)
#pragma unroll
                        for (int q = 0; q < Q; q++) old[g][q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    }
                }
#pragma unroll
                for (int g = 0; g < G; g++) {
                    if (k0 + g < run) {
                        float4 *hrow = reinterpret_cast<float4 *>(hist + (size_t)hrow_idx[g] * N);
#pragma unroll
                        for (int q = 0; q < Q; q++) {
                            const float mv[4] = { m[g][q].x, m[g][q].y, m[g][q].z, m[g][q].w };
                            const float ov[4] = { old[g][q].x, old[g][q].y, old[g][q].z, old[g][q].w };
#pragma unroll
                            for (int u = 0; u < 4; u++) {
                                const int j = 4 * q + u;
                                if (detect) {
                                    const float rel = s[j] > 0 ? mv[u] / s[j] : 0.0f;
                                    if (rel > thr && VALID_BIN(BIN_OF(j))) bad = true;   // safety net
                                }
                                const float d = s[j] - ov[u];
                                s[j] = d + mv[u];
                            }
                            hrow[q * kFastThreads + tid] = m[g][q];
                        }
                    }
                }
                hidx = hx;
                prm = px;
            }
#pragma unroll
            for (int j = 0; j < J; j++) s_sum[BIN_OF(j)] = s[j];
            if (bad) atomicOr(&sh.abort, 1);
        } else if (cmd == CMD_VALIDATE) {
            // the prefilter lists are complete only while pre[b] <= 0.9*thr*sum[b]
            bool bad = false;
            float4 pv[Q];
#pragma unroll
            for (int q = 0; q < Q; q++) pv[q] = reinterpret_cast<const float4 *>(pre)[q * kFastThreads + tid];
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const float pq[4] = { pv[q].x, pv[q].y, pv[q].z, pv[q].w };
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const float sv = s_sum[BIN_OF(4 * q + u)];
                    bad |= (sv > 0) & (pq[u] > 0.9f * thr * sv);
                }
            }
            if (bad) atomicOr(&sh.abort, 2);
        } else if (cmd == CMD_ZERO) {
#pragma unroll
            for (int j = 0; j < J; j++) s_sum[BIN_OF(j)] = 0.0f;
        }
        __syncthreads();
        if (tid < 64) abort_code |= sh.abort;
        t_cmd[cmd & 3] += IRDM_TICK() - t_l1;
        n_cmd[cmd & 3]++;
    }

    if (tid < 64) {
        if (lane < n_act) s_act[lane].last_active = r_la;
        WAVE_SYNC();
        FLUSH_GONE();
        if (lane == 0) {
#ifdef IRDM_SCAN_PROFILE
            long long *dbg = reinterpret_cast<long long *>(status + 4);
            dbg[0] = IRDM_TICK() - t_begin; dbg[1] = t_lead; dbg[2] = t_cmd[1]; dbg[3] = t_cmd[2]; dbg[4] = t_cmd[3];
            dbg[5] = n_cmd[1]; dbg[6] = n_cmd[2]; dbg[7] = n_cmd[3]; dbg[8] = n_cplx; dbg[9] = n_sparse;
            dbg[10] = t_s1; dbg[11] = t_s2; dbg[12] = t_s3; dbg[13] = t_cal; dbg[14] = t_cA; dbg[15] = t_cB; (void)t_s4; (void)t_fe;
#endif
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
#pragma unroll
    for (int j = 0; j < J; j++) sum_g[BIN_OF(j)] = s_sum[BIN_OF(j)];
#undef WAVE_SYNC
#undef BIN_OF
#undef VALID_BIN
#undef MASK_RANGE
#undef MASK_ALL_ONES
#undef PUSH_GONE
#undef FLUSH_GONE
}

size_t scan_fast_lds_bytes(int n)
{
    return (size_t)n * 4 + (size_t)n + sizeof(ActiveBurst) * kFastMaxActive + sizeof(PeakCand) * kListCap +
           sizeof(ListEntry) * kStageCap + kStageCap + sizeof(GoneBurst) * kGoneLds + sizeof(FastShared) + 16;
}

int launch_detect_scan_fast(const DetParams &P, DetState *st, float *sum, float *hist, const float *mag,
                            int n_frames, const unsigned *counts, const unsigned *goff,
                            const ListEntry *compact, const float *pre, GoneBurst *gone, int gone_cap,
                            int *status, hipStream_t stream)
{
    const int Q = P.n / (4 * kFastThreads);
    const size_t lds = scan_fast_lds_bytes(P.n);
#define IRDM_LAUNCH_FAST(JJ)                                                                     \
    do {                                                                                         \
        (void)hipFuncSetAttribute((const void *)detect_scan_fast_kernel<JJ>,                     \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
        hipLaunchKernelGGL((detect_scan_fast_kernel<JJ>), dim3(1), dim3(kFastThreads), lds,       \
                           stream, P, st, sum, hist, mag, n_frames, counts, goff, compact, pre,  \
                           gone, gone_cap, status);                                              \
    } while (0)
    switch (Q) {
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
