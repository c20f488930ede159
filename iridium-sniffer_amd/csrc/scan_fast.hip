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
//     (simd_baseline_update) -- dense and bin-parallel, in runs of frames with
//     an empty prefilter list, each thread also re-checking its own bins
//     exactly (safety net: a crossing the list did not announce aborts the
//     kernel, the host restores the pre-chunk state and runs the dense scan).
//     Two forms of the same kernel: MC = false, the updates are commands that
//     all wavefronts of the leader's workgroup execute; MC = true (default on
//     MI355X), the leader publishes them as 64-bit operation words and
//     updater workgroups on CUs of their own execute them, owning the sums.
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
    IRDM_DETECTOR_PRIO();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // 0.5: the list stays complete while the live sum is >= 0.556 of this reference (CMD_VALIDATE checks
    // pre <= 0.9*thr*sum).  Measured on the bench scene: per-bin sums move by +-20 % within a chunk (burst
    // leading edges below threshold enter the history), so 0.7 already trips the guard on every chunk.
    if (i < n) pre[i] = 0.5f * thr * sum[i];
}

// one workgroup per frame: bins with mag > pre[bin] -> (bin, mag), unordered
__global__ __launch_bounds__(256) void prefilter_kernel(const float *__restrict__ mag,
                                                        const float *__restrict__ pre, int n,
                                                        unsigned *__restrict__ counts,
                                                        ListEntry *__restrict__ entries, int n_frames, int cap)
{
    IRDM_DETECTOR_PRIO();
    __shared__ int cnt;
    const int tid = threadIdx.x;
    for (int frame = blockIdx.x; frame < n_frames; frame += gridDim.x) {
        if (tid == 0) cnt = 0;
        __syncthreads();
        const float4 *m4 = reinterpret_cast<const float4 *>(mag + (size_t)frame * n);
        const float4 *p4 = reinterpret_cast<const float4 *>(pre);
        ListEntry *out = entries + (size_t)frame * cap;
        for (int q = tid; q < n / 4; q += 256) {
            const float4 m = m4[q], p = p4[q];
            const float mv[4] = { m.x, m.y, m.z, m.w }, pv[4] = { p.x, p.y, p.z, p.w };
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (mv[u] > pv[u]) {
                    const int slot = atomicAdd(&cnt, 1);
                    if (slot < cap) {
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
        // bin (14 bits) | frame index within the scanned range << 14: the scan regroups entries by frame
        for (unsigned i = threadIdx.x; i < c; i += 256) {
            ListEntry e = src[i];
            e.bin |= frame << 14;
            dst[i] = e;
        }
    }
}

int launch_prefilter(const float *sum, float thr, float *pre, const float *mag, int n,
                     unsigned *counts, ListEntry *entries, unsigned *goff, ListEntry *compact,
                     int n_frames, hipStream_t stream)
{
    if (n_frames <= 0) return 0;
    hipLaunchKernelGGL(prefilter_threshold_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, sum, thr, pre, n);
    const int grid = n_frames < 8192 ? n_frames : 8192;
    hipLaunchKernelGGL(prefilter_kernel, dim3(grid), dim3(256), 0, stream, mag, pre, n, counts, entries, n_frames, kListCap);
    hipLaunchKernelGGL(list_offsets_kernel, dim3(1), dim3(1024), 0, stream, counts, goff, n_frames);
    hipLaunchKernelGGL(list_compact_kernel, dim3(grid), dim3(256), 0, stream, counts, goff, entries, compact, n_frames);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// the per-frame lists alone (scan_band.hip reads them in place, no compaction).  smin != nullptr: a retry after the
// band scan found the lists stale -- the threshold only goes down, to 0.45 * thr * (smallest sum the bin went through)
__global__ void prefilter_lower_kernel(const float *__restrict__ smin, float thr, float *__restrict__ pre, int n)
{
    IRDM_DETECTOR_PRIO();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pre[i] = fminf(pre[i], 0.45f * thr * smin[i]);
}

int launch_prefilter_threshold(const float *sum, float thr, float *pre, int n, hipStream_t stream)
{
    hipLaunchKernelGGL(prefilter_threshold_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, sum, thr, pre, n);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_prefilter_lists(const float *sum, float thr, float *pre, const float *smin, const float *mag, int n,
                           unsigned *counts, ListEntry *entries, int n_frames, int cap, hipStream_t stream)
{
    if (n_frames <= 0) return 0;
    if (smin)
        hipLaunchKernelGGL(prefilter_lower_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, smin, thr, pre, n);
    else
        hipLaunchKernelGGL(prefilter_threshold_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, sum, thr, pre, n);
    const int grid = n_frames < 8192 ? n_frames : 8192;
    hipLaunchKernelGGL(prefilter_kernel, dim3(grid), dim3(256), 0, stream, mag, pre, n, counts, entries, n_frames, cap);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- wavefront reductions on the DPP network (no LDS round trips: __shfl_xor compiles to
// ds_bpermute, ~100+ cycles each for a lone wavefront) ----
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v, unsigned identity)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, 0xf, 0xf, false);
}
#define IRDM_DPP_REDUCE(v, OP, IDENT)                                  \
    do {                                                               \
        v = OP(v, dpp_u32<0x111>(v, IDENT));   /* row_shr:1  */        \
        v = OP(v, dpp_u32<0x112>(v, IDENT));   /* row_shr:2  */        \
        v = OP(v, dpp_u32<0x114>(v, IDENT));   /* row_shr:4  */        \
        v = OP(v, dpp_u32<0x118>(v, IDENT));   /* row_shr:8  */        \
        v = OP(v, dpp_u32<0x142>(v, IDENT));   /* row_bcast:15 */      \
        v = OP(v, dpp_u32<0x143>(v, IDENT));   /* row_bcast:31 */      \
    } while (0)
__device__ __forceinline__ unsigned op_min_u32(unsigned a, unsigned b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned op_or_u32(unsigned a, unsigned b) { return a | b; }
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    IRDM_DPP_REDUCE(v, op_min_u32, 0xffffffffu);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_or_u32(unsigned v)
{
    IRDM_DPP_REDUCE(v, op_or_u32, 0u);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// max of 64-bit keys (hi word compared first)
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k)
{
    unsigned lo = (unsigned)k, hi = (unsigned)(k >> 32);
#define IRDM_STEP(CTRL)                                                                  \
    do {                                                                                 \
        const unsigned olo = dpp_u32<CTRL>(lo, 0u), ohi = dpp_u32<CTRL>(hi, 0u);         \
        const bool take = ohi > hi || (ohi == hi && olo > lo);                           \
        lo = take ? olo : lo;                                                            \
        hi = take ? ohi : hi;                                                            \
    } while (0)
    IRDM_STEP(0x111); IRDM_STEP(0x112); IRDM_STEP(0x114); IRDM_STEP(0x118); IRDM_STEP(0x142); IRDM_STEP(0x143);
#undef IRDM_STEP
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, 63) << 32) |
           (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)lo, 63);
}

// Values that are wave-uniform by construction but arrive through a vector load (LDS / global) must be moved to
// SGPRs explicitly: otherwise hipcc treats every branch of the leader's state machine as divergent (exec-mask
// bookkeeping on each transition instead of a scalar compare-and-branch).
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ unsigned long long uni(unsigned long long v)
{
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
           (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

enum { CMD_EXIT = 0, CMD_BULK = 1, CMD_VALIDATE = 2, CMD_ZERO = 3, CMD_CROSS = 4 };
constexpr int kFastMaxActive = 64;      // active bursts live in the leader's lanes; more -> dense fallback
constexpr int kStageCap = 2048;         // list entries staged in LDS per batch
constexpr int kStageFrames = 32;        // frames per batch: one bit per frame in the 32-bit frame masks
constexpr int kMaxBins = 2048;          // distinct crossing bins per batch
constexpr int kCandCap = 512;
constexpr int kGoneLds = 64;            // gone records buffered in LDS between flushes
constexpr int kFastThreads = 512;       // 8 wavefronts (2 per SIMD, 256 VGPRs each): wavefront 0 leads, all execute dense commands

// G consecutive baseline updates of one thread's bins (simd_baseline_update + memcpy, burst_detect.c:441-452):
// sum = (sum - oldest) + mag per frame, the magnitude row replaces the oldest history row.  The history rows of a
// group are distinct (G <= 512), so reads never alias the group's writes.
template <int Q, int G>
__device__ __forceinline__ void bulk_group(const float *__restrict__ mag, float *__restrict__ hist, int N, int f0,
                                           int &hidx, int &prm, float (&s)[4 * Q], int detect, float thr, int tid,
                                           int half_bw, int dc, bool &bad)
{
    float4 m[G][Q], old[G][Q];
    int row[G], prm_g[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
        row[g] = hidx;
        prm_g[g] = prm;
        const float4 *mrow = reinterpret_cast<const float4 *>(mag + (size_t)(f0 + g) * N);
        const float4 *hrow = reinterpret_cast<const float4 *>(hist + (size_t)hidx * N);
#pragma unroll
        for (int q = 0; q < Q; q++) m[g][q] = mrow[q * kFastThreads + tid];
#pragma unroll
        for (int q = 0; q < Q; q++) old[g][q] = hrow[q * kFastThreads + tid];
        if (++hidx == kHistory) { prm = 1; hidx = 0; }
    }
#pragma unroll
    for (int g = 0; g < G; g++) {
        float4 *hrow = reinterpret_cast<float4 *>(hist + (size_t)row[g] * N);
#pragma unroll
        for (int q = 0; q < Q; q++) {
            // rows not yet rewritten since a reset read as zero (:623-624)
            const float4 o = prm_g[g] ? old[g][q] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            const float mv[4] = { m[g][q].x, m[g][q].y, m[g][q].z, m[g][q].w };
            const float ov[4] = { o.x, o.y, o.z, o.w };
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = 4 * q + u;
                if (detect) {
                    // safety net: mag / sum > thr is impossible while mag <= 0.99 * thr * sum;
                    // only the rare near-threshold bins pay for the exact division
                    if (mv[u] > 0.99f * thr * s[j]) {
                        const float rel = s[j] > 0 ? mv[u] / s[j] : 0.0f;
                        const int b = (q * kFastThreads + tid) * 4 + u;
                        if (rel > thr && b >= half_bw && b < N - half_bw && !(b >= dc - 3 && b <= dc + 3)) bad = true;
                    }
                }
                const float d = s[j] - ov[u];
                s[j] = d + mv[u];
            }
            hrow[q * kFastThreads + tid] = m[g][q];
        }
    }
}

// ---- multi-CU form (MC): the baseline updates leave the leader's workgroup ----
// The leader (workgroup 0) publishes every baseline operation -- a run of quiet frames, a forced update, a reset, a
// list-validity check -- as ONE 64-bit word in device memory and keeps going; W updater workgroups (bins partitioned
// between them, sums resident in their LDS) execute the operations in order, write the new sums of their bins to
// sum_g and count the operation in done[w].  The leader needs sums only for the few listed bins of the next
// crossing test: it waits for done[*] == published and reads exactly those bins with agent-scope loads.
// Everything that crosses workgroups -- operation words, sums, counters -- moves through agent-scope (sc1) atomic
// loads / stores, which complete at the device's coherence point; "sums acknowledged (vmcnt 0), then the counter" is
// the only ordering needed, so there is no release / acquire fence (L2 write-back / invalidate) on either side.
// Every spin is bounded: a timeout aborts the chunk (dense fallback).
enum { MCF_PRIMED = 1, MCF_DETECT = 2, MCF_ZERO = 4, MCF_EXIT = 8, MCF_VALIDATE = 16 };
constexpr unsigned long long kMcValid = 1ull << 63;
constexpr int kMcDoneStride = 16;        // one cache line (64 B) per updater's counter
constexpr int kMcMaxUpdaters = 32;
constexpr int kMcSpinLimit = 4000000;    // bounded spins: >= 1 s, far beyond any chunk's scan
__host__ __device__ inline unsigned long long mc_pack(int f0, int run, int hist_idx, int flags)
{
    return kMcValid | ((unsigned long long)(unsigned)f0 << 32) | ((unsigned long long)(unsigned)run << 20) |
           ((unsigned long long)(unsigned)hist_idx << 8) | (unsigned long long)(unsigned)flags;
}

// G consecutive baseline updates of ONE float4 group of bins (the same arithmetic, in the same order, as bulk_group)
template <int G>
__device__ __forceinline__ void mc_group(const float *__restrict__ mag, float *__restrict__ hist, int N, int f0,
                                         int &hidx, int &prm, float4 &s4, int g4, int detect, float thr,
                                         int half_bw, int dc, bool &bad)
{
    float4 m[G], old[G];
    int row[G], prm_g[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
        row[g] = hidx;
        prm_g[g] = prm;
        m[g] = reinterpret_cast<const float4 *>(mag + (size_t)(f0 + g) * N)[g4];
        old[g] = reinterpret_cast<const float4 *>(hist + (size_t)hidx * N)[g4];
        if (++hidx == kHistory) { prm = 1; hidx = 0; }
    }
    float s[4] = { s4.x, s4.y, s4.z, s4.w };
#pragma unroll
    for (int g = 0; g < G; g++) {
        const float4 o = prm_g[g] ? old[g] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const float mv[4] = { m[g].x, m[g].y, m[g].z, m[g].w };
        const float ov[4] = { o.x, o.y, o.z, o.w };
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (detect) {
                if (mv[u] > 0.99f * thr * s[u]) {
                    const float rel = s[u] > 0 ? mv[u] / s[u] : 0.0f;
                    const int b = 4 * g4 + u;
                    if (rel > thr && b >= half_bw && b < N - half_bw && !(b >= dc - 3 && b <= dc + 3)) bad = true;
                }
            }
            const float d = s[u] - ov[u];
            s[u] = d + mv[u];
        }
        reinterpret_cast<float4 *>(hist + (size_t)row[g] * N)[g4] = m[g];
    }
    s4 = make_float4(s[0], s[1], s[2], s[3]);
}

struct FastShared {
    int cmd, f0, run, detect, hist_idx, primed;
    int abort;
    int n_bins_old, n_bins;   // CMD_CROSS: distinct crossing bins recorded in s_bins (in: to clear, out: new count)
};

enum { S_TOP = 0, S_CPLX_A = 1, S_CPLX_B = 2, S_FRAME_END = 3 };

// ---------------------------------------------------------------------------
// Sparse detector scan, "frame-mask" form.
//
// Cost model measured on gfx950: a lone wavefront running branchy sequential code retires roughly
// one instruction per 10-20 cycles, so the design minimises the leader's instruction count:
//   * a batch = up to 32 frames; for every bin that crosses the threshold somewhere in the batch,
//     s_crossT[bin] holds a 32-bit mask of the frames in which it crosses (exact test, built by one
//     pass over the batch's staged list entries);
//   * an active burst (one per leader lane) gets its hit mask in O(1): crossT[cb-1]|crossT[cb]|crossT[cb+1]
//     (update_bursts, burst_detect.c:458-469);
//   * the frame of the next event (a burst expiring / growing too long, :498-505, or a peak candidate
//     appearing, :522-548) is found with bit arithmetic on those masks -- no per-frame loop;
//   * only event frames (about two per burst) run the list logic of delete_gone_bursts /
//     create_new_bursts (:490-632).
// Dense per-bin work (baseline updates while no burst is active, :438-454) is a command executed by
// all eight wavefronts.  Q = float4 groups per thread; thread t owns bins (q*512 + t)*4 .. +3.
// ---------------------------------------------------------------------------
template <int Q, bool MC>
__global__ __launch_bounds__(kFastThreads) void detect_scan_fast_kernel(
    DetParams P, DetState *__restrict__ st, float *__restrict__ sum_g, float *__restrict__ hist,
    const float *__restrict__ mag, int n_frames, const unsigned *__restrict__ counts,
    const unsigned *__restrict__ goff, const ListEntry *__restrict__ compact,
    const float *__restrict__ pre, GoneBurst *__restrict__ gone, int gone_cap, int *__restrict__ status,
    unsigned long long *__restrict__ mc_ops, int mc_ops_cap, unsigned *__restrict__ mc_done)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int J = 4 * Q;
    const int N = P.n;
    if constexpr (MC) {
        if (blockIdx.x > 0) {
            // ================= updater workgroup w of W: float4 groups [g_lo, g_hi) of every row =================
            const int w = blockIdx.x - 1, W = gridDim.x - 1;
            const int tid_ = threadIdx.x;
            const int per = (N / 4 + W - 1) / W;
            const int g_lo = w * per, g_hi = g_lo + per < N / 4 ? g_lo + per : N / 4;
            const int cnt = g_hi > g_lo ? g_hi - g_lo : 0;
            float4 *s_loc = reinterpret_cast<float4 *>(smem_raw);          // cnt sums
            float4 *p_loc = s_loc + per;                                   // cnt prefilter references
            __shared__ unsigned long long s_op;
            for (int i = tid_; i < cnt; i += kFastThreads) {
                s_loc[i] = reinterpret_cast<const float4 *>(sum_g)[g_lo + i];
                p_loc[i] = reinterpret_cast<const float4 *>(pre)[g_lo + i];
            }
            __syncthreads();
            const float thr_ = P.threshold;
            const int half_bw_ = P.width / 2, dc_ = N / 2;
            for (unsigned seen = 0;; seen++) {
                if (tid_ == 0) {
                    unsigned long long op = 0ull;
                    if ((int)seen < mc_ops_cap) {
                        int spins = 0;
                        while ((op = __hip_atomic_load(&mc_ops[seen], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0ull) {
                            if (++spins > kMcSpinLimit) break;
                            __builtin_amdgcn_s_sleep(1);
                        }
                        // the operation word has been seen (the loop exit waited for its load): nothing that follows may
                        // be moved in front of it by the compiler -- the loads of the operation's rows and sums are relaxed
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    if (op == 0ull) {                                      // the leader went away: give up
                        atomicOr(&status[0], 256);
                        op = mc_pack(0, 0, 0, MCF_EXIT);
                    }
                    s_op = op;
                }
                __syncthreads();
                const unsigned long long op = s_op;
                const int flags = (int)(op & 0xffu);
                if (flags & MCF_EXIT) break;
                const int f0 = (int)((op >> 32) & 0x7fffffffu), run = (int)((op >> 20) & 0xfffu);
                const int hidx0 = (int)((op >> 8) & 0xfffu);
                bool bad = false, stale = false;
                for (int i = tid_; i < cnt; i += kFastThreads) {
                    float4 s4 = s_loc[i];
                    if (flags & MCF_ZERO) {
                        s4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    } else {
                        int hidx = hidx0, prm = flags & MCF_PRIMED ? 1 : 0, k0 = 0;
                        const int det = flags & MCF_DETECT ? 1 : 0;
                        for (; run - k0 >= 8; k0 += 8)
                            mc_group<8>(mag, hist, N, f0 + k0, hidx, prm, s4, g_lo + i, det, thr_, half_bw_, dc_, bad);
                        for (; run - k0 >= 2; k0 += 2)
                            mc_group<2>(mag, hist, N, f0 + k0, hidx, prm, s4, g_lo + i, det, thr_, half_bw_, dc_, bad);
                        for (; k0 < run; k0++)
                            mc_group<1>(mag, hist, N, f0 + k0, hidx, prm, s4, g_lo + i, det, thr_, half_bw_, dc_, bad);
                    }
                    if (flags & MCF_VALIDATE) {
                        // the prefilter lists are complete only while pre[b] <= 0.9*thr*sum[b]
                        const float4 pq = p_loc[i];
                        stale |= (s4.x > 0) & (pq.x > 0.9f * thr_ * s4.x);
                        stale |= (s4.y > 0) & (pq.y > 0.9f * thr_ * s4.y);
                        stale |= (s4.z > 0) & (pq.z > 0.9f * thr_ * s4.z);
                        stale |= (s4.w > 0) & (pq.w > 0.9f * thr_ * s4.w);
                    }
                    if (run > 0 || (flags & MCF_ZERO)) {
                        s_loc[i] = s4;
                        // agent-scope (write-through) stores: complete at the device's coherence point when acknowledged
                        unsigned *dst = reinterpret_cast<unsigned *>(sum_g) + 4 * (size_t)(g_lo + i);
                        __hip_atomic_store(dst + 0, __float_as_uint(s4.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(dst + 1, __float_as_uint(s4.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(dst + 2, __float_as_uint(s4.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(dst + 3, __float_as_uint(s4.w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (bad) atomicOr(&status[0], 1);
                if (stale) atomicOr(&status[0], 2);
                // every wavefront waits for the acknowledgement of its own stores (vmcnt 0: an sc1 store is acknowledged
                // when it is visible at agent scope), then the workgroup meets: all sums of this workgroup are visible
                // before the counter moves, so the counter needs no release fence (no L2 write-back of the XCD's
                // unrelated dirty lines on the leader's critical path).  The wait must be explicit: __syncthreads()
                // alone compiles to `s_waitcnt lgkmcnt(0); s_barrier` here -- the stores could still be in flight.
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid_ == 0)
                    __hip_atomic_store(&mc_done[w * kMcDoneStride], seen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
    }
    float *s_sum = reinterpret_cast<float *>(smem_raw);                                   // N
    unsigned *s_crossT = reinterpret_cast<unsigned *>(s_sum + N);                         // N frame masks
    unsigned *s_mbits = s_crossT + N;                                                     // N/32: 1 = not masked
    ListEntry *s_ent = reinterpret_cast<ListEntry *>(s_mbits + N / 32);                   // kStageCap
    PeakCand *s_cand = reinterpret_cast<PeakCand *>(s_ent + kStageCap);                   // kCandCap
    GoneBurst *s_gone = reinterpret_cast<GoneBurst *>(s_cand + kCandCap);                 // kGoneLds
    ActiveBurst *s_act = reinterpret_cast<ActiveBurst *>(s_gone + kGoneLds);              // kFastMaxActive
    unsigned short *s_bins = reinterpret_cast<unsigned short *>(s_act + kFastMaxActive);  // kMaxBins
    FastShared &sh = *reinterpret_cast<FastShared *>(s_bins + kMaxBins);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const bool leader = uni(tid) < 64;                   // wavefront 0 (scalar predicate)
    const float thr = P.threshold;
    const int half_bw = P.width / 2;
    const int dc = N / 2;
    const int log_n = P.log_n;
    const uint64_t index0 = uni((unsigned long long)st->index);
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const int gap_frames = (P.post_len + N - 1) >> log_n;   // frames without a hit until last_active + post_len <= index

    // NOTE: never `volatile` LDS pointers (they compile to system-coherent FLAT accesses, microseconds each);
    // LDS ops of one wavefront execute in order, a compiler barrier between phases is all that is needed.
    // workgroup barrier that orders LDS only: the history-row stores of a bulk command stay in flight (each thread
    // re-reads only rows it wrote itself, in program order), __syncthreads() would wait ~1 us for their acknowledgement
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    // branch-layout hint: the rare paths (aborts, capacity checks, squelch, forced updates) are moved out of the
    // leader's straight-line code
#define RARE(c) __builtin_expect(!!(c), 0)
#define LIKELY(c) __builtin_expect(!!(c), 1)
#define WAVE_SYNC() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)
#define BIN_OF(j) ((((j) >> 2) * kFastThreads + tid) * 4 + ((j) & 3))
#define VALID_BIN(b) ((b) >= half_bw && (b) < N - half_bw && !((b) >= dc - 3 && (b) <= dc + 3))
#define UNMASKED(b) ((s_mbits[(b) >> 5] >> ((b) & 31)) & 1u)
    // set (SETV=1) or clear (SETV=0) the mask bits of [cb-half_bw, cb+half_bw] (mask_burst, :473-480)
#define MASK_EDIT(cb, SETV)                                                                         \
    do {                                                                                            \
        int lo_ = (cb) - half_bw, hi_ = (cb) + half_bw;                                             \
        if (lo_ < 0) lo_ = 0;                                                                       \
        if (hi_ >= N) hi_ = N - 1;                                                                  \
        for (int w_ = (lo_ >> 5) + lane; w_ <= (hi_ >> 5); w_ += 64) {                              \
            const int b0_ = w_ << 5;                                                                \
            const int l_ = lo_ > b0_ ? lo_ - b0_ : 0, h_ = hi_ < b0_ + 31 ? hi_ - b0_ : 31;         \
            const unsigned m_ = (h_ == 31 ? ~0u : ((1u << (h_ + 1)) - 1u)) & ~((1u << l_) - 1u);   \
            if (SETV) atomicOr(&s_mbits[w_], m_); else atomicAnd(&s_mbits[w_], ~m_);  /* no return value: no LDS round trip */ \
        }                                                                                           \
    } while (0)
#define MASK_ALL_ONES() do { for (int i_ = lane; i_ < N / 32; i_ += 64) s_mbits[i_] = ~0u; } while (0)
#define RL64(v, l) (((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)((v) >> 32), (l)) << 32) | \
                    (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(v), (l)))
#define PUSH_GONE(b, stopv, slot)                                                                  \
    do {                                                                                           \
        const unsigned ls_ = (slot) - gone_base;                                                   \
        if (ls_ < (unsigned)kGoneLds) {                                                            \
            GoneBurst g_;                                                                          \
            g_.id = (b).id; g_.start = (b).start; g_.stop = (stopv); g_.last_active = (b).last_active; \
            g_.center_bin = (b).center_bin; g_.peak_rel = (b).peak_rel; g_.base_sum = (b).base_sum;    \
            g_.pad = 0;                                                                            \
            s_gone[ls_] = g_;                                                                      \
        }                                                                                          \
    } while (0)
    // PUSH_GONE runs inside per-lane branches; the overflow check is made on the uniform counter afterwards
    // (abort_code must stay a scalar: it steers the state machine)
#define CHECK_GONE() do { if (n_gone - gone_base > (unsigned)kGoneLds) abort_code |= 16; } while (0)
    // gone records collect in LDS (no global store, hence no vmcnt wait, inside the frame loop)
#define FLUSH_GONE()                                                                               \
    do {                                                                                           \
        const unsigned cnt_ = n_gone - gone_base;                                                  \
        for (unsigned i_ = lane; i_ < cnt_ && i_ < (unsigned)kGoneLds; i_ += 64)                   \
            if ((int)(gone_base + i_) < gone_cap) gone[gone_base + i_] = s_gone[i_];               \
        gone_base = n_gone;                                                                        \
    } while (0)

    // MC: publish one baseline operation (single 64-bit word, no fence, no wait)
#define MC_PUBLISH(F0, RUN, FLAGS)                                                                     \
    do {                                                                                               \
        if (RARE((int)n_pub >= mc_ops_cap - 1)) { abort_code |= 128; }                                 \
        else {                                                                                         \
            if (lane == 0)                                                                             \
                __hip_atomic_store(&mc_ops[n_pub], mc_pack((F0), (RUN), hist_idx, (FLAGS) | (primed ? MCF_PRIMED : 0)), \
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                        \
            n_pub++;                                                                                   \
        }                                                                                              \
    } while (0)
    // MC: RUN baseline updates starting at frame F0 -- what CMD_BULK and the bookkeeping after it do
#define MC_BULK(F0, RUN, DET)                                                                          \
    do {                                                                                               \
        MC_PUBLISH(F0, RUN, (DET) ? MCF_DETECT : 0);                                                   \
        const int tot_ = hist_idx + (RUN);                                                             \
        if (tot_ >= kHistory) primed = 1;                                                              \
        hist_idx = tot_ % kHistory;                                                                    \
        cross_valid = false;                                                                           \
    } while (0)
    // MC: every published operation has been executed by every updater (their sums are in sum_g)
#define MC_WAIT()                                                                                      \
    do {                                                                                               \
        if (n_ack != n_pub) {                                                                          \
            const int W_ = (int)gridDim.x - 1;                                                         \
            int spins_ = 0;                                                                            \
            for (;;) {                                                                                 \
                unsigned d_ = n_pub;                                                                   \
                if (lane < W_) d_ = __hip_atomic_load(&mc_done[lane * kMcDoneStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
                if (__all((int)(d_ - n_pub) >= 0)) break;                                              \
                if (++spins_ > kMcSpinLimit) { abort_code |= 128; break; }                             \
                __builtin_amdgcn_s_sleep(1);                                                           \
            }                                                                                          \
            /* the counters have been seen: the MC_SUM loads that follow are relaxed atomics to other  \
               addresses, which the compiler could otherwise hoist above the spin exit */              \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
            n_ack = n_pub;                                                                             \
        }                                                                                              \
    } while (0)
#define MC_SUM(bin) __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned *>(sum_g) + (bin), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))

    // ---- load carried state ----
#pragma unroll
    for (int j = 0; j < J; j++) {
        if constexpr (!MC) s_sum[BIN_OF(j)] = sum_g[BIN_OF(j)];     // MC: s_sum only caches the bins read from sum_g
        s_crossT[BIN_OF(j)] = 0u;
    }
    for (int i = tid; i < N / 32; i += kFastThreads) s_mbits[i] = ~0u;
    const int n_act_in = uni(st->n_act);
    if (tid == 0) {
        sh.cmd = CMD_EXIT;
        sh.abort = n_act_in > kFastMaxActive ? 4 : 0;
    }
    for (int i = tid; i < n_act_in && i < kFastMaxActive; i += kFastThreads) s_act[i] = st->act[i];
    __syncthreads();

    // ---- leader state (registers of wavefront 0) ----
    // Active bursts live in fixed lane slots: lane s mirrors slot s while bit s of `occ` is set.
    // The reference's list order (burst_detect.c:148-160) is creation order == ascending id.
    int hist_idx = uni(st->hist_idx), primed = uni(st->primed), squelch = uni(st->squelch);
    unsigned long long occ = 0;
    {
        const int na = n_act_in < kFastMaxActive ? n_act_in : kFastMaxActive;
        occ = na >= 64 ? ~0ull : ((1ull << na) - 1ull);
    }
    unsigned n_gone = uni(st->n_gone);
    unsigned gone_base = n_gone;
    unsigned long long burst_id = uni((unsigned long long)st->burst_id);
    int abort_code = uni(sh.abort);
    int r_cb = 0;
    float r_peak = 0.0f, r_base = 0.0f;
    uint64_t r_la = 0, r_start = 0, r_id = 0;
    if ((occ >> lane) & 1) {
        r_peak = s_act[lane].peak_rel;
        r_base = s_act[lane].base_sum;
        r_cb = s_act[lane].center_bin;
        r_la = s_act[lane].last_active;
        r_start = s_act[lane].start;
        r_id = s_act[lane].id;
    }
    if (leader) {
        unsigned long long o = occ;
        while (o) {
            const int sl = __builtin_ctzll(o);
            o &= o - 1;
            MASK_EDIT(__builtin_amdgcn_readlane(r_cb, sl), 0);
            WAVE_SYNC();
        }
    }
    int sb = 0, snf = 0;                   // staged batch: frames [sb, sb+snf)
    unsigned r_off = 0;                    // lane k: offset of frame sb+k in s_ent (lane snf: total)
    int n_bins = 0;                        // distinct crossing bins recorded in s_bins
    bool cross_cmd_done = false;           // CMD_CROSS just rebuilt s_crossT / s_bins for the frames >= f
    bool cross_valid = false;              // s_crossT holds the exact crossings of frames >= f of the batch
    bool hc_valid = false;                 // H / C below are current
    unsigned H = 0;                        // lane s: frames of the batch in which slot s sees a crossing near its centre
    unsigned C = 0;                        // frames of the batch that hold a peak candidate under the current mask
    int f = 0, state = S_TOP;
    unsigned n_pub = 0, n_ack = 0;         // MC: operations published / known to be completed by every updater
    int e0 = 0, e1 = 0, n_cand = 0, hist_before = 0;
    bool any_cand = false, was_quiet = false;
    unsigned long long ev_del = 0;
    long long tk[14] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0};
    int nk[14] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0};
    long long bulk_frames = 0;
    (void)bulk_frames;
    const long long t_begin = IRDM_TICK();
    (void)t_begin;
#define TK(i, t0) do { tk[i] += IRDM_TICK() - (t0); nk[i]++; } while (0)

    for (;;) {
        const long long t_l0 = IRDM_TICK();
        if (leader) {
            // ================= leader step: run until a dense command is needed =================
            int cmd = -1, c_f0 = 0, c_run = 0, c_detect = 0;
            while (cmd < 0) {
                if (RARE(abort_code || (state == S_TOP && f >= n_frames))) { cmd = CMD_EXIT; break; }
                if (state == S_TOP) {
                    const long long tT_ = IRDM_TICK();
                    if (RARE(!primed)) {
                        // update_filters_pre returns 0 (:427-428): updates only, up to the priming frame
                        int run = kHistory - hist_idx;
                        if (run > n_frames - f) run = n_frames - f;
                        if constexpr (MC) { MC_BULK(f, run, 0); f += run; continue; }
                        cmd = CMD_BULK; c_f0 = f; c_run = run; c_detect = 0;
                        f += run;
                        break;
                    }
                    if (RARE(f < sb || f >= sb + snf)) {
                        const long long t0_ = IRDM_TICK();
                        // ---- stage the compact lists of up to kStageFrames frames starting at f ----
                        const int nf = n_frames - f < kStageFrames ? n_frames - f : kStageFrames;
                        unsigned g = 0, c = 0;
                        if (lane <= nf) g = goff[f + lane];
                        if (lane < nf) c = counts[f + lane];
                        if (RARE(__any(c > (unsigned)kListCap))) { abort_code |= 8; continue; }
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
                        cross_valid = false;
                        TK(0, t0_);
                    }
                    const int k0 = f - sb;
                    was_quiet = occ == 0;
                    hist_before = hist_idx;

                    if (RARE(occ == 0)) {
                        // ---------------- quiet ----------------
                        // frames with an empty list are updated in bulk, every thread re-checking its own
                        // bins exactly (safety net)
                        const unsigned nxt = __shfl_down(r_off, 1);
                        const bool nonempty = lane >= k0 && lane < snf && nxt != r_off;
                        const unsigned long long nz = __ballot(nonempty) >> k0;
                        const int run = nz ? __builtin_ctzll(nz) : snf - k0;
                        if (run > 0) {
                            squelch = squelch > run ? squelch - run : 0;
                            if constexpr (MC) { MC_BULK(f, run, 1); f += run; continue; }
                            cmd = CMD_BULK; c_f0 = f; c_run = run; c_detect = 1;
                            f += run;
                            break;
                        }
                        // a listed frame: exact test of ITS entries against the live sums; with no burst
                        // active the mask is all ones, so a valid crossing is a peak (:529-548)
                        e0 = __builtin_amdgcn_readlane((int)r_off, k0);
                        e1 = __builtin_amdgcn_readlane((int)r_off, k0 + 1);
                        n_cand = 0;
                        if constexpr (MC) MC_WAIT();
                        for (int base = e0; base < e1; base += 64) {
                            const int i = base + lane;
                            bool cand = false;
                            PeakCand c;
                            c.rel = 0.0f; c.bin = 0;
                            if (i < e1) {
                                const ListEntry e = s_ent[i];
                                const int bin = e.bin & 0x3FFF;
                                float sv;
                                if constexpr (MC) { sv = MC_SUM(bin); s_sum[bin] = sv; } else { sv = s_sum[bin]; }
                                c.rel = sv > 0 ? e.mag / sv : 0.0f;
                                c.bin = bin;
                                cand = c.rel > thr && VALID_BIN(bin);
                            }
                            const unsigned long long cm = __ballot(cand);
                            if (cand && n_cand + __popcll(cm & lt_mask) < kCandCap) s_cand[n_cand + __popcll(cm & lt_mask)] = c;
                            n_cand += __popcll(cm);
                        }
                        if (RARE(n_cand > kCandCap)) { abort_code |= 32; continue; }
                        WAVE_SYNC();
                        if (n_cand == 0) {
                            if (squelch > 0) squelch--;                           // :629-630
                            state = S_FRAME_END;
                            continue;
                        }
                        ev_del = 0;
                        state = S_CPLX_B;                                         // nothing to delete
                        continue;
                    }

                    // ---------------- busy ----------------
                    const int total = __builtin_amdgcn_readlane((int)r_off, snf);
                    if (RARE(!cross_valid)) {
                        const long long t0_ = IRDM_TICK();
                        // s_crossT[bin] = frames (>= f) of the batch in which `bin` crosses: exact
                        // simd_relative_mag + `> threshold` with the (frozen) baseline.  One pass over the
                        // batch's staged entries: a dense command when there are enough of them to pay for
                        // the two barriers, the leader alone otherwise.
                        const int ef = __builtin_amdgcn_readlane((int)r_off, k0);
                        if constexpr (MC) { if (!cross_cmd_done) MC_WAIT(); }
                        if (!cross_cmd_done && total - ef + n_bins > 192) {
                            cmd = CMD_CROSS; c_f0 = ef; c_run = total; c_detect = sb;
                            break;
                        }
                        if (!cross_cmd_done) {
                        for (int i = lane; i < n_bins; i += 64) s_crossT[s_bins[i]] = 0u;
                        WAVE_SYNC();
                        n_bins = 0;
                        for (int base = ef; base < total; base += 64) {
                            const int i = base + lane;
                            bool first = false;
                            int bin = 0;
                            if (i < total) {
                                const ListEntry e = s_ent[i];
                                bin = e.bin & 0x3FFF;
                                const int k = (e.bin >> 14) - sb;
                                float sv;
                                if constexpr (MC) { sv = MC_SUM(bin); s_sum[bin] = sv; } else { sv = s_sum[bin]; }
                                const float rel = sv > 0 ? e.mag / sv : 0.0f;
                                if (rel > thr) first = atomicOr(&s_crossT[bin], 1u << k) == 0u;
                            }
                            const unsigned long long fm = __ballot(first);
                            if (first && n_bins + __popcll(fm & lt_mask) < kMaxBins)
                                s_bins[n_bins + __popcll(fm & lt_mask)] = (unsigned short)bin;
                            n_bins += __popcll(fm);
                        }
                        }
                        cross_cmd_done = false;
                        if (RARE(n_bins > kMaxBins)) { abort_code |= 64; continue; }
                        WAVE_SYNC();
                        cross_valid = true;
                        hc_valid = false;
                        TK(1, t0_);
                    }
                    const long long t2_ = IRDM_TICK();
                    if (!hc_valid) {
                        H = 0;
                        if ((occ >> lane) & 1) {
                            H = s_crossT[r_cb];
                            if (r_cb > 0) H |= s_crossT[r_cb - 1];
                            if (r_cb < N - 1) H |= s_crossT[r_cb + 1];
                        }
                        unsigned acc = 0;
                        for (int i = lane; i < n_bins; i += 64) {
                            const int bin = s_bins[i];
                            if (UNMASKED(bin) && VALID_BIN(bin)) acc |= s_crossT[bin];
                        }
                        C = wave_or_u32(acc);
                        hc_valid = true;
                        TK(2, t2_);
                    }
                    const long long t3_ = IRDM_TICK();
                    // ---- find the next event frame with bit arithmetic on the frame masks ----
                    const unsigned rm = (snf >= 32 ? ~0u : ((1u << snf) - 1u)) & ~((1u << k0) - 1u);
                    const uint64_t idx_base = index0 + (uint64_t)sb * N;
                    int Ej = 32;
                    const unsigned Hm = H & rm;
                    if ((occ >> lane) & 1) {
                        // expiry (:505): first frame k with no hit in (k - gap, k] and k >= last_active + gap
                        unsigned D;
                        if (RARE(gap_frames >= 32)) {
                            D = Hm ? ~((1u << __builtin_ctz(Hm)) - 1u) : 0u;     // no expiry at or after a hit within one batch
                        } else {
                            D = Hm;
                            int filled = 1;
                            while (filled * 2 <= gap_frames) { D |= D << filled; filled *= 2; }
                            if (gap_frames > filled) D |= D << (gap_frames - filled);
                        }
                        const long long la_f = ((long long)(r_la - idx_base)) >> log_n;       // frames, may be < 0
                        const long long kmin = la_f + gap_frames;
                        const unsigned low = kmin <= 0 ? ~0u : (kmin >= 32 ? 0u : ~((1u << (int)kmin) - 1u));
                        const unsigned X = ~D & low & rm;
                        int Eexp = X ? __builtin_ctz(X) : 32;
                        // too long (:499-502): first hit frame whose index - start > max_len
                        int Elong = 32;
                        if (P.max_len > 0) {
                            const long long kl = (((long long)(r_start + (uint64_t)P.max_len - idx_base)) >> log_n) + 1;
                            const unsigned T = Hm & (kl <= 0 ? ~0u : (kl >= 32 ? 0u : ~((1u << (int)kl) - 1u)));
                            Elong = T ? __builtin_ctz(T) : 32;
                        }
                        Ej = Eexp < Elong ? Eexp : Elong;
                    }
                    const int Ed = (int)wave_min_u32((unsigned)Ej);
                    const unsigned Cm = C & rm;
                    const int Ec = Cm ? __builtin_ctz(Cm) : 32;
                    const int E = Ed < Ec ? Ed : Ec;
                    // update_bursts for the frames up to and including E
                    {
                        const unsigned up = Hm & (E >= 31 ? ~0u : ((1u << (E + 1)) - 1u));
                        if (up) r_la = idx_base + ((uint64_t)(31 - __builtin_clz(up)) << log_n);
                    }
                    TK(3, t3_);
                    if (RARE(E >= snf)) {
                        const int cnt = snf - k0;
                        squelch = squelch > cnt ? squelch - cnt : 0;                  // :629-630 per frame
                        f = sb + snf;
                        continue;
                    }
                    {
                        const int cnt = E - k0;
                        squelch = squelch > cnt ? squelch - cnt : 0;
                    }
                    f = sb + E;
                    ev_del = __ballot(((occ >> lane) & 1) && Ej == E);
                    any_cand = (Cm >> E) & 1u;
                    state = S_CPLX_A;
                    TK(11, tT_);
                    continue;
                }
                if (state == S_CPLX_A) {
                    const long long t4_ = IRDM_TICK();
                    // ---- a burst ends or may start in frame f (burst_detect.c:490-632), part A ----
                    const int k = f - sb;
                    e0 = __builtin_amdgcn_readlane((int)r_off, k);
                    e1 = __builtin_amdgcn_readlane((int)r_off, k + 1);
                    const uint64_t idx = index0 + (uint64_t)f * N;
                    n_cand = 0;
                    if (LIKELY(!any_cand && ev_del && (ev_del & (ev_del - 1)) == 0)) {
                        // ---- common case 1: exactly one burst ends, nothing can start ----
                        const int sl = __builtin_ctzll(ev_del);
                        const int cbd = __builtin_amdgcn_readlane(r_cb, sl);
                        bool force = false;
                        if (lane == sl) {
                            ActiveBurst b;
                            b.id = r_id; b.center_bin = r_cb; b.peak_rel = r_peak; b.start = r_start; b.base_sum = r_base; b.pad = 0;
                            b.last_active = r_la;
                            force = P.max_len > 0 && (r_la - r_start > (uint64_t)P.max_len);
                            PUSH_GONE(b, idx, n_gone);
                            H = 0;
                        }
                        n_gone += 1;
                        CHECK_GONE();
                        force = __any(force) != 0;
                        occ &= ~ev_del;
                        MASK_EDIT(cbd, 1);                                        // its range is free again ...
                        WAVE_SYNC();
                        {   // ... except where a surviving burst's range overlaps it
                            const bool ov = ((occ >> lane) & 1) && r_cb >= cbd - P.width - 1 && r_cb <= cbd + P.width + 1;
                            unsigned long long om = __ballot(ov);
                            while (om) {
                                const int s2 = __builtin_ctzll(om);
                                om &= om - 1;
                                MASK_EDIT(__builtin_amdgcn_readlane(r_cb, s2), 0);
                                WAVE_SYNC();
                            }
                        }
                        if (LIKELY(hc_valid && !force)) {
                            // candidate frames can only gain the later crossings of the freed bins
                            int lo_ = cbd - half_bw, hi_ = cbd + half_bw;
                            if (lo_ < 0) lo_ = 0;
                            if (hi_ >= N) hi_ = N - 1;
                            unsigned acc = 0;
                            for (int b_ = lo_ + lane; b_ <= hi_; b_ += 64)
                                if (UNMASKED(b_) && VALID_BIN(b_)) acc |= s_crossT[b_];
                            C |= wave_or_u32(acc);
                        } else {
                            hc_valid = false;
                        }
                        if (squelch > 0) squelch--;                               // create_new_bursts' else branch (:629-630)
                        state = S_FRAME_END;
                        TK(12, t4_);
                        if constexpr (MC) { if (RARE(force)) MC_BULK(f, 1, 0); continue; }
                        if (RARE(force)) { cmd = CMD_BULK; c_f0 = f; c_run = 1; c_detect = 0; break; }
                        continue;
                    }
                    if (LIKELY(any_cand && e1 - e0 <= 64 && !ev_del)) {
                        // ---- common case 2: bursts may start, none ends; the frame's entries fit the lanes ----
                        const int i = e0 + lane;
                        float c_rel = -1.0f;
                        int c_bin = 0;
                        if (i < e1) {
                            const ListEntry e = s_ent[i];
                            const int bin = e.bin & 0x3FFF;
                            if (((s_crossT[bin] >> k) & 1u) && UNMASKED(bin) && VALID_BIN(bin)) {
                                c_bin = bin;
                                c_rel = e.mag / s_sum[bin];
                            }
                        }
                        // create_new_bursts (:556-591) == repeated arg-max over the candidates not yet masked by a
                        // burst created in this frame (rel > 0: its IEEE bits order like the value)
                        while (true) {
                            unsigned long long key = c_rel > 0.0f
                                ? (((unsigned long long)__float_as_uint(c_rel) << 32) | (unsigned)(~c_bin)) : 0ull;
                            key = wave_max_u64(key);
                            if (key == 0ull) break;
                            const float br = __uint_as_float((unsigned)(key >> 32));
                            const int bb = (int)~(unsigned)key;
                            if (RARE(occ == ~0ull)) { abort_code |= 4; break; }
                            const int sl = __builtin_ctzll(~occ);
                            if (lane == sl) {
                                r_cb = bb;
                                r_peak = br;
                                r_base = s_sum[bb];
                                r_start = idx - (uint64_t)P.pre_len;
                                r_la = r_start;
                                r_id = burst_id;
                            }
                            occ |= 1ull << sl;
                            burst_id += 10;
                            MASK_EDIT(bb, 0);
                            if (c_bin >= bb - half_bw && c_bin <= bb + half_bw) c_rel = -1.0f;   // now masked
                            hc_valid = false;
                        }
                        WAVE_SYNC();
                        n_cand = 0;
                        state = S_CPLX_B;                                         // squelch check / decay only
                        TK(13, t4_);
                        continue;
                    }
                    if (any_cand) {
                        // peaks of this frame under the PREVIOUS frame's mask (remove_peaks_around_bursts, :522-525)
                        for (int base = e0; base < e1; base += 64) {
                            const int i = base + lane;
                            bool cand = false;
                            PeakCand c;
                            c.rel = 0.0f; c.bin = 0;
                            if (i < e1) {
                                const ListEntry e = s_ent[i];
                                const int bin = e.bin & 0x3FFF;
                                if (((s_crossT[bin] >> k) & 1u) && UNMASKED(bin) && VALID_BIN(bin)) {
                                    cand = true;
                                    c.bin = bin;
                                    c.rel = e.mag / s_sum[bin];
                                }
                            }
                            const unsigned long long cm = __ballot(cand);
                            if (cand && n_cand + __popcll(cm & lt_mask) < kCandCap) s_cand[n_cand + __popcll(cm & lt_mask)] = c;
                            n_cand += __popcll(cm);
                        }
                        if (RARE(n_cand > kCandCap)) { abort_code |= 32; continue; }
                        WAVE_SYNC();
                    }
                    // delete_gone_bursts (:490-518): emitted in list order == ascending id
                    bool force = false;
                    if (ev_del) {
                        const bool mine = (ev_del >> lane) & 1;
                        int rank = 0;
                        unsigned long long d2 = ev_del;
                        while (d2) {
                            const int sl = __builtin_ctzll(d2);
                            d2 &= d2 - 1;
                            rank += (RL64(r_id, sl) < r_id) ? 1 : 0;
                        }
                        if (mine) {
                            ActiveBurst b;
                            b.id = r_id; b.center_bin = r_cb; b.peak_rel = r_peak; b.start = r_start; b.base_sum = r_base; b.pad = 0;
                            b.last_active = r_la;
                            force = P.max_len > 0 && (r_la - r_start > (uint64_t)P.max_len);
                            PUSH_GONE(b, idx, n_gone + rank);
                        }
                        n_gone += __popcll(ev_del);
                        CHECK_GONE();
                        force = __any(force) != 0;
                        // update_burst_mask (:482-486): only the deleted bursts' ranges can change
                        d2 = ev_del;
                        while (d2) {
                            const int sl = __builtin_ctzll(d2);
                            d2 &= d2 - 1;
                            MASK_EDIT(__builtin_amdgcn_readlane(r_cb, sl), 1);
                            WAVE_SYNC();
                        }
                        occ &= ~ev_del;
                        d2 = occ;
                        while (d2) {
                            const int sl = __builtin_ctzll(d2);
                            d2 &= d2 - 1;
                            MASK_EDIT(__builtin_amdgcn_readlane(r_cb, sl), 0);
                            WAVE_SYNC();
                        }
                        hc_valid = false;
                    }
                    state = S_CPLX_B;
                    TK(4, t4_);
                    if (RARE(force)) {                                            // update_filters_post(d, 1)
                        if constexpr (MC) { MC_BULK(f, 1, 0); continue; }
                        cmd = CMD_BULK; c_f0 = f; c_run = 1; c_detect = 0;
                        break;
                    }
                    continue;
                }
                if (state == S_CPLX_B) {
                    const long long t5_ = IRDM_TICK();
                    const uint64_t index = index0 + (uint64_t)f * N;
                    // create_new_bursts (:556-591): descending magnitude, ties by ascending bin, skipping
                    // bins masked by bursts created earlier in the same frame == repeated arg-max
                    while (RARE(n_cand > 0)) {
                        // key = (rel bits, ~bin): rel > 0, so its IEEE bits order like the value; the larger key
                        // is the larger rel, ties the smaller bin
                        unsigned long long key = 0ull;
                        for (int k = lane; k < n_cand; k += 64) {
                            const PeakCand c = s_cand[k];
                            if (UNMASKED(c.bin)) {
                                const unsigned long long kk =
                                    ((unsigned long long)__float_as_uint(c.rel) << 32) | (unsigned)(~c.bin);
                                key = kk > key ? kk : key;
                            }
                        }
                        key = wave_max_u64(key);
                        if (key == 0ull) break;
                        const float br = __uint_as_float((unsigned)(key >> 32));
                        const int bb = (int)~(unsigned)key;
                        if (RARE(occ == ~0ull)) { abort_code |= 4; break; }
                        const int sl = __builtin_ctzll(~occ);
                        float base_v = s_sum[bb];
                        if constexpr (MC) {
                            // a forced update earlier in this frame (delete_gone_bursts, :508-512) moved the sums after
                            // the cache was filled: burst_detect.c:583 reads the updated one
                            if (RARE(hist_idx != hist_before)) { MC_WAIT(); base_v = MC_SUM(bb); }
                        }
                        if (lane == sl) {
                            r_cb = bb;
                            r_peak = br;
                            r_base = base_v;
                            r_start = index - (uint64_t)P.pre_len;
                            r_la = r_start;
                            r_id = burst_id;
                        }
                        occ |= 1ull << sl;
                        burst_id += 10;
                        MASK_EDIT(bb, 0);
                        WAVE_SYNC();
                        hc_valid = false;
                    }
                    if (RARE(abort_code)) continue;
                    bool reset = false;
                    if (RARE(P.max_bursts > 0 && __popcll(occ) > P.max_bursts)) { // squelch (:594-631)
                        const bool mine = ((occ >> lane) & 1) && r_start != index - (uint64_t)P.pre_len;
                        const unsigned long long om = __ballot(mine);
                        int rank = 0;
                        unsigned long long d2 = om;
                        while (d2) {
                            const int sl = __builtin_ctzll(d2);
                            d2 &= d2 - 1;
                            rank += (RL64(r_id, sl) < r_id) ? 1 : 0;
                        }
                        if (mine) {
                            ActiveBurst b;
                            b.id = r_id; b.center_bin = r_cb; b.peak_rel = r_peak; b.start = r_start; b.base_sum = r_base; b.pad = 0;
                            b.last_active = r_la;
                            PUSH_GONE(b, index, n_gone + rank);
                        }
                        n_gone += __popcll(om);
                        CHECK_GONE();
                        occ = 0;
                        MASK_ALL_ONES();
                        squelch += 3;
                        if (squelch >= 10) {
                            hist_idx = 0;
                            primed = 0;
                            squelch = 0;
                            reset = true;
                        }
                        hc_valid = false;
                    } else if (squelch > 0) {
                        squelch--;
                    }
                    WAVE_SYNC();
                    state = S_FRAME_END;
                    TK(5, t5_);
                    if (RARE(reset)) {
                        if constexpr (MC) { MC_PUBLISH(0, 0, MCF_ZERO); cross_valid = false; continue; }
                        cmd = CMD_ZERO;
                        break;
                    }
                    continue;
                }
                // S_FRAME_END: update_filters_post(d, 0) (:698)
                const long long tE_ = IRDM_TICK();
                state = S_TOP;
                if constexpr (MC) {
                    if (RARE(occ == 0 || !primed)) { MC_BULK(f, 1, 0); }
                    else if (RARE(was_quiet || hist_idx != hist_before)) { MC_PUBLISH(0, 0, MCF_VALIDATE); }
                } else {
                    if (RARE(occ == 0 || !primed)) { cmd = CMD_BULK; c_f0 = f; c_run = 1; c_detect = 0; }
                    else if (RARE(was_quiet || hist_idx != hist_before)) { cmd = CMD_VALIDATE; }
                }
                if (RARE(n_gone - gone_base > (unsigned)(kGoneLds - 40))) { WAVE_SYNC(); FLUSH_GONE(); }
                f++;
                TK(8, tE_);
            }
            const long long tP_ = IRDM_TICK();
            if (lane == 0) {
                sh.cmd = cmd; sh.f0 = c_f0; sh.run = c_run; sh.detect = c_detect;
                sh.hist_idx = hist_idx; sh.primed = primed;
                sh.n_bins_old = n_bins; sh.n_bins = 0;
            }
            if (cmd == CMD_BULK) {
                const int tot = hist_idx + c_run;
                if (tot >= kHistory) primed = 1;
                hist_idx = tot % kHistory;
                cross_valid = false;                 // the baseline moves
            } else if (cmd == CMD_ZERO) {
                cross_valid = false;
            }
            tk[7] += IRDM_TICK() - tP_;
        }
        const long long t_l1 = IRDM_TICK();
        tk[6] += t_l1 - t_l0;
        LDS_BARRIER();
        const int cmd = uni(sh.cmd);
        if (cmd == CMD_EXIT) break;

        // ================= dense command, all threads =================
        if (!MC && cmd == CMD_BULK) {
            // consecutive baseline updates (simd_baseline_update + memcpy, burst_detect.c:441-452)
            const int f0 = uni(sh.f0), run = uni(sh.run), detect = uni(sh.detect);
            int hidx = uni(sh.hist_idx), prm = uni(sh.primed);
            float s[J];
#pragma unroll
            for (int j = 0; j < J; j++) s[j] = s_sum[BIN_OF(j)];
            bool bad = false;
            // One CU moves ~64 B/clk through its L1, i.e. ~0.6 us per frame (mag row + history row in, history row
            // out); runs are short (4 frames on average), so groups are sized to the run -- 4, 2, then 1 frame(s)
            // with every load of the group in flight at once and nothing loaded twice.
            int k0 = 0;
            constexpr int GMAX = (64 / J) > 4 ? 4 : ((64 / J) > 0 ? (64 / J) : 1);
            if (GMAX >= 4)
                for (; run - k0 >= 4; k0 += 4)
                    bulk_group<Q, 4>(mag, hist, N, f0 + k0, hidx, prm, s, detect, thr, tid, half_bw, dc, bad);
            if (GMAX >= 2)
                for (; run - k0 >= 2; k0 += 2)
                    bulk_group<Q, 2>(mag, hist, N, f0 + k0, hidx, prm, s, detect, thr, tid, half_bw, dc, bad);
            for (; k0 < run; k0++)
                bulk_group<Q, 1>(mag, hist, N, f0 + k0, hidx, prm, s, detect, thr, tid, half_bw, dc, bad);
#pragma unroll
            for (int j = 0; j < J; j++) s_sum[BIN_OF(j)] = s[j];
            if (bad) atomicOr(&sh.abort, 1);
        } else if (!MC && cmd == CMD_VALIDATE) {
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
        } else if (!MC && cmd == CMD_ZERO) {
#pragma unroll
            for (int j = 0; j < J; j++) s_sum[BIN_OF(j)] = 0.0f;
        } else if (cmd == CMD_CROSS) {
            // rebuild the frame masks of the staged entries [ef, total) with the live baseline, 512 entries at a time
            const int ef = uni(sh.f0), total = uni(sh.run), sb_ = uni(sh.detect), nb_old = uni(sh.n_bins_old);
            for (int i = tid; i < nb_old; i += kFastThreads) s_crossT[s_bins[i]] = 0u;
            __syncthreads();
            for (int base = ef; base < total; base += kFastThreads) {
                const int i = base + tid;
                bool first = false;
                int bin = 0;
                if (i < total) {
                    const ListEntry e = s_ent[i];
                    bin = e.bin & 0x3FFF;
                    const int k = (e.bin >> 14) - sb_;
                    float sv;
                    if constexpr (MC) { sv = MC_SUM(bin); s_sum[bin] = sv; } else { sv = s_sum[bin]; }
                    const float rel = sv > 0 ? e.mag / sv : 0.0f;
                    if (rel > thr) first = atomicOr(&s_crossT[bin], 1u << k) == 0u;
                }
                const unsigned long long fm = __ballot(first);
                if (fm) {
                    int slot0 = 0;
                    if (lane == 0) slot0 = atomicAdd(&sh.n_bins, __popcll(fm));
                    slot0 = __builtin_amdgcn_readfirstlane(slot0);
                    const int slot = slot0 + __popcll(fm & lt_mask);
                    if (first && slot < kMaxBins) s_bins[slot] = (unsigned short)bin;
                }
            }
        }
        LDS_BARRIER();
        if (cmd == CMD_CROSS) { TK(9, t_l1); } else if (cmd == CMD_BULK) { TK(10, t_l1); }
        if (leader) {
            abort_code |= uni(sh.abort);
            if (cmd == CMD_CROSS) { n_bins = uni(sh.n_bins); cross_cmd_done = true; }
        }
        if (cmd == CMD_BULK) { nk[7]++; bulk_frames += uni(sh.run); }
    }

    if (leader) {
        WAVE_SYNC();
        FLUSH_GONE();
        // carried bursts go back in list order (ascending id)
        const bool mine = (occ >> lane) & 1;
        int rank = 0;
        unsigned long long d2 = occ;
        while (d2) {
            const int sl = __builtin_ctzll(d2);
            d2 &= d2 - 1;
            rank += (RL64(r_id, sl) < r_id) ? 1 : 0;
        }
        if (mine) {
            ActiveBurst b;
                            b.id = r_id; b.center_bin = r_cb; b.peak_rel = r_peak; b.start = r_start; b.base_sum = r_base; b.pad = 0;
            b.last_active = r_la;
            st->act[rank] = b;
        }
        if (lane == 0) {
#ifdef IRDM_SCAN_PROFILE
            long long *dbg = reinterpret_cast<long long *>(status + 4);
            dbg[0] = IRDM_TICK() - t_begin;
            for (int i = 0; i < 12; i++) { dbg[1 + i] = tk[i]; dbg[13 + i] = nk[i]; }
            dbg[29] = bulk_frames; dbg[25] = tk[12]; dbg[26] = nk[12]; dbg[27] = tk[13]; dbg[28] = nk[13];
#endif
            if constexpr (MC) {
                // the updaters leave; their verdicts (safety net, stale lists, timeout) are OR-ed into status[0] as well
                if ((int)n_pub < mc_ops_cap)
                    __hip_atomic_store(&mc_ops[n_pub], mc_pack(0, 0, 0, MCF_EXIT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (abort_code | sh.abort) atomicOr(&status[0], abort_code | sh.abort);
            } else {
                status[0] = abort_code | sh.abort;
            }
            if (n_gone > (unsigned)gone_cap) st->overflow = 1;
            st->index = index0 + (uint64_t)n_frames * N;
            st->burst_id = burst_id;
            st->hist_idx = hist_idx;
            st->primed = primed;
            st->squelch = squelch;
            st->n_act = __popcll(occ);
            st->n_gone = n_gone;
        }
    }
    __syncthreads();
    if constexpr (!MC) {
#pragma unroll
        for (int j = 0; j < J; j++) sum_g[BIN_OF(j)] = s_sum[BIN_OF(j)];
    }
#undef MC_PUBLISH
#undef MC_BULK
#undef MC_WAIT
#undef MC_SUM
#undef TK
#undef WAVE_SYNC
#undef RARE
#undef LIKELY
#undef LDS_BARRIER
#undef BIN_OF
#undef VALID_BIN
#undef UNMASKED
#undef MASK_EDIT
#undef MASK_ALL_ONES
#undef RL64
#undef PUSH_GONE
#undef CHECK_GONE
#undef FLUSH_GONE
}

size_t scan_fast_lds_bytes(int n)
{
    return (size_t)n * 4 + (size_t)n * 4 + (size_t)n / 8 + sizeof(ListEntry) * kStageCap + sizeof(PeakCand) * kCandCap +
           sizeof(GoneBurst) * kGoneLds + sizeof(ActiveBurst) * kFastMaxActive + 2 * kMaxBins + sizeof(FastShared) + 16;
}

int launch_detect_scan_fast(const DetParams &P, DetState *st, float *sum, float *hist, const float *mag,
                            int n_frames, const unsigned *counts, const unsigned *goff,
                            const ListEntry *compact, const float *pre, GoneBurst *gone, int gone_cap,
                            int *status, unsigned long long *mc_ops, int mc_ops_cap, unsigned *mc_done,
                            int mc_updaters, hipStream_t stream)
{
    const int Q = P.n / (4 * kFastThreads);
    if (Q < 1) return -1;
    const size_t lds = scan_fast_lds_bytes(P.n);
    // mc_updaters > 0: multi-CU form, 1 leader + mc_updaters workgroups (mc_ops / mc_done zeroed by the caller)
    const bool mc = mc_updaters > 0 && mc_ops && mc_done;
    if (mc && mc_updaters > kMcMaxUpdaters) return -1;
#define IRDM_LAUNCH_FAST(JJ)                                                                     \
    do {                                                                                         \
        if (mc) {                                                                                \
            (void)hipFuncSetAttribute((const void *)detect_scan_fast_kernel<JJ, true>,           \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
            hipLaunchKernelGGL((detect_scan_fast_kernel<JJ, true>), dim3(1 + mc_updaters),       \
                               dim3(kFastThreads), lds, stream, P, st, sum, hist, mag, n_frames, \
                               counts, goff, compact, pre, gone, gone_cap, status, mc_ops,       \
                               mc_ops_cap, mc_done);                                             \
        } else {                                                                                 \
            (void)hipFuncSetAttribute((const void *)detect_scan_fast_kernel<JJ, false>,          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
            hipLaunchKernelGGL((detect_scan_fast_kernel<JJ, false>), dim3(1), dim3(kFastThreads),\
                               lds, stream, P, st, sum, hist, mag, n_frames, counts, goff,       \
                               compact, pre, gone, gone_cap, status,                             \
                               (unsigned long long *)nullptr, 0, (unsigned *)nullptr);           \
        }                                                                                        \
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
