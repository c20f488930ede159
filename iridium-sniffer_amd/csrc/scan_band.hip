// scan_band.hip -- detector scan, band-parallel speculative form (burst_detect.c:426-632, :689-698).
//
// The reference walks the frames of a stream one after the other because (a) the noise floor is a float running
// sum that is updated only in frames that end with no burst active (update_filters_post, :438-454) or that delete
// an over-long burst (:516-517), and (b) bursts live across frames.  Everything else is local in frequency
// (band_core.hpp).  This file turns the walk into whole-chip passes over one chunk:
//
//   plan    (1 workgroup)  from the SPECULATED per-frame update vector u[f] in {0,1,2}: prefix sums, the list of
//                          update steps, which intermediate sums are needed (only those a listed bin is tested
//                          against), and -- from the second round on -- the verdict on the previous round
//   sums    (1 lane / bin) the exact recurrence sum = (sum - oldest) + mag (simd_baseline_update) along the
//                          planned steps, loads software-pipelined; writes the needed snapshots
//   cross   (1 WG / frame) relative magnitude = mag / sum > threshold (simd_relative_mag, exact division) for the
//                          bins the prefilter listed -> crossing bits, relative magnitudes, band occupancy
//   walk    (1 lane / band x 64-frame block) the state machine per band and activity segment (band_core.hpp)
//   (verify: the next round's plan checks that neighbouring bands agree on every burst within burst_width/2 of
//            their common boundary, one wavefront per boundary)
//
// A round is accepted when the update vector the bands produce equals the speculated one (then, by induction over
// the frames, every sum, crossing and burst equals the sequential result: the state at frame f depends only on
// u[0..f-1], which are the true ones if the round reproduces them), the boundaries agree (then the union of the
// bands' own bursts is a history in which every decision follows the reference's rules from its +-burst_width/2
// neighbourhood, i.e. the sequential history), and the bound on simultaneously active bursts stays below
// max_bursts (no squelch, :594).  Otherwise the produced vector becomes the next speculation; round 0 starts from
// "no update" (sums frozen at the chunk start).  Two or three rounds on the benchmark scenes.  Any doubt -- no fixed
// point within kBandRounds, a capacity, a stale prefilter reference, a possible squelch -- leaves the carried state
// untouched (nothing is written before `commit`) and the caller falls back to the sequential kernels.
//
//   commit  (1 workgroup)  ids by a global sort of the creation events (frame, descending relative magnitude,
//                          ascending bin -- the order of :551/:569), finished bursts in emission order, the bursts
//                          carried to the next chunk, DetState, the new sums
//   history (512 WGs)      the last <= 512 update frames' magnitude rows become the history ring
#include "common.hpp"
#include "types.hpp"
#include "kernels.hpp"
#include "band_core.hpp"

namespace irdm {

namespace {

constexpr int kPlanThreads = 1024;          // one workgroup (256 lanes measured +60 us per plan / commit launch: 0.57 -> 0.81 ms per scan)
constexpr int kSumDepth = 32;             // update steps per batch of the sums pass (two batches of loads in flight)

// exclusive scan of in[0..len) into out[0..len), total returned to every thread; one workgroup, kPlanThreads threads
// (each thread a contiguous run; wave scans on the shuffle network, the 16 wave totals through LDS)
__device__ int block_scan(const int32_t *in, int32_t *out, int len, int32_t *s_part)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (len + kPlanThreads - 1) / kPlanThreads;
    const int lo = tid * per, hi = min(lo + per, len);
    int s = 0;
    for (int i = lo; i < hi; i++) s += in[i];
    int incl = s;
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    if (lane == 63) s_part[wave] = incl;
    __syncthreads();
    int wave_base = 0, total = 0;
    for (int w = 0; w < kPlanThreads / 64; w++) {
        const int v = s_part[w];
        if (w < wave) wave_base += v;
        total += v;
    }
    int run = wave_base + incl - s;
    for (int i = lo; i < hi; i++) {
        const int v = in[i];
        out[i] = run;
        run += v;
    }
    __syncthreads();
    return total;
}

// bands i and i+1 must agree on every burst within burst_width/2 of their common boundary (one wavefront per boundary)
__device__ bool boundary_agrees(const BandParams &P, const BandWork &W, int i, int lane)
{
    const int X = (i + 1) * P.band_w;
    const BandRec *A = W.recs + (size_t)i * kBandRecCap, *B = W.recs + (size_t)(i + 1) * kBandRecCap;
    const int na = min((int)W.rec_count[i], kBandRecCap), nb = min((int)W.rec_count[i + 1], kBandRecCap);
    int bad = 0, ca = 0, cb = 0;
    for (int a = lane; a < na; a += 64) {
        if (A[a].cb < X - P.hw || A[a].cb >= X + P.hw) continue;
        ca++;
        bool found = false;
        for (int b = 0; b < nb && !found; b++) found = band_rec_same(A[a], B[b]);
        if (!found) bad = 1;
    }
    for (int b = lane; b < nb; b += 64)
        if (B[b].cb >= X - P.hw && B[b].cb < X + P.hw) cb++;
    for (int d = 32; d; d >>= 1) {
        ca += __shfl_xor(ca, d);
        cb += __shfl_xor(cb, d);
        bad |= __shfl_xor(bad, d);
    }
    return !(bad || ca != cb);
}

__global__ __launch_bounds__(kPlanThreads) void band_plan_kernel(BandParams P, BandWork W, const unsigned *__restrict__ counts,
                                                                 DetState *__restrict__ st, int round)
{
    IRDM_DETECTOR_PRIO();
    __shared__ int32_t s_part[kPlanThreads];
    __shared__ int s_mismatch, s_first, s_status, s_agree_fail;
    __shared__ unsigned s_flags;
    const int tid = threadIdx.x;
    BandCtl *ctl = W.ctl;
    if (round == 0) {
        // a new scan: control block, abort flags and the chunk's finished-burst count start from zero (what three
        // memset launches did before)
        if (tid < (int)(sizeof(BandCtl) / 4)) reinterpret_cast<uint32_t *>(ctl)[tid] = 0;
        if (tid == 0) {
            *W.flags = 0;
            st->n_gone = 0;
        }
        __syncthreads();
    }
    if (ctl->status != 0) return;
    const int F = P.n_frames;

    if (tid == 0) {
        s_mismatch = 0;
        s_first = 0x7fffffff;
        s_flags = 0;
        s_status = 0;
        s_agree_fail = 0;
    }
    __syncthreads();
    if (round == 0) {
        for (int f = tid; f < F; f += kPlanThreads) {
            W.uq[f] = 0;
            W.uf[f] = 0;
            if (counts[f] > (unsigned)P.list_cap) atomicOr(&s_flags, BAND_F_LIST);
        }
        if (tid == 0) ctl->h0 = st->hist_idx;
    } else {
        // verdict on the previous round
        for (int f = tid; f < F; f += kPlanThreads) {
            const int q = ((W.busy[f >> 6] >> (f & 63)) & 1) ? 0 : 1;
            const int fc = (int)((W.forced[f >> 6] >> (f & 63)) & 1);
            if (q != W.uq[f] || fc != W.uf[f]) {
                atomicAdd(&s_mismatch, 1);
                atomicMin(&s_first, f);
            }
        }
        for (int b = tid; b < P.occ_words; b += kPlanThreads)
            if (W.conc[b] >= (unsigned)P.max_bursts) atomicOr(&s_flags, BAND_F_SQUELCH);
        for (int i = tid >> 6; i + 1 < P.n_bands; i += kPlanThreads / 64)
            if (!boundary_agrees(P, W, i, tid & 63) && (tid & 63) == 0) atomicOr(&s_agree_fail, 1);
        __syncthreads();
        if (tid == 0) {
            unsigned fl = s_flags | *W.flags;
            int status = 0;
            if (fl) {
                status = 2;
            } else if (s_mismatch == 0) {
                if (s_agree_fail) {
                    fl |= BAND_F_AGREE;
                    status = 2;
                } else {
                    status = 1;
                }
            } else if (round >= kBandRounds) {
                fl |= BAND_F_ITER;
                status = 2;
            }
            ctl->flags = fl;
            ctl->mismatch = s_mismatch;
            ctl->first_mismatch = s_first;
            ctl->rounds = round;
            ctl->status = status;
            s_status = status;
        }
        __syncthreads();
        if (s_status != 0) return;
        for (int f = tid; f < F; f += kPlanThreads) {
            W.uq[f] = ((W.busy[f >> 6] >> (f & 63)) & 1) ? 0 : 1;
            W.uf[f] = (uint8_t)((W.forced[f >> 6] >> (f & 63)) & 1);
        }
    }
    __syncthreads();
    if (round == 0 && s_flags) {
        if (tid == 0) {
            ctl->flags = s_flags;
            ctl->status = 2;
        }
        return;
    }

    // ---- clear what the round accumulates ----
    for (int i = tid; i < P.n_bands * P.occ_words; i += kPlanThreads) W.occ[i] = 0;
    for (int i = tid; i < P.occ_words; i += kPlanThreads) {
        W.busy[i] = 0;
        W.forced[i] = 0;
        W.conc[i] = 0;
    }
    for (int i = tid; i < P.n_bands; i += kPlanThreads) W.rec_count[i] = 0;

    // ---- update steps ----
    for (int f = tid; f < F; f += kPlanThreads) W.tmp[f] = W.uq[f] + W.uf[f];
    __syncthreads();
    const int n_upd = block_scan(W.tmp, W.cnt_before, F, s_part);
    for (int k = tid; k <= n_upd; k += kPlanThreads) W.need[k] = 0;
    __syncthreads();
    for (int f = tid; f < F; f += kPlanThreads) {
        int k = W.cnt_before[f];
        if (counts[f] > 0) {
            W.need[k] = 1;
            if (W.uf[f]) W.need[k + 1] = 1;
        }
        if (W.uf[f]) W.upd_frame[k++] = f;      // the forced update comes first (:516-517, then :698)
        if (W.uq[f]) W.upd_frame[k] = f;
    }
    __syncthreads();
    const int n_snap = block_scan(W.need, W.tmp, n_upd + 1, s_part);
    for (int k = tid; k <= n_upd; k += kPlanThreads) W.snap_slot[k] = W.need[k] ? W.tmp[k] : -1;
    __syncthreads();
    const int h0 = ctl->h0;
    // (the sums pass reads whole batches of step descriptors: pad them with no-ops that touch valid memory)
    for (int k = n_upd + tid; k < n_upd + 2 * kSumDepth + 1; k += kPlanThreads) W.steps[k] = make_int4(0, 0, -1, 0);
    for (int k = tid; k < n_upd; k += kPlanThreads) {
        W.snap_after[k] = W.snap_slot[k + 1];
        // the row an update replaces: one of the carried history for the first 512 steps, after that the magnitude
        // row written 512 steps earlier (the ring is only materialised by the commit)
        W.old_row[k] = k < kHistory ? -(((h0 + k) % kHistory) + 1) : W.upd_frame[k - kHistory];
        W.steps[k] = make_int4(W.upd_frame[k], W.old_row[k], W.snap_after[k], 0);
    }
    for (int f = tid; f < F; f += kPlanThreads) {
        const int k = W.cnt_before[f];
        const int pre = counts[f] > 0 ? W.snap_slot[k] : -1;
        W.slot_pre[f] = pre;
        W.slot_post[f] = (counts[f] > 0 && W.uf[f]) ? W.snap_slot[k + 1] : pre;
    }
    if (tid == 0) {
        ctl->n_upd = n_upd;
        ctl->n_snap = n_snap;
        ctl->rounds = round + 1;
        if (n_snap > W.snap_cap) {
            ctl->flags = BAND_F_SNAP;
            ctl->status = 2;
        }
    }
}

// ---- sums: one lane per bin along the planned update steps ----

__global__ __launch_bounds__(64) void band_sum_kernel(BandParams P, BandWork W, const float *__restrict__ mag,
                                                      const float *__restrict__ hist, const float *__restrict__ sum,
                                                      const float *__restrict__ pre, float *__restrict__ smin_out,
                                                      const int4 *__restrict__ steps, float *__restrict__ snap)
{
    IRDM_DETECTOR_PRIO();
    const BandCtl *ctl = W.ctl;
    if (ctl->status != 0) return;
    const int b = blockIdx.x * 64 + threadIdx.x;
    const int N = P.n;
    const int n = ctl->n_upd;
    // steps: (frame, old row, snapshot slot after the step, -) per update step, padded with no-ops; a kernel argument of
    // its own (like snap) so that the compiler knows nothing here writes it and fetches it with scalar loads
    float s = sum[b], smin = s;
    const int slot0 = W.snap_slot[0];
    if (slot0 >= 0) snap[(size_t)slot0 * N + b] = s;

    // Two batches of kSumDepth steps are in flight: the loads of batch i+1 are issued before batch i is consumed.  The
    // step descriptors are wave-uniform and read a batch at a time (scalar loads of 16 bytes per step, unguarded: the
    // plan pads the list), so that the per-step work is two vector loads and two float operations.
    float nwA[kSumDepth], olA[kSumDepth], nwB[kSumDepth], olB[kSumDepth];
    int slA[kSumDepth], slB[kSumDepth];
#define IRDM_SUM_LOAD(nw, ol, sl, k0)                                                                   \
    _Pragma("unroll") for (int j = 0; j < kSumDepth; j++) {                                             \
        const int4 st = steps[(k0) + j];                                                                \
        const float *po = st.y < 0 ? hist + (size_t)(-st.y - 1) * N : mag + (size_t)st.y * N;           \
        nw[j] = mag[(size_t)st.x * N + b];                                                              \
        ol[j] = po[b];                                                                                  \
        sl[j] = st.z;                                                                                   \
    }
#define IRDM_SUM_CONSUME(nw, ol, sl, k0)                                                                \
    _Pragma("unroll") for (int j = 0; j < kSumDepth; j++) {                                             \
        if ((k0) + j < n) {                                                                             \
            const float d = s - ol[j]; /* simd_baseline_update: two separately rounded operations */    \
            s = d + nw[j];                                                                              \
            smin = fminf(smin, s);                                                                      \
            if (sl[j] >= 0) snap[(size_t)sl[j] * N + b] = s;                                            \
        }                                                                                               \
    }
    IRDM_SUM_LOAD(nwA, olA, slA, 0)
    for (int k0 = 0; k0 < n; k0 += 2 * kSumDepth) {
        IRDM_SUM_LOAD(nwB, olB, slB, k0 + kSumDepth)
        IRDM_SUM_CONSUME(nwA, olA, slA, k0)
        if (k0 + 2 * kSumDepth < n) { IRDM_SUM_LOAD(nwA, olA, slA, k0 + 2 * kSumDepth) }
        IRDM_SUM_CONSUME(nwB, olB, slB, k0 + kSumDepth)
    }
#undef IRDM_SUM_LOAD
#undef IRDM_SUM_CONSUME
    W.sum_new[b] = s;
    smin_out[b] = smin;
    // the prefilter listed mag > pre = 0.5 * thr * sum_at_chunk_start; a crossing needs mag > thr * sum (within one
    // rounding), so the lists are complete while pre <= 0.9 * thr * (smallest sum the bin went through)
    if (!(pre[b] <= 0.9f * P.thr * smin)) atomicOr(W.flags, BAND_F_STALE);
}

// ---- crossing bits of one frame ----
__global__ __launch_bounds__(256) void band_cross_kernel(BandParams P, BandWork W, const unsigned *__restrict__ counts,
                                                         const ListEntry *__restrict__ entries)
{
    IRDM_DETECTOR_PRIO();
    __shared__ uint32_t s_bits[16384 / 32];
    if (W.ctl->status != 0) return;
    const int f = blockIdx.x, tid = threadIdx.x, N = P.n;
    const unsigned c = counts[f];
    if (c == 0 || c > (unsigned)P.list_cap) return;
    for (int w = tid; w < N / 32; w += 256) s_bits[w] = 0;
    __syncthreads();
    const float *__restrict__ srow = W.snap + (size_t)W.slot_pre[f] * N;
    const ListEntry *__restrict__ e = entries + (size_t)f * P.list_cap;
    float *__restrict__ rq = W.relq + (size_t)f * N;
    for (unsigned i = tid; i < c; i += 256) {
        const int bin = e[i].bin;
        const float base = srow[bin];
        const float rel = base > 0 ? e[i].mag / base : 0.0f;      // simd_relative_mag (simd_generic.c:137-145)
        if (rel > P.thr) {
            atomicOr(&s_bits[bin >> 5], 1u << (bin & 31));
            rq[bin] = rel;
        }
    }
    __syncthreads();
    uint32_t *__restrict__ out = reinterpret_cast<uint32_t *>(W.cross) + (size_t)f * (N / 32);
    for (int w = tid; w < N / 32; w += 256) out[w] = s_bits[w];
    if (tid < P.n_bands) {
        const int w0 = (tid * P.band_w - P.band_w / 2) / 32, nw = 2 * P.band_w / 32;
        uint32_t any = 0;
        for (int k = 0; k < nw; k++) {
            const int w = w0 + k;
            if (w >= 0 && w < N / 32) any |= s_bits[w];
        }
        if (any) atomicOr(reinterpret_cast<unsigned long long *>(&W.occ[(size_t)tid * P.occ_words + (f >> 6)]), 1ull << (f & 63));
    }
}

// ---- walk: lane = band, workgroup = 64-frame block ----
template <int NW>
__global__ __launch_bounds__(64) void band_walk_kernel(BandParams P, BandWork W, BandIO io, const DetState *__restrict__ st)
{
    IRDM_DETECTOR_PRIO();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if (W.ctl->status != 0) return;
    io.act_in = st->act;
    io.n_act_in = st->n_act;
    const int band = threadIdx.x, blk = blockIdx.x;
    if (band >= P.n_bands) return;
    int64_t *l_start = reinterpret_cast<int64_t *>(smem_raw);
    int64_t *l_la = l_start + kBandSlots * 64;
    int32_t *l_cb = reinterpret_cast<int32_t *>(l_la + kBandSlots * 64);
    int32_t *l_cf = l_cb + kBandSlots * 64;
    int32_t *l_seq = l_cf + kBandSlots * 64;
    float *l_rel = reinterpret_cast<float *>(l_seq + kBandSlots * 64);
    float *l_base = l_rel + kBandSlots * 64;
    BandSlots S{ l_start + band, l_la + band, l_cb + band, l_cf + band, l_seq + band, l_rel + band, l_base + band, 64 };

    bool carried = false;
    if (blk == 0) {
        BandWalker<NW> w(P, io, S, band);
        if (w.load_carried() > 0) {
            carried = true;
            w.run(0, true);
        }
    }
    uint64_t starts = band_segment_starts(io.occ + (size_t)band * P.occ_words, blk, P.gap, carried);
    while (starts) {
        const int j = __builtin_ctzll(starts);
        starts &= starts - 1;
        BandWalker<NW> w(P, io, S, band);
        w.run(64 * blk + j, false);
    }
}

// ---- commit ----
__global__ __launch_bounds__(kPlanThreads) void band_commit_kernel(BandParams P, BandWork W, DetState *__restrict__ st,
                                                                   float *__restrict__ sum, GoneBurst *__restrict__ gone,
                                                                   int gone_cap)
{
    IRDM_DETECTOR_PRIO();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint64_t *s_key = reinterpret_cast<uint64_t *>(smem_raw);                    // kBandMaxTotal
    uint16_t *s_rank = reinterpret_cast<uint16_t *>(s_key + kBandMaxTotal);      // kBandMaxTotal
    __shared__ unsigned s_n, s_carried, s_gone;
    BandCtl *ctl = W.ctl;
    if (ctl->status != 1) return;
    const int tid = threadIdx.x;
    if (tid == 0) {
        s_n = 0;
        s_carried = 0;
        s_gone = 0;
    }
    __syncthreads();
    // the bands' own bursts (16 threads per band, only over the records that exist)
    for (int band = tid >> 4; band < P.n_bands; band += kPlanThreads / 16) {
        {
            const int cnt = min((int)W.rec_count[band], kBandRecCap);
            for (int j = tid & 15; j < cnt; j += 16) {
                const int i = band * kBandRecCap + j;
                const BandRec &r = W.recs[i];
                if (!(r.flags & 1)) continue;
                const unsigned at = atomicAdd(&s_n, 1u);
                if (at < (unsigned)kBandMaxTotal) W.tot[at] = i;
                if (r.cf < 0) atomicAdd(&s_carried, 1u);
                if (r.stop >= 0) atomicAdd(&s_gone, 1u);
            }
        }
    }
    __syncthreads();
    const int n = (int)s_n, n_carried = (int)s_carried, n_gone = (int)s_gone;
    if (n > kBandMaxTotal || n_gone > gone_cap) {
        if (tid == 0) {
            ctl->flags |= n > kBandMaxTotal ? BAND_F_TOTAL : BAND_F_GONECAP;
            ctl->n_gone = n_gone;
            ctl->status = 2;
        }
        return;
    }
    // creation order: carried bursts keep their place, new ones by (frame, descending rel, ascending bin) (:551, :569)
    for (int t = tid; t < n; t += kPlanThreads) {
        const BandRec &r = W.recs[W.tot[t]];
        uint64_t key;
        if (r.cf < 0) {
            key = (uint64_t)r.seq;
        } else {
            uint32_t rb;
            __builtin_memcpy(&rb, &r.rel, 4);
            key = (1ull << 63) | ((uint64_t)r.cf << 46) | ((uint64_t)(0xffffffffu - rb) << 14) | (uint64_t)r.cb;
        }
        s_key[t] = key;
    }
    __syncthreads();
    for (int t = tid; t < n; t += kPlanThreads) {
        const uint64_t k = s_key[t];
        int rank = 0;
        for (int u = 0; u < n; u++) rank += s_key[u] < k ? 1 : 0;
        s_rank[t] = (uint16_t)rank;
    }
    __syncthreads();
    const uint64_t id0 = st->burst_id;
    for (int t = tid; t < n; t += kPlanThreads) {
        const BandRec &r = W.recs[W.tot[t]];
        W.ids[t] = r.cf < 0 ? st->act[r.seq].id : id0 + 10ull * (uint64_t)(s_rank[t] - n_carried);
    }
    // emission order: by the frame that deleted the burst, within a frame in list (= creation) order (:490-514);
    // bursts still active follow in creation order
    for (int t = tid; t < n; t += kPlanThreads) {
        const BandRec &r = W.recs[W.tot[t]];
        s_key[t] = r.stop >= 0 ? (((uint64_t)((r.stop - (int64_t)P.idx0) / P.n)) << 32) | s_rank[t]
                               : (1ull << 63) | s_rank[t];
    }
    __syncthreads();
    for (int t = tid; t < n; t += kPlanThreads) {
        const uint64_t k = s_key[t];
        int pos = 0;
        for (int u = 0; u < n; u++) pos += s_key[u] < k ? 1 : 0;
        const BandRec &r = W.recs[W.tot[t]];
        const uint64_t id = W.ids[t];
        if (r.stop >= 0) {
            GoneBurst g;
            g.id = id; g.start = (uint64_t)r.start; g.stop = (uint64_t)r.stop; g.last_active = (uint64_t)r.last_active;
            g.center_bin = r.cb; g.peak_rel = r.rel; g.base_sum = r.base; g.pad = 0;
            gone[pos] = g;
        } else {
            ActiveBurst a;
            a.id = id; a.start = (uint64_t)r.start; a.last_active = (uint64_t)r.last_active;
            a.center_bin = r.cb; a.peak_rel = r.rel; a.base_sum = r.base; a.pad = 0;
            st->act[pos - n_gone] = a;
        }
    }
    for (int b = tid; b < P.n; b += kPlanThreads) sum[b] = W.sum_new[b];
    if (tid == 0) {
        const int F = P.n_frames;
        st->index += (uint64_t)F * (uint64_t)P.n;
        st->burst_id = id0 + 10ull * (uint64_t)(n - n_carried);
        st->hist_idx = (ctl->h0 + ctl->n_upd) % kHistory;
        st->squelch = st->squelch > F ? st->squelch - F : 0;        // :629-630, once per frame
        st->n_act = n - n_gone;
        st->n_gone = (uint32_t)n_gone;
        ctl->n_gone = n_gone;
        ctl->n_total = n;
        ctl->committed = 1;
    }
}

constexpr int kExportBlocks = 32;

__global__ __launch_bounds__(256) void band_history_kernel(BandParams P, BandWork W, const float *__restrict__ mag,
                                                           float *__restrict__ hist, const DetState *__restrict__ st,
                                                           const uint32_t *__restrict__ gone, int gone_cap,
                                                           uint32_t *__restrict__ hp_gone, uint32_t *__restrict__ hp_hdr,
                                                           uint32_t *__restrict__ hp_ctl)
{
    IRDM_DETECTOR_PRIO();
    const BandCtl *ctl = W.ctl;
    // the scan's verdict and records go to the host whatever the verdict is (types.hpp, gone_export_body)
    if (hp_hdr && blockIdx.x < kExportBlocks)
        gone_export_body(st, gone, gone_cap, hp_gone, hp_hdr, reinterpret_cast<const uint32_t *>(ctl), hp_ctl,
                         (int)(sizeof(BandCtl) / 4), kExportBlocks);
    if (ctl->status != 1 || !ctl->committed) return;
    const int k = ctl->n_upd - 1 - (int)blockIdx.x;
    if (k < 0) return;
    const float4 *src = reinterpret_cast<const float4 *>(mag + (size_t)W.upd_frame[k] * P.n);
    float4 *dst = reinterpret_cast<float4 *>(hist + (size_t)((ctl->h0 + k) % kHistory) * P.n);
    for (int i = threadIdx.x; i < P.n / 4; i += 256) dst[i] = src[i];
}

}  // namespace

int band_list_cap(int n) { return n < kBandListCap ? n : kBandListCap; }

int band_scan_supported(const DetParams &D, BandParams *out, int n_frames, uint64_t idx0)
{
    BandParams P;
    P.n = D.n;
    P.nw64 = D.n / 64;
    P.n_frames = n_frames;
    P.occ_words = (n_frames + 63) / 64;
    P.hw = D.width / 2;
    P.pre_len = D.pre_len;
    P.post_len = D.post_len;
    P.max_len = D.max_len;
    P.max_bursts = D.max_bursts;
    P.band_w = P.hw <= 20 ? 128 : 256;
    P.n_bands = D.n / P.band_w;
    P.gap = (D.post_len + D.n - 1) / D.n;
    P.thr = D.threshold;
    P.idx0 = idx0;
    P.list_cap = band_list_cap(D.n);
    if (out) *out = P;
    // what the kernels assume: at most 64 bands (one wavefront), the halo wider than two masks, a segment cut within
    // one 64-frame word, at least one whole band
    if (D.n < 2048 || D.n > 16384 || P.n_bands < 1 || P.n_bands > 64) return 0;
    if (P.hw < 1 || 2 * P.hw + 8 > P.band_w / 2) return 0;
    if (P.gap < 1 || P.gap >= 64) return 0;
    if (D.max_bursts <= 0 || D.max_bursts > kMaxActive - 64) return 0;
    return 1;
}

size_t band_work_bytes(int n, size_t max_chunk)
{
    const size_t F = max_chunk / (size_t)n + 2;
    size_t b = 0;
    auto add = [&](size_t x) { b += (x + 255) & ~(size_t)255; };
    add(sizeof(BandCtl));
    add(F); add(F);                                  // uq, uf
    add(4 * (F + 2)); add(4 * (2 * F + 4));          // cnt_before, tmp
    add(4 * (2 * F + 4)); add(4 * (2 * F + 4)); add(4 * (2 * F + 4));   // upd_frame, old_row, snap_after
    add(16 * (2 * F + 4 + 3 * kSumDepth));                              // steps
    add(4 * (2 * F + 4)); add(4 * (2 * F + 4));      // need, snap_slot
    add(4 * F); add(4 * F);                          // slot_pre, slot_post
    add(F * (size_t)n / 8);                          // cross
    add(F * (size_t)n * 4);                          // relq
    add((F + 2) * (size_t)n * 4);                    // snap
    add(64 * ((F + 63) / 64) * 8);                   // occ
    add(((F + 63) / 64) * 8); add(((F + 63) / 64) * 8); add(((F + 63) / 64) * 4);   // busy, forced, conc
    add(sizeof(BandRec) * 64 * kBandRecCap); add(4 * 64);                            // recs, rec_count
    add(4 * (size_t)n); add(4 * kBandMaxTotal); add(8 * kBandMaxTotal); add(256);    // sum_new, tot, ids, flags
    return b;
}

int band_work_carve(BandWork *W, void *base, int n, size_t max_chunk)
{
    const size_t F = max_chunk / (size_t)n + 2;
    unsigned char *p = static_cast<unsigned char *>(base);
    auto take = [&](size_t x) { void *q = p; p += (x + 255) & ~(size_t)255; return q; };
    W->ctl = static_cast<BandCtl *>(take(sizeof(BandCtl)));
    W->uq = static_cast<uint8_t *>(take(F));
    W->uf = static_cast<uint8_t *>(take(F));
    W->cnt_before = static_cast<int32_t *>(take(4 * (F + 2)));
    W->tmp = static_cast<int32_t *>(take(4 * (2 * F + 4)));
    W->upd_frame = static_cast<int32_t *>(take(4 * (2 * F + 4)));
    W->old_row = static_cast<int32_t *>(take(4 * (2 * F + 4)));
    W->snap_after = static_cast<int32_t *>(take(4 * (2 * F + 4)));
    W->steps = static_cast<int4 *>(take(16 * (2 * F + 4 + 3 * kSumDepth)));
    W->need = static_cast<int32_t *>(take(4 * (2 * F + 4)));
    W->snap_slot = static_cast<int32_t *>(take(4 * (2 * F + 4)));
    W->slot_pre = static_cast<int32_t *>(take(4 * F));
    W->slot_post = static_cast<int32_t *>(take(4 * F));
    W->cross = static_cast<uint64_t *>(take(F * (size_t)n / 8));
    W->relq = static_cast<float *>(take(F * (size_t)n * 4));
    W->snap = static_cast<float *>(take((F + 2) * (size_t)n * 4));
    W->snap_cap = (int)(F + 2);
    W->occ = static_cast<uint64_t *>(take(64 * ((F + 63) / 64) * 8));
    W->busy = static_cast<uint64_t *>(take(((F + 63) / 64) * 8));
    W->forced = static_cast<uint64_t *>(take(((F + 63) / 64) * 8));
    W->conc = static_cast<uint32_t *>(take(((F + 63) / 64) * 4));
    W->recs = static_cast<BandRec *>(take(sizeof(BandRec) * 64 * kBandRecCap));
    W->rec_count = static_cast<uint32_t *>(take(4 * 64));
    W->sum_new = static_cast<float *>(take(4 * (size_t)n));
    W->tot = static_cast<uint32_t *>(take(4 * kBandMaxTotal));
    W->ids = static_cast<uint64_t *>(take(8 * kBandMaxTotal));
    W->flags = static_cast<uint32_t *>(take(256));
    return 0;
}

// Enqueue the whole band scan of one chunk on `stream`.  Nothing of the carried state (st, sum, hist) is written
// unless BandCtl::status ends as 1 (accepted); the caller reads the control block afterwards.
int launch_band_scan(const DetParams &D, BandWork W, DetState *st, float *sum, float *hist, const float *mag,
                     int n_frames, uint64_t idx0, const unsigned *counts, const ListEntry *entries, const float *pre,
                     float *smin, GoneBurst *gone, int gone_cap, int round_begin, int round_end, GoneBurst *hp_gone,
                     uint32_t *hp_hdr, void *hp_ctl, int hp_cap, hipStream_t stream)
{
    // Rounds [round_begin, round_end) and the verdict on the last of them.  The host enqueues rounds 0 .. kBandFirst - 1
    // (two or three suffice on every scene measured; a round that is not needed is four empty launches) and, if the
    // verdict is still open (status 0, no flags), the rest up to kBandRounds with round_begin = kBandFirst: everything a
    // round needs from the one before lives in the workspace.
    BandParams P;
    if (!band_scan_supported(D, &P, n_frames, idx0) || n_frames < 1) return -1;
    // (round 0's plan pass resets the control block, the flags and the finished-burst count)
    BandIO io;
    io.cross = W.cross;
    io.occ = W.occ;
    io.relq = W.relq;
    io.snap = W.snap;
    io.slot_post = W.slot_post;
    io.act_in = nullptr;        // the walk kernel reads the carried bursts from *st
    io.n_act_in = 0;
    io.recs = W.recs;
    io.rec_count = W.rec_count;
    io.busy = W.busy;
    io.forced = W.forced;
    io.conc = W.conc;
    io.flags = W.flags;
    const size_t walk_lds = (size_t)kBandSlots * 64 * (8 + 8 + 4 + 4 + 4 + 4 + 4);
    const size_t commit_lds = (size_t)kBandMaxTotal * (8 + 2);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)band_walk_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)walk_lds);
        (void)hipFuncSetAttribute((const void *)band_walk_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)walk_lds);
        (void)hipFuncSetAttribute((const void *)band_commit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)commit_lds);
        attr_done = true;
    }
    for (int round = round_begin; round <= round_end; round++) {
        // (a continuation starts behind the plan its predecessor's verdict pass already made)
        if (round > round_begin || round_begin == 0)
            hipLaunchKernelGGL(band_plan_kernel, dim3(1), dim3(kPlanThreads), 0, stream, P, W, counts, st, round);
        if (round == round_end) break;
        hipLaunchKernelGGL(band_sum_kernel, dim3(P.n / 64), dim3(64), 0, stream, P, W, mag, hist, sum, pre, smin, W.steps, W.snap);
        hipLaunchKernelGGL(band_cross_kernel, dim3(n_frames), dim3(256), 0, stream, P, W, counts, entries);
        if (P.band_w == 128)
            hipLaunchKernelGGL((band_walk_kernel<4>), dim3(P.occ_words), dim3(64), walk_lds, stream, P, W, io, st);
        else
            hipLaunchKernelGGL((band_walk_kernel<8>), dim3(P.occ_words), dim3(64), walk_lds, stream, P, W, io, st);
    }
    hipLaunchKernelGGL(band_commit_kernel, dim3(1), dim3(kPlanThreads), commit_lds, stream, P, W, st, sum, gone, gone_cap);
    static_assert(kHistory >= kExportBlocks, "the export rides on the history pass's first workgroups");
    hipLaunchKernelGGL(band_history_kernel, dim3(kHistory), dim3(256), 0, stream, P, W, mag, hist, st,
                       reinterpret_cast<const uint32_t *>(gone), hp_cap < gone_cap ? hp_cap : gone_cap,
                       reinterpret_cast<uint32_t *>(hp_gone), hp_hdr, static_cast<uint32_t *>(hp_ctl));
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace irdm
