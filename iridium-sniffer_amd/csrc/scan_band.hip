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
//   walk    (1 wavefront / band and activity segment) the state machine per band (band_wave.hpp; band_core.hpp is the
//                          same machine with a lane per band: the CPU test's and option band_walk_wave 0's form)
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
//   commit  (1 workgroup; inside the plan pass that accepts the round)  ids by a global sort of the creation events (frame, descending relative magnitude,
//                          ascending bin -- the order of :551/:569), finished bursts in emission order, the bursts
//                          carried to the next chunk, DetState, the new sums
//   history (512 WGs)      the last <= 512 update frames' magnitude rows become the history ring
#include <atomic>
#include <mutex>
#include "common.hpp"
#include "types.hpp"
#include "kernels.hpp"
#include "band_core.hpp"
#include "band_wave.hpp"

namespace irdm {

namespace {

// a launch voided by its own first pass (see BandParams::chained): bar[5] carries its serial number
__device__ __forceinline__ bool band_void(const BandParams &P, const BandWork &W)
{
    return __hip_atomic_load(W.bar + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)P.serial;
}

// diagnostic timeline (BandParams::tl_sel >= 0): earliest start and latest end over the workgroups of a pass
struct TlScope {
    unsigned long long *lo, *hi;
    __device__ __forceinline__ TlScope(const BandParams &P, const BandWork &W, int slot) : lo(nullptr), hi(nullptr)
    {
        if (P.tl_sel >= 0 && (threadIdx.x & 63) == 0 && slot >= 0 && slot < kBandTlSlots) {      // (every wavefront)
            lo = W.tl + (size_t)P.tl_sel * 2 * kBandTlSlots + slot;
            hi = lo + kBandTlSlots;
            atomicMin(lo, (unsigned long long)wall_clock64());
        }
    }
    // (the pass proper is over: what the workgroup does from here on is stamped by another scope)
    __device__ __forceinline__ void leave()
    {
        if (hi) atomicMax(hi, (unsigned long long)wall_clock64());
        hi = nullptr;
    }
    __device__ __forceinline__ ~TlScope() { leave(); }
};

constexpr int kPlanThreads = 1024;          // one workgroup (256 lanes measured +60 us per plan / commit launch: 0.57 -> 0.81 ms per scan)
constexpr int kSumDepth = 32;             // update steps per batch of the sums pass (two batches of loads in flight)

// exclusive scan of in[0..len) into out[0..len), total returned to every thread; one workgroup of NT threads
// (each thread a contiguous run; wave scans on the shuffle network, the wave totals through LDS)
template <int NT>
__device__ int block_scan(const int32_t *in, int32_t *out, int len, int32_t *s_part)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (len + NT - 1) / NT;
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
    for (int w = 0; w < NT / 64; w++) {
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
        // (only B's records in the same zone can match: the centre bin is compared first)
        for (int b = 0; b < nb && !found; b++) found = B[b].cb == A[a].cb && band_rec_same(A[a], B[b]);
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

// The same test for every boundary at once (16 lanes per boundary, one workgroup of >= 1008 threads): the wavefront form
// above walks band i+1's records one dependent load after the other for every record of band i in the zone -- 29 us
// alone and 49 us in run of a plan pass.  Here the records of both bands inside the zone (a handful: bursts whose centre
// lies within burst_width/2 of the boundary) are listed in LDS first -- band i by index, band i+1 by (centre bin, low
// word of the start, index) -- and each zone record of band i looks its partner up in that list; only a candidate whose
// key matches is fetched and compared in full.  More than kZoneCap zone records on one side: the plain search.
struct ZoneEnt {
    uint32_t cb, start_lo, idx;
};
constexpr int kZoneCap = 32;
constexpr size_t kZoneLdsBytes = 64 * kZoneCap * (sizeof(ZoneEnt) + sizeof(uint32_t)) + 128 * sizeof(uint32_t);

// (nt: threads of the workgroup, a multiple of 16; 16 lanes per boundary, the boundaries in passes of nt / 16)
__device__ void boundaries_agree_lds(const BandParams &P, const BandWork &W, uint8_t *lds, int *s_fail, int nt)
{
    ZoneEnt *zb = reinterpret_cast<ZoneEnt *>(lds);                           // [64][kZoneCap]
    uint32_t *za = reinterpret_cast<uint32_t *>(zb + 64 * kZoneCap);           // [64][kZoneCap]
    uint32_t *zc = za + 64 * kZoneCap;                                         // [64][2]
    const int tid = threadIdx.x, gl = tid & 15;
    for (int t = tid; t < 128; t += nt) zc[t] = 0;
    __syncthreads();
    for (int i = tid >> 4; i + 1 < P.n_bands; i += nt >> 4) {
        const int X = (i + 1) * P.band_w;
        const BandRec *A = W.recs + (size_t)i * kBandRecCap, *B = W.recs + (size_t)(i + 1) * kBandRecCap;
        const int na = min((int)W.rec_count[i], kBandRecCap), nb = min((int)W.rec_count[i + 1], kBandRecCap);
        const int nmax = max(na, nb);
        for (int k = gl; k < nmax; k += 16) {
            // (both bands' loads of a stride are in flight together)
            const int cba = k < na ? A[k].cb : -0x40000000;
            const int cbb = k < nb ? B[k].cb : -0x40000000;
            const uint32_t stb = k < nb ? (uint32_t)B[k].start : 0u;
            if (cba >= X - P.hw && cba < X + P.hw) {
                const uint32_t at = atomicAdd(&zc[2 * i], 1u);
                if (at < (uint32_t)kZoneCap) za[i * kZoneCap + at] = (uint32_t)k;
            }
            if (cbb >= X - P.hw && cbb < X + P.hw) {
                const uint32_t at = atomicAdd(&zc[2 * i + 1], 1u);
                if (at < (uint32_t)kZoneCap) zb[i * kZoneCap + at] = ZoneEnt{ (uint32_t)cbb, stb, (uint32_t)k };
            }
        }
    }
    __syncthreads();
    int bad = 0;
    for (int i = tid >> 4; i + 1 < P.n_bands; i += nt >> 4) {
        const int X = (i + 1) * P.band_w;
        const BandRec *A = W.recs + (size_t)i * kBandRecCap, *B = W.recs + (size_t)(i + 1) * kBandRecCap;
        const int na = min((int)W.rec_count[i], kBandRecCap), nb = min((int)W.rec_count[i + 1], kBandRecCap);
        const int ca = (int)zc[2 * i], cb = (int)zc[2 * i + 1];
        if (ca != cb) {
            bad = 1;
        } else if (ca > kZoneCap) {
            for (int a = gl; a < na; a += 16) {
                if (A[a].cb < X - P.hw || A[a].cb >= X + P.hw) continue;
                bool found = false;
                for (int b = 0; b < nb && !found; b++) found = B[b].cb == A[a].cb && band_rec_same(A[a], B[b]);
                if (!found) bad = 1;
            }
        } else {
            for (int k = gl; k < ca; k += 16) {
                const BandRec ra = A[za[i * kZoneCap + k]];
                bool found = false;
                for (int j = 0; j < cb && !found; j++) {
                    const ZoneEnt e = zb[i * kZoneCap + j];
                    if (e.cb == (uint32_t)ra.cb && e.start_lo == (uint32_t)ra.start) found = band_rec_same(ra, B[e.idx]);
                }
                if (!found) bad = 1;
            }
        }
    }
    if (bad) atomicOr(s_fail, 1);
}

// step descriptor of an update step: frame uf, the row it replaces (old < 0: row -old - 1 of the carried history, else the
// magnitude row of frame old), snapshot slot after the step (< 0: none).  Rows are at most 64 KB and there are at most
// max_chunk / n + 2 of them: every offset is far below 4 GB.
__device__ __forceinline__ SumStep make_step(const BandParams &P, int uf, int old, int slot)
{
    SumStep s;
    const uint32_t row = (uint32_t)P.n * 4u;
    s.nw_off = (uint32_t)uf * row;
    s.ol_off = (uint32_t)(old < 0 ? -old - 1 : old) * row;
    s.snap_off = slot >= 0 ? (uint32_t)slot * row : ~0u;
    s.pad = 0;
    return s;
}
__device__ __forceinline__ SumStep noop_step()
{
    SumStep s;
    s.nw_off = 0;
    s.ol_off = 0;
    s.snap_off = ~0u;
    s.pad = 0;
    return s;
}

// shared scratch of the plan pass
struct PlanShared {
    int32_t part[kPlanThreads / 64 + 1];
    int mismatch, first, status, agree_fail;
    unsigned flags;
    int kstar;
};

constexpr int kPlanLdsFrames = 8192;     // frames the LDS form of the plan holds (need flags + snapshot slots: 80 KB)
constexpr size_t kPlanLdsBytes = ((2 * kPlanLdsFrames + 2 + 15) & ~15) + sizeof(int32_t) * (2 * kPlanLdsFrames + 2);
static_assert(kZoneLdsBytes <= kPlanLdsBytes, "the boundary test's lists live in the plan's LDS before the plan needs it");

// lds: kPlanLdsBytes of dynamic LDS, or nullptr (the general path through the workspace arrays)
// u_busy / u_forced: the busy / forced bitmaps the round's update vector is taken from -- W's own (what the previous
// round's walk produced), or, for the first round of a scan fed by a speculation pass (P.spec_in, round 1), that pass's
template <int NT>
__device__ void band_plan_body(const BandParams &P, const BandWork &W, const unsigned *__restrict__ counts,
                               DetState *__restrict__ st, int round, PlanShared &sh, uint8_t *lds,
                               const uint64_t *u_busy = nullptr, const uint64_t *u_forced = nullptr)
{
    // first: this pass opens the scan (round 0, or round 1 of a scan whose round 0 was a speculation pass elsewhere)
    const bool first = round == 0 || (P.spec_in && round == 1);
    if (!(P.spec_in && round == 1) || u_busy == nullptr) {
        u_busy = W.busy;
        u_forced = W.forced;
    }
    constexpr int kPlanThreads = NT;      // (shadows the launch constant: every stride below is the workgroup's size)
    int32_t *s_part = sh.part;
    int &s_mismatch = sh.mismatch, &s_first = sh.first, &s_status = sh.status, &s_agree_fail = sh.agree_fail;
    unsigned &s_flags = sh.flags;
    const int tid = threadIdx.x;
    BandCtl *ctl = W.ctl;
    if (first) {
        // the scan in front committed (bar[4], set by its commit pass)?  A chained launch that finds it did not leaves
        // everything -- workspace, control block, carried state -- as it is: the host continues or redoes that scan
        if (tid == 0) {
            const unsigned prev = W.bar[4];
            s_status = (P.chained && prev != 1u) ? 1 : 0;
            if (s_status) __hip_atomic_store(W.bar + 5, (unsigned)P.serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else W.bar[4] = 0;
        }
        __syncthreads();
        if (s_status) return;
        __syncthreads();
    } else if (band_void(P, W)) {
        return;
    }
    if (first) {
        // a new scan: control block, abort flags and the chunk's finished-burst count start from zero (what three
        // memset launches did before)
        if (tid < (int)(sizeof(BandCtl) / 4)) reinterpret_cast<uint32_t *>(ctl)[tid] = 0;
        if (tid == 0) {
            *W.flags = 0;
            st->n_gone = 0;
        }
        __syncthreads();
        if (round != 0 && tid == 0) ctl->h0 = st->hist_idx;       // (round 0 notes it with its plan below)
    }
    if (ctl->status != 0) return;
    const int F = P.n_frames;
    // phase stamps of the pass (diagnostic: exported with the control block, stat keys "plan_tp_<i>")
    const uint64_t t_begin = wall_clock64();
    uint32_t *tp = ctl->tp + (round == 1 ? 0 : 8);
    auto stamp = [&](int i) {
        if (tid == 0 && round >= 1) tp[i] = (uint32_t)(wall_clock64() - t_begin);
    };

    if (tid == 0) {
        s_mismatch = 0;
        s_first = 0x7fffffff;
        s_flags = 0;
        s_status = 0;
        s_agree_fail = 0;
    }
    __syncthreads();
    if (round == 0) {
        // Round 0 speculates "no update in this chunk": no update steps, and one snapshot -- the carried sums -- if any
        // frame has list entries.  Everything the general path below derives from u is a constant then; written in one
        // pass (the general path's ten dependent passes took 100 us of one workgroup in run for this).
        int any = 0;
        for (int f = tid; f < F; f += kPlanThreads) {
            const unsigned c = counts[f];
            W.uq[f] = 0;
            W.uf[f] = 0;
            W.cnt_before[f] = 0;
            const int slot = c > 0 ? 0 : -1;
            W.slot_pre[f] = slot;
            W.slot_post[f] = slot;
            any |= c > 0 ? 1 : 0;
            if (c > (unsigned)P.list_cap) atomicOr(&s_flags, BAND_F_LIST);
        }
        if (any) atomicOr(&s_mismatch, 1);          // (s_mismatch doubles as "some frame has entries" here)
        for (int i = tid; i < P.n_bands * P.occ_words; i += kPlanThreads) W.occ[i] = 0;
        for (int i = tid; i < P.occ_words; i += kPlanThreads) {
            W.busy[i] = 0;
            W.forced[i] = 0;
            W.conc[i] = 0;
        }
        for (int i = tid; i < P.n_bands; i += kPlanThreads) W.rec_count[i] = 0;
        for (int k = tid; k < 2 * kSumDepth + 1; k += kPlanThreads) W.steps[k] = noop_step();
        __syncthreads();
        if (tid == 0) {
            const int some = s_mismatch;
            ctl->h0 = st->hist_idx;
            W.need[0] = some;
            W.snap_slot[0] = some ? 0 : -1;
            ctl->n_upd = 0;
            ctl->n_snap = some;
            ctl->rounds = 1;
            if (s_flags) {
                ctl->flags = s_flags;
                ctl->status = 2;
            }
        }
        return;
    } else if (first) {
        // no previous round of this scan to judge: the update vector comes from the speculation pass.  What round 0's plan
        // checks of the lists is checked here.
        for (int f = tid; f < F; f += kPlanThreads)
            if (counts[f] > (unsigned)P.list_cap) atomicOr(&s_flags, BAND_F_LIST);
        __syncthreads();
        if (s_flags) {
            if (tid == 0) {
                ctl->flags = s_flags;
                ctl->rounds = 1;
                ctl->status = 2;
            }
            return;
        }
        // (the planned vector, as the next round's verdict compares it with what this round produces)
        for (int f = tid; f < F; f += kPlanThreads) {
            W.uq[f] = ((u_busy[f >> 6] >> (f & 63)) & 1) ? 0 : 1;
            W.uf[f] = (uint8_t)((u_forced[f >> 6] >> (f & 63)) & 1);
        }
        if (P.selfcheck & 32) {
            // (test hook) the guess spoilt in ONE late frame: this round's verdict finds it, the next round's sums pass has a
            // long unchanged prefix to restart behind
            __syncthreads();
            if (tid == 0 && F >= 8) W.uq[(3 * F) / 4] ^= 1;
            __syncthreads();
        }
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
        __syncthreads();
        stamp(0);
        if (P.selfcheck & 2) {
            // (test hook) band i+1's view of every burst near its lower boundary is made to differ from band i's
            for (int i = 1; i < P.n_bands; i++) {
                BandRec *B = W.recs + (size_t)i * kBandRecCap;
                const int X = i * P.band_w, nb = min((int)W.rec_count[i], kBandRecCap);
                for (int k = tid; k < nb; k += kPlanThreads)
                    if (B[k].cb >= X - P.hw && B[k].cb < X + P.hw) B[k].last_active += 1;
            }
            __syncthreads();
        }
        if (lds != nullptr) {
            boundaries_agree_lds(P, W, lds, &s_agree_fail, NT);
            if (P.selfcheck & 1) {
                // (test hook) the wavefront form must give the same answer
                __shared__ int s_other;
                if (tid == 0) s_other = 0;
                __syncthreads();
                for (int i = tid >> 6; i + 1 < P.n_bands; i += kPlanThreads / 64)
                    if (!boundary_agrees(P, W, i, tid & 63) && (tid & 63) == 0) atomicOr(&s_other, 1);
                __syncthreads();
                if (tid == 0 && s_other != s_agree_fail) atomicOr(&s_flags, BAND_F_CHECK);
            }
        } else {
            for (int i = tid >> 6; i + 1 < P.n_bands; i += kPlanThreads / 64)
                if (!boundary_agrees(P, W, i, tid & 63) && (tid & 63) == 0) atomicOr(&s_agree_fail, 1);
        }
        __syncthreads();
        stamp(1);
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

    if (lds != nullptr && F <= kPlanLdsFrames) {
        // ---- the same plan with everything between the passes in registers and LDS (F <= 8192 frames: 10 / 12 MHz
        // chunks up to 64 Mi samples): a thread owns a contiguous run of frames, the scans run over one count per thread,
        // the need flags and snapshot slots live in LDS.  The general path below does ten passes through global
        // arrays, each a round trip through the L2 beside the per-burst chains' traffic (44 us alone, 100-140 us in run).
        constexpr int FPT = kPlanLdsFrames / NT;              // frames per thread
        uint8_t *s_need = lds;                                   // [2 F + 2]
        int32_t *s_slot = reinterpret_cast<int32_t *>(lds + ((2 * kPlanLdsFrames + 2 + 15) & ~15));   // [2 F + 2]
        const int f0 = tid * FPT;
        int uqb = 0, ufb = 0, cnz = 0, c_t = 0;
#pragma unroll
        for (int j = 0; j < FPT; j++) {
            const int f = f0 + j;
            if (f < F) {
                int q = ((u_busy[f >> 6] >> (f & 63)) & 1) ? 0 : 1;
                const int fc = (int)((u_forced[f >> 6] >> (f & 63)) & 1);
                if ((P.selfcheck & 32) && first && round == 1 && F >= 8 && f == (3 * F) / 4) q ^= 1;       // (test hook, as above)
                uqb |= q << j;
                ufb |= fc << j;
                cnz |= (counts[f] > 0 ? 1 : 0) << j;
                c_t += q + fc;
            }
        }
        __syncthreads();                      // (every thread has read the bitmaps: they may be cleared)
        stamp(2);
        for (int i = tid; i < P.n_bands * P.occ_words; i += kPlanThreads) W.occ[i] = 0;
        for (int i = tid; i < P.occ_words; i += kPlanThreads) {
            W.busy[i] = 0;
            W.forced[i] = 0;
            W.conc[i] = 0;
        }
        for (int i = tid; i < P.n_bands; i += kPlanThreads) W.rec_count[i] = 0;
        // exclusive scan of the per-thread update counts
        const int lane = tid & 63, wave = tid >> 6;
        auto scan1 = [&](int v, int &total) {
            int incl = v;
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(incl, d);
                if (lane >= d) incl += o;
            }
            if (lane == 63) s_part[wave] = incl;
            __syncthreads();
            int wb = 0, tot = 0;
            for (int w = 0; w < NT / 64; w++) {
                const int o = s_part[w];
                if (w < wave) wb += o;
                tot += o;
            }
            __syncthreads();
            total = tot;
            return wb + incl - v;
        };
        int n_upd = 0;
        const int base = scan1(c_t, n_upd);
        stamp(3);
        // (a snapshot of every 64th state whether a frame needs it or not: where a later round's sums pass may restart --
        // the same states in every round, so the slots of an unchanged prefix keep their numbers)
        for (int k = tid; k <= n_upd + 1; k += kPlanThreads) s_need[k] = (P.sum_restart && k > 0 && (k & 63) == 0) ? 1 : 0;
        __syncthreads();
        {
            int k = base;
#pragma unroll
            for (int j = 0; j < FPT; j++) {
                const int f = f0 + j;
                if (f < F) {
                    const int q = (uqb >> j) & 1, fc = (ufb >> j) & 1;
                    if (f == s_first) sh.kstar = k;    // (the update steps in front of the first frame whose u changed)
                    if ((cnz >> j) & 1) {
                        s_need[k] = 1;
                        if (fc) s_need[k + 1] = 1;
                    }
                    if (fc) W.upd_frame[k++] = f;      // the forced update comes first (:516-517, then :698)
                    if (q) W.upd_frame[k++] = f;
                }
            }
        }
        __syncthreads();
        stamp(4);
        // snapshot slots: exclusive scan of the need flags, a contiguous run of steps per thread
        const int kpt = (n_upd + 1 + NT - 1) / NT;
        const int k0 = tid * kpt, k1 = min(k0 + kpt, n_upd + 1);
        int nn = 0;
        for (int k = k0; k < k1; k++) nn += s_need[k];
        int n_snap = 0;
        int run = scan1(nn, n_snap);
        for (int k = k0; k < k1; k++) {
            const int nd = s_need[k];
            s_slot[k] = nd ? run : -1;
            run += nd;
        }
        if (tid == 0) W.snap_slot[0] = s_need[0] ? 0 : -1;      // (the sums pass reads slot 0 from the workspace)
        __syncthreads();
        stamp(5);
        const int h0 = ctl->h0;
        for (int k = n_upd + tid; k < n_upd + 2 * kSumDepth + 1; k += kPlanThreads) W.steps[k] = noop_step();
        for (int k = tid; k < n_upd; k += kPlanThreads) {
            const int uf_k = W.upd_frame[k];
            const int old = k < kHistory ? -(((h0 + k) % kHistory) + 1) : W.upd_frame[k - kHistory];
            W.steps[k] = make_step(P, uf_k, old, s_slot[k + 1]);
        }
        {
            int k = base;
#pragma unroll
            for (int j = 0; j < FPT; j++) {
                const int f = f0 + j;
                if (f < F) {
                    const int q = (uqb >> j) & 1, fc = (ufb >> j) & 1, has = (cnz >> j) & 1;
                    const int pre = has ? s_slot[k] : -1;
                    W.slot_pre[f] = pre;
                    W.slot_post[f] = (has && fc) ? s_slot[k + 1] : pre;
                    k += q + fc;
                }
            }
        }
        __syncthreads();
        stamp(6);
        if (tid == 0) {
            ctl->n_upd = n_upd;
            ctl->n_snap = n_snap;
            ctl->rounds = round + 1;
            if (n_snap > W.snap_cap) {
                ctl->flags = BAND_F_SNAP;
                ctl->status = 2;
            }
            // Restart (rounds >= 2 of a scan, i.e. a previous sums pass ran on a plan that agrees with this one up to the first
            // frame whose u changed): that frame has k* update steps in front of it; the steps below k* replace the same rows
            // with the same rows, and the previous pass stored the state after step 64 m - 1 as a snapshot for every m.  The
            // sums pass starts at the last such state at or below k*.  (simd_baseline_update's order per bin is untouched:
            // the prefix IS the previous round's, bit for bit.)
            int kr = 0;
            if (P.sum_restart && !first && round >= 2 && s_first != 0x7fffffff && s_first < F) kr = (sh.kstar / 64) * 64;
            if (kr > n_upd) kr = (n_upd / 64) * 64;
            if (kr >= 64 && s_slot[kr] >= 0) {
                ctl->k_restart = kr;
                ctl->restart_slot = s_slot[kr];
                ctl->n_restarts++;
            } else {
                ctl->k_restart = 0;
                ctl->restart_slot = -1;
            }
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
    const int n_upd = block_scan<NT>(W.tmp, W.cnt_before, F, s_part);
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
    const int n_snap = block_scan<NT>(W.need, W.tmp, n_upd + 1, s_part);
    for (int k = tid; k <= n_upd; k += kPlanThreads) W.snap_slot[k] = W.need[k] ? W.tmp[k] : -1;
    __syncthreads();
    const int h0 = ctl->h0;
    // (the sums pass reads whole batches of step descriptors: pad them with no-ops that touch valid memory)
    for (int k = n_upd + tid; k < n_upd + 2 * kSumDepth + 1; k += kPlanThreads) W.steps[k] = noop_step();
    for (int k = tid; k < n_upd; k += kPlanThreads) {
        W.snap_after[k] = W.snap_slot[k + 1];
        // the row an update replaces: one of the carried history for the first 512 steps, after that the magnitude
        // row written 512 steps earlier (the ring is only materialised by the commit)
        W.old_row[k] = k < kHistory ? -(((h0 + k) % kHistory) + 1) : W.upd_frame[k - kHistory];
        W.steps[k] = make_step(P, W.upd_frame[k], W.old_row[k], W.snap_after[k]);
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

__device__ void band_commit_body(const BandParams &P, const BandWork &W, DetState *__restrict__ st,
                                 float *__restrict__ sum, GoneBurst *__restrict__ gone, int gone_cap,
                                 unsigned char *smem_raw);
// A verdict "accepted" is followed by the commit in the same workgroup -- one launch and its wait less on the scan's critical
// path (a plan pass without its LDS, selfcheck bit 4, leaves the commit to band_commit_kernel).
template <int NT>
__global__ __launch_bounds__(NT) void band_plan_kernel(BandParams P, BandWork W, const unsigned *__restrict__ counts,
                                                       DetState *__restrict__ st, int round, float *__restrict__ sum,
                                                       GoneBurst *__restrict__ gone, int gone_cap, const float *__restrict__ pre,
                                                       float *__restrict__ smin_out, const uint64_t *u_busy,
                                                       const uint64_t *u_forced)
{
    IRDM_DETECTOR_PRIO();
    TlScope tl(P, W, 4 * round);
    __shared__ PlanShared sh;
    extern __shared__ __attribute__((aligned(16))) unsigned char plan_lds[];
    static_assert(kPlanLdsBytes >= (size_t)kBandMaxTotal * 10, "the fused commit sorts in the plan pass's LDS");
    band_plan_body<NT>(P, W, counts, st, round, sh, (P.selfcheck & 16) ? nullptr : plan_lds, u_busy, u_forced);
    if (round == 0 && pre != nullptr) {
        // Round 0 has no update steps: its sums pass would copy the carried sums into snapshot 0 and check the lists'
        // levels against them (band_sum_body with n_upd = 0) -- N loads and stores, done here by the workgroup that is
        // resident anyway instead of a launch of its own (3 us alone, 20 us + 17 us of waiting in front of it in run).
        __syncthreads();
        if (!band_void(P, W) && W.ctl->status == 0) {
            const int slot0 = W.snap_slot[0];
            for (int b = (int)threadIdx.x; b < P.n; b += NT) {
                const float sv = sum[b];
                if (slot0 >= 0) W.snap[(size_t)slot0 * P.n + b] = sv;
                W.sum_new[b] = sv;
                smin_out[b] = sv;
                if (!(pre[b] <= 0.9f * P.thr * sv)) atomicOr(W.flags, BAND_F_STALE);
            }
        }
    }
    if (round >= 1 && !(P.selfcheck & 16)) {
        __syncthreads();
        if (!band_void(P, W) && W.ctl->status == 1) band_commit_body(P, W, st, sum, gone, gone_cap, plan_lds);
    }
}

// ---- sums: one lane per bin along the planned update steps ----

// buffer resource over [p, p + bytes): raw (stride 0), 32-bit float data format, out-of-range reads return zero
__device__ __forceinline__ __amdgpu_buffer_rsrc_t band_rsrc(const void *p, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)(bytes > 0x7fffffffu ? 0x7fffffffu : bytes), 0x00020000);
}

// (b: this lane's bin)
__device__ __forceinline__ void band_sum_body(const BandParams &P, const BandWork &W, const float *__restrict__ mag,
                                              const float *__restrict__ hist, const float *__restrict__ sum,
                                              const float *__restrict__ pre, float *__restrict__ smin_out,
                                              const SumStep *steps, float *snap, int b)
{
    const BandCtl *ctl = W.ctl;
    const int N = P.n;
    const int n = ctl->n_upd;
    // (k_begin > 0: the steps below it are the previous round's, whose pass stored the state in front of step k_begin; the
    // smallest sum so far is then the previous pass's -- over all of ITS steps, a lower bound: the list check stays safe)
    const int k_begin = ctl->k_restart;
    float s = sum[b], smin = s;
    const int slot0 = W.snap_slot[0];
    if (k_begin > 0) {
        s = snap[(size_t)ctl->restart_slot * N + b];
        smin = __builtin_fminf(smin_out[b], s);
    } else if (slot0 >= 0) {
        snap[(size_t)slot0 * N + b] = s;
    }
    // This pass is one wavefront per 64 bins with nothing to hide behind, so what a step costs is its instruction count:
    // with row numbers in the descriptors the 64-bit address arithmetic (13 scalar and one vector instruction per load)
    // was most of the 140 ns a step took.  The rows are read as buffer loads instead -- resource = the whole array, the
    // lane's byte offset in a register that never changes, the row's byte offset straight from the descriptor in the
    // scalar-offset operand -- which leaves two loads, three float operations and the snapshot test per step.
    const size_t row = (size_t)N * 4;
    const __amdgpu_buffer_rsrc_t r_mag = band_rsrc(mag, (size_t)P.n_frames * row);
    const __amdgpu_buffer_rsrc_t r_hist = band_rsrc(hist, (size_t)kHistory * row);
    const __amdgpu_buffer_rsrc_t r_snap = band_rsrc(snap, (size_t)W.snap_cap * row);
    const int boff = b * 4;

    // Two batches of kSumDepth steps are in flight: the loads of batch i+1 are issued before batch i is consumed.  A batch
    // lies on one side of step kHistory (512 = 16 batches), where the replaced rows change from the carried history ring
    // to the chunk's own magnitude rows.
    // The step descriptors are wave-uniform.  Read with scalar loads (rounds 2-3: four groups of sixteen s_load per batch,
    // each waited for before its rows could be asked for) they were a third of the pass: four scalar round trips per 32
    // steps.  Now lane j of a register quadruple holds the descriptor of step k0 + j -- ONE vector load per 64 steps,
    // issued 128 steps ahead (unguarded: the plan pads the list with 2 kSumDepth + 1 no-ops and the workspace has room
    // behind it) -- and a step's offsets reach the loads' scalar-offset operand by v_readlane; the steps that store a
    // snapshot are a ballot.  (2 bursts per Msample, 10 MHz: 429 -> 260 us for the first sums pass.  Measured without
    // further gain: 16 / 32 bins per wavefront, sibling wavefronts reading the rows ahead, and a consumer wavefront of three
    // issue slots per step fed through LDS by three loader wavefronts that also store the snapshots, -13 % --
    // profiles/r4_sched_experiments.txt: with long lists the pass moves two rows in and a snapshot row out per step.)
    static_assert(kHistory % kSumDepth == 0, "a batch of steps must not straddle the history ring's length");
    static_assert(kSumDepth == 32, "two batches per 64 descriptors");
    static_assert(sizeof(SumStep) == 16, "a descriptor is one 16-byte load");
    const int lane = (int)(threadIdx.x & 63);
    const uint4 *steps4 = reinterpret_cast<const uint4 *>(steps);
    float nwA[kSumDepth], olA[kSumDepth], nwB[kSumDepth], olB[kSumDepth];
#define IRDM_SUM_LOAD(NWv, OLv, D, half, k0)                                                            \
    {                                                                                                   \
        const __amdgpu_buffer_rsrc_t r_old = (k0) < kHistory ? r_hist : r_mag;                          \
        _Pragma("unroll") for (int j = 0; j < kSumDepth; j++) {                                         \
            const int o_nw = __builtin_amdgcn_readlane((int)D.x, (half) * kSumDepth + j);               \
            const int o_ol = __builtin_amdgcn_readlane((int)D.y, (half) * kSumDepth + j);               \
            NWv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_mag, boff, o_nw, 0)); \
            OLv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_old, boff, o_ol, 0)); \
        }                                                                                               \
    }
#define IRDM_SUM_STEP(NWv, OLv, D, half)                                                                \
    {                                                                                                   \
        const float dd = s - OLv[j]; /* simd_baseline_update: two separately rounded operations */      \
        s = dd + NWv[j];                                                                                \
        smin = __builtin_fminf(smin, s); /* (one v_min_f32; the sums are not NaNs) */                                                                     \
        if ((snaps >> ((half) * kSumDepth + j)) & 1)                                                    \
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, s), r_snap, boff,             \
                                                  __builtin_amdgcn_readlane((int)D.z, (half) * kSumDepth + j), 0); \
    }
    // (a whole batch inside the list runs without the per-step test of the list's end)
#define IRDM_SUM_CONSUME(NWv, OLv, D, half, k0)                                                         \
    {                                                                                                   \
        const unsigned long long snaps = __builtin_amdgcn_ballot_w64(D.z != ~0u);                       \
        if ((k0) + kSumDepth <= n) {                                                                    \
            _Pragma("unroll") for (int j = 0; j < kSumDepth; j++) IRDM_SUM_STEP(NWv, OLv, D, half)      \
        } else {                                                                                        \
            _Pragma("unroll") for (int j = 0; j < kSumDepth; j++)                                       \
                if ((k0) + j < n) IRDM_SUM_STEP(NWv, OLv, D, half)                                      \
        }                                                                                               \
    }
    uint4 d = steps4[k_begin + lane], dn = steps4[k_begin + 2 * kSumDepth + lane];
    IRDM_SUM_LOAD(nwA, olA, d, 0, k_begin)
    for (int k0 = k_begin; k0 < n; k0 += 2 * kSumDepth) {
        const bool more = k0 + 2 * kSumDepth < n;
        // (the descriptors of the 64 steps after the next 64, always: by the time they are needed they are older than
        // every row in flight, so no wait is spent on them; past the list's padding this reads allocated words nobody uses)
        const uint4 dnn = steps4[k0 + 4 * kSumDepth + lane];
        IRDM_SUM_LOAD(nwB, olB, d, 1, k0 + kSumDepth)
        IRDM_SUM_CONSUME(nwA, olA, d, 0, k0)
        if (more) { IRDM_SUM_LOAD(nwA, olA, dn, 0, k0 + 2 * kSumDepth) }
        IRDM_SUM_CONSUME(nwB, olB, d, 1, k0 + kSumDepth)
        d = dn;
        dn = dnn;
    }
#undef IRDM_SUM_LOAD
#undef IRDM_SUM_CONSUME
#undef IRDM_SUM_STEP
    W.sum_new[b] = s;
    smin_out[b] = smin;
    // the prefilter listed mag > pre = 0.5 * thr * sum_at_chunk_start; a crossing needs mag > thr * sum (within one
    // rounding), so the lists are complete while pre <= 0.9 * thr * (smallest sum the bin went through)
    if (!(pre[b] <= 0.9f * P.thr * smin)) atomicOr(W.flags, BAND_F_STALE);
}

// 64 bins per wavefront (32 and 16 measured the same, 10 MHz, 1000 and 7600 update steps: the pass is bound by what a step
// costs one wavefront, not by the cache lines a CU has in flight)
__global__ __launch_bounds__(64) void band_sum_kernel(BandParams P, BandWork W, const float *__restrict__ mag,
                                                      const float *__restrict__ hist, const float *__restrict__ sum,
                                                      const float *__restrict__ pre, float *__restrict__ smin_out,
                                                      const SumStep *__restrict__ steps, float *__restrict__ snap)
{
    IRDM_DETECTOR_PRIO();
    TlScope tl(P, W, 4 * (W.ctl->rounds - 1) + 1);
    if (band_void(P, W) || W.ctl->status != 0) return;
    band_sum_body(P, W, mag, hist, sum, pre, smin_out, steps, snap, (int)blockIdx.x * 64 + (int)threadIdx.x);
}

// ---- the walk with a wavefront per band and activity segment (band_wave.hpp) ----
// A fixed grid of wavefronts shares out the (band, 64-frame block) pairs; a pair has work if a segment starts inside the
// block (or, block 0, if bursts are carried into the band).  64 pairs are tested at once, a lane each.
constexpr int kWalkWaveGroups = 256;         // workgroups of 4 wavefronts

template <int NW>
__global__ __launch_bounds__(256) void band_walk_wave_kernel(BandParams P, BandWork W, BandIO io, const DetState *__restrict__ st)
{
    IRDM_DETECTOR_PRIO();
    TlScope tl(P, W, 4 * (W.ctl->rounds - 1) + 3);
    if (band_void(P, W) || W.ctl->status != 0) return;
    io.act_in = st->act;
    io.n_act_in = (int32_t)wv_first((uint32_t)st->n_act);
    const int lane = threadIdx.x & 63;
    const int gw = (int)wv_first((uint32_t)(blockIdx.x * 4 + (threadIdx.x >> 6))), n_waves = (int)gridDim.x * 4;
    const int n_pairs = P.n_bands * P.occ_words;
    unsigned long long *tl_stat = nullptr;
    if (P.tl_sel >= 0)
        tl_stat = W.tl + (size_t)P.tl_sel * 2 * kBandTlSlots + kBandTlSlots + (W.ctl->rounds - 1 == 0 ? 26 : 28);
    int wave_events = 0;
    const unsigned long long t_wave = tl_stat ? wall_clock64() : 0;
    for (int p0 = gw; p0 < n_pairs; p0 += 64 * n_waves) {
        const int pl = p0 + lane * n_waves;
        bool work = false;
        if (pl < n_pairs) {
            // (a wavefront's pairs: one time block, different bands -- busy bands stay busy for a whole chunk)
            const int band = pl / P.occ_words, blk = pl % P.occ_words;
            work = blk == 0 || band_segment_starts(io.occ + (size_t)band * P.occ_words, blk, P.gap, false) != 0;
        }
        uint64_t todo = __builtin_amdgcn_ballot_w64(work);
        while (todo) {
            const int j = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int p = p0 + j * n_waves;
            const int band = p / P.occ_words, blk = p % P.occ_words;
            bool carried = false;
            int events = 0;
            if (blk == 0) {
                WaveWalker<NW> w(P, io, band, lane);
                if (w.load_carried() > 0) {
                    carried = true;
                    w.run(0, true);
                    events += w.n_events;
                }
            }
            uint64_t starts = wv_first64(band_segment_starts(io.occ + (size_t)band * P.occ_words, blk, P.gap, carried));
            while (starts) {
                const int q = __builtin_ctzll(starts);
                starts &= starts - 1;
                WaveWalker<NW> w(P, io, band, lane);
                w.run(64 * blk + q, false);
                events += w.n_events;
            }
            wave_events += events;
            if (tl_stat && lane == 0) {
                atomicMax(tl_stat, (unsigned long long)events);
                atomicAdd(tl_stat + 1, (unsigned long long)events);
            }
        }
    }
    if (tl_stat && lane == 0 && W.ctl->rounds - 1 != 0) {
        // (slots 30 / 31: the busiest wavefront's events, the longest wavefront's time in 10 ns ticks)
        unsigned long long *x = W.tl + (size_t)P.tl_sel * 2 * kBandTlSlots + kBandTlSlots + 30;
        atomicMax(x, (unsigned long long)wave_events);
        atomicMax(x + 1, wall_clock64() - t_wave);
    }
}

__device__ __forceinline__ void band_cross_wave(const BandParams &P, const BandWork &W, unsigned c,
                                                const ListEntry *__restrict__ entries, int f, uint32_t *s_bits, int lane)
{
    const int N = P.n;
    for (int w = lane; w < N / 32; w += 64) s_bits[w] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float *srow = W.snap + (size_t)W.slot_pre[f] * N;
    const ListEntry *e = entries + (size_t)f * P.list_cap;
    float *rq = W.relq + (size_t)f * N;
    for (unsigned i = lane; i < c; i += 64) {
        const int bin = e[i].bin;
        const float base = srow[bin];
        const float rel = base > 0 ? e[i].mag / base : 0.0f;      // simd_relative_mag (simd_generic.c:137-145)
        if (rel > P.thr) {
            atomicOr(&s_bits[bin >> 5], 1u << (bin & 31));
            rq[bin] = rel;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t *out = reinterpret_cast<uint32_t *>(W.cross) + (size_t)f * (N / 32);
    for (int w = lane; w < N / 32; w += 64) out[w] = s_bits[w];
    if (lane < P.n_bands) {
        const int w0 = (lane * P.band_w - P.band_w / 2) / 32, nw = 2 * P.band_w / 32;
        uint32_t any = 0;
        for (int k = 0; k < nw; k++) {
            const int w = w0 + k;
            if (w >= 0 && w < N / 32) any |= s_bits[w];
        }
        if (any) atomicOr(reinterpret_cast<unsigned long long *>(&W.occ[(size_t)lane * P.occ_words + (f >> 6)]), 1ull << (f & 63));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// The crossing pass as a fixed grid of wavefronts that walk their frames (a wavefront per frame with list entries, the
// empty frames skipped 64 at a time): beside the per-burst chains' kernels the one-workgroup-per-frame form spent its
// time asking the dispatcher for 8192 workgroups (10 us alone, 50-150 us in run).
constexpr int kCrossGroups = 256;

__global__ __launch_bounds__(256) void band_cross_w_kernel(BandParams P, BandWork W, const unsigned *__restrict__ counts,
                                                           const ListEntry *__restrict__ entries)
{
    IRDM_DETECTOR_PRIO();
    TlScope tl(P, W, 4 * (W.ctl->rounds - 1) + 2);
    __shared__ uint32_t s_bits_all[4][16384 / 32];
    if (band_void(P, W) || W.ctl->status != 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = (int)blockIdx.x * 4 + wave, n_waves = (int)gridDim.x * 4;
    uint32_t *s_bits = s_bits_all[wave];
    for (int f0 = gw; f0 < P.n_frames; f0 += 64 * n_waves) {
        const int fl = f0 + lane * n_waves;
        const unsigned cl = fl < P.n_frames ? counts[fl] : 0u;
        uint64_t todo = __builtin_amdgcn_ballot_w64(cl != 0 && cl <= (unsigned)P.list_cap);
        while (todo) {
            const int j = __builtin_ctzll(todo);
            todo &= todo - 1;
            const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)cl, j);
            band_cross_wave(P, W, c, entries, f0 + j * n_waves, s_bits, lane);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// band_spec: round 0 as a SPECULATION PASS beside the previous chunk's scan.
//
// Round 0 of a scan exists to guess the update vector u (which frames end quiet, which force an update): it speculates
// "no update", lets the bands walk the chunk on the frozen sums and hands what they produce to round 1, whose verdict --
// produced u == planned u, neighbouring bands agree, no squelch -- is what makes a result exact.  Nothing of round 0 is
// kept but that guess.  And a guess needs no exact inputs: run on the sums as chunk k's round 1 leaves them (available
// after its sums pass) and on the bursts the PREVIOUS speculation pass left active at its chunk's end, the guess for chunk
// k + 1 can be made on a second workspace while chunk k's round 1 is still crossing, walking and committing.  The scan of
// chunk k + 1 then opens with round 1 (BandParams::spec_in): plan . sums . cross . walk . verdict + commit -- half the
// dependent launches per chunk on the stream whose back-to-back scans are the pipeline's period.  A wrong guess (a carried
// burst the speculation did not know) costs a further round like any other mismatch; the verdict never looks at where a
// plan came from.
// ---------------------------------------------------------------------------------------------------------------
// (256 threads and no LDS to speak of: four wavefronts find room beside the decimator's resident grid, where a
// 1024-thread workgroup waited for that launch to drain -- 210-220 us "long" in the kernel trace, ending with the decimator)
__global__ __launch_bounds__(256) void band_spec_prep_kernel(BandParams P, BandWork S, const unsigned *__restrict__ counts,
                                                              const float *__restrict__ sum_src, DetState *__restrict__ st_spec,
                                                              int have_prev)
{
    IRDM_DETECTOR_PRIO();
    const int tid = threadIdx.x, F = P.n_frames;
    __shared__ unsigned s_n;
    if (tid == 0) s_n = 0;
    __syncthreads();
    // what the previous speculation pass left active at the end of ITS chunk is what this one takes over at the start of
    // its own (records still hold that pass's walk; ids do not matter to a guess)
    if (have_prev)
        for (int band = tid >> 4; band < P.n_bands; band += 256 / 16) {
            const int cnt = min((int)S.rec_count[band], kBandRecCap);
            for (int j = tid & 15; j < cnt; j += 16) {
                const BandRec &r = S.recs[(size_t)band * kBandRecCap + j];
                if (!(r.flags & 1) || r.stop >= 0) continue;
                const unsigned at = atomicAdd(&s_n, 1u);
                if (at < (unsigned)kMaxActive) {
                    ActiveBurst a;
                    a.id = 0; a.start = (uint64_t)r.start; a.last_active = (uint64_t)r.last_active;
                    a.center_bin = r.cb; a.peak_rel = r.rel; a.base_sum = r.base; a.pad = 0;
                    st_spec->act[at] = a;
                }
            }
        }
    __syncthreads();
    if (tid == 0) st_spec->n_act = (uint32_t)min(s_n, (unsigned)kMaxActive);
    // round 0's plan on the speculation workspace: no update steps, one snapshot -- the sums handed in
    if (tid < (int)(sizeof(BandCtl) / 4)) reinterpret_cast<uint32_t *>(S.ctl)[tid] = 0;
    for (int f = tid; f < F; f += 256) {
        const int slot = counts[f] > 0 ? 0 : -1;
        S.slot_pre[f] = slot;
        S.slot_post[f] = slot;
    }
    for (int i = tid; i < P.n_bands * P.occ_words; i += 256) S.occ[i] = 0;
    for (int i = tid; i < P.occ_words; i += 256) {
        S.busy[i] = 0;
        S.forced[i] = 0;
        S.conc[i] = 0;
    }
    for (int i = tid; i < P.n_bands; i += 256) S.rec_count[i] = 0;
    for (int b = tid; b < P.n; b += 256) S.snap[b] = sum_src[b];
    __syncthreads();
    if (tid == 0) {
        *S.flags = 0;
        S.ctl->rounds = 1;
        S.snap_slot[0] = 0;
    }
}

// ---- commit ----
// ascending bitonic sort of np = 2^m (key, payload) pairs in LDS by the whole workgroup (keys are unique; padding = ~0)
__device__ void lds_bitonic_sort(uint64_t *key, uint16_t *val, int np)
{
    for (int k = 2; k <= np; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < np / 2; i += blockDim.x) {
                const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo | j;
                const uint64_t a = key[lo], b = key[hi];
                if ((a > b) == ((lo & k) == 0)) {
                    key[lo] = b;
                    key[hi] = a;
                    const uint16_t va = val[lo];
                    val[lo] = val[hi];
                    val[hi] = va;
                }
            }
            __syncthreads();
        }
    }
}

// (one workgroup of blockDim.x threads; smem_raw: kBandMaxTotal * 10 bytes of LDS)
__device__ void band_commit_body(const BandParams &P, const BandWork &W, DetState *__restrict__ st,
                                 float *__restrict__ sum, GoneBurst *__restrict__ gone, int gone_cap,
                                 unsigned char *smem_raw)
{
    const int kPlanThreads = (int)blockDim.x;      // (shadows the launch constant: every stride below is the workgroup's size)
    uint64_t *s_key = reinterpret_cast<uint64_t *>(smem_raw);                    // kBandMaxTotal
    uint16_t *s_val = reinterpret_cast<uint16_t *>(s_key + kBandMaxTotal);       // kBandMaxTotal
    uint32_t *rank_of = W.rank;                                                  // record -> place in creation order
    __shared__ unsigned s_n, s_carried, s_gone;
    BandCtl *ctl = W.ctl;
    if (band_void(P, W) || ctl->status != 1 || ctl->committed) return;
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
        s_val[t] = (uint16_t)t;
    }
    // (the two orders are sorts: the all-pairs count this replaces took 0.3-0.6 ms of one workgroup at 2600 bursts)
    int np = 2;
    while (np < n) np <<= 1;
    for (int t = n + tid; t < np; t += kPlanThreads) {
        s_key[t] = ~0ull;
        s_val[t] = 0xffff;
    }
    __syncthreads();
    lds_bitonic_sort(s_key, s_val, np);
    for (int q = tid; q < n; q += kPlanThreads) rank_of[s_val[q]] = (uint32_t)q;
    __syncthreads();
    const uint64_t id0 = st->burst_id;
    for (int t = tid; t < n; t += kPlanThreads) {
        const BandRec &r = W.recs[W.tot[t]];
        W.ids[t] = r.cf < 0 ? st->act[r.seq].id : id0 + 10ull * (uint64_t)((int)rank_of[t] - n_carried);
    }
    // emission order: by the frame that deleted the burst, within a frame in list (= creation) order (:490-514);
    // bursts still active follow in creation order
    for (int t = tid; t < n; t += kPlanThreads) {
        const BandRec &r = W.recs[W.tot[t]];
        s_key[t] = r.stop >= 0 ? (((uint64_t)((r.stop - (int64_t)P.idx0) >> P.log_n)) << 32) | rank_of[t]
                               : (1ull << 63) | rank_of[t];
        s_val[t] = (uint16_t)t;
    }
    for (int t = n + tid; t < np; t += kPlanThreads) {
        s_key[t] = ~0ull;
        s_val[t] = 0xffff;
    }
    __syncthreads();
    lds_bitonic_sort(s_key, s_val, np);
    for (int pos = tid; pos < n; pos += kPlanThreads) {
        const int t = s_val[pos];
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
        W.bar[4] = 1;                // a scan chained behind this one may run
    }
}

__global__ __launch_bounds__(kPlanThreads) void band_commit_kernel(BandParams P, BandWork W, DetState *__restrict__ st,
                                                                   float *__restrict__ sum, GoneBurst *__restrict__ gone,
                                                                   int gone_cap)
{
    IRDM_DETECTOR_PRIO();
    TlScope tl(P, W, 24);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    band_commit_body(P, W, st, sum, gone, gone_cap, smem_raw);
}

constexpr int kExportBlocks = 32;

__global__ __launch_bounds__(256) void band_history_kernel(BandParams P, BandWork W, const float *__restrict__ mag,
                                                           float *__restrict__ hist, const DetState *__restrict__ st,
                                                           const uint32_t *__restrict__ gone, int gone_cap,
                                                           uint32_t *__restrict__ hp_gone, uint32_t *__restrict__ hp_hdr,
                                                           uint32_t *__restrict__ hp_ctl)
{
    IRDM_DETECTOR_PRIO();
    TlScope tl(P, W, 25);
    const BandCtl *ctl = W.ctl;
    if (band_void(P, W)) {
        // (the host drains and ignores a void launch; its export slot says so for whoever looks)
        if (hp_ctl && blockIdx.x == 0 && threadIdx.x == 0) {
            __hip_atomic_store(hp_ctl + 2, (uint32_t)BAND_F_CHAIN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // flags
            __hip_atomic_store(hp_ctl + 0, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);                      // status
            __threadfence_system();
        }
        return;
    }
    // the scan's verdict and records go to the host whatever the verdict is (types.hpp, gone_export_body)
    if (hp_hdr && blockIdx.x < kExportBlocks)
        gone_export_body(st, gone, gone_cap, hp_gone, hp_hdr, reinterpret_cast<const uint32_t *>(ctl), hp_ctl,
                         (int)(sizeof(BandCtl) / 4), kExportBlocks);
    if (ctl->status != 1 || !ctl->committed) return;
    const int k = ctl->n_upd - 1 - (int)blockIdx.x;
    if (k < 0) return;
    const float4 *src = reinterpret_cast<const float4 *>(mag + (size_t)W.upd_frame[k] * P.n);
    float4 *dst = reinterpret_cast<float4 *>(hist + (size_t)((ctl->h0 + k) % kHistory) * P.n);
    // (eight loads in flight per lane before the first store: a row at 8192 points in one round trip instead of eight
    // dependent ones -- 48 us for 2 x 16 MB was 0.7 TB/s)
    for (int i0 = threadIdx.x; i0 < P.n / 4; i0 += 8 * 256) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (i0 + 256 * u < P.n / 4) v[u] = src[i0 + 256 * u];
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (i0 + 256 * u < P.n / 4) dst[i0 + 256 * u] = v[u];
    }
}

}  // namespace

int band_list_cap(int n) { return n < kBandListCap ? n : kBandListCap; }

int band_scan_supported(const DetParams &D, BandParams *out, int n_frames, uint64_t idx0)
{
    BandParams P;
    P.n = D.n;
    P.log_n = 31 - __builtin_clz((unsigned)(D.n > 0 ? D.n : 1));
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
    if (D.n < 2048 || D.n > 16384 || D.n != (1 << P.log_n) || P.n_bands < 1 || P.n_bands > 64) return 0;
    if (P.hw < 1 || 2 * P.hw + 8 > P.band_w / 2) return 0;
    if (P.gap < 1 || P.gap >= 64) return 0;
    if (D.max_bursts <= 0 || D.max_bursts > kMaxActive - 64) return 0;
    return 1;
}

// spec: the workspace of the speculation passes (band_spec): one snapshot row instead of F + 2
size_t band_work_bytes(int n, size_t max_chunk, bool spec)
{
    const size_t F = max_chunk / (size_t)n + 2;
    size_t b = 0;
    auto add = [&](size_t x) { b += (x + 255) & ~(size_t)255; };
    add(sizeof(BandCtl));
    add(F); add(F);                                  // uq, uf
    add(4 * (F + 2)); add(4 * (2 * F + 4));          // cnt_before, tmp
    add(4 * (2 * F + 4)); add(4 * (2 * F + 4)); add(4 * (2 * F + 4));   // upd_frame, old_row, snap_after
    add(sizeof(SumStep) * (2 * F + 4 + 3 * kSumDepth + 6 * kSumDepth));   // steps (+ the descriptor read-ahead of the sums pass)
    add(4 * (2 * F + 4)); add(4 * (2 * F + 4));      // need, snap_slot
    add(4 * F); add(4 * F);                          // slot_pre, slot_post
    add(F * (size_t)n / 8);                          // cross
    add(F * (size_t)n * 4);                          // relq
    add((spec ? 2 : F + 2 + F / 32 + 4) * (size_t)n * 4);         // snap (+ the restart points: a state in 64)
    add(64 * ((F + 63) / 64) * 8);                   // occ
    add(((F + 63) / 64) * 8); add(((F + 63) / 64) * 8); add(((F + 63) / 64) * 4);   // busy, forced, conc
    add(sizeof(BandRec) * 64 * kBandRecCap); add(4 * 64);                            // recs, rec_count
    add(4 * (size_t)n); add(4 * kBandMaxTotal); add(8 * kBandMaxTotal); add(256);    // sum_new, tot, ids, flags
    add(256); add(4 * kBandMaxTotal);                                                // bar, rank
    add(8 * 4 * kBandTlSlots);                                                       // tl
    return b;
}

int band_work_carve(BandWork *W, void *base, int n, size_t max_chunk, bool spec)
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
    W->steps = static_cast<SumStep *>(take(sizeof(SumStep) * (2 * F + 4 + 3 * kSumDepth + 6 * kSumDepth)));
    W->need = static_cast<int32_t *>(take(4 * (2 * F + 4)));
    W->snap_slot = static_cast<int32_t *>(take(4 * (2 * F + 4)));
    W->slot_pre = static_cast<int32_t *>(take(4 * F));
    W->slot_post = static_cast<int32_t *>(take(4 * F));
    W->cross = static_cast<uint64_t *>(take(F * (size_t)n / 8));
    W->relq = static_cast<float *>(take(F * (size_t)n * 4));
    W->snap = static_cast<float *>(take((spec ? 2 : F + 2 + F / 32 + 4) * (size_t)n * 4));
    W->snap_cap = (int)(spec ? 2 : F + 2 + F / 32 + 4);
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
    W->bar = static_cast<unsigned *>(take(256));
    W->rank = static_cast<uint32_t *>(take(4 * kBandMaxTotal));
    W->tl = static_cast<unsigned long long *>(take(8 * 4 * kBandTlSlots));
    return 0;
}

// (one counter for all contexts of the process, fed from any thread: two launches never share a serial number)
static unsigned next_launch_serial()
{
    static std::atomic<unsigned> launch_serial{ 0 };
    unsigned serial = launch_serial.fetch_add(1) + 1;
    if (serial == 0) serial = launch_serial.fetch_add(1) + 1;       // (never 0: the idle value of bar[5])
    return serial;
}

// The speculation pass of a chunk (band_spec, above) on `stream`: prep . cross . walk on the workspace S.  sum_src: the sums
// the crossings are tested against (the scan workspace's sum_new: what the previous chunk's round 1 computed); have_prev: S
// still holds the previous speculation pass's records (its bursts active at the end become this pass's carried bursts).
int launch_band_spec(const DetParams &D, BandWork S, DetState *st_spec, const float *sum_src, int n_frames, uint64_t idx0,
                     const unsigned *counts, const ListEntry *entries, int have_prev, hipStream_t stream, const BandTune &tune)
{
    BandParams P;
    if (!band_scan_supported(D, &P, n_frames, idx0) || n_frames < 1) return -1;
    P.serial = (int32_t)next_launch_serial();
    P.selfcheck = tune.selfcheck & 8;
    BandIO io;
    io.cross = S.cross;
    io.occ = S.occ;
    io.relq = S.relq;
    io.snap = S.snap;
    io.slot_post = S.slot_post;
    io.act_in = nullptr;
    io.n_act_in = 0;
    io.recs = S.recs;
    io.rec_count = S.rec_count;
    io.busy = S.busy;
    io.forced = S.forced;
    io.conc = S.conc;
    io.flags = S.flags;
    hipLaunchKernelGGL(band_spec_prep_kernel, dim3(1), dim3(256), 0, stream, P, S, counts, sum_src, st_spec, have_prev);
    hipLaunchKernelGGL(band_cross_w_kernel, dim3(kCrossGroups), dim3(256), 0, stream, P, S, counts, entries);
    if (P.band_w == 128)
        hipLaunchKernelGGL((band_walk_wave_kernel<4>), dim3(kWalkWaveGroups), dim3(256), 0, stream, P, S, io, st_spec);
    else
        hipLaunchKernelGGL((band_walk_wave_kernel<8>), dim3(kWalkWaveGroups), dim3(256), 0, stream, P, S, io, st_spec);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Enqueue the band scan of one chunk on `stream`: rounds [round_begin, round_end) and the verdict on the last of them.  Nothing
// of the carried state (st, sum, hist) is written unless BandCtl::status ends as 1 (accepted); the caller reads the control
// block (exported to pinned host memory by the history pass's first workgroups) afterwards.  The host enqueues rounds
// 0 .. kBandFirst - 1 (two or three suffice on every scene measured; a round that is not needed is four empty launches) and,
// if the verdict is still open (status 0, no flags), the rest up to kBandRounds with round_begin = kBandFirst: everything a
// round needs from the one before lives in the workspace.
//   spec       a speculation pass made this chunk's round 0 on that workspace (launch_band_spec): the scan opens with round 1,
//              whose plan takes the update vector from spec's bitmaps (round_begin must be 1)
//   sums_done  recorded behind the first sums pass of this launch (its sum_new is what the NEXT chunk's speculation pass
//              tests against)
//   gate_*     time-chunk sharding, irdm_expect_history: the 512-frame history of the previous chunk is still on its way from
//              the previous rank.  Round 0 speculates "no update" and reads none of it; the first sums pass that does (round 1)
//              starts behind a one-lane kernel that waits until the HOST has published `gate_seq` (the history has arrived
//              in the caller's receive buffer) and the copy of it into the ring, both on this stream
// A launch per pass: plan . [sums . cross . walk . plan]* . history -- the forms with fewer launches (the next plan pass run by
// the walk pass's last workgroup; every round inside one cooperative launch), plan passes launched ahead on a second stream and
// the history copy on a side stream were built, exact, and measured slower (docs/rounds/, profiles/r5_tail_ab.json,
// profiles/r6_option_ab.json); they are not part of the product.
int launch_band_scan(const DetParams &D, BandWork W, DetState *st, float *sum, float *hist, const float *mag,
                     int n_frames, uint64_t idx0, const unsigned *counts, const ListEntry *entries, const float *pre,
                     float *smin, GoneBurst *gone, int gone_cap, int round_begin, int round_end, GoneBurst *hp_gone,
                     uint32_t *hp_hdr, void *hp_ctl, int hp_cap, int chained, int tl_sel, hipStream_t stream,
                     const BandTune &tune, const uint32_t *gate_flag, uint32_t gate_seq, uint32_t *gate_err, const void *gate_src,
                     size_t gate_bytes, const BandWork *spec, hipEvent_t sums_done)
{
    if (spec && round_begin != 1) return -1;
    const uint64_t *ub = spec ? spec->busy : nullptr, *uf = spec ? spec->forced : nullptr;
    bool sums_noted = sums_done == nullptr;
    BandParams P;
    if (!band_scan_supported(D, &P, n_frames, idx0) || n_frames < 1) return -1;
    P.serial = (int32_t)next_launch_serial();
    P.chained = chained;
    P.spec_in = spec ? 1 : 0;
    P.sum_restart = tune.sum_restart;
    P.selfcheck = tune.selfcheck;
    P.tl_sel = tune.timeline && tl_sel >= 0 && tl_sel < 2 ? tl_sel : -1;
    if (P.tl_sel >= 0 && (round_begin == 0 || spec)) {
        unsigned long long *half = W.tl + (size_t)P.tl_sel * 2 * kBandTlSlots;
        (void)hipMemsetAsync(half, 0xff, 8 * kBandTlSlots, stream);
        (void)hipMemsetAsync(half + kBandTlSlots, 0, 8 * kBandTlSlots, stream);
    }
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
    const size_t commit_lds = (size_t)kBandMaxTotal * (8 + 2);
    // (per launch: the attribute belongs to the device the calling thread is on)
    (void)hipFuncSetAttribute((const void *)band_commit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)commit_lds);
    (void)hipFuncSetAttribute((const void *)band_plan_kernel<kPlanThreads>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPlanLdsBytes);
    for (int round = round_begin; round <= round_end; round++) {
        // (a continuation starts behind the plan its predecessor's verdict pass already made)
        if (round > round_begin || round_begin == 0 || spec) {
            const size_t plan_lds = (P.selfcheck & 16) ? 0 : kPlanLdsBytes;
            const float *pre0 = round == 0 ? pre : nullptr;      // round 0's sums pass (no update steps) inside its plan pass
            hipLaunchKernelGGL(band_plan_kernel<kPlanThreads>, dim3(1), dim3(kPlanThreads), plan_lds, stream, P, W, counts, st, round, sum,
                               gone, gone_cap, pre0, smin, ub, uf);
        }
        if (round == round_end) break;
        if (gate_flag && round == (round_begin > 1 ? round_begin : 1)) {
            if (launch_wait_host_flag(gate_flag, gate_seq, gate_err, stream) != 0) return -1;
            if (hipMemcpyAsync(hist, gate_src, gate_bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return -1;
        }
        if (round >= 1)
            hipLaunchKernelGGL(band_sum_kernel, dim3(P.n / 64), dim3(64), 0, stream, P, W, mag, hist, sum, pre, smin, W.steps, W.snap);
        if (!sums_noted && round >= 1) {
            // (the first real sums pass of the scan: its sum_new is what the next chunk's speculation pass tests against)
            if (hipEventRecord(sums_done, stream) != hipSuccess) return -1;
            sums_noted = true;
        }
        hipLaunchKernelGGL(band_cross_w_kernel, dim3(kCrossGroups), dim3(256), 0, stream, P, W, counts, entries);
        if (P.band_w == 128)
            hipLaunchKernelGGL((band_walk_wave_kernel<4>), dim3(kWalkWaveGroups), dim3(256), 0, stream, P, W, io, st);
        else
            hipLaunchKernelGGL((band_walk_wave_kernel<8>), dim3(kWalkWaveGroups), dim3(256), 0, stream, P, W, io, st);
    }
    // (the commit is part of the accepting plan pass, unless that pass ran without its LDS)
    if (P.selfcheck & 16)
        hipLaunchKernelGGL(band_commit_kernel, dim3(1), dim3(kPlanThreads), commit_lds, stream, P, W, st, sum, gone, gone_cap);
    static_assert(kHistory >= kExportBlocks, "the export rides on the history pass's first workgroups");
    hipLaunchKernelGGL(band_history_kernel, dim3(kHistory), dim3(256), 0, stream, P, W, mag, hist, st,
                       reinterpret_cast<const uint32_t *>(gone), hp_cap < gone_cap ? hp_cap : gone_cap,
                       reinterpret_cast<uint32_t *>(hp_gone), hp_hdr, static_cast<uint32_t *>(hp_ctl));
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace irdm
