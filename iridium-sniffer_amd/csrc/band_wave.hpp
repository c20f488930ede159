// band_wave.hpp -- the band state machine of band_core.hpp with one WAVEFRONT per band and activity segment
// (burst_detect.c:458-632: update_bursts, masks, extract_peaks, delete_gone_bursts, create_new_bursts).
//
// band_core.hpp's BandWalker is one LANE per band: the 64 lanes of a wavefront walk 64 different bands' segments in
// lock step, so every frame of the longest lane costs the union of what any lane does in it -- the peak loop, the
// deletions, every loop over the active bursts at its longest trip count -- and the slots of 64 lanes need 55 KB of
// LDS.  Measured (device timeline, 10 MHz, 667 bursts per chunk): 14 500 events per pass spread over 8192 lanes, the
// longest lane 31 events, the pass 90 us: 2.9 us per event of the longest lane; at 12 MHz / 2600 bursts 48 events in
// 345 us.  Here the wavefront IS the band: lane i holds burst slot i (kBandSlots <= 32) in registers, everything that
// is one value per band (masks, crossing words, the valid set, the frame counters) is wave-uniform and lives in scalar
// registers, and the loops over the active bursts are single vector instructions plus a ballot or a DPP reduction.  No
// LDS.  The decisions, their order and every emitted record are those of BandWalker (which stays: the CPU test drives
// it against the oracle, the cooperative kernel uses it, and option band_walk_wave 0 selects it): same segments, same
// events, same arithmetic.
#pragma once
#include "band_core.hpp"

namespace irdm {

static_assert(kBandSlots <= 32, "a burst slot per lane, reductions over the first two rows of 16 lanes");

__device__ __forceinline__ uint32_t wv_first(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t wv_first64(uint64_t v)
{
    return ((uint64_t)wv_first((uint32_t)(v >> 32)) << 32) | wv_first((uint32_t)v);
}
// (lane: wave-uniform)
__device__ __forceinline__ uint64_t wv_lane64(uint64_t v, int lane)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}
// uniform loads: every lane reads the same address, the value is moved to scalar registers
__device__ __forceinline__ uint64_t wv_load64(const uint64_t *p) { return wv_first64(*p); }
__device__ __forceinline__ int32_t wv_load32(const int32_t *p) { return (int32_t)wv_first((uint32_t)*p); }

// minimum of v over lanes 0..31 (rows of 16 lanes: shifts by 1, 2, 4, 8 leave a row's minimum in its last lane)
__device__ __forceinline__ int32_t wv_min32(int32_t v)
{
    constexpr int kMax = 0x7fffffff;
    v = min(v, __builtin_amdgcn_update_dpp(kMax, v, 0x111, 0xf, 0xf, false));      // row_shr:1
    v = min(v, __builtin_amdgcn_update_dpp(kMax, v, 0x112, 0xf, 0xf, false));      // row_shr:2
    v = min(v, __builtin_amdgcn_update_dpp(kMax, v, 0x114, 0xf, 0xf, false));      // row_shr:4
    v = min(v, __builtin_amdgcn_update_dpp(kMax, v, 0x118, 0xf, 0xf, false));      // row_shr:8
    return min(__builtin_amdgcn_readlane(v, 15), __builtin_amdgcn_readlane(v, 31));
}

template <int NW>
struct WaveWalker {
    const BandParams &P;
    const BandIO &io;
    const uint64_t *occ;                     // this band's occupancy words
    int lane;
    // ---- wave-uniform ----
    int band, e0, word0;
    int own_lo, own_hi, rep_lo, rep_hi;      // relative to e0
    uint64_t M[NW], elig[NW];
    uint32_t valid;
    int last_occ;
    int acc_blk;
    uint32_t acc_max;
    int busy_blk;                            // 64-frame block whose busy bits are collected in busy_acc (-1: none)
    uint64_t busy_acc;
    int n_events;
    int occ_blk;                             // 64-frame block whose occupancy word is cached
    uint64_t occ_w;
    int cwu_f;                               // frame whose crossing words skim() left in cwu (-1: none)
    uint64_t cwu[NW];
    // ---- per lane: burst slot `lane` ----
    int64_t s_start, s_la;
    int32_t s_cb, s_cf, s_seq;
    float s_rel, s_base;

    __device__ __forceinline__ WaveWalker(const BandParams &p, const BandIO &i, int b, int ln) : P(p), io(i), lane(ln), band(b)
    {
        const int H = P.band_w / 2;
        occ = io.occ + (size_t)band * P.occ_words;
        e0 = band * P.band_w - H;
        word0 = e0 >> 6;
        own_lo = H;
        own_hi = H + P.band_w;
        rep_lo = own_lo - P.hw;
        rep_hi = own_hi + P.hw;
        uint64_t dc[NW];
#pragma unroll
        for (int k = 0; k < NW; k++) M[k] = elig[k] = dc[k] = 0;
        bm_set_range<NW>(elig, P.hw - e0, P.n - P.hw - 1 - e0);
        bm_set_range<NW>(dc, P.n / 2 - 3 - e0, P.n / 2 + 3 - e0);
#pragma unroll
        for (int k = 0; k < NW; k++) elig[k] &= ~dc[k];
        valid = 0;
        last_occ = -1;
        acc_blk = -1;
        acc_max = 0;
        busy_blk = -1;
        busy_acc = 0;
        n_events = 0;
        occ_blk = -1;
        occ_w = 0;
        cwu_f = -1;
#pragma unroll
        for (int k = 0; k < NW; k++) cwu[k] = 0;
        s_start = s_la = 0;
        s_cb = s_cf = s_seq = 0;
        s_rel = s_base = 0.0f;
    }

    __device__ __forceinline__ bool mine() const { return lane < 32 && ((valid >> (lane & 31)) & 1u); }

    __device__ __forceinline__ uint64_t occ_word(int blk)
    {
        if (blk != occ_blk) {
            occ_w = wv_load64(occ + blk);
            occ_blk = blk;
        }
        return occ_w;
    }

    // first occupied frame in [a, b] (clipped to the scan) or INT32_MAX (band_core.hpp: band_next_occ)
    __device__ __forceinline__ int next_occ(int a, int b)
    {
        if (a < 0) a = 0;
        if (b >= P.n_frames) b = P.n_frames - 1;
        while (a <= b) {
            const int w = a >> 6;
            const uint64_t v = occ_word(w) & (~0ull << (a & 63));
            if (v) {
                const int f = 64 * w + __builtin_ctzll(v);
                return f <= b ? f : 0x7fffffff;
            }
            a = 64 * (w + 1);
        }
        return 0x7fffffff;
    }

    __device__ __forceinline__ void rebuild_mask()
    {
#pragma unroll
        for (int k = 0; k < NW; k++) M[k] = 0;
        for (uint32_t v = valid; v; v &= v - 1) {
            const int r = __builtin_amdgcn_readlane(s_cb, __builtin_ctz(v)) - e0;
            bm_set_range<NW>(M, r - P.hw, r + P.hw);
        }
    }

    // records of the slots in `who` (a ballot) that lie in the band's reporting range; stop < 0: still active
    __device__ __forceinline__ void emit(uint64_t who, int64_t stop, bool lng_lane)
    {
        const int r = s_cb - e0;
        const bool me = ((who >> lane) & 1) && r >= rep_lo && r < rep_hi;
        const uint64_t em = __builtin_amdgcn_ballot_w64(me);
        if (em == 0) return;
        uint32_t at0 = 0;
        if (lane == 0) at0 = band_add32(&io.rec_count[band], (uint32_t)__builtin_popcountll(em));
        at0 = wv_first(at0);
        if (me) {
            const uint32_t at = at0 + (uint32_t)__builtin_popcountll(em & ((1ull << lane) - 1));
            if (at >= (uint32_t)kBandRecCap) {
                band_or32(io.flags, BAND_F_RECS);
            } else {
                BandRec g;
                g.start = s_start;
                g.last_active = s_la;
                g.stop = stop;
                g.cf = s_cf;
                g.cb = s_cb;
                g.rel = s_rel;
                g.base = s_base;
                g.flags = ((r >= own_lo && r < own_hi) ? 1 : 0) | (lng_lane ? 2 : 0);
                g.seq = s_seq;
                io.recs[(size_t)band * kBandRecCap + at] = g;
            }
        }
    }

    __device__ __forceinline__ uint32_t owned_count() const
    {
        const int r = s_cb - e0;
        return (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(mine() && r >= own_lo && r < own_hi));
    }

    __device__ __forceinline__ void flush_conc()
    {
        if (acc_blk >= 0 && acc_max > 0 && lane == 0) band_add32(&io.conc[acc_blk], acc_max);
        acc_blk = -1;
        acc_max = 0;
    }

    // The busy bits of a 64-frame block go out once, when the walk leaves the block: 64 bands OR into the same few words,
    // and device-scope atomics on one address are served one after the other at the memory side -- with an atomic per
    // event (36 000 per pass at 12 MHz / 2600 bursts on 64 addresses) the kernel's last atomics landed 200-300 us after
    // its last wavefront had finished, and the next pass waited for them.
    __device__ __forceinline__ void flush_busy()
    {
        if (busy_blk >= 0 && busy_acc && lane == 0) band_or64(&io.busy[busy_blk], busy_acc);
        busy_blk = -1;
        busy_acc = 0;
    }

    // frames a..b end with c of this band's own bursts active
    __device__ __forceinline__ void account(int a, int b, uint32_t c)
    {
        if (c == 0 || a > b) return;
        for (int blk = a >> 6; blk <= (b >> 6); blk++) {
            const int lo = blk == (a >> 6) ? (a & 63) : 0, hi = blk == (b >> 6) ? (b & 63) : 63;
            const uint64_t bits = (~0ull >> (63 - hi)) & (~0ull << lo);
            if (blk != busy_blk) {
                flush_busy();
                busy_blk = blk;
                busy_acc = bits;
            } else {
                busy_acc |= bits;
            }
            if (blk != acc_blk) {
                flush_conc();
                acc_blk = blk;
                acc_max = c;
            } else if (c > acc_max) {
                acc_max = c;
            }
        }
    }

    // bursts carried into the chunk whose centre lies in the extended range, in the order of DetState::act; returns
    // their number
    __device__ __forceinline__ int load_carried()
    {
        int n = 0;
        bool full = false;
        for (int i0 = 0; i0 < io.n_act_in && !full; i0 += 64) {
            const int i = i0 + lane;
            int r = -1;
            if (i < io.n_act_in) r = io.act_in[i].center_bin - e0;
            uint64_t in = __builtin_amdgcn_ballot_w64(r >= 0 && r < 64 * NW);
            while (in) {
                const int j = i0 + __builtin_ctzll(in);
                in &= in - 1;
                if (n >= kBandSlots) {
                    if (lane == 0) band_or32(io.flags, BAND_F_SLOTS);
                    full = true;
                    break;
                }
                if (lane == n) {
                    s_start = (int64_t)io.act_in[j].start;
                    s_la = (int64_t)io.act_in[j].last_active;
                    s_cb = io.act_in[j].center_bin;
                    s_cf = -1;
                    s_seq = j;
                    s_rel = io.act_in[j].peak_rel;
                    s_base = io.act_in[j].base_sum;
                }
                valid |= 1u << n;
                n++;
            }
        }
        rebuild_mask();
        return n;
    }

    // frame after `f` at which something can happen, INT32_MAX at the end of the segment
    __device__ __forceinline__ int next_event(int f)
    {
        if (valid == 0) return next_occ(f + 1, last_occ + P.gap);
        // earliest frame at which an active burst satisfies last_active + post_len <= index (:505)
        const int64_t num = s_la + (int64_t)P.post_len - (int64_t)P.idx0;
        int64_t e64 = num <= 0 ? 0 : (num + P.n - 1) >> P.log_n;
        if (e64 > 0x7fffffff) e64 = 0x7fffffff;
        const int ef = wv_min32(mine() ? (int32_t)e64 : 0x7fffffff);
        const int nf = next_occ(f + 1, ef);
        return nf < ef ? nf : ef;
    }

    // the crossing words of frame f, one per lane k < NW (0 outside the spectrum); rows of unoccupied frames hold
    // nothing defined and are only looked at when the frame is occupied
    __device__ __forceinline__ uint64_t fetch_cw(int f) const
    {
        const int w = word0 + lane;
        uint64_t v = 0;
        if (lane < NW && w >= 0 && w < P.nw64 && f < P.n_frames) v = io.cross[(size_t)f * P.nw64 + w];
        return v;
    }

    __device__ __forceinline__ void process(int f)
    {
        const int64_t index = (int64_t)P.idx0 + (int64_t)f * P.n;
        const bool occupied = (occ_word(f >> 6) >> (f & 63)) & 1;
        uint64_t cw[NW];
        if (cwu_f == f) {
            // (skim() stopped at this frame: its words are at hand, zero if the frame is not occupied)
#pragma unroll
            for (int k = 0; k < NW; k++) cw[k] = cwu[k];
        } else {
            const uint64_t cwv = fetch_cw(f);
#pragma unroll
            for (int k = 0; k < NW; k++) cw[k] = occupied ? wv_lane64(cwv, k) : 0;
        }
        if (occupied) last_occ = f;
        const int r = s_cb - e0;                      // (this lane's slot; meaningless unless mine())

        // update_bursts (:458-469): centre bin or a neighbour above the threshold
        if (valid && bm_any<NW>(cw)) {
            uint64_t hit[NW];
#pragma unroll
            for (int k = 0; k < NW; k++) {
                hit[k] = cw[k] | (cw[k] << 1) | (cw[k] >> 1);
                if (k > 0) hit[k] |= cw[k - 1] >> 63;
                if (k < NW - 1) hit[k] |= cw[k + 1] << 63;
            }
            if (mine() && bm_test<NW>(hit, r)) s_la = index;
        }
        // remove_peaks_around_bursts + extract_peaks (:522-552): the mask still holds the bursts this frame deletes
        uint64_t pk[NW];
#pragma unroll
        for (int k = 0; k < NW; k++) pk[k] = cw[k] & ~M[k] & elig[k];

        // delete_gone_bursts (:490-518)
        if (valid) {
            const bool lng = P.max_len > 0 && (s_la - s_start > (int64_t)P.max_len);
            const bool gone = mine() && (s_la + (int64_t)P.post_len <= index || lng);
            const uint64_t del = __builtin_amdgcn_ballot_w64(gone);
            if (del) {
                if (__builtin_amdgcn_ballot_w64(gone && lng && r >= own_lo && r < own_hi) && lane == 0)
                    band_or64(&io.forced[f >> 6], 1ull << (f & 63));
                emit(del, index, lng);
                valid &= ~(uint32_t)del;
                rebuild_mask();                           // update_burst_mask (:482-486)
            }
        }

        // create_new_bursts (:556-591): peaks in descending relative magnitude (ties: ascending bin, the order a
        // stable sort leaves them in), each masking +-burst_width/2 around itself
        while (bm_any<NW>(pk)) {
            float best = -1.0f;
            int best_r = 0x7fffffff;
            const uint64_t bit = 1ull << lane;
#pragma unroll
            for (int k = 0; k < NW; k++) {
                if (pk[k] & bit) {
                    const float v = io.relq[(size_t)f * P.n + (e0 + 64 * k + lane)];
                    if (v > best) {
                        best = v;
                        best_r = 64 * k + lane;
                    }
                }
            }
            // the largest value, among equals the lowest bin
            for (int d = 32; d; d >>= 1) {
                const float ov = __shfl_xor(best, d);
                const int orr = __shfl_xor(best_r, d);
                if (ov > best || (ov == best && orr < best_r)) {
                    best = ov;
                    best_r = orr;
                }
            }
            best_r = (int)wv_first((uint32_t)best_r);
            best = __uint_as_float(wv_first(__float_as_uint(best)));
            if (best_r == 0x7fffffff) break;          // cannot happen: crossing bits carry rel > threshold > 0
            const int slot = __builtin_ctz(~valid);
            if (slot >= kBandSlots) {
                if (lane == 0) band_or32(io.flags, BAND_F_SLOTS);
                break;
            }
            const int cbin = e0 + best_r;
            if (lane == slot) {
                s_start = index - (int64_t)P.pre_len;
                s_la = index - (int64_t)P.pre_len;
                s_cb = cbin;
                s_cf = f;
                s_seq = 0;
                s_rel = best;
                s_base = io.snap[(size_t)io.slot_post[f] * P.n + cbin];
            }
            valid |= 1u << slot;
            bm_set_range<NW>(M, best_r - P.hw, best_r + P.hw);
#pragma unroll
            for (int k = 0; k < NW; k++) pk[k] &= ~M[k];
        }
    }

    // OR of H << d for d in [0, g)
    static __device__ __forceinline__ uint64_t smear(uint64_t h, int g)
    {
        int have = 1;
        while (2 * have <= g) {
            h |= h << have;
            have *= 2;
        }
        if (have < g) h |= h << (g - have);
        return h;
    }

    // Most events change nothing but a burst's last_active: the frames of a burst between its creation and its end.
    // skim(f) looks at the 64 frames from f on at once, a lane per frame -- which of them carry a peak outside the
    // masks (a new burst), and per active burst in which of them its centre is hit, when it would run out
    // (:505, gap frames after its last hit) and whether a hit comes after max_burst_len (:499) -- and returns the first
    // frame that needs process(); the frames in front of it only move last_active (and last_occ), which is applied
    // here.  f must be the next event frame, valid != 0.  Exact: between two frames that create or delete a burst the
    // masks and the set of bursts are constant, an event frame in between does nothing but (:458-469), and every frame
    // at which the walk would delete a burst is an event frame (the minimum of the expiries is).
    __device__ __forceinline__ int skim(int f)
    {
        while (f < P.n_frames) {
            const int wl = P.n_frames - f < 64 ? P.n_frames - f : 64;
            const int blk = f >> 6, sh = f & 63;
            uint64_t om = occ_word(blk) >> sh;
            if (sh && blk + 1 < P.occ_words) om |= wv_load64(occ + blk + 1) << (64 - sh);
            if (wl < 64) om &= (1ull << wl) - 1;
            const bool occ_j = (om >> lane) & 1;
            uint64_t cw[NW];
#pragma unroll
            for (int k = 0; k < NW; k++) {
                const int w = word0 + k;
                cw[k] = (occ_j && w >= 0 && w < P.nw64) ? io.cross[(size_t)(f + lane) * P.nw64 + w] : 0;
            }
            bool pkj = false;
#pragma unroll
            for (int k = 0; k < NW; k++) pkj |= (cw[k] & ~M[k] & elig[k]) != 0;
            uint64_t stop = __builtin_amdgcn_ballot_w64(pkj);
            uint64_t hit[NW];
#pragma unroll
            for (int k = 0; k < NW; k++) {
                hit[k] = cw[k] | (cw[k] << 1) | (cw[k] >> 1);
                if (k > 0) hit[k] |= cw[k - 1] >> 63;
                if (k < NW - 1) hit[k] |= cw[k + 1] << 63;
            }
            // this lane's slot: the frame at which it runs out as last_active stands, the first frame at which a hit
            // makes it longer than max_burst_len
            const int64_t num = s_la + (int64_t)P.post_len - (int64_t)P.idx0;
            int64_t e64 = num <= 0 ? 0 : (num + P.n - 1) >> P.log_n;
            if (e64 > 0x7fffffff) e64 = 0x7fffffff;
            int64_t l64 = ((s_start + (int64_t)P.max_len - (int64_t)P.idx0) >> P.log_n) + 1;
            if (l64 > 0x7fffffff || P.max_len <= 0) l64 = 0x7fffffff;
            if (l64 < -0x40000000) l64 = -0x40000000;
            const bool over = P.max_len > 0 && (s_la - s_start > (int64_t)P.max_len);
            const int my_exp = (int)e64, my_lng = over ? -0x40000000 : (int)l64;
            uint64_t my_h = 0;
            for (uint32_t v = valid; v; v &= v - 1) {
                const int sl = __builtin_ctz(v);
                const int r = __builtin_amdgcn_readlane(s_cb, sl) - e0;
                const int ex = __builtin_amdgcn_readlane(my_exp, sl), lg = __builtin_amdgcn_readlane(my_lng, sl);
                const uint64_t H = __builtin_amdgcn_ballot_w64(bm_test<NW>(hit, r));
                uint64_t alive = smear(H, P.gap);
                const int d0 = ex - f;
                if (d0 >= 64) alive = ~0ull;
                else if (d0 > 0) alive |= (1ull << d0) - 1;
                stop |= ~alive;
                const int l0 = lg - f;
                if (l0 <= -0x20000000) stop |= 1;                  // (already longer: the frame we were called with)
                else if (l0 < 64) stop |= l0 <= 0 ? H : (H & (~0ull << l0));
                if (lane == sl) my_h = H;
            }
            if (wl < 64) stop &= (1ull << wl) - 1;
            const int e = stop ? __builtin_ctzll(stop) : 64;
            const uint64_t below = e >= 64 ? ~0ull : (1ull << e) - 1;
            const uint64_t hb = my_h & below;
            if (mine() && hb) s_la = (int64_t)P.idx0 + (int64_t)(f + 63 - __builtin_clzll(hb)) * P.n;
            const uint64_t ob = om & below;
            if (ob) {
                last_occ = f + 63 - __builtin_clzll(ob);
                n_events += __builtin_popcountll(ob);
            }
            if (e < wl) {
                cwu_f = f + e;
#pragma unroll
                for (int k = 0; k < NW; k++) cwu[k] = wv_lane64(cw[k], e);
                return f + e;
            }
            f += wl;
        }
        return f;
    }

    // walk one segment: from frame f_start, or (carried; load_carried() > 0 was called) from the bursts handed over
    // at the chunk boundary
    __device__ __forceinline__ void run(int f_start, bool carried)
    {
        int f;
        if (carried) {
            last_occ = -1;
            const uint32_t c0 = owned_count();
            f = next_event(-1);
            if (P.selfcheck & 8 ? false : (valid != 0 && f < P.n_frames)) f = skim(f);
            account(0, (f < P.n_frames ? f : P.n_frames) - 1, c0);
        } else {
            f = f_start;
        }
        while (f < P.n_frames) {
            process(f);
            n_events++;
            const uint32_t c = owned_count();
            int nf = next_event(f);
            // (between this frame and the next one that creates or deletes a burst, valid and so c stay as they are)
            if (P.selfcheck & 8 ? false : (valid != 0 && nf < P.n_frames)) nf = skim(nf);
            account(f, (nf < P.n_frames ? nf : P.n_frames) - 1, c);
            if (valid == 0 && nf == 0x7fffffff) break;
            f = nf;
        }
        flush_conc();
        flush_busy();
        // still active at the end of the chunk: handed to the next one
        emit(__builtin_amdgcn_ballot_w64(mine()), -1, false);
    }
};

}  // namespace irdm
