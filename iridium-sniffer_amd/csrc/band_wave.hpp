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
    int n_events;
    int occ_blk;                             // 64-frame block whose occupancy word is cached
    uint64_t occ_w;
    int pre_f;                               // frame whose crossing words were requested ahead (-1: none)
    // ---- per lane: burst slot `lane` ----
    int64_t s_start, s_la;
    int32_t s_cb, s_cf, s_seq;
    float s_rel, s_base;
    uint64_t pre_cwv;                        // lane k < NW: crossing word k of frame pre_f

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
        n_events = 0;
        occ_blk = -1;
        occ_w = 0;
        pre_f = -1;
        pre_cwv = 0;
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

    // frames a..b end with c of this band's own bursts active
    __device__ __forceinline__ void account(int a, int b, uint32_t c)
    {
        if (c == 0 || a > b) return;
        for (int blk = a >> 6; blk <= (b >> 6); blk++) {
            const int lo = blk == (a >> 6) ? (a & 63) : 0, hi = blk == (b >> 6) ? (b & 63) : 63;
            if (lane == 0) band_or64(&io.busy[blk], (~0ull >> (63 - hi)) & (~0ull << lo));
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
        const uint64_t cwv = pre_f == f ? pre_cwv : fetch_cw(f);
        // (most events are consecutive frames of a burst: the next frame's words are on their way while this one is
        // worked on)
        pre_f = f + 1;
        pre_cwv = fetch_cw(f + 1);
        uint64_t cw[NW];
#pragma unroll
        for (int k = 0; k < NW; k++) cw[k] = occupied ? wv_lane64(cwv, k) : 0;
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

    // walk one segment: from frame f_start, or (carried; load_carried() > 0 was called) from the bursts handed over
    // at the chunk boundary
    __device__ __forceinline__ void run(int f_start, bool carried)
    {
        int f;
        if (carried) {
            last_occ = -1;
            const uint32_t c0 = owned_count();
            f = next_event(-1);
            account(0, (f < P.n_frames ? f : P.n_frames) - 1, c0);
        } else {
            f = f_start;
        }
        while (f < P.n_frames) {
            process(f);
            n_events++;
            const uint32_t c = owned_count();
            const int nf = next_event(f);
            account(f, (nf < P.n_frames ? nf : P.n_frames) - 1, c);
            if (valid == 0 && nf == 0x7fffffff) break;
            f = nf;
        }
        flush_conc();
        // still active at the end of the chunk: handed to the next one
        emit(__builtin_amdgcn_ballot_w64(mine()), -1, false);
    }
};

}  // namespace irdm
