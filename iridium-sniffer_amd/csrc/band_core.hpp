// band_core.hpp -- the detector state machine of ONE BIN BAND over ONE ACTIVITY SEGMENT
// (burst_detect.c:458-632: update_bursts, masks, extract_peaks, delete_gone_bursts, create_new_bursts).
//
// Why a band can be walked on its own (DESIGN.md "The detector scan"): given the per-frame crossing bits
// (relative magnitude > threshold, which needs the running baseline sums and nothing else), every decision of the
// reference's state machine is local in frequency -- a burst looks at center_bin +-1 (:458-469) and masks
// center_bin +- burst_width/2 (:473-480), the greedy peak loop (:556-591) only interacts through those masks.
// The only global couplings are (i) "no burst active anywhere => the baseline is updated" (:440), (ii) the forced
// update after an over-long burst (:516-517), (iii) the squelch count (:594) and (iv) burst ids (:569).  The band
// scan (scan_band.hip) speculates the per-frame update vector of (i)+(ii), derives exact sums and crossing bits from
// it, lets every band walk its own bins plus a halo, and accepts the result only if the update vector the bands
// produce equals the speculated one, neighbouring bands agree on every burst within burst_width/2 of their common
// boundary, and the squelch bound cannot have been reached; ids come from a global sort afterwards.
//
// A band's state is its set of active bursts, so after ceil(post_len / N) frames without a crossing in the band's
// range (:505: last_active + post_len <= index) the state is empty whatever happened before: such frames cut the
// time axis of a band into independent SEGMENTS.  One lane walks one segment, event driven (occupied frames and
// expiry frames only).
//
// This header is host- and device-compilable: tests/band_host.cpp drives the same code on the CPU against the oracle.
#pragma once
#include <stdint.h>
#include "types.hpp"

#if defined(__HIPCC__)
#define IRDM_HD __host__ __device__ __forceinline__
#else
#define IRDM_HD inline
#endif

namespace irdm {

constexpr int kBandSlots = 24;        // simultaneously active bursts in one band's extended range (512 bins / 28 = 18)
constexpr int kBandRecCap = 1024;     // burst records per band and chunk
constexpr int kBandMaxTotal = 8192;   // owned burst records per chunk that the commit step can order

// abort reasons (BandCtl::flags)
enum : uint32_t {
    BAND_F_SLOTS = 1,        // more simultaneous bursts in a band than kBandSlots
    BAND_F_RECS = 2,         // more records in a band than kBandRecCap
    BAND_F_LIST = 4,         // a prefilter list overflowed (kBandListCap)
    BAND_F_STALE = 8,        // the running sum fell below what the prefilter threshold assumed
    BAND_F_SQUELCH = 16,     // the bound on simultaneously active bursts reaches max_bursts (:594)
    BAND_F_AGREE = 32,       // neighbouring bands disagree on a burst near their boundary
    BAND_F_ITER = 64,        // the update vector did not reach its fixed point
    BAND_F_TOTAL = 128,      // more records than kBandMaxTotal
    BAND_F_GONECAP = 256,    // more finished bursts than the caller's record buffer holds
    BAND_F_SNAP = 512,       // more sum snapshots than the buffer holds
    BAND_F_CHAIN = 2048,     // a chained launch found that the scan in front of it had not committed (it wrote nothing)
    BAND_F_CHECK = 4096,     // (option band_selfcheck) the two forms of the boundary test gave different answers
    BAND_F_COOP = 1024,      // the cooperative kernel's grid barrier timed out (not every workgroup became resident)
};

struct BandParams {
    int32_t n, nw64;             // FFT size, u64 words per row of crossing bits
    int32_t log_n;               // n == 1 << log_n
    int32_t n_frames, occ_words; // frames in this scan, ceil(n_frames / 64)
    int32_t hw;                  // burst_width / 2
    int32_t pre_len, post_len, max_len, max_bursts;
    int32_t band_w, n_bands;     // owned bins per band; extended range = band_w / 2 bins on either side
    int32_t gap;                 // ceil(post_len / n): frames after the last crossing at which a burst is gone
    int32_t list_cap;            // entries per frame in the prefilter lists
    float thr;
    uint64_t idx0;               // absolute sample index of frame 0
    int32_t serial = 0;          // launch number: a launch that finds itself void (below) marks BandWork::bar[5] with it
    int32_t selfcheck = 0;       // test hook (option band_selfcheck): bit 0 run both forms of the boundary test and compare,
                                 // bit 1 spoil band i+1's copy of every record in a boundary zone first (both must object),
                                 // bit 3 the wavefront walk without its 64-frame look-ahead (band_wave.hpp: skim),
                                 // bit 4 the plan pass without its LDS (through the workspace arrays, wavefront boundary test),
                                 // bit 5 the speculation pass's guess spoilt in one late frame (a round more, its sums pass restarted)
    int32_t tl_sel = -1;         // >= 0: the passes stamp their first workgroup's start and last one's end into half tl_sel of
                                 // BandWork::tl (diagnostic, option band_timeline)
    int32_t chained = 0;         // 1: enqueued behind a band scan whose verdict the host had not seen: valid only if that
                                 // scan committed (BandWork::bar[4]), else every pass of this launch returns untouched
    int32_t sum_restart = 1;     // 1: a later round's sums pass restarts at the last stored state in front of the first frame whose u
                                 // changed (a snapshot every 64 update steps); 0: every sums pass walks all steps (option band_sum_restart)
    int32_t spec_in = 0;         // 1: this scan has no round 0 of its own -- a speculation pass on a second workspace, run beside
                                 // the previous chunk's scan, produced the update vector round 1 starts from (scan_band.hip, band_spec)
};

struct BandRec {
    int64_t start, last_active, stop;   // stop < 0: still active at the end of the chunk
    int32_t cf;                         // frame in which it was created; -1: carried in (seq = its place in DetState::act)
    int32_t cb;
    float rel, base;                    // relative magnitude at creation (:572), baseline_sum[center] then (:583)
    int32_t flags;                      // bit 0: centre inside the band's own bins, bit 1: ended by max_burst_len
    int32_t seq;
};

IRDM_HD uint32_t band_float_bits(float v)
{
    uint32_t u;
    __builtin_memcpy(&u, &v, 4);
    return u;
}

IRDM_HD bool band_rec_same(const BandRec &a, const BandRec &b)
{
    return a.start == b.start && a.last_active == b.last_active && a.stop == b.stop && a.cf == b.cf &&
           a.cb == b.cb && a.seq == b.seq && (a.flags & 2) == (b.flags & 2) &&
           band_float_bits(a.rel) == band_float_bits(b.rel) && band_float_bits(a.base) == band_float_bits(b.base);
}

struct BandIO {
    const uint64_t *cross;       // [n_frames][nw64] crossing bits (rows of unoccupied frames are undefined)
    const uint64_t *occ;         // [n_bands][occ_words] frames with a crossing inside the band's extended range
    const float *relq;           // [n_frames][n] relative magnitude where the crossing bit is set
    const float *snap;           // [n_snap][n] baseline sums
    const int32_t *slot_post;    // [n_frames] snapshot holding baseline_sum as create_new_bursts sees it (:583)
    const ActiveBurst *act_in;   // bursts active at the start of the chunk
    int32_t n_act_in;
    BandRec *recs;               // [n_bands][kBandRecCap]
    uint32_t *rec_count;         // [n_bands]
    uint64_t *busy, *forced;     // [occ_words] frames that end with a burst active / that force a baseline update
    uint32_t *conc;              // [occ_words] bound on the bursts active at once within 64 frames
    uint32_t *flags;
};

// per-lane slot storage (device: LDS, field-major, one column per lane; host: plain arrays)
struct BandSlots {
    int64_t *start, *la;
    int32_t *cb, *cf, *seq;
    float *rel, *base;
    int stride;
};

#if defined(__HIP_DEVICE_COMPILE__)
IRDM_HD void band_or64(uint64_t *p, uint64_t v) { atomicOr(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v); }
IRDM_HD void band_or32(uint32_t *p, uint32_t v) { atomicOr(p, v); }
IRDM_HD uint32_t band_add32(uint32_t *p, uint32_t v) { return atomicAdd(p, v); }
#else
IRDM_HD void band_or64(uint64_t *p, uint64_t v) { *p |= v; }
IRDM_HD void band_or32(uint32_t *p, uint32_t v) { *p |= v; }
IRDM_HD uint32_t band_add32(uint32_t *p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
#endif

template <int NW>
IRDM_HD void bm_set_range(uint64_t (&m)[NW], int a, int b)       // bits a..b inclusive, clipped to the mask
{
#pragma unroll
    for (int k = 0; k < NW; k++) {
        int lo = a - 64 * k, hi = b - 64 * k;
        if (lo < 0) lo = 0;
        if (hi > 63) hi = 63;
        if (lo <= hi) m[k] |= (~0ull >> (63 - hi)) & (~0ull << lo);
    }
}

template <int NW>
IRDM_HD int bm_test(const uint64_t (&m)[NW], int r)              // 0 <= r < 64 * NW
{
    uint64_t w = 0;
#pragma unroll
    for (int k = 0; k < NW; k++)
        if (k == (r >> 6)) w = m[k];
    return (int)((w >> (r & 63)) & 1);
}

template <int NW>
IRDM_HD bool bm_any(const uint64_t (&m)[NW])
{
    uint64_t w = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) w |= m[k];
    return w != 0;
}

// first occupied frame in [a, b] (clipped to the scan) or INT32_MAX
IRDM_HD int band_next_occ(const uint64_t *occ, int a, int b, int n_frames)
{
    if (a < 0) a = 0;
    if (b >= n_frames) b = n_frames - 1;
    while (a <= b) {
        const int w = a >> 6;
        uint64_t v = occ[w] & (~0ull << (a & 63));
        if (v) {
            const int f = 64 * w + __builtin_ctzll(v);
            return f <= b ? f : 0x7fffffff;
        }
        a = 64 * (w + 1);
    }
    return 0x7fffffff;
}

// segment starts inside time block `blk` (bit i = frame 64*blk + i): occupied, and no occupied frame among the `gap`
// before it; with bursts carried into the band, frame -1 counts as occupied (the carried segment owns what follows)
IRDM_HD uint64_t band_segment_starts(const uint64_t *occ, int blk, int gap, bool carried)
{
    const uint64_t cur = occ[blk], prev = blk > 0 ? occ[blk - 1] : 0;
    uint64_t smear = 0;
    for (int d = 1; d <= gap; d++) smear |= (cur << d) | (prev >> (64 - d));
    if (blk == 0 && carried) smear |= gap >= 64 ? ~0ull : ((1ull << gap) - 1);
    return cur & ~smear;
}

template <int NW>
struct BandWalker {
    const BandParams &P;
    const BandIO &io;
    BandSlots S;
    int band, e0, word0;
    int own_lo, own_hi, rep_lo, rep_hi;      // relative to e0
    uint64_t M[NW], elig[NW];
    uint32_t valid;
    int last_occ;
    int acc_blk;
    uint32_t acc_max;
    int n_events = 0;            // frames processed (diagnostic)

    IRDM_HD BandWalker(const BandParams &p, const BandIO &i, BandSlots s, int b) : P(p), io(i), S(s), band(b)
    {
        const int H = P.band_w / 2;
        e0 = band * P.band_w - H;
        word0 = e0 >> 6;                     // e0 is a multiple of 64 (may be negative)
        own_lo = H;
        own_hi = H + P.band_w;
        rep_lo = own_lo - P.hw;
        rep_hi = own_hi + P.hw;
#pragma unroll
        for (int k = 0; k < NW; k++) M[k] = elig[k] = 0;
        // extract_peaks (:529-552): bins [half_bw, N - half_bw) without the DC notch of +-3 bins
        uint64_t dc[NW];
#pragma unroll
        for (int k = 0; k < NW; k++) dc[k] = 0;
        bm_set_range<NW>(elig, P.hw - e0, P.n - P.hw - 1 - e0);
        bm_set_range<NW>(dc, P.n / 2 - 3 - e0, P.n / 2 + 3 - e0);
#pragma unroll
        for (int k = 0; k < NW; k++) elig[k] &= ~dc[k];
        valid = 0;
        last_occ = -1;
        acc_blk = -1;
        acc_max = 0;
    }

    IRDM_HD void rebuild_mask()
    {
#pragma unroll
        for (int k = 0; k < NW; k++) M[k] = 0;
        for (uint32_t v = valid; v; v &= v - 1) {
            const int i = __builtin_ctz(v);
            const int r = S.cb[i * S.stride] - e0;
            bm_set_range<NW>(M, r - P.hw, r + P.hw);
        }
    }

    IRDM_HD void emit(int i, int64_t stop, int lng)
    {
        const int r = S.cb[i * S.stride] - e0;
        if (r < rep_lo || r >= rep_hi) return;
        const uint32_t at = band_add32(&io.rec_count[band], 1);
        if (at >= (uint32_t)kBandRecCap) {
            band_or32(io.flags, BAND_F_RECS);
            return;
        }
        BandRec g;
        g.start = S.start[i * S.stride];
        g.last_active = S.la[i * S.stride];
        g.stop = stop;
        g.cf = S.cf[i * S.stride];
        g.cb = S.cb[i * S.stride];
        g.rel = S.rel[i * S.stride];
        g.base = S.base[i * S.stride];
        g.flags = ((r >= own_lo && r < own_hi) ? 1 : 0) | (lng ? 2 : 0);
        g.seq = S.seq[i * S.stride];
        io.recs[(size_t)band * kBandRecCap + at] = g;
    }

    IRDM_HD uint32_t owned_count() const
    {
        uint32_t c = 0;
        for (uint32_t v = valid; v; v &= v - 1) {
            const int r = S.cb[__builtin_ctz(v) * S.stride] - e0;
            c += (r >= own_lo && r < own_hi) ? 1 : 0;
        }
        return c;
    }

    // frames a..b end with c of this band's own bursts active
    IRDM_HD void account(int a, int b, uint32_t c)
    {
        if (c == 0 || a > b) return;
        for (int blk = a >> 6; blk <= (b >> 6); blk++) {
            const int lo = blk == (a >> 6) ? (a & 63) : 0, hi = blk == (b >> 6) ? (b & 63) : 63;
            band_or64(&io.busy[blk], (~0ull >> (63 - hi)) & (~0ull << lo));
            if (blk != acc_blk) {
                flush_conc();
                acc_blk = blk;
                acc_max = c;
            } else if (c > acc_max) {
                acc_max = c;
            }
        }
    }

    IRDM_HD void flush_conc()
    {
        if (acc_blk >= 0 && acc_max > 0) band_add32(&io.conc[acc_blk], acc_max);
        acc_blk = -1;
        acc_max = 0;
    }

    // bursts carried into the chunk whose centre lies in the extended range; returns their number
    IRDM_HD int load_carried()
    {
        int n = 0;
        for (int i = 0; i < io.n_act_in; i++) {
            const int r = io.act_in[i].center_bin - e0;
            if (r < 0 || r >= 64 * NW) continue;
            if (n >= kBandSlots) {
                band_or32(io.flags, BAND_F_SLOTS);
                break;
            }
            S.start[n * S.stride] = (int64_t)io.act_in[i].start;
            S.la[n * S.stride] = (int64_t)io.act_in[i].last_active;
            S.cb[n * S.stride] = io.act_in[i].center_bin;
            S.cf[n * S.stride] = -1;
            S.seq[n * S.stride] = i;
            S.rel[n * S.stride] = io.act_in[i].peak_rel;
            S.base[n * S.stride] = io.act_in[i].base_sum;
            valid |= 1u << n;
            n++;
        }
        rebuild_mask();
        return n;
    }

    // earliest frame at which one of the active bursts satisfies last_active + post_len <= index (:505)
    IRDM_HD int min_expiry() const
    {
        int64_t best = 0x7fffffff;
        for (uint32_t v = valid; v; v &= v - 1) {
            const int i = __builtin_ctz(v);
            const int64_t num = S.la[i * S.stride] + (int64_t)P.post_len - (int64_t)P.idx0;
            // (n is a power of two: a 64-bit division here was most of what an event cost a lane)
            const int64_t ef = num <= 0 ? 0 : (num + P.n - 1) >> P.log_n;
            if (ef < best) best = ef;
        }
        return (int)best;
    }

    // frame after `f` at which something can happen, INT32_MAX at the end of the segment
    IRDM_HD int next_event(int f)
    {
        const uint64_t *occ = io.occ + (size_t)band * P.occ_words;
        if (valid == 0) return band_next_occ(occ, f + 1, last_occ + P.gap, P.n_frames);
        const int ef = min_expiry();
        const int nf = band_next_occ(occ, f + 1, ef, P.n_frames);
        return nf < ef ? nf : ef;
    }

    IRDM_HD void process(int f)
    {
        const int64_t index = (int64_t)P.idx0 + (int64_t)f * P.n;
        const uint64_t *occ = io.occ + (size_t)band * P.occ_words;
        const bool occupied = (occ[f >> 6] >> (f & 63)) & 1;
        uint64_t cw[NW];
#pragma unroll
        for (int k = 0; k < NW; k++) {
            const int w = word0 + k;
            cw[k] = (occupied && w >= 0 && w < P.nw64) ? io.cross[(size_t)f * P.nw64 + w] : 0;
        }
        if (occupied) last_occ = f;

        // update_bursts (:458-469): centre bin or a neighbour above the threshold
        if (valid && bm_any<NW>(cw)) {
            uint64_t hit[NW];
#pragma unroll
            for (int k = 0; k < NW; k++) {
                hit[k] = cw[k] | (cw[k] << 1) | (cw[k] >> 1);
                if (k > 0) hit[k] |= cw[k - 1] >> 63;
                if (k < NW - 1) hit[k] |= cw[k + 1] << 63;
            }
            for (uint32_t v = valid; v; v &= v - 1) {
                const int i = __builtin_ctz(v);
                if (bm_test<NW>(hit, S.cb[i * S.stride] - e0)) S.la[i * S.stride] = index;
            }
        }
        // remove_peaks_around_bursts + extract_peaks (:522-552): the mask still holds the bursts this frame deletes
        uint64_t pk[NW];
#pragma unroll
        for (int k = 0; k < NW; k++) pk[k] = cw[k] & ~M[k] & elig[k];

        // delete_gone_bursts (:490-518)
        bool deleted = false;
        for (uint32_t v = valid; v; v &= v - 1) {
            const int i = __builtin_ctz(v);
            const int64_t la = S.la[i * S.stride];
            const bool lng = P.max_len > 0 && (la - S.start[i * S.stride] > (int64_t)P.max_len);
            if (la + (int64_t)P.post_len <= index || lng) {
                const int r = S.cb[i * S.stride] - e0;
                if (lng && r >= own_lo && r < own_hi) band_or64(&io.forced[f >> 6], 1ull << (f & 63));
                emit(i, index, lng ? 1 : 0);
                valid &= ~(1u << i);
                deleted = true;
            }
        }
        if (deleted) rebuild_mask();      // update_burst_mask (:482-486)

        // create_new_bursts (:556-591): peaks in descending relative magnitude (ties: ascending bin, the order a
        // stable sort leaves them in), each masking +-burst_width/2 around itself
        while (bm_any<NW>(pk)) {
            float best = -1.0f;
            int best_r = -1;
#pragma unroll
            for (int k = 0; k < NW; k++) {
                uint64_t w = pk[k];
                const float *row = io.relq + (size_t)f * P.n + (e0 + 64 * k);
                while (w) {
                    // four candidates per round (independent loads; a short tail repeats its last candidate)
                    const int j0 = __builtin_ctzll(w);
                    w &= w - 1;
                    const int j1 = w ? __builtin_ctzll(w) : j0;
                    w &= w ? w - 1 : 0;
                    const int j2 = w ? __builtin_ctzll(w) : j1;
                    w &= w ? w - 1 : 0;
                    const int j3 = w ? __builtin_ctzll(w) : j2;
                    w &= w ? w - 1 : 0;
                    const float v0 = row[j0], v1 = row[j1], v2 = row[j2], v3 = row[j3];
                    if (v0 > best) { best = v0; best_r = 64 * k + j0; }
                    if (v1 > best) { best = v1; best_r = 64 * k + j1; }
                    if (v2 > best) { best = v2; best_r = 64 * k + j2; }
                    if (v3 > best) { best = v3; best_r = 64 * k + j3; }
                }
            }
            if (best_r < 0) break;          // cannot happen: crossing bits carry rel > threshold > 0
            int slot = __builtin_ctz(~valid);
            if (slot >= kBandSlots) {
                band_or32(io.flags, BAND_F_SLOTS);
                break;
            }
            const int cbin = e0 + best_r;
            S.start[slot * S.stride] = index - (int64_t)P.pre_len;
            S.la[slot * S.stride] = index - (int64_t)P.pre_len;
            S.cb[slot * S.stride] = cbin;
            S.cf[slot * S.stride] = f;
            S.seq[slot * S.stride] = 0;
            S.rel[slot * S.stride] = best;
            S.base[slot * S.stride] = io.snap[(size_t)io.slot_post[f] * P.n + cbin];
            valid |= 1u << slot;
            bm_set_range<NW>(M, best_r - P.hw, best_r + P.hw);
#pragma unroll
            for (int k = 0; k < NW; k++) pk[k] &= ~M[k];
        }
    }

    // walk one segment: from frame f_start, or (carried; load_carried() > 0 was called) from the bursts handed over
    // at the chunk boundary
    IRDM_HD void run(int f_start, bool carried)
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
        for (uint32_t v = valid; v; v &= v - 1) emit(__builtin_ctz(v), -1, 0);
    }
};

}  // namespace irdm
