// band_host.cpp -- TEST INFRASTRUCTURE: drives the product's per-band state machine (csrc/band_core.hpp, the same
// source the gfx950 walk kernel compiles) on the CPU, with the plan / sums / cross / verify / commit passes of
// csrc/scan_band.hip restated sequentially around it, so that the band-parallel speculative scan can be checked
// against the oracle's detector without a GPU (tests/test_band_host.py).  Built by the test with g++.
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "../iridium-sniffer_amd/csrc/band_core.hpp"
// the walk pass as the device runs it (a wavefront per band and segment, csrc/band_wave.hpp) on an emulated wavefront
#include "wave_emul.hpp"
#include "../iridium-sniffer_amd/csrc/band_wave.hpp"

using namespace irdm;

namespace {

struct Host {
    DetParams D;
    DetState st;
    std::vector<float> sum, hist;       // [N], [512][N]
    int rounds_total = 0, chunks = 0;
    uint32_t last_flags = 0;
};

int g_walker = 0;        // 0: BandWalker (a lane per band), 1: WaveWalker on the emulated wavefront, 2: the same without skim()

// band_walk_wave_kernel's treatment of one (band, block) pair, all 64 lanes in lock step
template <int NW>
void walk_all_wave(const BandParams &P, BandIO &io)
{
    for (int blk = 0; blk < P.occ_words; blk++) {
        for (int band = 0; band < P.n_bands; band++) {
            if (blk != 0 && band_segment_starts(io.occ + (size_t)band * P.occ_words, blk, P.gap, false) == 0) continue;
            wave_emul::run([&](int lane) {
                bool carried = false;
                if (blk == 0) {
                    WaveWalker<NW> w(P, io, band, lane);
                    if (w.load_carried() > 0) {
                        carried = true;
                        w.run(0, true);
                    }
                }
                uint64_t starts = wv_first64(band_segment_starts(io.occ + (size_t)band * P.occ_words, blk, P.gap, carried));
                while (starts) {
                    const int q = __builtin_ctzll(starts);
                    starts &= starts - 1;
                    WaveWalker<NW> w(P, io, band, lane);
                    w.run(64 * blk + q, false);
                }
            });
        }
    }
}

template <int NW>
void walk_all(const BandParams &P, BandIO &io)
{
    std::vector<int64_t> s_start(kBandSlots), s_la(kBandSlots);
    std::vector<int32_t> s_cb(kBandSlots), s_cf(kBandSlots), s_seq(kBandSlots);
    std::vector<float> s_rel(kBandSlots), s_base(kBandSlots);
    BandSlots S{ s_start.data(), s_la.data(), s_cb.data(), s_cf.data(), s_seq.data(), s_rel.data(), s_base.data(), 1 };
    for (int blk = 0; blk < P.occ_words; blk++) {
        for (int band = 0; band < P.n_bands; band++) {
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
    }
}

// one chunk of n_frames primed frames; returns 1 accepted, 2 aborted (state untouched)
int band_chunk(Host &H, const float *mag, int n_frames, std::vector<GoneBurst> &gone_out, int max_rounds,
               int band_w_override, std::vector<float> &pre_io, std::vector<float> &smin_out)
{
    const DetParams &D = H.D;
    const int N = D.n, F = n_frames;
    BandParams P;
    P.selfcheck = g_walker == 2 ? 8 : 0;
    P.n = N; P.log_n = 31 - __builtin_clz((unsigned)N); P.nw64 = N / 64; P.n_frames = F; P.occ_words = (F + 63) / 64; P.hw = D.width / 2;
    P.pre_len = D.pre_len; P.post_len = D.post_len; P.max_len = D.max_len; P.max_bursts = D.max_bursts;
    P.band_w = band_w_override ? band_w_override : (P.hw <= 20 ? 128 : 256);
    P.list_cap = std::min(kBandListCap, N); P.n_bands = N / P.band_w; P.gap = (D.post_len + N - 1) / N; P.thr = D.threshold; P.idx0 = H.st.index;
    const int OW = P.occ_words;

    // prefilter lists (scan_fast.hip prefilter_kernel); pre_io: the caller lowers it and retries after BAND_F_STALE
    std::vector<float> &pre = pre_io;
    std::vector<std::vector<ListEntry>> lists(F);
    for (int f = 0; f < F; f++)
        for (int b = 0; b < N; b++)
            if (mag[(size_t)f * N + b] > pre[b]) lists[f].push_back(ListEntry{ b, mag[(size_t)f * N + b] });

    std::vector<uint8_t> uq(F, 0), uf(F, 0);
    std::vector<uint64_t> cross((size_t)F * P.nw64), occ((size_t)P.n_bands * OW), busy(OW), forced(OW);
    std::vector<uint32_t> conc(OW), rec_count(P.n_bands);
    std::vector<float> relq((size_t)F * N, 0.0f);
    std::vector<BandRec> recs((size_t)P.n_bands * kBandRecCap);
    std::vector<float> sum_new(N);
    std::vector<int32_t> upd_frame, cnt_before(F + 1), slot_pre(F), slot_post(F);
    std::vector<std::vector<float>> snaps;
    uint32_t flags = 0;
    const int h0 = H.st.hist_idx;
    int n_upd = 0;

    for (int round = 0;; round++) {
        if (round > 0) {
            int mismatch = 0;
            for (int f = 0; f < F; f++) {
                const int q = ((busy[f >> 6] >> (f & 63)) & 1) ? 0 : 1, fc = (int)((forced[f >> 6] >> (f & 63)) & 1);
                if (q != uq[f] || fc != uf[f]) mismatch++;
            }
            for (int b = 0; b < OW; b++)
                if (conc[b] >= (uint32_t)P.max_bursts) flags |= BAND_F_SQUELCH;
            if (flags) { H.last_flags = flags; return 2; }
            if (mismatch == 0) {
                // verify boundaries
                bool bad = false;
                for (int i = 0; i + 1 < P.n_bands && !bad; i++) {
                    const int X = (i + 1) * P.band_w;
                    const BandRec *A = &recs[(size_t)i * kBandRecCap], *B = &recs[(size_t)(i + 1) * kBandRecCap];
                    int ca = 0, cb = 0;
                    for (uint32_t a = 0; a < rec_count[i]; a++) {
                        if (A[a].cb < X - P.hw || A[a].cb >= X + P.hw) continue;
                        ca++;
                        bool found = false;
                        for (uint32_t b = 0; b < rec_count[i + 1] && !found; b++) found = band_rec_same(A[a], B[b]);
                        if (!found) bad = true;
                    }
                    for (uint32_t b = 0; b < rec_count[i + 1]; b++)
                        if (B[b].cb >= X - P.hw && B[b].cb < X + P.hw) cb++;
                    if (ca != cb) bad = true;
                }
                if (bad) { H.last_flags = BAND_F_AGREE; return 2; }
                H.rounds_total += round;
                break;
            }
            if (round >= max_rounds) { H.last_flags = BAND_F_ITER; return 2; }
            for (int f = 0; f < F; f++) {
                uq[f] = ((busy[f >> 6] >> (f & 63)) & 1) ? 0 : 1;
                uf[f] = (uint8_t)((forced[f >> 6] >> (f & 63)) & 1);
            }
        } else {
            for (int f = 0; f < F; f++)
                if (lists[f].size() > (size_t)std::min(kBandListCap, N)) { H.last_flags = BAND_F_LIST; return 2; }
        }
        std::fill(occ.begin(), occ.end(), 0); std::fill(busy.begin(), busy.end(), 0);
        std::fill(forced.begin(), forced.end(), 0); std::fill(conc.begin(), conc.end(), 0);
        std::fill(rec_count.begin(), rec_count.end(), 0);
        // plan
        upd_frame.clear();
        for (int f = 0; f < F; f++) {
            cnt_before[f] = (int)upd_frame.size();
            if (uf[f]) upd_frame.push_back(f);
            if (uq[f]) upd_frame.push_back(f);
        }
        n_upd = (int)upd_frame.size();
        // sums (every state kept: the host has the memory and it keeps this restatement trivial)
        snaps.assign(1, H.sum);
        {
            std::vector<float> s = H.sum;
            std::vector<float> smin = s;
            for (int k = 0; k < n_upd; k++) {
                const float *nw = mag + (size_t)upd_frame[k] * N;
                const float *ol = k < kHistory ? &H.hist[(size_t)((h0 + k) % kHistory) * N]
                                               : mag + (size_t)upd_frame[k - kHistory] * N;
                for (int b = 0; b < N; b++) {
                    const float d = s[b] - ol[b];
                    s[b] = d + nw[b];
                    smin[b] = std::min(smin[b], s[b]);
                }
                snaps.push_back(s);
            }
            sum_new = s;
            for (int b = 0; b < N; b++)
                if (!(pre[b] <= 0.9f * D.threshold * smin[b])) flags |= BAND_F_STALE;
            smin_out = smin;
        }
        std::vector<float> snap_flat;   // BandIO wants one array: flatten lazily, only rows that are referenced
        std::vector<int32_t> row_of(n_upd + 2, -1);
        auto slot_for = [&](int k) {
            if (row_of[k] < 0) {
                row_of[k] = (int)(snap_flat.size() / N);
                snap_flat.insert(snap_flat.end(), snaps[k].begin(), snaps[k].end());
            }
            return row_of[k];
        };
        snap_flat.reserve((size_t)N * 64);
        for (int f = 0; f < F; f++) {
            if (lists[f].empty()) { slot_pre[f] = slot_post[f] = -1; continue; }
            slot_pre[f] = slot_for(cnt_before[f]);
            slot_post[f] = uf[f] ? slot_for(cnt_before[f] + 1) : slot_pre[f];
        }
        // cross
        for (int f = 0; f < F; f++) {
            if (lists[f].empty()) continue;
            uint64_t *row = &cross[(size_t)f * P.nw64];
            memset(row, 0, sizeof(uint64_t) * P.nw64);
            const float *srow = &snap_flat[(size_t)slot_pre[f] * N];
            for (const ListEntry &e : lists[f]) {
                const float base = srow[e.bin];
                const float rel = base > 0 ? e.mag / base : 0.0f;
                if (rel > D.threshold) {
                    row[e.bin >> 6] |= 1ull << (e.bin & 63);
                    relq[(size_t)f * N + e.bin] = rel;
                }
            }
            for (int band = 0; band < P.n_bands; band++) {
                const int w0 = (band * P.band_w - P.band_w / 2) / 64, nw = 2 * P.band_w / 64;
                uint64_t any = 0;
                for (int k = 0; k < nw; k++)
                    if (w0 + k >= 0 && w0 + k < P.nw64) any |= row[w0 + k];
                if (any) occ[(size_t)band * OW + (f >> 6)] |= 1ull << (f & 63);
            }
        }
        // walk
        BandIO io;
        io.cross = cross.data(); io.occ = occ.data(); io.relq = relq.data(); io.snap = snap_flat.data();
        io.slot_post = slot_post.data(); io.act_in = H.st.act; io.n_act_in = H.st.n_act;
        io.recs = recs.data(); io.rec_count = rec_count.data(); io.busy = busy.data(); io.forced = forced.data();
        io.conc = conc.data(); io.flags = &flags;
        if (g_walker && P.band_w == 128) walk_all_wave<4>(P, io);
        else if (g_walker && P.band_w == 256) walk_all_wave<8>(P, io);
        else if (P.band_w == 128) walk_all<4>(P, io);
        else if (P.band_w == 256) walk_all<8>(P, io);
        else return -1;
    }

    // commit (scan_band.hip band_commit_kernel)
    std::vector<const BandRec *> tot;
    for (int band = 0; band < P.n_bands; band++)
        for (uint32_t i = 0; i < rec_count[band]; i++)
            if (recs[(size_t)band * kBandRecCap + i].flags & 1) tot.push_back(&recs[(size_t)band * kBandRecCap + i]);
    auto key1 = [&](const BandRec *r) -> uint64_t {
        if (r->cf < 0) return (uint64_t)r->seq;
        return (1ull << 63) | ((uint64_t)r->cf << 46) | ((uint64_t)(0xffffffffu - band_float_bits(r->rel)) << 14) | (uint64_t)r->cb;
    };
    std::sort(tot.begin(), tot.end(), [&](const BandRec *a, const BandRec *b) { return key1(a) < key1(b); });
    int n_carried = 0;
    for (const BandRec *r : tot) n_carried += r->cf < 0;
    std::vector<uint64_t> ids(tot.size());
    for (size_t t = 0; t < tot.size(); t++)
        ids[t] = tot[t]->cf < 0 ? H.st.act[tot[t]->seq].id : H.st.burst_id + 10ull * (uint64_t)((int)t - n_carried);
    std::vector<size_t> order(tot.size());
    for (size_t t = 0; t < tot.size(); t++) order[t] = t;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
        const int64_t sa = tot[a]->stop >= 0 ? tot[a]->stop : INT64_MAX, sb = tot[b]->stop >= 0 ? tot[b]->stop : INT64_MAX;
        return sa < sb;
    });
    std::vector<ActiveBurst> act;
    for (size_t t : order) {
        const BandRec *r = tot[t];
        if (r->stop >= 0) {
            GoneBurst g;
            g.id = ids[t]; g.start = (uint64_t)r->start; g.stop = (uint64_t)r->stop; g.last_active = (uint64_t)r->last_active;
            g.center_bin = r->cb; g.peak_rel = r->rel; g.base_sum = r->base; g.pad = 0;
            gone_out.push_back(g);
        } else {
            ActiveBurst a;
            a.id = ids[t]; a.start = (uint64_t)r->start; a.last_active = (uint64_t)r->last_active;
            a.center_bin = r->cb; a.peak_rel = r->rel; a.base_sum = r->base; a.pad = 0;
            act.push_back(a);
        }
    }
    for (size_t i = 0; i < act.size(); i++) H.st.act[i] = act[i];
    H.st.n_act = (int)act.size();
    H.st.burst_id += 10ull * (uint64_t)((int)tot.size() - n_carried);
    H.st.index += (uint64_t)F * N;
    H.st.squelch = H.st.squelch > F ? H.st.squelch - F : 0;
    for (int k = std::max(0, n_upd - kHistory); k < n_upd; k++)
        memcpy(&H.hist[(size_t)((h0 + k) % kHistory) * N], mag + (size_t)upd_frame[k] * N, sizeof(float) * N);
    H.st.hist_idx = (h0 + n_upd) % kHistory;
    H.sum = sum_new;
    H.chunks++;
    return 1;
}

}  // namespace

extern "C" {

// which form of the walk the next scans use (g_walker)
void band_host_set_walker(int w) { g_walker = w; }

// mag: [n_frames][n] magnitude frames of a stream from its first sample.  The first 512 frames prime the baseline
// (burst_detect.c:427-428, :448-452); the rest is scanned in chunks of chunk_frames.  Returns the number of finished
// bursts written to out (emission order), or -(flags) if a chunk aborted.  stats: [0] rounds, [1] chunks.
int band_host_scan(const float *mag, int n_frames, int n, int pre_len, int post_len, int width, int max_bursts,
                   int max_len, float threshold, int chunk_frames, int max_rounds, int band_w, GoneBurst *out,
                   int out_cap, float *sum_out, int *stats)
{
    Host H;
    H.D.n = n; H.D.log_n = 0; H.D.pre_len = pre_len; H.D.post_len = post_len; H.D.width = width;
    H.D.max_bursts = max_bursts; H.D.max_len = max_len; H.D.threshold = threshold;
    memset(&H.st, 0, sizeof(H.st));
    H.sum.assign(n, 0.0f);
    H.hist.assign((size_t)kHistory * n, 0.0f);
    if (n_frames < kHistory) return -1;
    for (int f = 0; f < kHistory; f++) {
        const float *m = mag + (size_t)f * n;
        for (int b = 0; b < n; b++) {
            const float d = H.sum[b] - 0.0f;
            H.sum[b] = d + m[b];
        }
        memcpy(&H.hist[(size_t)f * n], m, sizeof(float) * n);
    }
    H.st.index = (uint64_t)kHistory * n;
    H.st.primed = 1;
    std::vector<GoneBurst> gone;
    stats[3] = 0;
    for (int f0 = kHistory; f0 < n_frames; f0 += chunk_frames) {
        const int F = std::min(chunk_frames, n_frames - f0);
        std::vector<float> pre(n), smin;
        for (int b = 0; b < n; b++) pre[b] = 0.5f * threshold * H.sum[b];
        int rc = 0;
        for (int attempt = 0; attempt < 3; attempt++) {
            rc = band_chunk(H, mag + (size_t)f0 * n, F, gone, max_rounds, band_w, pre, smin);
            if (rc == 1 || rc < 0 || H.last_flags != BAND_F_STALE) break;
            // the noise floor fell below what the lists assumed: lower the list threshold where it did and redo
            for (int b = 0; b < n; b++) pre[b] = std::min(pre[b], 0.45f * threshold * smin[b]);
            stats[3]++;
        }
        if (rc != 1) return rc < 0 ? -1 : -(int)H.last_flags - 1000;
    }
    if ((int)gone.size() > out_cap) return -2;
    for (size_t i = 0; i < gone.size(); i++) out[i] = gone[i];
    memcpy(sum_out, H.sum.data(), sizeof(float) * n);
    stats[0] = H.rounds_total;
    stats[1] = H.chunks;
    stats[2] = H.st.n_act;
    return (int)gone.size();
}

}
