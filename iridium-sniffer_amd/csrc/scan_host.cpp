// scan_host.cpp -- the detector scan of a chunk from the host's side: which scan a chunk gets (band scan; the sequential forms
// as fallbacks), the chained launch behind the scan in flight, the speculation pass of the next chunk, settling (continuation,
// stale-list retry, fallbacks, record-buffer growth) and the hand-over of the finished bursts to a batch context.
#include "pipeline.hpp"

namespace irdmh {

// ---- detector scan of one chunk: sparse kernel with the dense kernel as exact fallback ----
// scan_launch only enqueues (detector stream; the scan kernels themselves hop to sstream, which pipeline_depth 1
// confines to one CU); scan_finish waits, falls back to the dense scan if the sparse one aborted, and fetches the
// finished bursts into h_gone.  pipeline_depth 0 calls them back to back; pipeline_depth 1 calls scan_finish at the
// start of the NEXT feed, so the detector of chunk k runs while the host returns, the caller produces chunk k+1 and
// the FFT of chunk k+1 executes.
// (every pass of a band scan, the sequential scans, snapshots and the state export / import run on the detector's one stream:
// they are ordered by it)
int hist_fence(irdm_pipeline *) { return 0; }

uint32_t next_scan_seq(irdm_pipeline *p)
{
    if (++p->seq_counter == 0) ++p->seq_counter;
    return p->seq_counter;
}

int scan_hop_in(irdm_pipeline *p)
{
    if (p->sstream == p->stream) return 0;
    IRDM_HIP_CHECK(hipEventRecord(p->ev_scan_in, p->stream));
    IRDM_HIP_CHECK(hipStreamWaitEvent(p->sstream, p->ev_scan_in, 0));
    return 0;
}

int scan_hop_out(irdm_pipeline *p)
{
    if (p->sstream == p->stream) return 0;
    IRDM_HIP_CHECK(hipEventRecord(p->ev_scan_out, p->sstream));
    IRDM_HIP_CHECK(hipStreamWaitEvent(p->stream, p->ev_scan_out, 0));
    return 0;
}

int scan_dense(irdm_pipeline *p, const float *mag, int n_frames, bool timed)
{
    if (hist_fence(p) != 0 || scan_hop_in(p) != 0) return -1;
    if (timed) IRDM_HIP_CHECK(hipEventRecord(p->ev_sk[0], p->sstream));
    if (launch_detect_scan(p->P, p->d_state, p->d_sum, p->d_hist, mag, n_frames, p->d_gone, p->gone_cap,
                           p->d_cand_a, p->d_cand_b, p->sstream) != 0)
        return -1;
    if (timed) IRDM_HIP_CHECK(hipEventRecord(p->ev_sk[1], p->sstream));
    if (scan_hop_out(p) != 0) return -1;
    p->stat_dense_frames += n_frames;
    return 0;
}

// which scan a chunk gets: scan_mode 0 = the band scan where the geometry allows it (else the sparse leader scan,
// else dense), 1 dense, 2 / 3 the sparse leader scan on one CU / with updater workgroups, 4 band
int scan_pick(const irdm_pipeline *p)
{
    if (p->scan_mode == 1) return 0;
    if ((p->scan_mode == 0 || p->scan_mode == 4) && p->band_ok) return 2;
    return p->P.n >= 2048 ? 1 : 0;          // the sparse kernel's lanes own 2048-bin quarters (scan_fast.hip)
}

int scan_snapshot(irdm_pipeline *p)
{
    const DetParams &P = p->P;
    // snapshot of the carried state (a few tens of MB, D2D): restored if the sparse scan aborts or the burst-record
    // buffer turns out too small (scan_finish then redoes the chunk)
    if (hist_fence(p) != 0) return -1;
    IRDM_HIP_CHECK(hipMemcpyAsync(p->d_sum_bak, p->d_sum, sizeof(float) * P.n, hipMemcpyDeviceToDevice, p->stream));
    IRDM_HIP_CHECK(hipMemcpyAsync(p->d_hist_bak, p->d_hist, sizeof(float) * (size_t)kHistory * P.n,
                                  hipMemcpyDeviceToDevice, p->stream));
    IRDM_HIP_CHECK(hipMemcpyAsync(p->d_state_bak, p->d_state, sizeof(DetState), hipMemcpyDeviceToDevice, p->stream));
    return 0;
}

int scan_restore(irdm_pipeline *p)
{
    const DetParams &P = p->P;
    if (hist_fence(p) != 0) return -1;
    IRDM_HIP_CHECK(hipMemcpyAsync(p->d_sum, p->d_sum_bak, sizeof(float) * P.n, hipMemcpyDeviceToDevice, p->stream));
    IRDM_HIP_CHECK(hipMemcpyAsync(p->d_hist, p->d_hist_bak, sizeof(float) * (size_t)kHistory * P.n,
                                  hipMemcpyDeviceToDevice, p->stream));
    IRDM_HIP_CHECK(hipMemcpyAsync(p->d_state, p->d_state_bak, sizeof(DetState), hipMemcpyDeviceToDevice, p->stream));
    return 0;
}

// the band scan proper over the primed frames [done, n_frames) of the chunk; retry = 1: the lists went stale, rebuild
// them against the lowered reference first
int scan_band_enqueue_at(irdm_pipeline *p, const float *mag, int n_frames, int done, int retry, bool more_rounds,
                                uint64_t c0, const irdm_pipeline::FeedSlot *feed, int sel, int first, int chained,
                                uint32_t seq, uint64_t chunk_no, bool use_spec)
{
    (void)seq; (void)chunk_no;
    const DetParams &P = p->P;
    const float *mag_rest = mag + (size_t)done * P.n;
    const uint64_t idx0 = c0 + (uint64_t)done * (uint64_t)P.n;           // chunks start on frame boundaries
    // the candidate lists: K1's (whole chunk, frame 0 first: only when nothing of the chunk was primed away), else the
    // prefilter pass; a retry rebuilds them in the same buffers against the lowered levels
    const bool from_k1 = feed && feed->lists && done == 0;
    const int ls = from_k1 && p->depth ? (int)(feed - p->fs) : 0;
    float *pre = from_k1 ? p->k1_pre[ls] : p->d_pre;
    unsigned *counts = from_k1 ? p->k1_counts[ls] : p->d_counts;
    ListEntry *entries = from_k1 ? p->k1_entries[ls] : p->d_entries;
    int *pin = p->h_pin_set[sel];
    GoneBurst *hpg = p->hp_gone_set[sel];
    if (more_rounds) {
        // the first rounds left the verdict open: the remaining rounds, on the same lists and workspace
        return launch_band_scan(P, p->band, p->d_state, p->d_sum, p->d_hist, mag_rest, n_frames - done, idx0, counts, entries, pre,
                                p->d_smin, p->d_gone, p->gone_cap, first, kBandRounds, hpg,
                                reinterpret_cast<uint32_t *>(pin + 64), pin + 96, p->hp_gone_cap, 0, sel, p->stream, p->band_tune);
    }
    if (!from_k1 || retry) {
        if (launch_prefilter_lists(p->d_sum, P.threshold, pre, retry ? p->d_smin : nullptr, mag_rest, P.n, counts,
                                   entries, n_frames - done, band_list_cap(P.n), p->stream) != 0)
            return -1;
    } else {
        p->stat_k1_lists++;
    }
    IRDM_HIP_CHECK(hipEventRecord(p->ev_sk_set[sel][0], p->stream));
    const bool gate = p->gate_armed && !retry && p->hp_gate_dev;
    if (gate) {
        p->gate_armed = false;
        p->gate_open_pending = true;
    }
    // (use_spec: this chunk's round 0 was made by a speculation pass, spec_enqueue: the scan opens with round 1)
    if (launch_band_scan(P, p->band, p->d_state, p->d_sum, p->d_hist, mag_rest, n_frames - done, idx0, counts,
                         entries, pre, p->d_smin, p->d_gone, p->gone_cap, use_spec ? 1 : 0, first, hpg,
                         reinterpret_cast<uint32_t *>(pin + 64), pin + 96, p->hp_gone_cap, chained, sel, p->stream, p->band_tune,
                         gate ? p->hp_gate_dev : nullptr, p->gate_seq, gate ? p->hp_gate_dev + 1 : nullptr,
                         p->gate_src, sizeof(float) * (size_t)kHistory * P.n,
                         use_spec ? &p->band_spec : nullptr, p->ev_sums1) != 0)
        return -1;
    if (use_spec) p->stat_spec_scans++;
    IRDM_HIP_CHECK(hipEventRecord(p->ev_sk_set[sel][1], p->stream));
    // (the control block reaches the host with the records: scan_export)
    return 0;
}

// ... of the scan in flight (fl_*)
int scan_band_enqueue(irdm_pipeline *p, const float *mag, int n_frames, int done, int retry, bool more_rounds)
{
    if (!more_rounds && !retry) p->fl_band_first = p->band_first ? p->band_first : p->band_auto;
    return scan_band_enqueue_at(p, mag, n_frames, done, retry, more_rounds, p->fl_c0, p->fl_feed, p->out_sel, p->fl_band_first, 0,
                                p->fl_seq, p->fl_no);
}

// the sequential forms: the sparse leader scan with the dense kernel as its exact fallback, or the dense kernel alone
int scan_legacy_enqueue(irdm_pipeline *p, const float *mag, int n_frames, int done, bool sparse)
{
    const DetParams &P = p->P;
    if (hist_fence(p) != 0) return -1;
    if (sparse) {
        IRDM_HIP_CHECK(hipMemsetAsync(p->d_status, 0, sizeof(int) * 64, p->stream));
        if (done < n_frames) {
            const float *mag_rest = mag + (size_t)done * P.n;
            if (launch_prefilter(p->d_sum, P.threshold, p->d_pre, mag_rest, P.n, p->d_counts, p->d_entries,
                                 p->d_goff, p->d_compact, n_frames - done, p->stream) != 0)
                return -1;
            // the leader and every updater need a CU of their own (one workgroup's LDS fills more than half a CU): more
            // workgroups than the scan stream has CUs would wait for each other until the bounded spins give up
            const int upd = (p->scan_mode == 3 || (p->scan_mode != 2 && p->mc_auto))
                                ? std::min(p->mc_updaters, p->scan_cus - 1) : 0;
            const int mc_words = (int)std::min<size_t>((size_t)p->mc_ops_cap, 3 * (size_t)(n_frames - done) + 64);
            if (upd > 0) {
                // at most 3 operations per frame + the exit word
                IRDM_HIP_CHECK(hipMemsetAsync(p->d_mc_ops, 0, sizeof(unsigned long long) * (size_t)mc_words, p->stream));
                IRDM_HIP_CHECK(hipMemsetAsync(p->d_mc_done, 0, sizeof(unsigned) * 32 * 16, p->stream));
            }
            if (scan_hop_in(p) != 0) return -1;
            IRDM_HIP_CHECK(hipEventRecord(p->ev_sk[0], p->sstream));
            if (launch_detect_scan_fast(P, p->d_state, p->d_sum, p->d_hist, mag_rest, n_frames - done,
                                        p->d_counts, p->d_goff, p->d_compact, p->d_pre, p->d_gone,
                                        p->gone_cap, p->d_status, p->d_mc_ops, mc_words, p->d_mc_done, upd, p->sstream) != 0)
                return -1;
            IRDM_HIP_CHECK(hipEventRecord(p->ev_sk[1], p->sstream));
            if (scan_hop_out(p) != 0) return -1;
        }
        IRDM_HIP_CHECK(hipMemcpyAsync(p->h_pin, p->d_status, sizeof(int) * 64, hipMemcpyDeviceToHost, p->stream));
    } else if (done < n_frames) {
        if (scan_dense(p, mag + (size_t)done * P.n, n_frames - done, true) != 0) return -1;
    }
    return 0;
}

// the scan's records and header words into pinned host memory, behind whatever the scan stream holds
int scan_export(irdm_pipeline *p)
{
    const bool band = p->fl_mode == 2 && p->fl_band_ran;
    return launch_gone_export(p->d_state, p->d_gone, std::min(p->gone_cap, p->hp_gone_cap), p->hp_gone,
                              reinterpret_cast<uint32_t *>(p->h_pin + 64), band ? p->band.ctl : nullptr, p->h_pin + 96,
                              (int)sizeof(BandCtl), p->stream);
}

// the export targets the next scan_finish reads
void scan_select_outputs(irdm_pipeline *p, int sel)
{
    p->out_sel = sel;
    p->h_pin = p->h_pin_set[sel];
    p->hp_gone = p->hp_gone_set[sel];
    p->ev_sk[0] = p->ev_sk_set[sel][0];
    p->ev_sk[1] = p->ev_sk_set[sel][1];
    p->ev_end = p->ev_end_set[sel];
}

// scan_chain: chunk k's band scan enqueued BEHIND chunk k-1's, before the host has seen that one's verdict.  The two
// are on the same stream, so the GPU starts scan k the moment scan k-1 ends; without this the scan engine idled for the
// host's wake-up from the wait plus the enqueue of the first pass (0.2-0.3 ms of a 1.4 ms period, and the scans in
// sequence ARE the period).  Safe because a band scan writes nothing of the carried state before its commit and commits
// only as the last thing it does: the chained scan's first pass checks on the device that its predecessor committed
// (BandWork::bar[4]) and declines itself otherwise (BAND_F_CHAIN), the host sees the predecessor's trouble when it
// settles it, drains the declined launch and launches again the ordinary way.  Exports go to the other set of pinned
// targets.
// (no: the chunk's number -- this feed's, or, from the end of the previous feed, the next one's)
int scan_chain_try(irdm_pipeline *p, irdm_pipeline::FeedSlot &f, uint64_t no)
{
    p->chain_pending = false;
    // (only with K1's own candidate lists: the prefilter pass that builds them otherwise writes the one set of buffers
    // the scan in front may still need for a continuation or a retry, and it runs before the chained launch's check)
    if (!p->fl_active || p->fl_mode != 2 || !p->fl_band_ran || !p->host_primed || scan_pick(p) != 2 ||
        f.frames < 1 || !f.lists)
        return 0;
    const int sel = p->out_sel ^ 1;
    IRDM_HIP_CHECK(hipStreamWaitEvent(p->stream, f.ev_k1, 0));
    memset(p->h_pin_set[sel] + 96, 0, sizeof(BandCtl));
    p->chain_band_first = p->band_first ? p->band_first : p->band_auto;
    p->chain_seq = next_scan_seq(p);
    // a speculation pass for exactly this chunk (spec_enqueue, at the end of the previous feed)?  Then round 0 is done: the
    // scan waits for that pass and opens with round 1.  (Only here, in the chained launch: a scan that is launched again
    // after its predecessor's trouble, a retry or a continuation finds the speculation workspace taken by the next pass.)
    const bool use_spec = p->band_spec_opt && p->d_band_spec && p->spec_for_no == no && p->chain_band_first >= 2 &&
                          f.frames == p->spec_frames && !p->gate_armed;
    if (use_spec) IRDM_HIP_CHECK(hipStreamWaitEvent(p->stream, p->ev_spec_done, 0));
    if (scan_band_enqueue_at(p, f.mag, f.frames, 0, 0, false, f.c0, &f, sel, p->chain_band_first, 1, p->chain_seq, no, use_spec) != 0) return -1;
    IRDM_HIP_CHECK(hipEventRecord(p->ev_end_set[sel], p->stream));
    p->chain_pending = true;
    p->chain_no = no;
    p->chain_sel = sel;
    p->stat_chained++;
    return 0;
}

// The speculation pass of the NEXT chunk (feed slot `nx`, chunk number `no`; K1 and its candidate lists are enqueued or
// done), on its own stream: behind K1 of that chunk and behind the first sums pass of the scan just enqueued -- the sums it
// tests against -- which is also behind that scan's plan pass, the one reader of the workspace this pass overwrites.
int spec_enqueue(irdm_pipeline *p, irdm_pipeline::FeedSlot &nx, uint64_t no)
{
    if (!p->band_spec_opt || !p->d_band_spec || !p->host_primed || scan_pick(p) != 2 || nx.frames < 1 || !nx.lists) return 0;
    const int ls = (int)(&nx - p->fs);
    IRDM_HIP_CHECK(hipStreamWaitEvent(p->stream_spec, nx.ev_k1, 0));
    IRDM_HIP_CHECK(hipStreamWaitEvent(p->stream_spec, p->ev_sums1, 0));
    const int have_prev = p->spec_for_no != ~0ull && p->spec_for_no + 1 == no;
    if (launch_band_spec(p->P, p->band_spec, p->d_state_spec, p->band.sum_new, nx.frames, nx.c0, p->k1_counts[ls], p->k1_entries[ls],
                         have_prev, p->stream_spec, p->band_tune) != 0)
        return -1;
    IRDM_HIP_CHECK(hipEventRecord(p->ev_spec_done, p->stream_spec));
    p->spec_for_no = no;
    p->spec_frames = nx.frames;
    p->stat_spec_passes++;
    return 0;
}

int scan_launch(irdm_pipeline *p, const float *mag, int n_frames, uint64_t c1)
{
    if (p->chain_pending) {
        // enqueued by scan_chain_try (a band scan of a primed detector from frame 0): the bookkeeping only
        p->chain_pending = false;
        p->fl_mode = 2;
        p->fl_sparse = false;
        p->fl_mag = mag;
        p->fl_frames = n_frames;
        p->fl_c1 = c1;
        p->fl_c0 = p->total_samples;
        p->fl_no = p->chunk_no;
        p->fl_done = 0;
        p->fl_band_ran = true;
        p->fl_band_first = p->chain_band_first;
        p->fl_seq = p->chain_seq;
        scan_select_outputs(p, p->chain_sel);
        p->fl_active = true;
        return 0;
    }
    p->fl_mode = scan_pick(p);
    // (the band scan zeroes the chunk's finished-burst count in its first pass; the priming frames and the sequential
    // scans append to it)
    if (p->fl_mode != 2 || !p->host_primed) IRDM_HIP_CHECK(hipMemsetAsync(&p->d_state->n_gone, 0, sizeof(uint32_t), p->stream));
    p->fl_sparse = p->fl_mode == 1;
    p->fl_mag = mag;
    p->fl_frames = n_frames;
    p->fl_c1 = c1;
    p->fl_c0 = p->total_samples;
    p->fl_no = p->chunk_no;
    p->fl_seq = next_scan_seq(p);
    // stream start: the first 512 frames only prime the baseline (burst_detect.c:427-428) -- dense kernel, no bursts
    int done = 0;
    if (!p->host_primed && p->fl_mode != 0) {
        done = std::min(n_frames, kHistory - p->host_hist_idx);
        if (scan_dense(p, mag, done, done == n_frames) != 0) return -1;
    }
    p->fl_done = done;
    // (a snapshot, where one is taken, is the state AFTER the priming frames: a redo restarts at frame `done`)
    if (p->fl_mode == 2) {
        // nothing of the carried state is written before the band scan's commit: no snapshot
        memset(p->h_pin + 96, 0, sizeof(BandCtl));
        p->fl_band_ran = done < n_frames;
        if (done < n_frames) {
            if (scan_band_enqueue(p, mag, n_frames, done, 0) != 0) return -1;
        } else {
            reinterpret_cast<BandCtl *>(p->h_pin + 96)->status = 1;       // the chunk was all priming
        }
    } else {
        if (scan_snapshot(p) != 0) return -1;
        if (scan_legacy_enqueue(p, mag, n_frames, done, p->fl_mode == 1) != 0) return -1;
    }
    // (the band scan's last pass has exported its records and control block already)
    if (!(p->fl_mode == 2 && p->fl_band_ran) && scan_export(p) != 0) return -1;
    IRDM_HIP_CHECK(hipEventRecord(p->ev_end, p->stream));
    p->fl_active = true;
    return 0;
}

int scan_finish(irdm_pipeline *p, int *n_gone_out)
{
    *n_gone_out = 0;
    if (!p->fl_active) return 0;
    p->fl_active = false;
    auto now_us = [] {
        struct timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
    };
    double tq0 = now_us(), tq1;
    // (the scan's own end, not the stream's: the next chunk's scan may be enqueued behind it already)
    IRDM_HIP_CHECK(hipEventSynchronize(p->ev_end));
    if (p->gate_open_pending) {
        // the scan waited for the previous chunk's history (irdm_expect_history): whatever runs from here on -- more
        // rounds, a retry, a sequential fallback -- reads it too
        p->gate_open_pending = false;
        if (p->hp_gate[1]) {
            fprintf(stderr, "irdm_hip: the detector scan waited for a history import that never came (irdm_expect_history)\n");
            p->hp_gate[1] = 0;
            return -1;
        }
    }
    tq1 = now_us(); p->host_us[6] += tq1 - tq0; tq0 = tq1;          // [6] waiting for the scan itself
    int redo_from = p->fl_done;       // where a dense redo restarts (the priming frames are never redone)
    bool redone = false;              // something ran after the export the launch enqueued
    if (p->fl_mode == 2) {
        const BandCtl *ctl = reinterpret_cast<const BandCtl *>(p->h_pin + 96);
        int tries = 0;
        auto more_rounds = [&]() -> int {
            // verdict still open after the rounds enqueued up front: run the rest
            if (ctl->status != 0 || ctl->flags != 0 || !p->fl_band_ran) return 0;
            redone = true;
            p->stat_band_extra++;
            if (scan_band_enqueue(p, p->fl_mag, p->fl_frames, p->fl_done, 0, true) != 0) return -1;
            if (scan_export(p) != 0) return -1;
            IRDM_HIP_CHECK(hipStreamSynchronize(p->stream));
            return 0;
        };
        if (more_rounds() != 0) return -1;
        while (ctl->status != 1 && ctl->flags == BAND_F_STALE && tries < 2) {
            // a bin's running sum fell below what the prefilter lists assumed (the noise floor dropped by more than
            // 1.8x inside the chunk): rebuild the lists against the lowest sums seen and scan again
            tries++;
            redone = true;
            p->stat_band_retries++;
            if (scan_band_enqueue(p, p->fl_mag, p->fl_frames, p->fl_done, 1) != 0) return -1;
            if (scan_export(p) != 0) return -1;
            IRDM_HIP_CHECK(hipStreamSynchronize(p->stream));
            if (more_rounds() != 0) return -1;
        }
        p->stat_band_rounds += (uint64_t)ctl->rounds;
        p->stat_band_steps += (uint64_t)(ctl->n_upd > 0 ? ctl->n_upd : 0);      // (update steps of the last round: what the sums pass walked)
        if (p->band_tune.timeline && p->fl_band_ran) {
            // (diagnostic) the passes' device timeline of this scan: durations, and the idle time in front of each pass
            unsigned long long tl[2 * kBandTlSlots];
            IRDM_HIP_CHECK(hipMemcpy(tl, p->band.tl + (size_t)p->out_sel * 2 * kBandTlSlots, sizeof(tl), hipMemcpyDeviceToHost));
            for (int i = 26; i < 32; i++) p->stat_tl_dur[i] += tl[kBandTlSlots + i];      // (event counts of the walk passes)
            unsigned long long prev_end = 0;
            for (int i = 0; i < 26; i++) {
                const unsigned long long lo = tl[i], hi = tl[kBandTlSlots + i];
                if (lo == ~0ull || hi == 0 || hi < lo) continue;
                p->stat_tl_dur[i] += hi - lo;
                if (prev_end && lo > prev_end) p->stat_tl_gap[i] += lo - prev_end;
                p->stat_tl_n[i]++;
                prev_end = hi;
            }
        }
        for (int i = 0; i < 16; i++) p->stat_plan_tp[i] += ctl->tp[i];
        p->stat_sum_restarts += (uint64_t)(ctl->n_restarts > 0 ? ctl->n_restarts : 0);
        if (ctl->status == 1) {
            p->band_auto = std::min(std::max(ctl->rounds, 2), kBandRounds);
            p->stat_band_chunks++;
            p->stat_fast_chunks++;
        } else {
            // declined (possible squelch, capacities, no fixed point, ...): the carried state is untouched, the
            // sequential kernels take the chunk
            p->stat_band_aborts++;
            p->stat_fallbacks++;
            redone = true;
            p->last_band_flags = ctl->flags;
            if (getenv("IRDM_SCAN_DEBUG"))
                fprintf(stderr, "irdm_hip: band scan declined the chunk (flags 0x%x, %d rounds, %d mismatches from frame %d) -> sequential scan\n",
                        ctl->flags, ctl->rounds, ctl->mismatch, ctl->first_mismatch);
            p->fl_mode = p->P.n >= 2048 && p->scan_mode != 4 ? 1 : 0;
            p->fl_sparse = p->fl_mode == 1;
            IRDM_HIP_CHECK(hipMemsetAsync(&p->d_state->n_gone, 0, sizeof(uint32_t), p->stream));
            if (scan_snapshot(p) != 0) return -1;
            if (scan_legacy_enqueue(p, p->fl_mag, p->fl_frames, p->fl_done, p->fl_mode == 1) != 0) return -1;
            IRDM_HIP_CHECK(hipEventRecord(p->ev[2], p->stream));
            IRDM_HIP_CHECK(hipStreamSynchronize(p->stream));
        }
    }
    if (p->fl_mode == 1) {
        const int status = p->h_pin[0];
        if (getenv("IRDM_SCAN_DEBUG")) {
            const long long *d = reinterpret_cast<const long long *>(p->h_pin + 4);
            fprintf(stderr, "scan dbg (10ns ticks): all=%lld leader=%lld | stage=%lld(%lld) cross=%lld(%lld) hc=%lld(%lld) find=%lld(%lld) "
                            "partA=%lld(%lld) partB=%lld(%lld) | publish=%lld(nbulk %lld) frame_end=%lld(%lld) | cmdcross=%lld(%lld) cmdbulk=%lld(%lld) busytop_total=%lld(%lld) fast1=%lld(%lld) fast2=%lld(%lld) bulk_frames=%lld\n",
                    d[0], d[7], d[1], d[13], d[2], d[14], d[3], d[15], d[4], d[16], d[5], d[17], d[6], d[18], d[8], d[20], d[9], d[21],
                    d[10], d[22], d[11], d[23], d[12], d[24], d[25], d[26], d[27], d[28], d[29]);
        }
        if (status != 0) {
            // a list overflowed, went stale, or missed a crossing: redo the chunk with the dense scan
            p->stat_fallbacks++;
            redone = true;
            if (getenv("IRDM_SCAN_DEBUG")) fprintf(stderr, "irdm_hip: sparse scan aborted with status 0x%x -> dense scan\n", status);
            if (scan_restore(p) != 0) return -1;
            if (scan_dense(p, p->fl_mag + (size_t)redo_from * p->P.n, p->fl_frames - redo_from, true) != 0) return -1;
            IRDM_HIP_CHECK(hipEventRecord(p->ev[2], p->stream));
        } else {
            p->stat_fast_chunks++;
        }
    }
    volatile uint32_t *counters = reinterpret_cast<volatile uint32_t *>(p->h_pin + 64);
    volatile int32_t *hdr = p->h_pin + 66;
    p->settle_clean = !redone;
    if (redone) {
        if (scan_export(p) != 0) return -1;
        IRDM_HIP_CHECK(hipStreamSynchronize(p->stream));
    }
    tq1 = now_us(); p->host_us[7] += tq1 - tq0; tq0 = tq1;          // [7] retries / fallbacks + the counters' round trip
    p->host_hist_idx = hdr[0];
    p->host_primed = hdr[1];
    int n_gone = (int)counters[0];
    if (counters[1] || n_gone > p->gone_cap) {
        p->settle_clean = false;
        // more finished bursts in this chunk than the record buffer holds (the reference's lists grow without bound,
        // burst_detect.c:148-154): grow it, restore the pre-chunk state and redo the chunk with the dense scan
        const int want = std::max(n_gone, p->gone_cap) + 4096;
        GoneBurst *bigger = dev_alloc<GoneBurst>((size_t)want);
        if (!bigger) {
            fprintf(stderr, "irdm_hip: %d bursts in one chunk and no memory for their records\n", n_gone);
            return -1;
        }
        (void)hipFree(p->d_gone);
        p->d_gone = bigger;
        p->gone_cap = want;
        p->h_gone.resize(want);
        p->stat_fallbacks++;
        if (p->gone_cap > p->hp_gone_cap) {
            // (no other scan can be exporting: one chained behind this one has declined itself and was drained by the
            // retry's stream synchronise above)
            p->hp_gone_cap = p->gone_cap;
            for (int s = 0; s < 2; s++) {
                (void)hipHostFree(p->hp_gone_set[s]);
                p->hp_gone_set[s] = nullptr;
                if (hipHostMalloc(reinterpret_cast<void **>(&p->hp_gone_set[s]), sizeof(GoneBurst) * (size_t)p->hp_gone_cap, hipHostMallocDefault) != hipSuccess)
                    return -1;
            }
            p->hp_gone = p->hp_gone_set[p->out_sel];
        }
        if (scan_restore(p) != 0) return -1;
        if (scan_dense(p, p->fl_mag + (size_t)redo_from * p->P.n, p->fl_frames - redo_from, true) != 0) return -1;
        if (scan_export(p) != 0) return -1;
        IRDM_HIP_CHECK(hipStreamSynchronize(p->stream));
        p->host_hist_idx = hdr[0];
        p->host_primed = hdr[1];
        n_gone = (int)counters[0];
        if (counters[1] || n_gone > p->gone_cap) {
            fprintf(stderr, "irdm_hip: detector capacity exceeded (%d bursts in one chunk, cap %d)\n", n_gone, p->gone_cap);
            return -1;
        }
    }
    if (n_gone > 0) memcpy(p->h_gone.data(), p->hp_gone, sizeof(GoneBurst) * n_gone);
    tq1 = now_us(); p->host_us[8] += tq1 - tq0; tq0 = tq1;          // [8] the burst records' round trip
    float ms = 0;
    // the scan proper (band passes, the sparse kernel, or the dense one when it ran instead)
    p->last_ms[1] = hipEventElapsedTime(&ms, p->ev_sk[0], p->ev_sk[1]) == hipSuccess ? ms : -1.0f;
    p->last_frames = p->fl_frames;
    p->d_mag_last = p->fl_mag;
    // burst_detect.c:739: counted where the detector hands the burst over -- here, when the scan settles -- so that the
    // count is complete for a state export while the bursts' per-burst chains are still in flight
    p->tagged += (uint64_t)n_gone;
    *n_gone_out = n_gone;
    return 0;
}

// pipeline_depth 1: a detector scan still in flight is completed and its bursts become the pending list
int settle(irdm_pipeline *p)
{
    if (!p->fl_active) return 0;
    pipeline_enter(p);
    const uint64_t c1 = p->fl_c1;
    int n_gone = 0;
    if (scan_finish(p, &n_gone) != 0) return -1;
    p->pend_gone.assign(p->h_gone.begin(), p->h_gone.begin() + n_gone);
    p->has_pending = true;
    p->pend_c1 = c1;
    p->pend_no = p->fl_no;
    return 0;
}

// control-plane calls (state export / import, probes, stage-level entry points): the detector settled and every stream
// idle -- the pipeline's streams are non-blocking, a null-stream copy orders against none of them
int quiesce(irdm_pipeline *p)
{
    pipeline_enter(p);
    if (settle(p) != 0) return -1;
    IRDM_HIP_CHECK(hipDeviceSynchronize());
    return 0;
}

// pipeline_depth >= 1: the finished bursts of the previously scanned chunk (pend_gone) go through the per-burst
// stages on the next batch context, reading the history ring only; nothing waits here.  A chunk with more bursts than
// burst_cap is worked off synchronously, batch by batch, except for its last batch.
int deferred_enqueue(irdm_pipeline *p)
{
    if (!p->has_pending) return 0;
    BatchCtx &b = p->bc[p->pend_no % p->n_bc];
    b.chunk_no = p->pend_no;
    const SampleSource src = make_source(p, nullptr, 0, p->pend_c1);
    const int n = (int)p->pend_gone.size();
    int base = 0;
    // the chain reads the ring: it must hold the chunk these bursts come from (ev_ring: a seeded history)
    IRDM_HIP_CHECK(hipStreamWaitEvent(b.stream, p->ev_ring, 0));
    IRDM_HIP_CHECK(hipStreamWaitEvent(b.stream, p->fs[p->pend_no % kFeedSlots].ev_copy, 0));
    // ... and K1 of the newest chunk goes first (k1_first 1), or K1 and its ring copy (2): a detector scan waits for
    // it, and K1 next to the decimator took 1.0-1.6 ms instead of 0.24 ms
    if (p->begin_no > 0) {
        const irdm_pipeline::FeedSlot &newest = p->fs[(p->begin_no - 1) % kFeedSlots];
        if (p->k1_first >= 2) IRDM_HIP_CHECK(hipStreamWaitEvent(b.stream, newest.ev_copy, 0));
        else if (p->k1_first == 1) IRDM_HIP_CHECK(hipStreamWaitEvent(b.stream, newest.ev_k1, 0));
    }
    while (n - base > p->burst_cap) {
        if (process_bursts(p, b, src, p->pend_gone.data() + base, p->burst_cap) != 0) return -1;
        base += p->burst_cap;
    }
    if (bursts_enqueue(p, b, src, p->pend_gone.data() + base, n - base) != 0) return -1;
    p->has_pending = false;
    return 0;
}

// wait for the context's batch (if any) and emit its records; returns the number of bursts emitted, -1 on error
int deferred_finish(irdm_pipeline *p, BatchCtx &b)
{
    return bursts_finish(p, b);
}

}  // namespace irdmh
