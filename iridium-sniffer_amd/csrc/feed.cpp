// feed.cpp -- the feeding calls (irdm_feed_begin / _end / _device / _host, irdm_flush, irdm_advance), the polls, and the
// buffer helpers for hosts without HIP headers.
#include "pipeline.hpp"

namespace irdmh {

extern "C" int irdm_flush(irdm_pipeline_t *p)
{
    if (!p) return -1;
    if (!p->depth) return 0;
    if (p->begin_no != p->end_no) return -1;        // a chunk handed over with irdm_feed_begin is still pending
    pipeline_enter(p);
    if (settle(p) != 0) return -1;
    int emitted = 0;
    // records leave in chunk order: the batches in flight, oldest first, then the pending bursts of the last scan
    for (;;) {
        BatchCtx *oldest = nullptr;
        for (int i = 0; i < p->n_bc; i++)
            if (p->bc[i].n > 0 && (!oldest || p->bc[i].chunk_no < oldest->chunk_no)) oldest = &p->bc[i];
        if (!oldest) break;
        const int e = deferred_finish(p, *oldest);
        if (e < 0) return -1;
        emitted += e;
    }
    if (p->has_pending) {
        BatchCtx &b = p->bc[p->pend_no % p->n_bc];
        if (deferred_enqueue(p) != 0) return -1;
        const int e = deferred_finish(p, b);
        if (e < 0) return -1;
        emitted += e;
    }
    return emitted;
}

// irdm_flush without the waiting: the detector scan in flight is settled and its bursts' per-burst chain ENQUEUED; records
// of batches that have finished come out, nothing else is waited for (a context that is still busy with an older batch is
// waited for only if the new chain needs that very context).  What a rank of a time-sharded stream calls at the end of a
// super-step: its chain then runs beside the next super-step's scatter, K1 and scan (sharding.TimeShard).  Returns the
// number of bursts whose records were emitted, -1 on error.
extern "C" int irdm_advance(irdm_pipeline_t *p)
{
    if (!p) return -1;
    if (!p->depth) return 0;
    if (p->begin_no != p->end_no) return -1;
    pipeline_enter(p);
    if (settle(p) != 0) return -1;
    int emitted = 0;
    auto oldest_of = [&]() -> BatchCtx * {
        BatchCtx *o = nullptr;
        for (int i = 0; i < p->n_bc; i++)
            if (p->bc[i].n > 0 && (!o || p->bc[i].chunk_no < o->chunk_no)) o = &p->bc[i];
        return o;
    };
    for (BatchCtx *o; (o = oldest_of()) != nullptr && (p->detect_only || hipStreamQuery(o->stream) == hipSuccess);) {
        const int e = deferred_finish(p, *o);
        if (e < 0) return -1;
        emitted += e;
    }
    if (p->has_pending) {
        BatchCtx &b = p->bc[p->pend_no % p->n_bc];
        while (b.n > 0) {            // (records leave in chunk order: everything older than the batch in the way goes first)
            const int e = deferred_finish(p, *oldest_of());
            if (e < 0) return -1;
            emitted += e;
        }
        if (deferred_enqueue(p) != 0) return -1;
    }
    return emitted;
}

// A feed in two halves.  irdm_feed_begin: everything that does not depend on the detector state -- K1 of the chunk and
// (pipeline_depth >= 1) its copy into the history ring.  irdm_feed_end: the detector scan and the per-burst work.  A
// time-sharded rank calls them around the arrival of the previous rank's state (sharding.py); irdm_feed_device is the
// two back to back.
extern "C" int irdm_feed_begin(irdm_pipeline_t *p, const void *d_iq, size_t n_samples, void *stream_v)
{
    if (!p || (!d_iq && n_samples)) return -1;
    if (p->begin_no - p->end_no > (p->depth ? kLookAhead : 0u)) return -1;      // two chunks of look-ahead, pipeline_depth >= 1 only
    if (p->stream_closed) {
        fprintf(stderr, "irdm_hip: stream already ended by a chunk that was not a multiple of feed_block\n");
        return -1;
    }
    if (n_samples > p->max_chunk) {
        fprintf(stderr, "irdm_hip: chunk of %zu samples exceeds max_chunk_samples %zu\n", n_samples, p->max_chunk);
        return -1;
    }
    if (n_samples % p->feed_block != 0) p->stream_closed = true;     // last, ragged chunk of the stream
    pipeline_enter(p);
    // order after the caller's stream (the producer of d_iq)
    hipStream_t caller = static_cast<hipStream_t>(stream_v);
    // stream == NULL: the chunk is already complete in memory, nothing to order against.  (Not the legacy null stream,
    // which would wait for every other stream including the detector scan in flight; and no event on a foreign stream
    // when it is not needed: streams share hardware queues, and an event recorded on a stream that shares one with the
    // detector's sits behind the scan -- measured: K1 of the next chunk then started only after the scan had ended.)
    p->caller_ordered = caller != nullptr;
    if (p->caller_ordered) {
        IRDM_HIP_CHECK(hipEventRecord(p->ev[8], caller));
        if (caller != p->fstream) IRDM_HIP_CHECK(hipStreamWaitEvent(p->fstream, p->ev[8], 0));
    }
    const DetParams &P = p->P;
    const uint64_t c0 = p->begun_samples, c1 = c0 + n_samples;
    const int n_frames = (int)(n_samples / (size_t)P.n);

    // K1 of this chunk.  pipeline_depth 1: on its own stream and into the other magnitude buffer, while the detector
    // scan of the previous chunk may still be running
    irdm_pipeline::FeedSlot &f = p->fs[p->begin_no % kFeedSlots];
    float *const mags[kFeedSlots] = { p->d_mag, p->d_mag2, p->d_mag3 };
    float *mag = p->depth ? mags[p->begin_no % kFeedSlots] : p->d_mag;
    // written in place (irdm_ingest_ptr)?  Then the ring already holds the chunk.
    const uint64_t pos = c0 % p->ring_len;
    const bool in_ring = p->depth && n_samples > 0 && pos + n_samples <= p->ring_len &&
                         d_iq == static_cast<const char *>(p->d_ring) + pos * p->bps;
    IRDM_HIP_CHECK(hipEventRecord(f.ev_start, p->fstream));
    // K1, with the band scan's candidate lists where the scan will want them: the reference levels are the running
    // sums as they are NOW (the previous chunk's scan may still be at work on them -- any levels do, the scan checks the
    // lists against the ones they were built with, scan_band.hip band_sum_kernel); not before the detector is primed
    // (no sums yet: every bin would be listed)
    const int ls = p->depth ? (int)(p->begin_no % kFeedSlots) : 0;       // (pipeline_depth 0: one chunk at a time, one set)
    f.lists = false;
    if (p->k1_lists && p->host_primed && scan_pick(p) == 2 && p->k1_pre[ls] && n_frames > 0) {
        if (launch_prefilter_threshold(p->d_sum, P.threshold, p->k1_pre[ls], P.n, p->fstream) != 0) return -1;
        const int rc = launch_fft_mag_lists(P.log_n, p->dev_fmt, d_iq, p->d_window, p->d_tw, mag, n_frames, p->k1_pre[ls],
                                            p->k1_counts[ls], p->k1_entries[ls], band_list_cap(P.n), p->fstream,
                                            p->kclk_rec(3 + ls % 3), p->fir_order);
        if (rc < 0) return -1;
        f.lists = rc == 0;
    }
    if (!f.lists && launch_fft_mag(P.log_n, p->dev_fmt, d_iq, p->d_window, p->d_tw, mag, n_frames, p->fstream,
                                   p->kclk_rec(3 + ls % 3), p->fir_order) != 0)
        return -1;
    IRDM_HIP_CHECK(hipEventRecord(f.ev_k1, p->fstream));
    if (n_frames > 0 && launch_kclk_fold(p->kclk_rec(3 + ls % 3), p->fstream) != 0) return -1;   // (behind the event the scan waits for)
    // this chunk into the history ring, behind K1 on its stream (the ring keeps the chunks the per-burst chains in
    // flight still read: the copy never overwrites them)
    if (p->depth && !in_ring && (ring_guard(p, c0, c1, p->fstream) != 0 || ring_update(p, d_iq, c0, c1, p->fstream) != 0)) return -1;
    IRDM_HIP_CHECK(hipEventRecord(f.ev_copy, p->fstream));
    f.iq = d_iq;
    f.c0 = c0;
    f.c1 = c1;
    f.mag = mag;
    f.frames = n_frames;
    f.in_ring = in_ring;
    p->begun_samples = c1;
    p->begin_no++;
    return 0;
}

extern "C" int irdm_feed_end(irdm_pipeline_t *p)
{
    if (!p || p->begin_no == p->end_no) return -1;
    pipeline_enter(p);
    irdm_pipeline::FeedSlot &f = p->fs[p->end_no % kFeedSlots];
    const void *d_iq = f.iq;
    const uint64_t c0 = f.c0, c1 = f.c1;
    float *mag = f.mag;
    const int n_frames = f.frames;
    float ms = 0;

    int emitted = 0;
    if (!p->depth) {
        int n_gone = 0;
        p->fl_feed = &f;
        if (scan_launch(p, mag, n_frames, c1) != 0 || scan_finish(p, &n_gone) != 0) return -1;
        p->last_bursts.clear();
        p->last_chunk = d_iq;
        p->last_chunk_start = c0;
        p->last_chunk_end = c1;
        const SampleSource src = make_source(p, d_iq, c0, c1);
        // record the stage events once so an empty chunk has valid timings
        for (int i = 0; i < 4; i++) IRDM_HIP_CHECK(hipEventRecord(p->bc[0].ev[i], p->bc[0].stream));
        if (process_bursts(p, p->bc[0], src, p->h_gone.data(), n_gone) != 0) return -1;
        if (ring_update(p, d_iq, c0, c1, p->stream) != 0) return -1;
        IRDM_HIP_CHECK(hipEventRecord(p->ev[7], p->stream));
        IRDM_HIP_CHECK(hipStreamSynchronize(p->stream));
        emitted = n_gone;
    } else {
        auto now_us = [] {
            struct timespec ts;
            clock_gettime(CLOCK_MONOTONIC, &ts);
            return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
        };
        double t0 = now_us(), t1;
#define IRDM_HOST_PHASE(i) do { t1 = now_us(); p->host_us[i] += t1 - t0; t0 = t1; } while (0)
        IRDM_HOST_PHASE(0);
        p->last_bursts.clear();
        // 0. if the oldest chain has already finished, its records are built NOW, while the previous chunk's detector
        //    scan is still running (0.3 ms of host work that would otherwise follow the wait for the scan)
        BatchCtx &oldest = p->bc[p->chunk_no % p->n_bc];
        bool finished_early = false;
        if (oldest.n > 0 && !p->detect_only && p->fl_active && hipStreamQuery(oldest.stream) == hipSuccess) {
            emitted = deferred_finish(p, oldest);
            if (emitted < 0) return -1;
            finished_early = true;
        }
        IRDM_HOST_PHASE(4);
        // 1. this chunk's band scan goes behind the previous chunk's (scan_chain_try), then the previous chunk's is
        //    settled and its bursts collected
        // (already chained at the end of the previous feed -- scan_chain_early, below -- unless that could not be done)
        if (!(p->chain_pending && p->chain_no == p->chunk_no) && scan_chain_try(p, f, p->chunk_no) != 0) return -1;
        if (settle(p) != 0) return -1;
        if (p->chain_pending && !p->settle_clean) {
            // the scan in front did not commit on its own: the chained launch has declined itself (nothing written)
            IRDM_HIP_CHECK(hipStreamSynchronize(p->stream));
            p->chain_pending = false;
            p->stat_chain_undone++;
        }
        IRDM_HOST_PHASE(1);
        // 2. this chunk's detector (needs K1's output) goes first: the next chunk's scan can only start when this one
        //    has ended, so every microsecond before its launch is added to the period
        IRDM_HIP_CHECK(hipStreamWaitEvent(p->stream, f.ev_k1, 0));
        p->fl_feed = &f;
        if (scan_launch(p, mag, n_frames, c1) != 0) return -1;
        IRDM_HOST_PHASE(3);
        // 3. the per-burst stages of the chunk just settled: enqueued on the idle batch context, nothing waits.  (The
        //    context of the chunk before that is still at work: its tail overlaps this one's FIR.)
        if (deferred_enqueue(p) != 0) return -1;
        IRDM_HOST_PHASE(2);
        // 3b. the next chunk, if its feed has begun (look-ahead): its round 0 as a speculation pass beside this chunk's scan
        if (p->begin_no > p->end_no + 1 && p->fl_mode == 2 && p->fl_band_ran &&
            spec_enqueue(p, p->fs[(p->end_no + 1) % kFeedSlots], p->chunk_no + 1) != 0)
            return -1;
        // 3c. ... and its scan, chained behind this chunk's, NOW: what follows -- the wait for the oldest chain, the records,
        //     the caller's polls and its next irdm_feed_begin -- took 0.4-0.8 ms, during which the scan's stream ran dry
        //     after every scan: the period was (that host time + a scan) / 2, not a scan.
        //     The same launch the next irdm_feed_end would make first thing -- it finds it done.
        if (p->begin_no > p->end_no + 1 && !p->chain_pending &&
            scan_chain_try(p, p->fs[(p->end_no + 1) % kFeedSlots], p->chunk_no + 1) != 0)
            return -1;
        // 4. results of the older batch: its context is the one the NEXT chunk's bursts will use
        if (!finished_early) {
            emitted = deferred_finish(p, oldest);
            if (emitted < 0) return -1;
        }
        IRDM_HOST_PHASE(4);
        // 5. the caller may overwrite d_iq once we return: K1 and the ring copy are done with it.  (A chunk written in
        //    place stays where it is; K1 is waited for only so that its time can be read.)
        IRDM_HIP_CHECK(hipEventSynchronize(f.in_ring ? f.ev_k1 : f.ev_copy));
        IRDM_HOST_PHASE(5);
#undef IRDM_HOST_PHASE
    }
    p->chunk_no++;
    p->end_no++;
    p->total_samples = c1;

    // [0] K1, [5] the whole call on the detector side; [1] is set by scan_finish, [2..4] by bursts_finish
    p->last_ms[0] = hipEventElapsedTime(&ms, f.ev_start, f.ev_k1) == hipSuccess ? ms : -1.0f;
    p->last_ms[5] = !p->depth && hipEventElapsedTime(&ms, f.ev_start, p->ev[7]) == hipSuccess ? ms : -1.0f;
    return emitted;
}

extern "C" int irdm_feed_device(irdm_pipeline_t *p, const void *d_iq, size_t n_samples, void *stream_v)
{
    if (irdm_feed_begin(p, d_iq, n_samples, stream_v) != 0) return -1;
    return irdm_feed_end(p);
}

// Where the producer of the next chunk (an H2D copy, a conversion kernel) may write it so that it needs no copy into
// the history ring: the ring slot of the absolute sample index the next irdm_feed_begin starts at.  NULL when the
// context keeps no ring copy (pipeline_depth 0) or the chunk would straddle the end of the ring (it cannot when every
// chunk but the last has max_chunk_samples: the ring is a whole number of them).  The slot is the producer's until it
// hands it over with irdm_feed_begin(p, ptr, n, stream); it is overwritten ring_len samples later.
extern "C" void *irdm_ingest_ptr(irdm_pipeline_t *p, size_t n_samples)
{
    if (!p || !p->depth || n_samples == 0 || n_samples > p->max_chunk) return nullptr;
    const uint64_t pos = p->begun_samples % p->ring_len;
    if (pos + n_samples > p->ring_len) return nullptr;
    return static_cast<char *>(p->d_ring) + pos * p->bps;
}

extern "C" void *irdm_ring_ptr(irdm_pipeline_t *p, uint64_t *len_samples)
{
    if (!p) return nullptr;
    if (len_samples) *len_samples = p->ring_len;
    return p->d_ring;
}

// Pinned host memory for irdm_feed_host callers that have no HIP headers (the C99 host): H2D copies from pinned
// memory are asynchronous DMA at PCIe rate; from pageable memory they are staged and block the host.
extern "C" void *irdm_host_alloc(size_t bytes)
{
    void *q = nullptr;
    if (hipHostMalloc(&q, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return q;
}

extern "C" void irdm_host_free(void *q)
{
    if (q) (void)hipHostFree(q);
}

extern "C" void *irdm_device_alloc(int device, size_t bytes)
{
    void *q = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(&q, bytes) != hipSuccess) return nullptr;
    return q;
}

extern "C" void irdm_device_free(void *q)
{
    if (q) (void)hipFree(q);
}

extern "C" int irdm_device_upload(void *dptr, const void *host, size_t bytes)
{
    if (!dptr || (!host && bytes)) return -1;
    IRDM_HIP_CHECK(hipMemcpy(dptr, host, bytes, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int irdm_device_copy(void *dst, const void *src, size_t bytes)
{
    if ((!dst || !src) && bytes) return -1;
    IRDM_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice));
    return 0;
}

extern "C" int irdm_feed_host(irdm_pipeline_t *p, const void *h_iq, size_t n_samples)
{
    if (!p || (!h_iq && n_samples)) return -1;
    if (n_samples > p->max_chunk) return -1;
    pipeline_enter(p);
    // throughput mode: the H2D copy lands in the chunk's slot of the history ring and the chunk is fed in place (no staging
    // buffer, no device-to-device copy behind K1)
    if (void *slot = irdm_ingest_ptr(p, n_samples)) {
        IRDM_HIP_CHECK(hipMemcpyAsync(slot, h_iq, n_samples * p->bps, hipMemcpyHostToDevice, p->fstream));
        return irdm_feed_device(p, slot, n_samples, p->fstream);
    }
    if (!p->d_stage) {
        if (hipMalloc(&p->d_stage, p->max_chunk * p->bps) != hipSuccess) return -1;
    }
    // Raw bytes in the configured format (the ci16 narrowing of main.c:245-246 happens in the kernels' load stage).
    // The copy goes on K1's stream, never the null stream: with pipeline_depth 1 the previous chunk's detector scan is
    // still running and must not be waited for.  irdm_feed_device returns only after K1 and the history-ring copy of
    // its chunk are done, so one staging buffer is enough.
    IRDM_HIP_CHECK(hipMemcpyAsync(p->d_stage, h_iq, n_samples * p->bps, hipMemcpyHostToDevice, p->fstream));
    return irdm_feed_device(p, p->d_stage, n_samples, p->fstream);
}


extern "C" int irdm_poll_chunk_marks(irdm_pipeline_t *p, irdm_chunk_mark_t *out, int max)
{
    if (!p || !out || max < 0) return -1;
    return drain(p->q_marks, out, max);
}

// chunks (in the order fed, counted from 0) below this number have all their records in the queues: nothing of theirs is
// in a scan in flight, a pending burst list or a batch context
extern "C" uint64_t irdm_chunks_complete(const irdm_pipeline_t *p)
{
    if (!p) return 0;
    uint64_t w = p->chunk_no;
    if (p->fl_active) w = std::min<uint64_t>(w, p->fl_no);
    if (p->has_pending) w = std::min<uint64_t>(w, p->pend_no);
    for (int i = 0; i < p->n_bc; i++)
        if (p->bc[i].n > 0) w = std::min<uint64_t>(w, p->bc[i].chunk_no);
    return w;
}

extern "C" int irdm_poll_demods_packed(irdm_pipeline_t *p, irdm_demod_packed_t *out, int max)
{
    if (!p || !out || max < 0) return -1;
    return drain(p->q_packed, out, max);
}

extern "C" int irdm_poll_bursts(irdm_pipeline_t *p, irdm_burst_t *out, int max)
{
    if (!p || !out || max < 0) return -1;
    return drain(p->q_bursts, out, max);
}

extern "C" int irdm_poll_frames(irdm_pipeline_t *p, irdm_frame_info_t *out, float *samples_out, int max)
{
    if (!p || !out || max < 0) return -1;
    int n = 0;
    while (n < max && !p->q_frames.empty()) {
        out[n] = p->q_frames.front();
        p->q_frames.pop_front();
        if (!p->q_frame_samples.empty()) {
            if (samples_out) {
                const std::vector<float> &s = p->q_frame_samples.front();
                memcpy(samples_out + (size_t)n * 2 * IRDM_MAX_FRAME_SAMPLES, s.data(), s.size() * sizeof(float));
            }
            p->q_frame_samples.pop_front();
        }
        n++;
    }
    return n;
}

extern "C" int irdm_poll_demods(irdm_pipeline_t *p, irdm_demod_t *out, int max)
{
    if (!p || !out || max < 0) return -1;
    return drain(p->q_demods, out, max);
}

extern "C" int irdm_last_magnitudes(irdm_pipeline_t *p, float *out, size_t max_frames)
{
    if (!p || !out) return -1;
    if (quiesce(p) != 0) return -1;
    const size_t nf = std::min<size_t>(max_frames, (size_t)p->last_frames);
    if (!nf) return 0;
    IRDM_HIP_CHECK(hipMemcpy(out, p->d_mag_last, nf * p->P.n * sizeof(float), hipMemcpyDeviceToHost));
    return (int)nf;
}

extern "C" int irdm_detector_stats(irdm_pipeline_t *p, irdm_detector_stats_t *out)
{
    if (!p || !out || quiesce(p) != 0) return -1;
    const DetParams &P = p->P;
    std::vector<float> sum((size_t)P.n);
    IRDM_HIP_CHECK(hipMemcpy(sum.data(), p->d_sum, sizeof(float) * (size_t)P.n, hipMemcpyDeviceToHost));
    DetState head;
    IRDM_HIP_CHECK(hipMemcpy(&head, p->d_state, offsetof(DetState, act), hipMemcpyDeviceToHost));
    const int n_act = head.n_act < 0 ? 0 : (head.n_act > kMaxActive ? kMaxActive : head.n_act);
    std::vector<ActiveBurst> act((size_t)n_act);
    if (n_act)
        IRDM_HIP_CHECK(hipMemcpy(act.data(), reinterpret_cast<const char *>(p->d_state) + offsetof(DetState, act),
                                 sizeof(ActiveBurst) * (size_t)n_act, hipMemcpyDeviceToHost));
    out->active_bursts = n_act;
    out->primed = head.primed;
    // burst_detect.c:363-380
    double s = 0;
    for (int i = 0; i < P.n; i++) s += sum[i];
    const float avg = (float)(s / ((double)P.n * kHistory));
    const float bin_width = (float)p->cfg.sample_rate / P.n;
    out->noise_floor_dbfs_hz = (avg > 0 && bin_width > 0) ? 10.0f * log10f(avg / bin_width) : -120.0f;
    // burst_detect.c:572-576: the running maximum of the magnitude a burst is created with
    float peak = p->peak_signal_db;
    for (const ActiveBurst &a : act) {
        const float m = 10.0f * log10f(a.peak_rel * kHistory * 1.72f);
        if (m > peak) peak = m;
    }
    out->peak_signal_db = peak;
    return 0;
}

extern "C" int irdm_baseline_sum(irdm_pipeline_t *p, float *out)
{
    if (!p || !out || quiesce(p) != 0) return -1;
    IRDM_HIP_CHECK(hipMemcpy(out, p->d_sum, p->P.n * sizeof(float), hipMemcpyDeviceToHost));
    return p->P.n;
}

extern "C" int irdm_burst_samples(irdm_pipeline_t *p, int burst_in_chunk, float *out, size_t max_samples)
{
    if (!p || !out || burst_in_chunk < 0 || burst_in_chunk >= (int)p->last_bursts.size() || (!p->depth && !p->last_chunk))
        return -1;
    const irdm_burst_t &r = p->last_bursts[burst_in_chunk];
    const size_t n = std::min<size_t>(std::min<size_t>(max_samples, r.num_samples), p->l_cap);
    // NOTE: valid only until the next feed (the chunk pointer and ring are read again)
    SampleSource src = p->depth ? make_source(p, nullptr, 0, r.avail_end)
                                : make_source(p, p->last_chunk, p->last_chunk_start, p->last_chunk_end);
    // the ring already holds the chunk tail; reading through the chunk pointer is equivalent
    if (launch_gather_burst(src, r.start, r.avail_end, (int)n, p->d_probe, p->stream) != 0) return -1;
    IRDM_HIP_CHECK(hipMemcpyAsync(out, p->d_probe, n * sizeof(float2), hipMemcpyDeviceToHost, p->stream));
    IRDM_HIP_CHECK(hipStreamSynchronize(p->stream));
    return (int)n;
}

}  // namespace irdmh
