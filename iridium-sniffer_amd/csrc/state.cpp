// state.cpp -- detector-state export / import and history seeding (time-chunk sharding), and the stage-level batch
// entry points (irdm_downmix_burst, irdm_qpsk_demod_batch, irdm_frame_decode_batch, irdm_ida_decode_batch).
#include "pipeline.hpp"

namespace irdmh {

// ---- detector-state hand-off for time-chunk sharding ----
struct StateHeader {
    uint64_t magic, n, hist, total_samples, tagged, start_time_ns;
    int32_t host_primed, host_hist_idx;
};

extern "C" size_t irdm_state_bytes(const irdm_pipeline_t *p)
{
    if (!p) return 0;
    return sizeof(StateHeader) + sizeof(DetState) + sizeof(float) * (size_t)p->P.n * (1 + kHistory);
}

extern "C" long long irdm_export_state(irdm_pipeline_t *p, void *buf, size_t cap)
{
    if (!p || !buf || cap < irdm_state_bytes(p) || quiesce(p) != 0) return -1;
    pipeline_enter(p);
    char *o = static_cast<char *>(buf);
    StateHeader h = { 0x4952444d53544154ull, (uint64_t)p->P.n, (uint64_t)kHistory, p->total_samples, p->tagged,
                      p->start_time_ns, p->host_primed, p->host_hist_idx };
    memcpy(o, &h, sizeof(h));
    o += sizeof(h);
    IRDM_HIP_CHECK(hipMemcpy(o, p->d_state, sizeof(DetState), hipMemcpyDeviceToHost));
    o += sizeof(DetState);
    IRDM_HIP_CHECK(hipMemcpy(o, p->d_sum, sizeof(float) * p->P.n, hipMemcpyDeviceToHost));
    o += sizeof(float) * p->P.n;
    IRDM_HIP_CHECK(hipMemcpy(o, p->d_hist, sizeof(float) * (size_t)kHistory * p->P.n, hipMemcpyDeviceToHost));
    return (long long)irdm_state_bytes(p);
}

// A detector state may only be replaced while no scan that ran on the OLD state is ahead of the caller: with two chunks begun
// ahead, or with the next chunk's scan already chained behind the one in flight (scan_chain_early), that scan has read -- or
// committed against -- the state the import is about to overwrite, and irdm_feed_end would book its results as the
// imported stream's.  The time-sharded callers begin one chunk ahead at most and import before its irdm_feed_end.
static inline bool import_allowed(const irdm_pipeline *p) { return !p->chain_pending && p->begin_no <= p->end_no + 1; }

extern "C" int irdm_import_state(irdm_pipeline_t *p, const void *buf, size_t n)
{
    if (!p || !buf || n < irdm_state_bytes(p) || !import_allowed(p) || quiesce(p) != 0) return -1;
    pipeline_enter(p);
    const char *i = static_cast<const char *>(buf);
    StateHeader h;
    memcpy(&h, i, sizeof(h));
    if (h.magic != 0x4952444d53544154ull || h.n != (uint64_t)p->P.n || h.hist != (uint64_t)kHistory) return -1;
    i += sizeof(h);
    IRDM_HIP_CHECK(hipMemcpy(p->d_state, i, sizeof(DetState), hipMemcpyHostToDevice));
    i += sizeof(DetState);
    IRDM_HIP_CHECK(hipMemcpy(p->d_sum, i, sizeof(float) * p->P.n, hipMemcpyHostToDevice));
    i += sizeof(float) * p->P.n;
    IRDM_HIP_CHECK(hipMemcpy(p->d_hist, i, sizeof(float) * (size_t)kHistory * p->P.n, hipMemcpyHostToDevice));
    if (p->begin_no == p->end_no) p->total_samples = p->begun_samples = h.total_samples;     // (a feed already begun has fixed its own position)
    p->tagged = h.tagged;
    p->start_time_ns = h.start_time_ns;
    p->host_primed = h.host_primed;
    p->host_hist_idx = h.host_hist_idx;
    return 0;
}

// The same blob in DEVICE memory (e.g. a torch tensor that RCCL sends to the next rank): no host bounce of the 16-32 MiB
// history, and no device-wide synchronisation -- only the detector has to have settled; K1 / the ring copy of the next
// chunk and the per-burst chains in flight do not touch the detector state.
extern "C" long long irdm_export_state_device(irdm_pipeline_t *p, void *d_buf, size_t cap)
{
    if (!p || !d_buf || cap < irdm_state_bytes(p)) return -1;
    pipeline_enter(p);
    if (settle(p) != 0 || hist_fence(p) != 0) return -1;
    char *o = static_cast<char *>(d_buf);
    const StateHeader h = { 0x4952444d53544154ull, (uint64_t)p->P.n, (uint64_t)kHistory, p->total_samples, p->tagged,
                            p->start_time_ns, p->host_primed, p->host_hist_idx };
    IRDM_HIP_CHECK(hipMemcpyAsync(o, &h, sizeof(h), hipMemcpyHostToDevice, p->stream));
    o += sizeof(h);
    IRDM_HIP_CHECK(hipMemcpyAsync(o, p->d_state, sizeof(DetState), hipMemcpyDeviceToDevice, p->stream));
    o += sizeof(DetState);
    IRDM_HIP_CHECK(hipMemcpyAsync(o, p->d_sum, sizeof(float) * p->P.n, hipMemcpyDeviceToDevice, p->stream));
    o += sizeof(float) * p->P.n;
    IRDM_HIP_CHECK(hipMemcpyAsync(o, p->d_hist, sizeof(float) * (size_t)kHistory * p->P.n, hipMemcpyDeviceToDevice, p->stream));
    IRDM_HIP_CHECK(hipStreamSynchronize(p->stream));
    return (long long)irdm_state_bytes(p);
}

extern "C" int irdm_import_state_device(irdm_pipeline_t *p, const void *d_buf, size_t n)
{
    if (!p || !d_buf || n < irdm_state_bytes(p) || !import_allowed(p)) return -1;
    pipeline_enter(p);
    if (settle(p) != 0 || hist_fence(p) != 0) return -1;
    const char *i = static_cast<const char *>(d_buf);
    StateHeader h;
    IRDM_HIP_CHECK(hipMemcpyAsync(&h, i, sizeof(h), hipMemcpyDeviceToHost, p->stream));
    IRDM_HIP_CHECK(hipStreamSynchronize(p->stream));
    if (h.magic != 0x4952444d53544154ull || h.n != (uint64_t)p->P.n || h.hist != (uint64_t)kHistory) return -1;
    i += sizeof(h);
    IRDM_HIP_CHECK(hipMemcpyAsync(p->d_state, i, sizeof(DetState), hipMemcpyDeviceToDevice, p->stream));
    i += sizeof(DetState);
    IRDM_HIP_CHECK(hipMemcpyAsync(p->d_sum, i, sizeof(float) * p->P.n, hipMemcpyDeviceToDevice, p->stream));
    i += sizeof(float) * p->P.n;
    IRDM_HIP_CHECK(hipMemcpyAsync(p->d_hist, i, sizeof(float) * (size_t)kHistory * p->P.n, hipMemcpyDeviceToDevice, p->stream));
    IRDM_HIP_CHECK(hipStreamSynchronize(p->stream));
    if (p->begin_no == p->end_no) p->total_samples = p->begun_samples = h.total_samples;     // (a feed already begun has fixed its own position)
    p->tagged = h.tagged;
    p->start_time_ns = h.start_time_ns;
    p->host_primed = h.host_primed;
    p->host_hist_idx = h.host_hist_idx;
    return 0;
}

// ---- the same hand-off in two parts, so that the 16-32 MiB history can FOLLOW the detector's head ----
// head = header + DetState + sums (65 KB at 12 MHz): everything round 0 of the band scan reads.  The history (512 x N
// floats) is first read by round 1's sums pass: a rank takes the head, enqueues its scan and imports the history when it
// arrives; the scan waits for it on the device (launch_band_scan's gate).
extern "C" size_t irdm_state_head_bytes(const irdm_pipeline_t *p)
{
    if (!p) return 0;
    return sizeof(StateHeader) + sizeof(DetState) + sizeof(float) * (size_t)p->P.n;
}

extern "C" int irdm_import_state_head_device(irdm_pipeline_t *p, const void *d_buf, size_t n)
{
    if (!p || !d_buf || n < irdm_state_head_bytes(p) || !import_allowed(p)) return -1;
    pipeline_enter(p);
    if (settle(p) != 0) return -1;
    const char *i = static_cast<const char *>(d_buf);
    StateHeader h;
    IRDM_HIP_CHECK(hipMemcpyAsync(&h, i, sizeof(h), hipMemcpyDeviceToHost, p->stream));
    IRDM_HIP_CHECK(hipStreamSynchronize(p->stream));
    if (h.magic != 0x4952444d53544154ull || h.n != (uint64_t)p->P.n || h.hist != (uint64_t)kHistory) return -1;
    i += sizeof(h);
    IRDM_HIP_CHECK(hipMemcpyAsync(p->d_state, i, sizeof(DetState), hipMemcpyDeviceToDevice, p->stream));
    i += sizeof(DetState);
    IRDM_HIP_CHECK(hipMemcpyAsync(p->d_sum, i, sizeof(float) * p->P.n, hipMemcpyDeviceToDevice, p->stream));
    IRDM_HIP_CHECK(hipStreamSynchronize(p->stream));
    if (p->begin_no == p->end_no) p->total_samples = p->begun_samples = h.total_samples;
    p->tagged = h.tagged;
    p->start_time_ns = h.start_time_ns;
    p->host_primed = h.host_primed;
    p->host_hist_idx = h.host_hist_idx;
    return 0;
}

// Call between irdm_import_state_head_device and irdm_feed_end.  d_hist_buf: device memory (irdm_state_bytes() -
// irdm_state_head_bytes() bytes) the history WILL be in.  1 = the scan that irdm_feed_end enqueues waits on the device --
// behind its round 0 -- until irdm_import_state_history_device(p, d_hist_buf, n) says the history has arrived there, and
// copies it into the context itself; the caller MUST make that call before anything settles the scan
// (irdm_export_state_device, irdm_flush, the next irdm_feed_end).  0 = this scan cannot wait (pipeline_depth 0, a
// detector that is not primed, a scan other than the band scan): import the history before irdm_feed_end.
extern "C" int irdm_expect_history(irdm_pipeline_t *p, const void *d_hist_buf)
{
    if (!p || !d_hist_buf || !p->hp_gate_dev || !p->depth || !p->host_primed || scan_pick(p) != 2 || p->begin_no == p->end_no)
        return 0;
    // (test hook band_first 1: the launch ends with round 0's verdict, before the pass the gate sits in front of -- the
    // continuation would run on a history that was never copied in)
    if (p->band_first == 1) return 0;
    p->gate_seq++;
    p->gate_src = d_hist_buf;
    p->gate_armed = true;
    return 1;
}

// d_hist_buf: the history part of the blob (behind irdm_state_head_bytes()), device memory, complete and visible to the
// device when this is called
extern "C" int irdm_import_state_history_device(irdm_pipeline_t *p, const void *d_hist_buf, size_t n)
{
    const size_t bytes = p ? sizeof(float) * (size_t)kHistory * p->P.n : 0;
    if (!p || !d_hist_buf || n < bytes) return -1;
    pipeline_enter(p);
    if (p->gate_open_pending) {
        // a scan is waiting for it: the word it polls is written by the HOST (no GPU work of ours that could queue up
        // behind the waiting kernel); the scan's stream copies the history in and goes on
        if (d_hist_buf != p->gate_src) return -1;
        __atomic_store_n(&p->hp_gate[0], p->gate_seq, __ATOMIC_RELEASE);
        return 0;
    }
    p->gate_armed = false;               // (announced, but the scan was never enqueued: the ordinary import)
    if (settle(p) != 0 || hist_fence(p) != 0) return -1;
    IRDM_HIP_CHECK(hipMemcpyAsync(p->d_hist, d_hist_buf, bytes, hipMemcpyDeviceToDevice, p->stream));
    IRDM_HIP_CHECK(hipStreamSynchronize(p->stream));
    return 0;
}

// the preceding samples from DEVICE memory (a chunk overlap received from the previous rank)
extern "C" int irdm_seed_history_device(irdm_pipeline_t *p, const void *d_iq, size_t n_samples, uint64_t abs_start)
{
    if (!p || (!d_iq && n_samples) || n_samples > abs_start || p->begin_no != p->end_no) return -1;
    pipeline_enter(p);
    if (n_samples > p->ring_len) {
        d_iq = static_cast<const char *>(d_iq) + (n_samples - p->ring_len) * p->bps;
        n_samples = p->ring_len;
    }
    // behind whatever the ring stream still has to do -- and behind the decimators in flight that still read the slots
    // (irdm_advance leaves the previous chunk's chain running); the per-burst chains wait for ev_ring before they read the ring
    if (ring_guard(p, abs_start - n_samples, abs_start, p->fstream) != 0) return -1;
    if (ring_update(p, d_iq, abs_start - n_samples, abs_start, p->fstream) != 0) return -1;
    IRDM_HIP_CHECK(hipEventRecord(p->ev_ring, p->fstream));
    IRDM_HIP_CHECK(hipStreamSynchronize(p->fstream));
    p->total_samples = p->begun_samples = abs_start;
    return 0;
}

extern "C" int irdm_seed_history(irdm_pipeline_t *p, const void *h_iq, size_t n_samples, uint64_t abs_start)
{
    if (!p || (!h_iq && n_samples) || n_samples > abs_start || p->begin_no != p->end_no) return -1;
    if (quiesce(p) != 0) return -1;
    if (n_samples > p->ring_len) {       // only the most recent ring_len samples can matter
        h_iq = static_cast<const char *>(h_iq) + (n_samples - p->ring_len) * p->bps;
        n_samples = p->ring_len;
    }
    const char *src = static_cast<const char *>(h_iq);
    uint64_t a0 = abs_start - n_samples;
    size_t done = 0;
    while (done < n_samples) {
        const uint64_t pos = (a0 + done) % p->ring_len;
        const size_t run = std::min<size_t>(n_samples - done, p->ring_len - pos);
        IRDM_HIP_CHECK(hipMemcpy(static_cast<char *>(p->d_ring) + pos * p->bps, src + done * p->bps, run * p->bps,
                                 hipMemcpyHostToDevice));
        done += run;
    }
    p->total_samples = p->begun_samples = abs_start;
    return 0;
}

extern "C" int irdm_downmix_burst(irdm_pipeline_t *p, const irdm_burst_t *info, const float *samples,
                                  size_t num_samples, irdm_frame_info_t *frame, float *frame_samples)
{
    if (!p || !info || !samples || !frame || p->dev_fmt != 2) return -1;
    if (num_samples > p->l_cap) return -1;
    if (quiesce(p) != 0) return -1;
    IRDM_HIP_CHECK(hipMemcpy(p->d_probe, samples, num_samples * sizeof(float2), hipMemcpyHostToDevice));
    // the burst window is presented as a "chunk" that starts at info->start
    SampleSource src = make_source(p, p->d_probe, info->start, info->start + num_samples);
    GoneBurst g;
    memset(&g, 0, sizeof(g));
    g.id = info->id;
    g.start = info->start;
    g.stop = info->start + num_samples - (uint64_t)p->P.pre_len;     // num_samples = stop + pre_len - start
    g.last_active = info->last_active;
    g.center_bin = info->center_bin;
    g.peak_rel = info->peak_rel;
    g.base_sum = info->base_sum;
    // keep the result queues of the stream untouched: run on private queues
    std::deque<irdm_burst_t> qb; std::deque<irdm_frame_info_t> qf; std::deque<std::vector<float>> qs;
    std::deque<irdm_demod_t> qd;
    qb.swap(p->q_bursts); qf.swap(p->q_frames); qs.swap(p->q_frame_samples); qd.swap(p->q_demods);
    const int keep = p->keep_frame_samples, dec = p->decode_frames, dec_ida = p->decode_ida, det = p->detect_only;
    p->decode_frames = 0;
    p->decode_ida = 0;
    p->detect_only = 0;          // a stage-B call on a detect-only context still runs stage B
    const int marks = p->chunk_marks;
    p->chunk_marks = 0;          // (the records go to private queues: no chunk mark for them)
    const uint64_t tagged = p->tagged;
    std::vector<irdm_burst_t> last; last.swap(p->last_bursts);
    p->keep_frame_samples = 1;
    const int rc = process_bursts(p, p->bc[0], src, &g, 1);
    int ret = -1;
    if (rc == 0 && !p->q_frames.empty()) {
        *frame = p->q_frames.front();
        frame->magnitude = info->magnitude;
        frame->noise = info->noise;
        if (frame->drop_reason == 0 && frame_samples && !p->q_frame_samples.empty())
            memcpy(frame_samples, p->q_frame_samples.front().data(), p->q_frame_samples.front().size() * sizeof(float));
        ret = frame->drop_reason == 0 ? 1 : 0;
    }
    p->q_bursts.swap(qb); p->q_frames.swap(qf); p->q_frame_samples.swap(qs); p->q_demods.swap(qd);
    p->keep_frame_samples = keep;
    p->chunk_marks = marks;
    p->detect_only = det;
    p->decode_frames = dec;
    p->decode_ida = dec_ida;
    p->tagged = tagged;
    p->last_bursts.swap(last);
    return ret;
}

extern "C" int irdm_qpsk_demod_batch(irdm_pipeline_t *p, const float *samples, const int *num_samples,
                                     const int *direction, int n, irdm_demod_t *out)
{
    if (!p || !samples || !num_samples || !direction || !out || n < 0) return -1;
    pipeline_enter(p);
    for (int base = 0; base < n; base += p->burst_cap) {
        const int nb = std::min(p->burst_cap, n - base);
        p->h_work.assign(nb, BurstWork());
        for (int i = 0; i < nb; i++) {
            if (num_samples[base + i] < 0 || num_samples[base + i] > kMaxFrameSamples) return -1;
            p->h_work[i].num_samples = num_samples[base + i];
            p->h_work[i].direction = direction[base + i];
            p->h_work[i].drop_reason = 0;
        }
        IRDM_HIP_CHECK(hipMemcpyAsync(p->d_work, p->h_work.data(), sizeof(BurstWork) * nb, hipMemcpyHostToDevice, p->stream));
        IRDM_HIP_CHECK(hipMemcpyAsync(p->d_frames, samples + (size_t)base * 2 * kMaxFrameSamples,
                                      sizeof(float2) * (size_t)nb * kMaxFrameSamples, hipMemcpyHostToDevice, p->stream));
        if (launch_demod(p->d_work, nb, p->d_frames, p->cfg.use_gardner, p->sps, p->d_demod_ws, p->d_demod,
                         p->stream) != 0)
            return -1;
        p->h_demod.resize(nb);
        IRDM_HIP_CHECK(hipMemcpyAsync(p->h_demod.data(), p->d_demod, sizeof(DemodOut) * nb, hipMemcpyDeviceToHost, p->stream));
        IRDM_HIP_CHECK(hipStreamSynchronize(p->stream));
        for (int i = 0; i < nb; i++) {
            const DemodOut &d = p->h_demod[i];
            irdm_demod_t &o = out[base + i];
            memset(&o, 0, sizeof(o));
            o.ok = d.ok;
            if (!d.ok) continue;
            o.direction = d.direction;
            o.confidence = d.confidence;
            o.level = d.level;
            o.n_symbols = d.n_symbols;
            o.n_payload_symbols = d.n_symbols - 12;
            o.n_bits = 2 * d.n_symbols;
            o.total_phase = d.total_phase;
            memcpy(o.bits, d.bits, sizeof(o.bits));
            memcpy(o.llr, d.llr, sizeof(o.llr));
        }
    }
    return 0;
}

extern "C" int irdm_poll_decoded(irdm_pipeline_t *p, irdm_decoded_t *out, int max)
{
    if (!p || !out || max < 0) return -1;
    return drain(p->q_decoded, out, max);
}

extern "C" int irdm_frame_decode_batch(irdm_pipeline_t *p, const irdm_demod_t *in, int n, int use_llr, irdm_decoded_t *out)
{
    if (!p || !in || !out || n < 0) return -1;
    pipeline_enter(p);
    std::vector<int> nbits;
    for (int base = 0; base < n; base += p->burst_cap) {
        const int nb = std::min(p->burst_cap, n - base);
        p->h_demod.assign(nb, DemodOut());
        nbits.assign(nb, 0);
        for (int i = 0; i < nb; i++) {
            const irdm_demod_t &f = in[base + i];
            if (f.n_bits < 0 || f.n_bits > kMaxBits) return -1;
            DemodOut &d = p->h_demod[i];
            d.ok = 1;
            d.n_symbols = f.n_bits / 2;
            memcpy(d.bits, f.bits, sizeof(d.bits));
            memcpy(d.llr, f.llr, sizeof(d.llr));
            nbits[i] = f.n_bits;
        }
        IRDM_HIP_CHECK(hipMemcpyAsync(p->d_demod, p->h_demod.data(), sizeof(DemodOut) * nb, hipMemcpyHostToDevice, p->stream));
        IRDM_HIP_CHECK(hipMemcpyAsync(p->d_nbits, nbits.data(), sizeof(int) * nb, hipMemcpyHostToDevice, p->stream));
        if (launch_frame_decode(p->d_demod, nb, p->d_syn_ra, p->d_syn_hdr, use_llr ? 1 : 0, p->d_nbits, p->d_decoded,
                                p->stream) != 0)
            return -1;
        p->h_decoded.resize(nb);
        IRDM_HIP_CHECK(hipMemcpyAsync(p->h_decoded.data(), p->d_decoded, sizeof(DecodedOut) * nb, hipMemcpyDeviceToHost, p->stream));
        IRDM_HIP_CHECK(hipStreamSynchronize(p->stream));
        for (int i = 0; i < nb; i++)
            out[base + i] = finish_decoded(p->h_decoded[i], in[base + i].id, in[base + i].timestamp,
                                           in[base + i].center_frequency);
    }
    return 0;
}

extern "C" int irdm_poll_ida(irdm_pipeline_t *p, irdm_ida_t *out, int max)
{
    if (!p || !out || max < 0) return -1;
    return drain(p->q_ida, out, max);
}

extern "C" int irdm_ida_decode_batch(irdm_pipeline_t *p, const irdm_demod_t *in, int n, int use_llr, irdm_ida_t *out)
{
    if (!p || !in || !out || n < 0) return -1;
    pipeline_enter(p);
    std::vector<int> nbits, dirs;
    for (int base = 0; base < n; base += p->burst_cap) {
        const int nb = std::min(p->burst_cap, n - base);
        p->h_demod.assign(nb, DemodOut());
        nbits.assign(nb, 0);
        dirs.assign(nb, 0);
        for (int i = 0; i < nb; i++) {
            const irdm_demod_t &f = in[base + i];
            if (f.n_bits < 0 || f.n_bits > kMaxBits) return -1;
            DemodOut &d = p->h_demod[i];
            d.ok = 1;
            d.n_symbols = f.n_bits / 2;
            memcpy(d.bits, f.bits, sizeof(d.bits));
            memcpy(d.llr, f.llr, sizeof(d.llr));
            nbits[i] = f.n_bits;
            dirs[i] = f.direction;
        }
        IRDM_HIP_CHECK(hipMemcpyAsync(p->d_demod, p->h_demod.data(), sizeof(DemodOut) * nb, hipMemcpyHostToDevice, p->stream));
        IRDM_HIP_CHECK(hipMemcpyAsync(p->d_nbits, nbits.data(), sizeof(int) * nb, hipMemcpyHostToDevice, p->stream));
        IRDM_HIP_CHECK(hipMemcpyAsync(p->d_dirs, dirs.data(), sizeof(int) * nb, hipMemcpyHostToDevice, p->stream));
        if (launch_ida_decode(p->d_demod, nb, p->d_syn_da, p->d_syn_l1, p->d_syn_l2, p->d_syn_l3, use_llr ? 1 : 0,
                              p->d_nbits, p->d_dirs, p->d_ida, p->stream) != 0)
            return -1;
        p->h_ida.resize(nb);
        IRDM_HIP_CHECK(hipMemcpyAsync(p->h_ida.data(), p->d_ida, sizeof(IdaOut) * nb, hipMemcpyDeviceToHost, p->stream));
        IRDM_HIP_CHECK(hipStreamSynchronize(p->stream));
        for (int i = 0; i < nb; i++) out[base + i] = finish_ida(p->h_ida[i], in[base + i]);
    }
    return 0;
}

}  // namespace irdmh
