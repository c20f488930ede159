// create.cpp -- a context's construction and destruction (irdm_create / irdm_destroy): derived geometry (burst_detect.c:174-323),
// host-side designs (host_design.cpp), device buffers, streams, batch contexts; the small getters.
#include "pipeline.hpp"

namespace irdmh {

void pipeline_free(irdm_pipeline *p)
{
    if (!p) return;
    void *ptrs[] = { p->d_window, p->d_hist, p->d_sum, p->d_mag, p->d_tw, p->d_tw4096, p->d_tw2048,
                     p->d_dl_fft, p->d_ul_fft, p->d_rot_incr, p->d_state, p->d_gone,
                     p->d_cand_a, p->d_cand_b, p->d_ring, p->d_stage, p->d_in_taps, p->d_noise_taps,
                     p->d_start_taps, p->d_rrc_taps, p->d_cfo_window, p->d_work, p->d_tiles, p->d_dec,
                     p->d_lpf, p->d_rrc_ws, p->d_frames, p->d_demod_ws, p->d_probe, p->d_demod, p->d_decoded, p->d_syn_ra,
                     p->d_syn_hdr, p->d_nbits, p->d_ida, p->d_syn_da, p->d_syn_l1, p->d_syn_l2, p->d_syn_l3, p->d_dirs,
                     p->d_fir_off, p->d_mag2, p->d_mag3, p->k1_pre[1], p->k1_pre[2], p->k1_counts[1], p->k1_counts[2], p->k1_entries[1], p->k1_entries[2],
                     p->k1_pre[0] != p->d_pre ? p->k1_pre[0] : nullptr, p->k1_counts[0] != p->d_counts ? p->k1_counts[0] : nullptr,
                     p->k1_entries[0] != p->d_entries ? p->k1_entries[0] : nullptr, p->d_counts, p->d_entries, p->d_goff, p->d_compact, p->d_pre, p->d_sum_bak, p->d_hist_bak, p->d_state_bak,
                     p->d_status, p->d_mc_ops, p->d_mc_done, p->d_band, p->d_smin, p->d_kclk, p->d_state_spec };
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    for (int s = 0; s < 2; s++) {
        if (p->h_pin_set[s]) (void)hipHostFree(p->h_pin_set[s]);
        if (p->hp_gone_set[s]) (void)hipHostFree(p->hp_gone_set[s]);
        if (p->ev_end_set[s]) (void)hipEventDestroy(p->ev_end_set[s]);
        for (auto &e : p->ev_sk_set[s])
            if (e) (void)hipEventDestroy(e);
    }
    if (p->cfo_thread.joinable()) {
        {
            std::lock_guard<std::mutex> lk(p->cfo_mu);
            p->cfo_quit = true;
        }
        p->cfo_cv.notify_one();
        p->cfo_thread.join();
    }
    for (int i = 0; i < kMaxBc; i++) {
        BatchCtx &b = p->bc[i];
        if (b.ev_cfo) (void)hipEventDestroy(b.ev_cfo);
        if (b.ev_rot) (void)hipEventDestroy(b.ev_rot);
        for (auto &e : b.ev)
            if (e) (void)hipEventDestroy(e);
        if (b.hp_flag) (void)hipHostFree(b.hp_flag);
        if (b.hp_rot_new) (void)hipHostFree(b.hp_rot_new);
        if (b.d_rot_new) (void)hipFree(b.d_rot_new);
        if (b.hp_work) (void)hipHostFree(b.hp_work);
        if (b.hp_tiles) (void)hipHostFree(b.hp_tiles);
        if (b.hp_demod) (void)hipHostFree(b.hp_demod);
        if (b.hp_packed) (void)hipHostFree(b.hp_packed);
        if (b.owns_buffers) {
            void *own[] = { b.d_work, b.d_tiles, b.d_dec, b.d_lpf, b.d_rrc_ws, b.d_frames, b.d_demod_ws, b.d_demod,
                            b.d_decoded, b.d_ida };
            for (void *q : own)
                if (q) (void)hipFree(q);
            if (b.stream) (void)hipStreamDestroy(b.stream);
        }
    }
    if (p->ev_ring) (void)hipEventDestroy(p->ev_ring);
    for (auto &f : p->fs) {
        if (f.ev_start) (void)hipEventDestroy(f.ev_start);
        if (f.ev_k1) (void)hipEventDestroy(f.ev_k1);
        if (f.ev_copy) (void)hipEventDestroy(f.ev_copy);
    }
    if (p->hp_gate) (void)hipHostFree(p->hp_gate);
    if (p->d_rot_table) (void)hipFree(p->d_rot_table);
    for (float2 *q : p->rot_retired) (void)hipFree(q);
    for (float2 *q : p->scratch_retired) (void)hipFree(q);
    for (void *q : p->tiles_retired) (void)hipFree(q);
    for (void *q : p->tiles_host_retired) (void)hipHostFree(q);
    if (p->d_rot_slot) (void)hipFree(p->d_rot_slot);
    if (p->stream_rot_pre) { (void)hipStreamSynchronize(p->stream_rot_pre); (void)hipStreamDestroy(p->stream_rot_pre); }
    if (p->ev_rot_pre) (void)hipEventDestroy(p->ev_rot_pre);
    if (p->d_rot_pre_news) (void)hipFree(p->d_rot_pre_news);
    if (p->stream_spec) (void)hipStreamDestroy(p->stream_spec);
    if (p->ev_sums1) (void)hipEventDestroy(p->ev_sums1);
    if (p->ev_spec_done) (void)hipEventDestroy(p->ev_spec_done);
    if (p->d_band_spec) (void)hipFree(p->d_band_spec);
    for (auto &e : p->ev)
        if (e) (void)hipEventDestroy(e);
    if (p->sstream && p->sstream != p->stream) (void)hipStreamDestroy(p->sstream);
    if (p->ev_scan_in) (void)hipEventDestroy(p->ev_scan_in);
    if (p->ev_scan_out) (void)hipEventDestroy(p->ev_scan_out);
    if (p->fstream && p->fstream != p->stream) (void)hipStreamDestroy(p->fstream);
    if (p->stream2) (void)hipStreamDestroy(p->stream2);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

extern "C" void irdm_destroy(irdm_pipeline_t *p) { pipeline_free(p); }


// A stream for a per-burst chain.  IRDM_CHAIN_CU_RESERVE=R in the environment (0 = off, the default): a CU mask that keeps
// the chains off R CUs of the device (the last CU of each 32-CU mask word in turn), so that the decimator's resident grid --
// seven 256-register wavefronts per CU for 0.35-0.45 ms per chunk, on every CU it may use -- cannot hold ALL of them: a
// 1024-thread workgroup (the scan's plan passes) needs a CU to itself and otherwise waits until the decimator's launch has
// drained (kernel trace).  A masked stream has the default priority, not the chains' low one.
bool chain_stream_create(irdm_pipeline *, hipStream_t *out, int prio)
{
    // (CU masks that keep the chains off 4-32 CUs -- hipExtStreamCreateWithCUMask -- were measured twice in round 5 and dropped:
    // the decimator's resident grid on a masked stream took 0.61-0.67 ms, 64-66 against 71-72 Gsamples/s, profiles/r5_cu_reserve*.json)
    return hipStreamCreateWithPriority(out, hipStreamNonBlocking, prio) == hipSuccess;
}

extern "C" irdm_pipeline_t *irdm_create(const irdm_config_t *cfg)
{
    if (!cfg || cfg->sample_rate <= 0) return nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        fprintf(stderr, "irdm_hip: no HIP device -- there is no CPU fallback in this library\n");
        return nullptr;
    }
    if (hipSetDevice(cfg->device) != hipSuccess) return nullptr;
    // IRDM_CREATE_DEBUG: where the time and the device memory of a context go (stderr)
    const bool dbg = getenv("IRDM_CREATE_DEBUG") != nullptr;
    size_t mem_free0 = 0, mem_total = 0;
    if (dbg) (void)hipMemGetInfo(&mem_free0, &mem_total);
    auto t_now = [] {
        struct timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
    };
    const double t_create0 = t_now();
    auto mark = [&](const char *what) {
        if (!dbg) return;
        size_t fr = 0, tot = 0;
        (void)hipMemGetInfo(&fr, &tot);
        fprintf(stderr, "irdm_create: %-28s %8.2f ms  %8.1f MB on the device\n", what, t_now() - t_create0,
                ((double)mem_free0 - (double)fr) / 1e6);
    };

    irdm_pipeline *p = new (std::nothrow) irdm_pipeline();
    if (!p) return nullptr;
    p->cfg = *cfg;
    const int fs = cfg->sample_rate;

    // ---- detector constants (burst_detect.c:180-226) ----
    DetParams &P = p->P;
    P.log_n = (int)round(log2(fs / 1000.0));
    P.n = 1 << P.log_n;
    P.pre_len = 2 * P.n;
    P.post_len = (int)(fs * 16e-3);
    const int burst_width_hz = 40000;                               // iridium.h:40
    P.width = burst_width_hz / (fs / P.n);
    P.max_bursts = (int)((fs / (float)burst_width_hz) * 0.8f);
    P.max_len = (int)(fs * 0.09);
    const float tdb = cfg->threshold_db > 0 ? cfg->threshold_db : 16.0f;
    P.threshold = powf(10.0f, tdb / 10.0f) / kHistory / 1.72f;
    if (P.n < kScanThreads || P.n > 16384 || P.max_bursts + P.n / (P.width > 0 ? P.width : 1) + 8 > kMaxActive) {
        fprintf(stderr, "irdm_hip: unsupported sample rate %d (fft_size %d)\n", fs, P.n);
        delete p;
        return nullptr;
    }
    p->feed_block = cfg->feed_block > 0 ? cfg->feed_block : 32768;
    if (p->feed_block % P.n != 0) {
        fprintf(stderr, "irdm_hip: feed_block %d must be a multiple of fft_size %d\n", p->feed_block, P.n);
        delete p;
        return nullptr;
    }
    p->dev_fmt = cfg->format;
    p->bps = p->dev_fmt == 2 ? 8 : (p->dev_fmt == 1 ? 4 : 2);
    p->max_chunk = cfg->max_chunk_samples ? cfg->max_chunk_samples : ((size_t)64 << 20);
    p->max_chunk = (p->max_chunk + p->feed_block - 1) / p->feed_block * p->feed_block;
    p->burst_cap = cfg->max_bursts_per_chunk > 0 ? cfg->max_bursts_per_chunk : 4096;
    p->gone_cap = p->burst_cap;
    p->start_time_ns = cfg->start_time_ns;
    if (p->start_time_ns == 0) {
        struct timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        p->start_time_ns = ts.tv_sec * 1000000000ULL + ts.tv_nsec;
    }

    // reference ring size (burst_detect.c:292-296)
    p->ref_ring = (uint64_t)P.max_len + P.pre_len + P.post_len + (uint64_t)P.n * 4;
    if (p->ref_ring < (uint64_t)2 * fs) p->ref_ring = (uint64_t)2 * fs;
    // longest possible burst window: stop - start < max_len + post_len + N, plus pre_len
    p->l_cap = (size_t)P.max_len + P.post_len + P.pre_len + 2 * (size_t)P.n;
    p->depth = cfg->pipeline_depth > 0 ? std::min(cfg->pipeline_depth, kMaxBc - 1) : 0;
    p->k1_first = 1;
    p->k1_lists = 1;
    p->band_first = 0;
    p->band_auto = kBandFirst;
    p->ring_len = p->ref_ring + p->l_cap + p->feed_block;
    // the per-burst chains in flight read the previous depth+1 chunks while this one and the next (look-ahead) arrive
    if (p->depth) p->ring_len += (size_t)(p->depth + 2 + kLookAhead) * p->max_chunk;
    p->ring_len = (p->ring_len + 15) / 16 * 16;     // 16-sample segments never straddle the wrap
    // whole chunks: a chunk written in place (irdm_ingest_ptr) is contiguous (max_chunk is a multiple of feed_block)
    if (p->depth) p->ring_len = (p->ring_len + p->max_chunk - 1) / p->max_chunk * p->max_chunk;
    p->n_ckpt = (int)(p->l_cap / kRotSeg) + 2;

    // ---- downmix constants (burst_downmix.c:223-373) ----
    p->out_rate = 10 * 25000;
    p->sps = (float)p->out_rate / 25000;
    p->search_depth = p->out_rate;
    p->pre_start = (int)(100 * 1e-6f * p->out_rate);
    p->decim = (int)roundf((float)fs / p->out_rate);
    if (p->decim < 1) p->decim = 1;
    p->dec_stride = (int)(p->l_cap / p->decim) + 8;

    std::vector<float> in_taps = design_lpf(1.0f, 10000000.0f, p->out_rate * 0.4f, p->out_rate * 0.2f);
    std::vector<float> noise_taps = design_lpf(1.0f, (float)p->out_rate, 40000.0f / 2.0f, 40000.0f);
    int box = (int)(p->sps * 2);
    if (box < 3) box = 3;
    std::vector<float> start_taps = design_box(box);
    std::vector<float> rrc = design_rrc(1.0f, (float)p->out_rate, 25000.0f, 0.4f, 51);
    std::vector<float> rc = design_rc((float)p->out_rate, 25000.0f, 0.4f, 51);
    std::vector<float> cfo_window = design_blackman(kCfoN);
    if ((int)in_taps.size() != kFirTaps) {
        delete p;
        return nullptr;
    }
    p->in_ntaps = (int)in_taps.size();
    p->noise_ntaps = (int)noise_taps.size();
    p->start_ntaps = (int)start_taps.size();
    p->rrc_ntaps = (int)rrc.size();
    std::vector<cfloat> dl = design_sync_template(rc, kCorrN, p->sps, false, &p->dl_len);
    std::vector<cfloat> ul = design_sync_template(rc, kCorrN, p->sps, true, &p->ul_len);

    std::vector<float> window = design_blackman(P.n);
    for (int i = 0; i < P.n; i++) window[i] /= 0.42f;               // burst_detect.c:249-250
    std::vector<cfloat> tw = design_twiddles(P.n), tw4096 = design_twiddles(kCfoTotal),
                        tw2048 = design_twiddles(kCorrN);
    std::vector<cfloat> rot_incr = design_rotator_incr(P.n);

    // Streams.  pipeline_depth 0: everything on one stream.  pipeline_depth >= 1: the detector (prefilter + scan) and K1
    // get streams of the highest priority, the per-burst chains (two batch contexts) streams of the lowest: the scan of
    // chunk k gates the per-burst work of chunk k, whereas a chain's result is not needed for two more feeds -- without
    // priorities the scan's small kernels queue up behind the FIR workgroups of two chains (measured: the host waited
    // 1.3 ms per feed for a 0.6 ms scan).  (Round 1 confined a sequential leader scan to CUs of its own with CU masks;
    // the band scan is wide and short, masks would only take CUs away from it.)
    mark("host designs");
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);      // numerically lower = higher priority
    bool ok = true;
    p->stream2 = nullptr;
    if (p->depth) {
        // (K1's stream one level below the scan's: streams of one priority share hardware queues, and with the scans
        // chained the detector's queue is never empty -- K1 of the next chunk sat behind a whole scan, 1.35 -> 1.9 ms)
        int prio_k1 = prio_hi < prio_lo - 1 ? prio_hi + 1 : prio_hi;
        if (const char *e = getenv("IRDM_K1_PRIO")) prio_k1 = atoi(e);
        ok = hipStreamCreateWithPriority(&p->stream, hipStreamNonBlocking, prio_hi) == hipSuccess &&
             hipStreamCreateWithPriority(&p->fstream, hipStreamNonBlocking, prio_k1) == hipSuccess &&
             chain_stream_create(p, &p->stream2, prio_lo);
        p->bstream = p->stream2;
    } else {
        ok = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) == hipSuccess;
        p->bstream = p->stream;
        p->fstream = p->stream;
    }
    p->sstream = p->stream;
    p->bstream_prio = prio_lo;
    p->ev_scan_in = p->ev_scan_out = nullptr;
    p->has_pending = false;
    p->pend_c1 = 0;
    p->h_pin = nullptr;
    for (int s = 0; s < 2; s++) {
        p->h_pin_set[s] = nullptr;
        p->hp_gone_set[s] = nullptr;
        p->ev_end_set[s] = nullptr;
        for (auto &e : p->ev_sk_set[s]) ok = ok && hipEventCreate(&e) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&p->ev_end_set[s], hipEventDisableTiming) == hipSuccess;
        ok = ok && hipHostMalloc(reinterpret_cast<void **>(&p->h_pin_set[s]), sizeof(int) * 128, hipHostMallocDefault) == hipSuccess;
        if (ok) memset(p->h_pin_set[s], 0, sizeof(int) * 128);
    }
    p->out_sel = 0;
    p->chain_pending = false;
    p->settle_clean = true;
    p->h_pin = p->h_pin_set[0];
    p->ev_sk[0] = p->ev_sk_set[0][0];
    p->ev_sk[1] = p->ev_sk_set[0][1];
    p->ev_end = p->ev_end_set[0];
    for (auto &e : p->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
#define UP(dst, vec) ok = ok && ((dst = reinterpret_cast<decltype(dst)>(dev_upload((vec).data(), (vec).size()))) != nullptr)
#define AL(dst, T, count) ok = ok && ((dst = dev_alloc<T>(count)) != nullptr)
    UP(p->d_window, window);
    UP(p->d_tw, tw);
    UP(p->d_tw4096, tw4096);
    UP(p->d_tw2048, tw2048);
    UP(p->d_dl_fft, dl);
    UP(p->d_ul_fft, ul);
    UP(p->d_rot_incr, rot_incr);
    UP(p->d_in_taps, in_taps);
    UP(p->d_noise_taps, noise_taps);
    UP(p->d_start_taps, start_taps);
    UP(p->d_rrc_taps, rrc);
    UP(p->d_cfo_window, cfo_window);
    {
        // byte offset of tap k in the decimator's polyphase LDS tile: slot (k % M, k / M)
        const int row = fir_tile_row(p->decim);
        std::vector<int> off(kFirTaps);
        for (int k = 0; k < kFirTaps; k++) off[k] = ((k % p->decim) * row + k / p->decim) * (int)sizeof(float2);
        UP(p->d_fir_off, off);
    }
    mark("streams, uploads");
    AL(p->d_hist, float, (size_t)kHistory * P.n);
    AL(p->d_sum, float, (size_t)P.n);
    AL(p->d_mag, float, p->max_chunk);
    if (p->depth) AL(p->d_mag2, float, p->max_chunk);
    if (p->depth) AL(p->d_mag3, float, p->max_chunk);
    AL(p->d_state, DetState, 1);
    AL(p->d_gone, GoneBurst, (size_t)p->gone_cap);
    AL(p->d_cand_a, PeakCand, (size_t)P.n);
    AL(p->d_cand_b, PeakCand, (size_t)P.n);
    mark("detector buffers");
    p->d_rot_table = nullptr;
    AL(p->d_work, BurstWork, (size_t)p->burst_cap);
    p->tiles_cap = (size_t)p->burst_cap * 64;
    AL(p->d_tiles, FirTile, (p->tiles_cap + 1) * kFirTileUnits);
    p->cfo_quit = false;
    // decimated and low-passed bursts: rows end to end by their actual length (BurstWork::dec_off), room for 1/16 of
    // burst_cap full-length windows to begin with (4096 bursts of 7 ms at 10 MHz; 0.12 GB per context instead of 1.8),
    // grown by doubling when a batch needs more (bursts_enqueue)
    p->scratch_init = std::max<size_t>((size_t)p->burst_cap * p->dec_stride / 16, (size_t)4 * p->dec_stride);
    AL(p->d_dec, float2, p->scratch_init);
    AL(p->d_lpf, float2, lpf_alloc(p->scratch_init));
    AL(p->d_rrc_ws, float2, (size_t)p->burst_cap * kFrameNeed);
    AL(p->d_frames, float2, (size_t)p->burst_cap * kMaxFrameSamples);
    AL(p->d_demod_ws, float2, (size_t)p->burst_cap * 2 * kMaxSymbols);
    AL(p->d_demod, DemodOut, (size_t)p->burst_cap);
    AL(p->d_decoded, DecodedOut, (size_t)p->burst_cap);
    AL(p->d_nbits, int, (size_t)p->burst_cap);
    {
        // build_syndrome_table (frame_decode.c:95-129): remainder of every 1- and 2-bit error pattern
        auto rem = [](unsigned poly, unsigned v) {
            if (!v) return 0u;
            const int pb = 32 - __builtin_clz(poly);
            for (int i = 31; i >= pb - 1; i--)
                if (v & (1u << i)) v ^= poly << (i - pb + 1);
            return v;
        };
        auto build = [&](unsigned poly, int nbits, int max_err, int size) {
            std::vector<int2> t((size_t)size, make_int2(-1, 0));
            for (int b1 = 0; b1 < nbits; b1++) {
                const unsigned v = 1u << b1, r = rem(poly, v);
                if (r < (unsigned)size) t[r] = make_int2(1, (int)v);
            }
            if (max_err >= 2)
                for (int b1 = 0; b1 < nbits; b1++)
                    for (int b2 = b1 + 1; b2 < nbits; b2++) {
                        const unsigned v = (1u << b1) | (1u << b2), r = rem(poly, v);
                        if (r < (unsigned)size && t[r].x < 0) t[r] = make_int2(2, (int)v);
                    }
            return t;
        };
        std::vector<int2> ra = build(1207u, 31, 2, 1024), hdr = build(29u, 7, 1, 16);
        UP(p->d_syn_ra, ra);
        UP(p->d_syn_hdr, hdr);
        // ida_decode_init (ida_decode.c:96-102)
        std::vector<int2> da = build(3545u, 31, 2, 2048), l1 = build(29u, 7, 1, 16), l2 = build(465u, 14, 1, 256),
                          l3 = build(41u, 26, 2, 32);
        UP(p->d_syn_da, da);
        UP(p->d_syn_l1, l1);
        UP(p->d_syn_l2, l2);
        UP(p->d_syn_l3, l3);
        AL(p->d_ida, IdaOut, (size_t)p->burst_cap);
        AL(p->d_dirs, int, (size_t)p->burst_cap);
    }
    AL(p->d_probe, float2, p->l_cap);
    {
        const size_t max_frames = p->max_chunk / P.n;
        AL(p->d_counts, unsigned, max_frames);
        AL(p->d_entries, ListEntry, max_frames * (size_t)std::max(kListCap, band_list_cap(P.n)));
        AL(p->d_goff, unsigned, max_frames + 1);
        AL(p->d_compact, ListEntry, max_frames * kListCap);
        AL(p->d_pre, float, (size_t)P.n);
        // pipeline_depth 0: one chunk at a time, K1's lists share the prefilter pass's buffers; otherwise a set per feed
        // slot (the prefilter pass of a fallback may run while K1 of a later chunk writes its lists)
        p->k1_pre[0] = p->d_pre;
        p->k1_counts[0] = p->d_counts;
        p->k1_entries[0] = p->d_entries;
        for (int i = 0; i < kFeedSlots && p->depth; i++) {
            AL(p->k1_pre[i], float, (size_t)P.n);
            AL(p->k1_counts[i], unsigned, max_frames);
            AL(p->k1_entries[i], ListEntry, max_frames * (size_t)std::max(kListCap, band_list_cap(P.n)));
        }
        AL(p->d_sum_bak, float, (size_t)P.n);
        AL(p->d_hist_bak, float, (size_t)kHistory * P.n);
        AL(p->d_state_bak, DetState, 1);
        AL(p->d_status, int, 64);
        p->mc_ops_cap = (int)(4 * max_frames + 64);
        AL(p->d_mc_ops, unsigned long long, (size_t)p->mc_ops_cap);
        AL(p->d_mc_done, unsigned, 32 * 16);
    }
    mark("per-burst scratch, lists");
    ok = ok && hipHostMalloc(reinterpret_cast<void **>(&p->hp_gate), 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
         hipHostGetDevicePointer(reinterpret_cast<void **>(&p->hp_gate_dev), p->hp_gate, 0) == hipSuccess;
    if (ok) memset(p->hp_gate, 0, 64);
    AL(p->d_kclk, unsigned long long, (size_t)(6 + kMaxBc - 3) * kKClkWords);
    if (ok) {
        std::vector<unsigned long long> init((size_t)(6 + kMaxBc - 3) * kKClkWords, 0ull);
        for (int r = 0; r < 6 + kMaxBc - 3; r++)
            for (int i = 0; i < 64; i++) init[(size_t)r * kKClkWords + i] = ~0ull;
        ok = hipMemcpy(p->d_kclk, init.data(), sizeof(unsigned long long) * init.size(), hipMemcpyHostToDevice) == hipSuccess;
    }
    p->band_ok = band_scan_supported(P, nullptr, 1, 0) != 0;
    if (p->band_ok) {
        AL(p->d_smin, float, (size_t)P.n);
        if (ok) ok = hipMalloc(&p->d_band, band_work_bytes(P.n, p->max_chunk)) == hipSuccess;
        if (ok) band_work_carve(&p->band, p->d_band, P.n, p->max_chunk);
        if (ok) ok = hipMemset(p->band.bar, 0, 256) == hipSuccess;         // (no scan has committed, no launch is void)
        if (ok && p->depth) {
            // the speculation passes' workspace (one snapshot row; 0.27 GB at 64 Mi-sample chunks, most of it the sparse
            // relative-magnitude plane), carried-burst list, stream and events
            const size_t sb = band_work_bytes(P.n, p->max_chunk, true);
            ok = hipMalloc(&p->d_band_spec, sb) == hipSuccess;
            if (ok) band_work_carve(&p->band_spec, p->d_band_spec, P.n, p->max_chunk, true);
            // (control words, record counts, the void marker: zero; the planes are written before they are read)
            if (ok) ok = hipMemset(p->band_spec.ctl, 0, sizeof(BandCtl)) == hipSuccess && hipMemset(p->band_spec.bar, 0, 256) == hipSuccess &&
                         hipMemset(p->band_spec.rec_count, 0, 4 * 64) == hipSuccess && hipMemset(p->band_spec.flags, 0, 256) == hipSuccess;
            AL(p->d_state_spec, DetState, 1);
            if (ok) ok = hipMemset(p->d_state_spec, 0, sizeof(DetState)) == hipSuccess;
            ok = ok && hipStreamCreateWithPriority(&p->stream_spec, hipStreamNonBlocking, prio_hi) == hipSuccess &&
                 hipEventCreateWithFlags(&p->ev_sums1, hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&p->ev_spec_done, hipEventDisableTiming) == hipSuccess;
        }
    }
    mark("band scan workspace");
    if (ok) ok = hipMalloc(&p->d_ring, p->ring_len * p->bps) == hipSuccess;
    mark("history ring");
    // batch contexts: [0] aliases the pipeline's per-burst scratch and runs on bstream; [1] (pipeline_depth >= 1) has
    // scratch and a stream of its own
    p->n_bc = p->depth ? std::min(p->depth + 1, kMaxBc) : 1;
    for (int i = 0; i < p->n_bc && ok; i++) {
        BatchCtx &b = p->bc[i];
        b.owner = p;
        b.n = 0;
        b.cfo_seq = 0;
        b.owns_buffers = i > 0;
        b.tiles_cap = p->tiles_cap;
        b.dec_cap = p->scratch_init;
        if (i == 0) {
            b.stream = p->bstream;
            b.d_work = p->d_work; b.d_tiles = p->d_tiles; b.d_dec = p->d_dec; b.d_lpf = p->d_lpf;
            b.d_rrc_ws = p->d_rrc_ws; b.d_frames = p->d_frames; b.d_demod_ws = p->d_demod_ws; b.d_demod = p->d_demod;
            b.d_decoded = p->d_decoded; b.d_ida = p->d_ida;
        } else {
            ok = ok && chain_stream_create(p, &b.stream, p->bstream_prio);
            AL(b.d_work, BurstWork, (size_t)p->burst_cap);
            AL(b.d_tiles, FirTile, (b.tiles_cap + 1) * kFirTileUnits);
            AL(b.d_dec, float2, p->scratch_init);
            AL(b.d_lpf, float2, lpf_alloc(p->scratch_init));
            AL(b.d_rrc_ws, float2, (size_t)p->burst_cap * kFrameNeed);
            AL(b.d_frames, float2, (size_t)p->burst_cap * kMaxFrameSamples);
            AL(b.d_demod_ws, float2, (size_t)p->burst_cap * 2 * kMaxSymbols);
            AL(b.d_demod, DemodOut, (size_t)p->burst_cap);
            AL(b.d_decoded, DecodedOut, (size_t)p->burst_cap);
            AL(b.d_ida, IdaOut, (size_t)p->burst_cap);
        }
        ok = ok && hipHostMalloc(reinterpret_cast<void **>(&b.hp_work), sizeof(BurstWork) * (size_t)p->burst_cap,
                                 hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
             hipHostGetDevicePointer(reinterpret_cast<void **>(&b.hp_work_dev), b.hp_work, 0) == hipSuccess;
        ok = ok && hipHostMalloc(reinterpret_cast<void **>(&b.hp_tiles), sizeof(FirTile) * b.tiles_cap, hipHostMallocDefault) == hipSuccess;
        ok = ok && hipHostMalloc(reinterpret_cast<void **>(&b.hp_demod), sizeof(DemodOut) * (size_t)p->burst_cap, hipHostMallocDefault) == hipSuccess;
        ok = ok && hipHostMalloc(reinterpret_cast<void **>(&b.hp_packed), sizeof(DemodPacked) * (size_t)p->burst_cap, hipHostMallocDefault) == hipSuccess;
        ok = ok && hipHostMalloc(reinterpret_cast<void **>(&b.hp_flag), 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
             hipHostGetDevicePointer(reinterpret_cast<void **>(&b.hp_flag_dev), b.hp_flag, 0) == hipSuccess;
        if (ok) memset(b.hp_flag, 0, 64);
        ok = ok && hipHostMalloc(reinterpret_cast<void **>(&b.hp_rot_new), sizeof(int4) * (size_t)p->burst_cap, hipHostMallocMapped) == hipSuccess &&
             hipHostGetDevicePointer(reinterpret_cast<void **>(&b.hp_rot_new_dev), b.hp_rot_new, 0) == hipSuccess;
        AL(b.d_rot_new, int4, (size_t)p->burst_cap);
        ok = ok && hipEventCreateWithFlags(&b.ev_cfo, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&b.ev_rot, hipEventDisableTiming) == hipSuccess;
        for (auto &e : b.ev) ok = ok && hipEventCreate(&e) == hipSuccess;
        b.h_cfreq.assign((size_t)p->burst_cap, 0.0);
    }
    p->hp_gone_cap = p->gone_cap;
    for (int s = 0; s < 2; s++)
        ok = ok && hipHostMalloc(reinterpret_cast<void **>(&p->hp_gone_set[s]), sizeof(GoneBurst) * (size_t)p->hp_gone_cap, hipHostMallocDefault) == hipSuccess;
    p->hp_gone = p->hp_gone_set[0];
    ok = ok && hipEventCreateWithFlags(&p->ev_ring, hipEventDisableTiming) == hipSuccess;
    for (auto &f : p->fs)
        ok = ok && hipEventCreate(&f.ev_start) == hipSuccess && hipEventCreate(&f.ev_k1) == hipSuccess &&
             hipEventCreateWithFlags(&f.ev_copy, hipEventDisableTiming) == hipSuccess;
    p->chunk_no = 0;
#undef UP
#undef AL
    if (!ok) {
        fprintf(stderr, "irdm_hip: device allocation failed\n");
        pipeline_free(p);
        return nullptr;
    }
    mark("batch contexts");
    ok = hipMemset(p->d_hist, 0, sizeof(float) * (size_t)kHistory * P.n) == hipSuccess &&
         hipMemset(p->d_sum, 0, sizeof(float) * P.n) == hipSuccess &&
         hipMemset(p->d_state, 0, sizeof(DetState)) == hipSuccess &&
         hipMemset(p->d_ring, 0, p->ring_len * p->bps) == hipSuccess;
    ok = ok && hipDeviceSynchronize() == hipSuccess;
    mark("memsets, sync");
    // rotator checkpoints: an arena of blocks, handed out as bursts need their centre bin's row (rot_rows_prepare); nothing
    // is built here
    p->rot_runs = (p->n_ckpt + kRotRun - 1) / kRotRun;
    p->rot_blocks_cap = std::min(P.n, 1024) * p->rot_runs;        // (what 1024 whole rows would take: 0.57 GB at 10 MHz)
    p->rot_blocks_used = 0;
    p->rot_rows_used = 0;
    p->rot_len_h.assign((size_t)P.n, 0);
    p->rot_want.assign((size_t)P.n, 0);
    ok = ok && (p->d_rot_table = dev_alloc<float2>((size_t)p->rot_blocks_cap * kRotRun)) != nullptr;
    ok = ok && (p->d_rot_slot = dev_alloc<int>((size_t)P.n * p->rot_runs)) != nullptr;
    ok = ok && hipMemset(p->d_rot_slot, 0xff, sizeof(int) * (size_t)P.n * p->rot_runs) == hipSuccess;
    p->rot_build_ctx.assign((size_t)P.n, -1);
    p->rot_build_gen.assign((size_t)P.n, 0);
#ifndef IRDM_HIP_EMULATED
    // (a throughput context: the rows of all bins in the background, now.  Not fatal: without the memory for it the rows
    // come on demand.  The CPU emulation runs the launch to completion on enqueue -- seconds per context -- and asks for it
    // by option where it tests it.)
    if (ok && p->depth >= 1 && !getenv("IRDM_NO_ROT_PREBUILD")) (void)rot_prebuild(p);
#endif
    mark("rotator row pool");
    if (!ok) {
        fprintf(stderr, "irdm_hip: device initialisation failed\n");
        pipeline_free(p);
        return nullptr;
    }
    p->h_gone.resize(p->gone_cap);
    p->total_samples = p->begun_samples = 0;
    p->begin_no = p->end_no = 0;
    p->tagged = 0;
    p->stream_closed = false;
    p->last_frames = 0;
    p->last_chunk = nullptr;
    p->keep_frame_samples = 0;
    p->chunk_marks = 0;
    p->scan_mode = 0;
    p->mc_updaters = 7;         // + the leader = 8 workgroups: the scan stream's 8 reserved CUs (pipeline_depth 1)
    {
        hipDeviceProp_t prop;
        const bool have = hipGetDeviceProperties(&prop, cfg->device) == hipSuccess;
        p->mc_auto = have && prop.multiProcessorCount >= 64;
        if (p->scan_cus == 0) p->scan_cus = have ? prop.multiProcessorCount : 1;
    }
    p->stat_fast_chunks = p->stat_fallbacks = p->stat_dense_frames = 0;
    p->stat_band_chunks = p->stat_band_rounds = p->stat_band_retries = p->stat_band_aborts = 0;
    p->last_band_flags = 0;
    p->fl_mode = 0;
    p->fl_done = 0;
    p->host_primed = 0;
    p->host_hist_idx = 0;
    // Does libm_port.hpp reproduce THIS host's cexpf?  (Every float of the step's range is compared by
    // tools/check_sincosf.cpp; this is the same question asked of the running process on a probe set: 2^18 offsets
    // across [-0.26, 0.26], the neighbourhoods of the quadrant boundaries, zero and the tiny-argument branch.)
    p->dev_cfo = true;
    {
        auto same = [](float off) {
            const cfloat h = fine_rotator_incr(off);
            const float phase_inc = -2.0f * (float)M_PI * off;
            float re, im;
            if (libm_cexpf_i<true>(phase_inc, &re, &im) != 0) return false;
            const float hr = h.real(), hi = h.imag();
            return memcmp(&re, &hr, 4) == 0 && memcmp(&im, &hi, 4) == 0;
        };
        bool all = true;
        for (int i = 0; i < (1 << 18) && all; i++) all = same(-0.26f + 0.52f * (float)i / (float)(1 << 18));
        const float edges[] = { 0.0f, -0.0f, 1e-45f, -1e-45f, 1e-39f, 3e-5f, -3e-5f, 0.125f, -0.125f, 0.25f, -0.25f, 0.2499999f, 0.1250001f };
        for (float e : edges) all = all && same(e);
        if (!all) {
            fprintf(stderr, "irdm_hip: this host's cexpf differs from the restated glibc routine: the fine-CFO step stays on the host\n");
            p->dev_cfo = false;
        }
    }
    p->dev_cfo_ok = p->dev_cfo;
    mark("libm self-check");
    p->cfo_thread = std::thread(cfo_helper_main, p);
    mark("done");
    return p;
}

// libm_port.hpp as the device executes it, for arbitrary arguments: re + i im = cexpf(i x[k]) (NaN outside |x| < 120)
extern "C" int irdm_sincosf_probe(int device, const float *x, size_t n, float *re, float *im)
{
    if (!x || !re || !im) return -1;
    if (!n) return 0;
    IRDM_HIP_CHECK(hipSetDevice(device));
    float *d = nullptr;
    IRDM_HIP_CHECK(hipMalloc(&d, 3 * n * sizeof(float)));
    int rc = hipMemcpy(d, x, n * sizeof(float), hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
    if (!rc) rc = launch_sincosf_probe(d, n, d + n, d + 2 * n, nullptr);
    if (!rc) rc = hipMemcpy(re, d + n, n * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
    if (!rc) rc = hipMemcpy(im, d + 2 * n, n * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
    (void)hipFree(d);
    return rc;
}

extern "C" uint64_t irdm_tagged_bursts(const irdm_pipeline_t *p) { return p ? p->tagged : 0; }
extern "C" size_t irdm_max_chunk_samples(const irdm_pipeline_t *p) { return p ? p->max_chunk : 0; }
extern "C" size_t irdm_bytes_per_sample(const irdm_pipeline_t *p) { return p ? p->bps : 0; }
// Samples a context that takes over a stream at some position must be given from in front of it (irdm_seed_history*): the
// reference's ring -- stale-slot reads reach one ring length back (burst_detect.c:292-296, :401-422) -- plus the longest
// burst window.
extern "C" size_t irdm_required_overlap(const irdm_pipeline_t *p)
{
    if (!p) return 0;
    return (size_t)(p->ref_ring + (uint64_t)p->P.max_len + (uint64_t)p->P.post_len + (uint64_t)p->P.pre_len + 2 * (uint64_t)p->P.n);
}
// K1 and the history-ring copy of every chunk handed over so far have read their input: the caller may write the buffers
// again.  (Host wait on the ingest stream; the detector and the per-burst chains are not waited for.)
extern "C" int irdm_wait_ingest(irdm_pipeline_t *p)
{
    if (!p) return -1;
    pipeline_enter(p);
    IRDM_HIP_CHECK(hipStreamSynchronize(p->fstream));
    return 0;
}
extern "C" uint64_t irdm_sample_count(const irdm_pipeline_t *p) { return p ? p->total_samples : 0; }
extern "C" int irdm_fft_size(const irdm_pipeline_t *p) { return p ? p->P.n : -1; }
extern "C" uint64_t irdm_start_time_ns(const irdm_pipeline_t *p) { return p ? p->start_time_ns : 0; }

// The stream a stage-level call (irdm_downmix_burst) belongs to: its centre frequency and the wall-clock time of its
// sample 0 (burst_data_t carries both per burst, burst_detect.h:40-48).  Host fields only; not while a feed is begun.
extern "C" int irdm_set_stream_origin(irdm_pipeline_t *p, double center_frequency, uint64_t start_time_ns)
{
    if (!p || p->begin_no != p->end_no) return -1;
    p->cfg.center_frequency = center_frequency;
    if (start_time_ns) p->start_time_ns = start_time_ns;
    return 0;
}

}  // namespace irdmh
