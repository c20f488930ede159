// chain.cpp -- the per-burst chain of a chunk (K4 .. K7): where its burst windows live (history ring), the rotator checkpoint
// arena (on demand / prebuilt), the enqueue of the chain's launches, and the records built from what it brings back
// (burst_downmix.c:643-797, qpsk_demod.c:393-535; dB fields and timestamps with the host's libm / integer arithmetic).
#include "pipeline.hpp"

namespace irdmh {

SampleSource make_source(const irdm_pipeline *p, const void *chunk, uint64_t c0, uint64_t c1)
{
    // chunk == nullptr: every sample comes from the history ring (pipeline_depth 1)
    SampleSource s;
    s.chunk = chunk;
    s.chunk_start = chunk ? c0 : ~0ull;
    s.chunk_end = c1;
    s.ring = p->d_ring;
    s.ring_len = p->ring_len;
    s.ref_ring = p->ref_ring;
    s.fmt = p->dev_fmt;
    return s;
}

// copy the chunk's tail into the history ring (absolute index % ring_len)
// Samples [a0, a1) are about to be written into the history ring on stream `st`: behind the decimator of every batch in
// flight that may still read the slots they land in.  Consecutive chunks of a stream never meet a batch in flight (the
// ring is sized for that); a rank of a time-sharded stream jumps `world` chunks ahead per super-step and may (section 6).
int ring_guard(irdm_pipeline *p, uint64_t a0, uint64_t a1, hipStream_t st)
{
    const uint64_t L = p->ring_len;
    if (a1 <= a0 || L == 0) return 0;
    for (int i = 0; i < p->n_bc; i++) {
        const BatchCtx &b = p->bc[i];
        if (b.n <= 0 || b.ring_hi <= b.ring_lo) continue;
        bool hit = a1 - a0 >= L || b.ring_hi - b.ring_lo >= L;
        if (!hit) {
            const uint64_t x0 = a0 % L, y0 = b.ring_lo % L;
            hit = (y0 + L - x0) % L < a1 - a0 || (x0 + L - y0) % L < b.ring_hi - b.ring_lo;
        }
        if (hit) {
            IRDM_HIP_CHECK(hipStreamWaitEvent(st, b.ev[1], 0));
            p->stat_ring_waits++;
        }
    }
    return 0;
}

int ring_update(irdm_pipeline *p, const void *d_iq, uint64_t c0, uint64_t c1, hipStream_t st)
{
    uint64_t a0 = c1 > p->ring_len ? std::max(c0, c1 - p->ring_len) : c0;
    while (a0 < c1) {
        const uint64_t pos = a0 % p->ring_len;
        const uint64_t run = std::min<uint64_t>(c1 - a0, p->ring_len - pos);
        // (hipMemcpyAsync, i.e. the DMA engines, BESIDE the kernels: a copy kernel over the whole chip measured 60.7-61.0 against
        // 64.7-65.4 Gsamples/s for chunks not fed in place, round 5)
        IRDM_HIP_CHECK(hipMemcpyAsync(static_cast<char *>(p->d_ring) + pos * p->bps, static_cast<const char *>(d_iq) + (a0 - c0) * p->bps,
                                      run * p->bps, hipMemcpyDeviceToDevice, st));
        a0 += run;
    }
    return 0;
}

// DecodedOut (device) -> irdm_decoded_t: lat / lon / alt with the host libm, exactly parse_ira's expressions
// (frame_decode.c:336-342); they stay zero when fewer than 63 data bits were assembled (:321-322)
irdm_decoded_t finish_decoded(const DecodedOut &d, uint64_t id, uint64_t timestamp, double frequency)
{
    irdm_decoded_t o;
    memset(&o, 0, sizeof(o));
    o.type = d.type;
    o.sat_id = d.sat_id;
    o.beam_id = d.beam_id;
    o.n_pages = d.n_pages;
    for (int k = 0; k < 3; k++) o.pos_xyz[k] = d.pos_xyz[k];
    for (int k = 0; k < 12; k++) { o.page_tmsi[k] = d.page_tmsi[k]; o.page_msc[k] = d.page_msc[k]; }
    o.timeslot = d.timeslot;
    o.sv_blocking = d.sv_blocking;
    o.bc_type = d.bc_type;
    o.iri_time = d.iri_time;
    o.bch_len = d.bch_len;
    if (d.type == 1 && d.bch_len >= 63) {
        const int x = d.pos_xyz[0], y = d.pos_xyz[1], z = d.pos_xyz[2];
        const double xy = sqrt((double)x * x + (double)y * y);
        o.lat = atan2((double)z, xy) * 180.0 / M_PI;
        o.lon = atan2((double)y, (double)x) * 180.0 / M_PI;
        o.alt = (int)(sqrt((double)x * x + (double)y * y + (double)z * z) * 4.0) - 6378 + 23;
    }
    o.id = id;
    o.timestamp = timestamp;               // decoded_frame_t.timestamp / .frequency (frame_decode.c:418-419)
    o.frequency = frequency;
    return o;
}

// format_lcw_header (ida_decode.c:405-539): "LCW(ft,T:<type>,C:<code>,<remaining bits>)" left-justified in 110 columns
// plus one space.  Host text formatting of the four integers the kernel returns.
int lcw_field(const char *b, int from, int to)
{
    int v = 0;
    for (int i = from; i < to; i++) v = (v << 1) | (b[i] - '0');
    return v;
}

void format_lcw_header(int ft, int lcw_ft, int lcw_code, uint32_t lcw3_val, char *out, size_t outsz)
{
    char b[32], code[128], rem[64], raw[128];
    const char *ty = "rsrvd";
    for (int i = 0; i < 21; i++) b[i] = (char)('0' + ((lcw3_val >> (20 - i)) & 1));
    b[21] = 0;
    snprintf(code, sizeof(code), "rsrvd(%d)", lcw_code);           // the default of the maint / acchl / hndof switches
    snprintf(rem, sizeof(rem), "%s", b);
    if (lcw_ft == 0) {
        ty = "maint";
        if (lcw_code == 0) {
            snprintf(code, sizeof(code), "sync[status:%d,dtoa:%d,dfoa:%d]", b[1] - '0', lcw_field(b, 3, 13), lcw_field(b, 13, 21));
            snprintf(rem, sizeof(rem), "%c|%c", b[0], b[2]);
        } else if (lcw_code == 1) {
            snprintf(code, sizeof(code), "switch[dtoa:%d,dfoa:%d]", lcw_field(b, 3, 13), lcw_field(b, 13, 21));
            snprintf(rem, sizeof(rem), "%.3s", b);
        } else if (lcw_code == 3) {
            snprintf(code, sizeof(code), "maint[2][lqi:%d,power:%d,f_dtoa:%d,f_dfoa:%d]", (b[1] - '0') * 2 + (b[2] - '0'),
                     lcw_field(b, 3, 6), lcw_field(b, 6, 13), lcw_field(b, 13, 20));
            snprintf(rem, sizeof(rem), "%c|%c", b[0], b[20]);
        } else if (lcw_code == 6) {
            snprintf(code, sizeof(code), "geoloc");
        } else if (lcw_code == 12) {
            snprintf(code, sizeof(code), "maint[1][lqi:%d,power:%d]", (b[19] - '0') * 2 + (b[20] - '0'), lcw_field(b, 16, 19));
            snprintf(rem, sizeof(rem), "%.16s", b);
        } else if (lcw_code == 15) {
            snprintf(code, sizeof(code), "<silent>");
        }
    } else if (lcw_ft == 1) {
        ty = "acchl";
        if (lcw_code == 1) {
            snprintf(code, sizeof(code), "acchl[msg_type:%01x,bloc_num:%01x,sapi_code:%01x,segm_list:%.8s]",
                     lcw_field(b, 1, 4), b[4] - '0', lcw_field(b, 5, 8), b + 8);
            snprintf(rem, sizeof(rem), "%c,%02x", b[0], lcw_field(b, 16, 21));
        }
    } else if (lcw_ft == 2) {
        ty = "hndof";
        if (lcw_code == 3) {
            snprintf(code, sizeof(code), "handoff_resp[cand:%c,denied:%d,ref:%d,slot:%d,sband_up:%d,sband_dn:%d,access:%d]",
                     (b[2] - '0') == 0 ? 'P' : 'S', b[3] - '0', b[4] - '0', 1 + (b[6] - '0') * 2 + (b[7] - '0'),
                     lcw_field(b, 8, 13), lcw_field(b, 13, 18), lcw_field(b, 18, 21) + 1);
            snprintf(rem, sizeof(rem), "%.2s,%c", b, b[5]);
        } else if (lcw_code == 12) {
            snprintf(code, sizeof(code), "handoff_cand");
            snprintf(rem, sizeof(rem), "%.11s,%.10s", b, b + 11);
        } else if (lcw_code == 15) {
            snprintf(code, sizeof(code), "<silent>");
        }
    } else {
        snprintf(code, sizeof(code), "<%d>", lcw_code);
    }
    snprintf(raw, sizeof(raw), "LCW(%d,T:%s,C:%s,%s)", ft, ty, code, rem);
    snprintf(out, outsz, "%-110s ", raw);
}

// IdaOut (device) -> irdm_ida_t: the fields ida_decode() copies from the demod record (ida_decode.c:641-648) and the
// LCW header text
irdm_ida_t finish_ida(const IdaOut &d, const irdm_demod_t &f)
{
    irdm_ida_t o;
    memset(&o, 0, sizeof(o));
    o.id = f.id;
    if (!d.ok) return o;
    o.ok = 1;
    o.ft = d.ft; o.lcw_ft = d.lcw_ft; o.lcw_code = d.lcw_code; o.ec_lcw = d.ec_lcw; o.lcw3_val = d.lcw3_val;
    o.da_ctr = d.da_ctr; o.da_len = d.da_len; o.cont = d.cont; o.crc_ok = d.crc_ok;
    o.stored_crc = d.stored_crc; o.computed_crc = d.computed_crc;
    o.fixederrs = d.fixederrs; o.payload_len = d.payload_len; o.bch_len = d.bch_len;
    memcpy(o.payload, d.payload, sizeof(o.payload));
    memcpy(o.bch_stream, d.bch_stream, sizeof(o.bch_stream));
    format_lcw_header(d.ft, d.lcw_ft, d.lcw_code, d.lcw3_val, o.lcw_header, sizeof(o.lcw_header));
    o.direction = f.direction;
    o.timestamp = f.timestamp;
    o.frequency = f.center_frequency;
    o.magnitude = f.magnitude;
    o.noise = f.noise;
    o.level = f.level;
    o.confidence = f.confidence;
    o.n_symbols = f.n_payload_symbols;
    return o;
}

// ---- per-burst stages (K4..K7) of one batch of finished bursts ----
// bursts_enqueue() only enqueues on the context's stream; bursts_finish() waits for the batch and turns it into result
// records.  Nothing in between blocks the host: the one step that needs the host libm -- cexpf of the fine CFO
// (burst_downmix.c:716-717) and the centre frequency that decides the frame-length rules (:719, :763-767) -- is done by
// a helper thread over a mapped pinned copy of the work records: the stream records an event, the helper waits for it,
// does the arithmetic and publishes a sequence number that a one-lane kernel on the stream is waiting for.
// the centre frequency of the finished frames when the libm step ran on the device (rot_phase_kernel): the same
// expression, from the records the chain brought back (only frames that passed every drop rule read it)
void cfreq_from_records(BatchCtx &b)
{
    irdm_pipeline *p = b.owner;
    const DetParams &P = p->P;
    const int fs = p->cfg.sample_rate;
    for (int i = 0; i < b.n; i++) {
        const BurstWork &w = b.hp_work[i];
        const float rel = (w.center_bin - P.n / 2) / (float)P.n;
        double cf = p->cfg.center_frequency;
        cf += rel * fs;                                                   // burst_downmix.c:663-671
        if (w.drop_reason == 0) cf += w.center_offset * p->out_rate;
        b.h_cfreq[i] = cf;
    }
}

void fine_cfo_host(BatchCtx &b)
{
    irdm_pipeline *p = b.owner;
    const DetParams &P = p->P;
    const int fs = p->cfg.sample_rate;
    for (int i = 0; i < b.n; i++) {
        BurstWork &w = b.hp_work[i];
        const float rel = (w.center_bin - P.n / 2) / (float)P.n;
        double cf = p->cfg.center_frequency;
        cf += rel * fs;                                                   // burst_downmix.c:663-671
        if (!w.drop_reason) {
            const cfloat inc = fine_rotator_incr(w.center_offset);
            w.incr_re = inc.real();
            w.incr_im = inc.imag();
            cf += w.center_offset * p->out_rate;
        }
        b.h_cfreq[i] = cf;
        w.simplex = cf > 1626000000 ? 1 : 0;                              // iridium.h:18
    }
}

void cfo_helper_main(irdm_pipeline *p)
{
    pipeline_enter(p);
    for (;;) {
        BatchCtx *b;
        {
            std::unique_lock<std::mutex> lk(p->cfo_mu);
            p->cfo_cv.wait(lk, [&] { return p->cfo_quit || !p->cfo_jobs.empty(); });
            if (p->cfo_quit) return;
            b = p->cfo_jobs.front();
            p->cfo_jobs.pop_front();
        }
        (void)hipEventSynchronize(b->ev_cfo);
        fine_cfo_host(*b);
        __atomic_store_n(b->hp_flag, b->cfo_seq, __ATOMIC_RELEASE);
    }
}

// Back to an empty on-demand arena of `blocks` blocks (create's state; also what the test hook rot_pool_rows asks for):
// only while no chain has built or read a row.
int rot_arena_reset(irdm_pipeline *p, long long blocks)
{
    if (p->stat_rot_builds != 0) return -1;
    if (p->stream_rot_pre) IRDM_HIP_CHECK(hipStreamSynchronize(p->stream_rot_pre));
    p->rot_pre_pending = false;
    p->rot_pre_runs = 0;
    float2 *pool = dev_alloc<float2>((size_t)blocks * kRotRun);
    if (!pool) return -1;
    (void)hipFree(p->d_rot_table);
    p->d_rot_table = pool;
    p->rot_blocks_cap = (int)blocks;
    p->rot_blocks_used = 0;
    p->rot_rows_used = 0;
    std::fill(p->rot_len_h.begin(), p->rot_len_h.end(), 0);
    std::fill(p->rot_want.begin(), p->rot_want.end(), 0);
    std::fill(p->rot_build_ctx.begin(), p->rot_build_ctx.end(), -1);
    IRDM_HIP_CHECK(hipMemset(p->d_rot_slot, 0xff, sizeof(int) * (size_t)p->P.n * p->rot_runs));
    return 0;
}

// Every centre bin's row as far as a burst of ordinary length needs it (window = 2 pre + post + 12 ms of signal: Iridium's
// frames are 8.3 ms, simplex 20.3 ms -- longer bursts extend their bin's row on demand as before), one lane per bin, ONE
// launch on a stream of its own behind create; the chains wait for its event until the host has seen it complete.  The
// arena holds these blocks (bin b: blocks b * runs ..) plus the on-demand margin it had.  Footprint: n bins x runs x 16 KB --
// 0.07 GB at 2 MHz, 1.3 GB at 10 MHz, 3.2 GB at 12 MHz (of 288) -- against 3-4 ms per chain in a stream's first seconds.
int rot_prebuild(irdm_pipeline *p)
{
    if (p->stat_rot_builds != 0 || p->rot_rows_used != 0 || p->rot_pre_runs != 0) return -1;
    const DetParams &P = p->P;
    const long long window = 2ll * P.pre_len + P.post_len + (long long)(0.012 * p->cfg.sample_rate);
    int runs = (int)((window / kRotSeg + 8 + kRotRun - 1) / kRotRun);
    if (runs > p->rot_runs) runs = p->rot_runs;
    if (runs < 1) return -1;
    const long long blocks = (long long)P.n * runs;
    const long long cap = blocks + (long long)std::min(P.n, 256) * p->rot_runs;
    if (cap > 0x7fffffffll / 2) return -1;
    if (!p->stream_rot_pre && hipStreamCreateWithFlags(&p->stream_rot_pre, hipStreamNonBlocking) != hipSuccess) return -1;
    if (!p->ev_rot_pre && hipEventCreateWithFlags(&p->ev_rot_pre, hipEventDisableTiming) != hipSuccess) return -1;
    if (!p->d_rot_pre_news && !(p->d_rot_pre_news = dev_alloc<int4>((size_t)P.n))) return -1;
    float2 *pool = dev_alloc<float2>((size_t)cap * kRotRun);
    if (!pool) return -1;                                   // (no memory for it: rows on demand, as without the option)
    (void)hipFree(p->d_rot_table);
    p->d_rot_table = pool;
    p->rot_blocks_cap = (int)cap;
    std::vector<int4> news((size_t)P.n);
    for (int b = 0; b < P.n; b++) news[(size_t)b] = int4{ b, 0, runs * kRotRun, b * runs };
    IRDM_HIP_CHECK(hipMemcpy(p->d_rot_pre_news, news.data(), sizeof(int4) * news.size(), hipMemcpyHostToDevice));
    if (launch_rotator_rows(p->d_rot_incr, p->d_rot_table, p->rot_runs, p->d_rot_pre_news, P.n, p->d_rot_slot, p->stream_rot_pre) != 0)
        return -1;
    IRDM_HIP_CHECK(hipEventRecord(p->ev_rot_pre, p->stream_rot_pre));
    std::fill(p->rot_len_h.begin(), p->rot_len_h.end(), runs * kRotRun);
    p->rot_blocks_used = (int)blocks;
    p->rot_rows_used = P.n;
    p->rot_pre_runs = runs;
    p->rot_pre_pending = true;
    return 0;
}

int rot_rows_prepare(irdm_pipeline *p, BatchCtx &b, int nb, hipStream_t st)
{
    if (p->rot_pre_pending) {
        // (the prebuilt rows: this chain reads them -- and may continue or, growing the arena, copy them)
        IRDM_HIP_CHECK(hipStreamWaitEvent(st, p->ev_rot_pre, 0));
        if (hipEventQuery(p->ev_rot_pre) == hipSuccess) p->rot_pre_pending = false;
    }
    // A row is built as far as the bursts on its bin have needed it so far (a window of n samples restores checkpoints
    // 0 .. n / 16), in runs of kRotRun checkpoints -- a block of the arena each --, and continued from its last
    // checkpoint when a longer burst comes: the recurrence is sequential, 9 ns a sample -- 12 ms for a whole row at
    // 12 MHz, 1-2 ms for a typical burst's share.
    // What this chain has to wait for: the builds, on other chains' streams, of the runs its bursts' bins already have
    // (its decimator reads them, its own build continues them), unless the build is known to be complete (bursts_finish
    // waited for that context's stream since).  A context's builds are ordered on its stream, so its latest event covers
    // them all.  Chains whose bursts share no bin with a build in flight do not wait for it: the builds of consecutive
    // chunks run side by side.  And this chain's own build waits only for what it continues (a row whose last run is being
    // built elsewhere) or copies (a growing arena); the builds its DECIMATOR needs are waited for behind its own build.
    p->rot_touched.clear();
    const int me = (int)(&b - p->bc);
    unsigned wait_mask = 0, wait_first = 0;
    int blocks_wanted = 0;
    for (int i = 0; i < nb; i++) {
        const BurstWork &w = b.hp_work[i];
        if (w.drop_reason) continue;
        const int bin = w.center_bin;
        if (bin < 0 || bin >= p->P.n) continue;
        const int owner = p->rot_build_ctx[(size_t)bin];
        if (owner >= 0 && owner != me && p->rot_build_gen[(size_t)bin] > p->rot_done_gen[owner]) wait_mask |= 1u << owner;
        int need = (w.n + kRotSeg - 1) / kRotSeg + 8;
        need = (need + kRotRun - 1) / kRotRun * kRotRun;
        if (need > p->rot_runs * kRotRun) need = p->rot_runs * kRotRun;
        const int have = std::max(p->rot_len_h[(size_t)bin], p->rot_want[(size_t)bin]);
        if (need > have) {
            if (have > 0 && owner >= 0 && owner != me && p->rot_build_gen[(size_t)bin] > p->rot_done_gen[owner]) wait_first |= 1u << owner;
            if (p->rot_want[(size_t)bin] == 0) p->rot_touched.push_back(bin);
            p->rot_want[(size_t)bin] = need;
            blocks_wanted += (need - have) / kRotRun;
        }
    }
    const bool grow = !p->rot_touched.empty() && p->rot_blocks_used + blocks_wanted > p->rot_blocks_cap;
    if (grow)                                    // (the copy below reads every block built so far)
        for (int c = 0; c < p->n_bc; c++)
            if (p->rot_gen[c] > p->rot_done_gen[c]) wait_first |= 1u << c;
    // (the events as they are NOW: this chain's own build below does not touch them)
    auto wait_for = [&](unsigned mask) -> int {
        for (int c = 0; c < p->n_bc; c++)
            if (c != me && ((mask >> c) & 1)) IRDM_HIP_CHECK(hipStreamWaitEvent(st, p->bc[c].ev_rot, 0));
        return 0;
    };
    if (p->rot_touched.empty()) return wait_for(wait_mask);
    if (wait_for(wait_first) != 0) return -1;
    if (grow) {
        // the arena is full: twice the blocks (at most a whole row per FFT bin), the blocks built so far copied over on this
        // chain's stream -- behind every build so far (the waits above) -- and the old arena kept for the chains in flight
        // that were launched with its address (block numbers stay what they are)
        const long long max_blocks = (long long)p->P.n * p->rot_runs;
        long long cap2 = p->rot_blocks_cap;
        while (cap2 < (long long)p->rot_blocks_used + blocks_wanted && cap2 < max_blocks) cap2 = std::min(2 * cap2, max_blocks);
        float2 *pool2 = nullptr;
        if (cap2 < (long long)p->rot_blocks_used + blocks_wanted ||
            hipMalloc(reinterpret_cast<void **>(&pool2), sizeof(float2) * (size_t)cap2 * kRotRun) != hipSuccess) {
            for (int bin : p->rot_touched) p->rot_want[(size_t)bin] = 0;
            fprintf(stderr, "irdm_hip: no memory for %lld blocks of rotator checkpoints\n", cap2);
            return -1;
        }
        IRDM_HIP_CHECK(hipMemcpyAsync(pool2, p->d_rot_table, sizeof(float2) * (size_t)p->rot_blocks_used * kRotRun,
                                      hipMemcpyDeviceToDevice, st));
        p->rot_retired.push_back(p->d_rot_table);
        p->d_rot_table = pool2;
        p->rot_blocks_cap = (int)cap2;
        p->stat_rot_grows++;
        // The copy is ordered on THIS chain's stream only, but the host switches to pool2 at once: a chain enqueued on
        // another context a moment later (0.3 ms at bench rates: inside the copy's window) whose bins all have a finished
        // owner would read -- or continue a row from -- blocks of pool2 the copy has not written yet.  So this build
        // becomes the owner of every row built so far: whoever touches one of them waits for ev_rot below (recorded behind
        // the copy) until bursts_finish has synchronised this context.
        for (int bin = 0; bin < p->P.n; bin++)
            if (p->rot_len_h[(size_t)bin] > 0) {
                p->rot_build_ctx[(size_t)bin] = me;
                p->rot_build_gen[(size_t)bin] = p->rot_gen[me] + 1;
            }
    }
    int n_new = 0;
    for (int bin : p->rot_touched) {
        const int from = p->rot_len_h[(size_t)bin], to = p->rot_want[(size_t)bin];
        if (from == 0) p->rot_rows_used++;
        b.hp_rot_new[n_new++] = int4{ bin, from, to, p->rot_blocks_used };
        p->rot_blocks_used += (to - from) / kRotRun;
        p->stat_rot_ckpts += (uint64_t)(to - from);
        p->rot_len_h[(size_t)bin] = to;
        p->rot_want[(size_t)bin] = 0;
        p->rot_build_ctx[(size_t)bin] = me;
        p->rot_build_gen[(size_t)bin] = p->rot_gen[me] + 1;
    }
    p->rot_gen[me]++;
    p->stat_rot_builds++;
    p->stat_rot_rows += (uint64_t)n_new;
    // (the list by copy kernel: a kernel's plain loads of mapped host memory may be served from stale L2 lines)
    if (launch_copy_words(b.d_rot_new, b.hp_rot_new_dev, sizeof(int4) * (size_t)n_new, st) != 0) return -1;
    if (launch_rotator_rows(p->d_rot_incr, p->d_rot_table, p->rot_runs, b.d_rot_new, n_new, p->d_rot_slot, st) != 0) return -1;
    IRDM_HIP_CHECK(hipEventRecord(b.ev_rot, st));                   // chains with bursts on these bins wait for it
    return wait_for(wait_mask & ~wait_first);                        // what the decimator behind this build reads
}

int bursts_enqueue(irdm_pipeline *p, BatchCtx &b, const SampleSource &src, const GoneBurst *gone_list, int nb)
{
    const DetParams &P = p->P;
    const int fs = p->cfg.sample_rate;
    b.n = nb;
    b.recs.assign(nb, irdm_burst_t());
    size_t n_tiles = 0, dec_need = 0;
    int max_dec_len = 0;         // the longest decimated burst of the batch: the tile grid of post_tiles_kernel
    // (the register-resident decimator also needs the chunk to start at a multiple of 8 samples: a caller's burst window
    // presented as a chunk -- irdm_downmix_burst -- may not; such sources take the LDS kernel)
    const int fir_aligned = p->ring_len % 8 == 0 && p->ref_ring % 8 == 0 && (src.chunk_start == ~0ull || src.chunk_start % 8 == 0);
    const int tile_out = fir_tile_out(p->decim, fir_aligned, p->fir_generic, p->fir_order);
    for (int i = 0; i < nb; i++) {
        const GoneBurst &g = gone_list[i];
        irdm_burst_t &r = b.recs[i];
        r.id = g.id; r.start = g.start; r.stop = g.stop; r.last_active = g.last_active;
        r.center_bin = g.center_bin;
        r.peak_rel = g.peak_rel; r.base_sum = g.base_sum;
        // burst_detect.c:572, :583-586 with the host libm
        r.magnitude = 10.0f * log10f(g.peak_rel * kHistory * 1.72f);
        if (r.magnitude > p->peak_signal_db) p->peak_signal_db = r.magnitude;      // burst_detect.c:575-576
        r.noise = 10.0f * log10f(g.base_sum / kHistory / ((float)P.n * P.n) / 1.72f /
                                 ((float)fs / P.n));
        r.num_samples = g.stop + (uint64_t)P.pre_len - g.start;          // burst_detect.c:708-712
        // the frame [stop, stop+N) was processed by the feed call that delivered its last sample
        uint64_t e = (g.stop + (uint64_t)P.n + p->feed_block - 1) / p->feed_block * p->feed_block;
        r.avail_end = std::min<uint64_t>(e, src.chunk_end);

        BurstWork &w = b.hp_work[i];
        memset(&w, 0, sizeof(w));
        w.start = g.start;
        w.avail_end = r.avail_end;
        w.center_bin = g.center_bin;
        int n = r.num_samples > (uint64_t)(2 * 1024 * 1024) ? 2 * 1024 * 1024 : (int)r.num_samples;
        if ((size_t)n > p->l_cap) {
            fprintf(stderr, "irdm_hip: burst window %d exceeds l_cap %zu\n", n, p->l_cap);
            return -1;
        }
        w.n = n;
        w.dec_len = 0;
        w.drop_reason = 0;
        if (r.num_samples < 100) {
            w.drop_reason = 1;                                           // burst_downmix.c:645
        } else {
            int n_out = (n - p->in_ntaps + 1) / p->decim;                // burst_downmix.c:423
            if (n_out < 0) n_out = 0;
            w.dec_len = n_out;
            if (n_out < 100) w.drop_reason = 2;                          // burst_downmix.c:677
        }
        w.tile_base = (int32_t)n_tiles;
        w.dec_off = (int32_t)dec_need;
        if (!w.drop_reason) {
            max_dec_len = std::max(max_dec_len, w.dec_len);
            n_tiles += (size_t)(w.dec_len + tile_out - 1) / tile_out;
            dec_need += ((size_t)w.dec_len + 15) & ~(size_t)15;           // rows start on 128-byte lines
        }
    }
    b.ring_lo = b.ring_hi = 0;
    if (p->detect_only || nb == 0) return 0;      // stage A alone: burst records, no downmix / demod
    {
        uint64_t lo = ~0ull, hi = 0;
        for (int i = 0; i < nb; i++) {
            const BurstWork &w = b.hp_work[i];
            if (w.drop_reason) continue;
            // (a window that ends behind what its feed block had delivered reads the slots one reference ring length back)
            const uint64_t back = w.start + (uint64_t)w.n > w.avail_end ? p->ref_ring : 0;
            lo = std::min(lo, w.start > back ? w.start - back : 0);
            hi = std::max(hi, w.start + (uint64_t)w.n);
        }
        if (lo < hi) { b.ring_lo = lo; b.ring_hi = hi; }
    }
    if (dec_need > p->stat_scratch_peak) p->stat_scratch_peak = dec_need;
    if (dec_need > b.dec_cap) {
        // more outputs than this context's scratch holds: twice as much (the context is idle -- its last batch has been
        // collected -- but a free would wait for the whole device, which a gated scan may keep busy until this thread
        // opens the gate; the outgrown buffers stay until the context is closed)
        const size_t cap2 = std::max(dec_need, 2 * b.dec_cap);
        if (cap2 > (size_t)0x7fffffff) {
            fprintf(stderr, "irdm_hip: %zu decimated samples in a batch of %d bursts\n", dec_need, nb);
            return -1;
        }
        float2 *d2 = dev_alloc<float2>(cap2), *l2 = dev_alloc<float2>(lpf_alloc(cap2));
        if (!d2 || !l2) {
            if (d2) (void)hipFree(d2);
            if (l2) (void)hipFree(l2);
            fprintf(stderr, "irdm_hip: no memory for %zu decimated samples per batch\n", cap2);
            return -1;
        }
        p->scratch_retired.push_back(b.d_dec);
        p->scratch_retired.push_back(b.d_lpf);
        b.d_dec = d2;
        b.d_lpf = l2;
        b.dec_cap = cap2;
        if (!b.owns_buffers) { p->d_dec = d2; p->d_lpf = l2; }
        p->stat_scratch_grows++;
    }
    const bool tile_list = fir_needs_tile_list(p->decim, fir_aligned, p->fir_generic) != 0;
    if (n_tiles > b.tiles_cap) {
        // (like the scratch above: no hipFree / hipHostFree here -- either waits for the whole device, and in the time-shard
        // flow a gated scan spins on the device until THIS thread has returned from irdm_feed_end and published the history:
        // the free would sit out the gate's two-second time limit and the scan would fail.  The outgrown lists stay until
        // the context is closed.)
        if (b.d_tiles) p->tiles_retired.push_back(b.d_tiles);
        if (b.hp_tiles) p->tiles_host_retired.push_back(b.hp_tiles);
        b.hp_tiles = nullptr;
        b.tiles_cap = n_tiles * 2;
        b.d_tiles = dev_alloc<FirTile>((b.tiles_cap + 1) * kFirTileUnits);
        if (!b.owns_buffers) p->d_tiles = b.d_tiles;
        if (!b.d_tiles ||
            hipHostMalloc(reinterpret_cast<void **>(&b.hp_tiles), sizeof(FirTile) * b.tiles_cap, hipHostMallocDefault) != hipSuccess)
            return -1;
        p->stat_tiles_grows++;
    }
    if (tile_list) {
        n_tiles = 0;
        for (int i = 0; i < nb; i++) {
            const BurstWork &w = b.hp_work[i];
            if (!w.drop_reason)
                for (int o = 0; o < w.dec_len; o += tile_out) b.hp_tiles[n_tiles++] = FirTile{ i, o };
        }
    }
    hipStream_t st = b.stream;
    if (rot_rows_prepare(p, b, nb, st) != 0) return -1;
    // (copies by kernel, here and at the end of the chain: the runtime's copy path answers late next to the chains'
    // kernels, and an H2D from pinned memory may block the enqueueing thread)
    if (launch_copy_words(b.d_work, b.hp_work_dev, sizeof(BurstWork) * nb, st) != 0) return -1;
    if (tile_list && n_tiles && launch_copy_words(b.d_tiles, b.hp_tiles, sizeof(FirTile) * n_tiles, st) != 0) return -1;
    IRDM_HIP_CHECK(hipEventRecord(b.ev[0], st));
    if (launch_fir_decimate(src, b.d_work, nb, b.d_tiles, b.tiles_cap, (int)n_tiles, p->decim, p->d_in_taps,
                            p->d_fir_off, p->d_rot_incr, p->d_rot_table, p->rot_runs, b.d_dec, st,
                            p->kclk_fir((int)(&b - p->bc)), p->d_rot_slot, p->fir_order, p->fir_generic) != 0)
        return -1;
    IRDM_HIP_CHECK(hipEventRecord(b.ev[1], st));
    if (launch_downmix_post1(b.d_work, nb, max_dec_len, b.d_dec, b.d_lpf, box_of(b.d_lpf, b.dec_cap), p->d_noise_taps,
                             p->noise_ntaps, p->d_start_taps, p->start_ntaps, p->search_depth,
                             p->pre_start, p->d_cfo_window, p->d_tw4096, p->dev_cfo ? nullptr : b.hp_work_dev, st,
                             p->kclk_fir((int)(&b - p->bc)), p->fir_order, p->post_generic) != 0)
        return -1;
    // host libm step, ordered on the stream: post1 has stored what the step reads into the burst's record in the mapped
    // pinned buffer (system scope), the helper thread runs behind this event and publishes a sequence number, a one-lane
    // kernel waits for it, and rot_phase_kernel picks the step's results up from the same records.
    // (default: the step is part of rot_phase_kernel -- libm_port.hpp -- and the chain never leaves the GPU; the host
    // form remains for a host whose libm the port does not reproduce, irdm_create checks, and as the test hook host_cfo)
    CfoStep cfo;
    cfo.on_device = p->dev_cfo ? 1 : 0;
    cfo.n_fft = P.n;
    cfo.sample_rate = fs;
    cfo.out_rate = p->out_rate;
    cfo.center_frequency = p->cfg.center_frequency;
    b.cfo_on_device = p->dev_cfo;
    if (!p->dev_cfo) {
        IRDM_HIP_CHECK(hipEventRecord(b.ev_cfo, st));
        b.cfo_seq++;
        {
            std::lock_guard<std::mutex> lk(p->cfo_mu);
            p->cfo_jobs.push_back(&b);
        }
        p->cfo_cv.notify_one();
        if (launch_wait_host_flag(b.hp_flag_dev, b.cfo_seq, b.hp_flag_dev + 1, st) != 0) return -1;
    }
    if (launch_downmix_post2(b.d_work, nb, b.d_lpf, p->d_rrc_taps, p->rrc_ntaps,
                             p->d_tw2048, p->d_dl_fft, p->d_ul_fft, p->dl_len, p->ul_len, p->sps,
                             b.d_rrc_ws, b.d_frames, p->dev_cfo ? nullptr : b.hp_work_dev, cfo, st, p->fir_order, p->post_generic) != 0)
        return -1;
    IRDM_HIP_CHECK(hipEventRecord(b.ev[2], st));
    // the chain's results: work records and demodulator output.  packed_records (136 bytes per burst instead of 4.5 KB: hard
    // bits 8 per byte, no LLRs): written into pinned host memory by the demodulator's last kernel itself
    b.packed = p->packed_records && !p->decode_frames && !p->decode_ida && !p->keep_frame_samples;
    if (launch_demod(b.d_work, nb, b.d_frames, p->cfg.use_gardner, p->sps, b.d_demod_ws, b.d_demod, st,
                     b.packed ? b.hp_packed : nullptr, b.packed ? b.hp_work_dev : nullptr) != 0)
        return -1;
    IRDM_HIP_CHECK(hipEventRecord(b.ev[3], st));
    if (b.packed) return 0;
    if (p->decode_frames) {
        // post-demod bit layer on the demodulator's device-resident output (frames that failed the unique word
        // have ok = 0 and decode to FRAME_UNKNOWN)
        if (launch_frame_decode(b.d_demod, nb, p->d_syn_ra, p->d_syn_hdr, 1, nullptr, b.d_decoded, st) != 0)
            return -1;
    }
    if (p->decode_ida) {
        if (launch_ida_decode(b.d_demod, nb, p->d_syn_da, p->d_syn_l1, p->d_syn_l2, p->d_syn_l3, 1, nullptr, nullptr,
                              b.d_ida, st) != 0)
            return -1;
    }
    // full records: one copy launch for both
    return launch_copy2_to_host(b.hp_work_dev, b.d_work, sizeof(BurstWork) * nb, b.hp_demod, b.d_demod, sizeof(DemodOut) * nb, st);
}


// returns the number of bursts whose records were emitted, -1 on error
int bursts_finish(irdm_pipeline *p, BatchCtx &b)
{
    if (!p->chunk_marks || b.n == 0) return bursts_finish_records(p, b);
    const size_t before[6] = { p->q_bursts.size(), p->q_frames.size(), p->q_demods.size(), p->q_packed.size(),
                               p->q_decoded.size(), p->q_ida.size() };
    const uint64_t chunk = b.chunk_no;
    const int rc = bursts_finish_records(p, b);
    if (rc < 0) return rc;
    irdm_chunk_mark_t m;
    m.chunk = chunk;
    m.n_bursts = (uint32_t)(p->q_bursts.size() - before[0]);
    m.n_frames = (uint32_t)(p->q_frames.size() - before[1]);
    m.n_demods = (uint32_t)(p->q_demods.size() - before[2]);
    m.n_packed = (uint32_t)(p->q_packed.size() - before[3]);
    m.n_decoded = (uint32_t)(p->q_decoded.size() - before[4]);
    m.n_ida = (uint32_t)(p->q_ida.size() - before[5]);
    p->q_marks.push_back(m);
    return rc;
}

int bursts_finish_records(irdm_pipeline *p, BatchCtx &b)
{
    const int nb = b.n;
    const int fs = p->cfg.sample_rate;
    if (nb == 0) return 0;
    if (p->detect_only) {
        b.n = 0;
        for (int i = 0; i < nb; i++) {
            p->q_bursts.push_back(b.recs[i]);
            p->last_bursts.push_back(b.recs[i]);
        }
        return nb;
    }
    IRDM_HIP_CHECK(hipStreamSynchronize(b.stream));
    p->rot_done_gen[(int)(&b - p->bc)] = p->rot_gen[(int)(&b - p->bc)];      // (its rotator checkpoint builds are complete)
    struct timespec ts_;
    clock_gettime(CLOCK_MONOTONIC, &ts_);
    const double t_rec0 = ts_.tv_sec * 1e6 + ts_.tv_nsec * 1e-3;
    if (b.cfo_on_device) cfreq_from_records(b);
    b.n = 0;                     // only now: the helper thread reads it while the chain is in flight
    if (b.hp_flag[1]) {
        fprintf(stderr, "irdm_hip: the host step of the per-burst chain did not answer\n");
        return -1;
    }
    float ms = 0;
    b.ms[0] = hipEventElapsedTime(&ms, b.ev[0], b.ev[1]) == hipSuccess ? ms : -1.0f;
    b.ms[1] = hipEventElapsedTime(&ms, b.ev[1], b.ev[2]) == hipSuccess ? ms : -1.0f;
    b.ms[2] = hipEventElapsedTime(&ms, b.ev[2], b.ev[3]) == hipSuccess ? ms : -1.0f;
    for (int i = 0; i < 3; i++) p->last_ms[2 + i] = b.ms[i];
    if (p->decode_frames) {
        p->h_decoded.resize(nb);
        IRDM_HIP_CHECK(hipMemcpyAsync(p->h_decoded.data(), b.d_decoded, sizeof(DecodedOut) * nb, hipMemcpyDeviceToHost, b.stream));
    }
    if (p->decode_ida) {
        p->h_ida.resize(nb);
        IRDM_HIP_CHECK(hipMemcpyAsync(p->h_ida.data(), b.d_ida, sizeof(IdaOut) * nb, hipMemcpyDeviceToHost, b.stream));
    }
    if (p->keep_frame_samples) {
        p->h_frames.resize((size_t)nb * kMaxFrameSamples * 2);
        IRDM_HIP_CHECK(hipMemcpyAsync(p->h_frames.data(), b.d_frames, sizeof(float2) * (size_t)nb * kMaxFrameSamples,
                                      hipMemcpyDeviceToHost, b.stream));
    }
    if (p->decode_frames || p->decode_ida || p->keep_frame_samples) IRDM_HIP_CHECK(hipStreamSynchronize(b.stream));
    if (b.packed) {
        // packed_records: burst records and the compact frame records only (no frame-info queue, no LLRs); the same
        // expressions as below for the timestamp (burst_downmix.c:659-660, :431-433, :783) and the refined frequency
        // (qpsk_demod.c:521-527)
        for (int i = 0; i < nb; i++) {
            const BurstWork &w = b.hp_work[i];
            const irdm_burst_t &r = b.recs[i];
            p->q_bursts.push_back(r);
            p->last_bursts.push_back(r);
            const DemodPacked &d = b.hp_packed[i];
            if (w.drop_reason != 0 || !d.ok) continue;
            uint64_t timestamp = p->start_time_ns + (uint64_t)((double)r.start / fs * 1e9);
            if (w.dec_len > 0) timestamp += (uint64_t)((p->in_ntaps / 2) * 1000000000ULL / fs);
            p->q_packed.emplace_back();
            irdm_demod_packed_t &o = p->q_packed.back();
            o.id = r.id;
            o.timestamp = timestamp + (uint64_t)((double)w.start_idx / p->out_rate * 1e9);
            o.direction = d.direction;
            o.magnitude = r.magnitude;
            o.noise = r.noise;
            o.confidence = d.confidence;
            o.level = d.level;
            o.n_symbols = d.n_symbols;
            o.n_payload_symbols = d.n_symbols - 12;
            o.n_bits = 2 * d.n_symbols;
            o.ok = 1;
            o.total_phase = d.total_phase;
            memcpy(o.bits, d.bits, sizeof(o.bits));
            if (d.n_symbols > 0) {
                const double duration = (double)d.n_symbols / 25000;
                o.center_frequency = b.h_cfreq[i] + d.total_phase / duration / M_PI / 2.0;
            } else {
                o.center_frequency = b.h_cfreq[i];
            }
        }
        clock_gettime(CLOCK_MONOTONIC, &ts_);
        p->host_us[9] += ts_.tv_sec * 1e6 + ts_.tv_nsec * 1e-3 - t_rec0;
        return nb;
    }
    for (int i = 0; i < nb; i++) {
        const BurstWork &w = b.hp_work[i];
        irdm_burst_t &r = b.recs[i];
        p->q_bursts.push_back(r);
        p->last_bursts.push_back(r);

        irdm_frame_info_t f;
        memset(&f, 0, sizeof(f));
        f.id = r.id;
        f.drop_reason = w.drop_reason;
        f.dec_len = w.dec_len;
        uint64_t timestamp = p->start_time_ns + (uint64_t)((double)r.start / fs * 1e9);   // :659-660
        if (w.dec_len > 0) timestamp += (uint64_t)((p->in_ntaps / 2) * 1000000000ULL / fs); // :431-433
        if (w.drop_reason == 0 || w.drop_reason >= 3) f.start = w.start_idx;
        if (w.drop_reason == 0 || w.drop_reason >= 4) {
            f.center_offset = w.center_offset;
            f.uw_start_idx = w.uw_start;
            f.corr_re = w.corr_re;
            f.corr_im = w.corr_im;
            f.direction = w.direction;
        }
        if (w.drop_reason == 0) {
            f.timestamp = timestamp + (uint64_t)((double)w.start_idx / p->out_rate * 1e9);  // :783
            f.center_frequency = b.h_cfreq[i];
            f.sample_rate = (float)p->out_rate;
            f.samples_per_symbol = p->sps;
            f.magnitude = r.magnitude;
            f.noise = r.noise;
            f.uw_start = w.uw_corr;
            f.num_samples = w.num_samples;
        }
        if (w.drop_reason == 0) {
            f.demod_ok = b.hp_demod[i].ok ? 1 : 0;
            f.demod_direction = b.hp_demod[i].ok ? b.hp_demod[i].direction : 0;      // DIR_UNDEF, qpsk_demod.c:444
        }
        p->q_frames.push_back(f);
        {
            // always one entry per frame record, so that the two queues stay paired whatever keep_frame_samples does
            std::vector<float> sv;
            if (p->keep_frame_samples && w.drop_reason == 0)
                sv.assign(p->h_frames.begin() + (size_t)i * kMaxFrameSamples * 2,
                          p->h_frames.begin() + (size_t)i * kMaxFrameSamples * 2 + 2 * (size_t)w.num_samples);
            p->q_frame_samples.push_back(std::move(sv));
        }
        if (w.drop_reason == 0 && b.hp_demod[i].ok) {
            const DemodOut &d = b.hp_demod[i];
            // built in place in the queue, and only the symbols the frame has are copied (a record is 4.5 KB; 667 of
            // them filled, copied and copied again cost the feeding thread 0.5 ms per chunk)
            p->q_demods.emplace_back();
            irdm_demod_t &o = p->q_demods.back();
            const size_t nbits = std::min<size_t>(sizeof(o.bits) / sizeof(o.bits[0]), (size_t)(d.n_symbols > 0 ? 2 * d.n_symbols : 0));
            memset(&o, 0, offsetof(irdm_demod_t, bits));
            o.id = r.id;
            o.timestamp = f.timestamp;
            o.direction = d.direction;
            o.magnitude = r.magnitude;
            o.noise = r.noise;
            o.confidence = d.confidence;
            o.level = d.level;
            o.n_symbols = d.n_symbols;
            o.n_payload_symbols = d.n_symbols - 12;
            o.n_bits = 2 * d.n_symbols;
            o.ok = 1;
            o.total_phase = d.total_phase;
            memcpy(o.bits, d.bits, nbits * sizeof(o.bits[0]));
            memset(o.bits + nbits, 0, sizeof(o.bits) - nbits * sizeof(o.bits[0]));
            memcpy(o.llr, d.llr, nbits * sizeof(o.llr[0]));
            memset(o.llr + nbits, 0, sizeof(o.llr) - nbits * sizeof(o.llr[0]));
            if (d.n_symbols > 0) {                                       // qpsk_demod.c:521-527
                const double duration = (double)d.n_symbols / 25000;
                o.center_frequency = f.center_frequency + d.total_phase / duration / M_PI / 2.0;
            } else {
                o.center_frequency = f.center_frequency;
            }
            if (p->decode_frames) p->q_decoded.push_back(finish_decoded(p->h_decoded[i], o.id, o.timestamp, o.center_frequency));
            if (p->decode_ida) p->q_ida.push_back(finish_ida(p->h_ida[i], o));
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &ts_);
    p->host_us[9] += ts_.tv_sec * 1e6 + ts_.tv_nsec * 1e-3 - t_rec0;     // [9] building the records (inside [4])
    return nb;
}

// all finished bursts of a chunk, synchronously, through context `b` (batches of at most burst_cap)
int process_bursts(irdm_pipeline *p, BatchCtx &b, const SampleSource &src, const GoneBurst *gone_list, int n_gone)
{
    for (int base = 0; base < n_gone; base += p->burst_cap) {
        const int nb = std::min(p->burst_cap, n_gone - base);
        if (bursts_enqueue(p, b, src, gone_list + base, nb) != 0 || bursts_finish(p, b) < 0) return -1;
    }
    return 0;
}

}  // namespace irdmh
