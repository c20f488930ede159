// api.cpp -- options and statistics of a context, the kernel clock, the RAW line printer (frame_output.c:160-199) and the
// --save-bursts file pair (qpsk_demod.c:339-389).
#include "pipeline.hpp"

namespace irdmh {

extern "C" int irdm_set_option(irdm_pipeline_t *p, const char *key, int value)
{
    if (!p || !key) return -1;
    // ---- what a caller chooses (include/irdm_hip.h documents every key) ----
    if (!strcmp(key, "keep_frame_samples")) { p->keep_frame_samples = value; return 0; }
    if (!strcmp(key, "packed_records")) { p->packed_records = value; return 0; }
    if (!strcmp(key, "chunk_marks")) { p->chunk_marks = value ? 1 : 0; if (!value) p->q_marks.clear(); return 0; }
    if (!strcmp(key, "decode_frames")) { p->decode_frames = value; return 0; }
    if (!strcmp(key, "decode_ida")) { p->decode_ida = value; return 0; }
    if (!strcmp(key, "detect_only")) { p->detect_only = value; return 0; }
    if (!strcmp(key, "fir_order") || !strcmp(key, "simd_order")) { p->fir_order = value ? 1 : 0; return 0; }
    if (!strcmp(key, "host_cfo")) { p->dev_cfo = p->dev_cfo_ok && value == 0; return 0; }      // 1: the fine-CFO libm step on the helper thread
    if (!strcmp(key, "scan_mode")) { p->scan_mode = value; return 0; }
    if (!strcmp(key, "kernel_clock")) { p->kernel_clock = value != 0; return 0; }
    if (!strcmp(key, "rot_prebuild")) {
        // 1: every centre bin's row in one background launch now (the default of a context with pipeline_depth >= 1);
        // 0: rows on demand only (the default otherwise); only before the first burst
        if (value) return p->rot_pre_runs ? 0 : rot_prebuild(p);
        return p->rot_pre_runs ? rot_arena_reset(p, (long long)std::min(p->P.n, 1024) * p->rot_runs) : 0;
    }
    // ---- diagnostic ----
    if (!strcmp(key, "band_timeline")) { p->band_tune.timeline = value != 0; return 0; }
    // ---- test hooks: paths a default run takes only on rare inputs ----
    if (!strcmp(key, "fir_generic")) { p->fir_generic = value != 0; return 0; }
    if (!strcmp(key, "post_generic")) { p->post_generic = value != 0; return 0; }
    if (!strcmp(key, "k1_lists")) { p->k1_lists = value; return 0; }
    if (!strcmp(key, "band_first")) { p->band_first = value < 0 ? 0 : value > kBandRounds ? kBandRounds : value; return 0; }
    if (!strcmp(key, "band_spec")) { p->band_spec_opt = value != 0; return 0; }
    if (!strcmp(key, "band_selfcheck")) { p->band_tune.selfcheck = value; return 0; }
    if (!strcmp(key, "rot_pool_rows")) {
        // (test hook) an empty on-demand rotator checkpoint arena with room for `value` whole rows (a prebuilt one is given
        // up); only before the first burst
        if (value < 1 || value > p->P.n) return -1;
        return rot_arena_reset(p, (long long)value * p->rot_runs);
    }
    if (!strcmp(key, "scratch_outputs")) {
        // (test hook) the decimated / low-passed scratch of every context with room for `value` outputs to begin with;
        // only while no batch is in flight
        if (value < 16) return -1;
        for (int i = 0; i < p->n_bc; i++)
            if (p->bc[i].n != 0) return -1;
        IRDM_HIP_CHECK(hipDeviceSynchronize());
        for (int i = 0; i < p->n_bc; i++) {
            BatchCtx &b = p->bc[i];
            float2 *d2 = dev_alloc<float2>((size_t)value), *l2 = dev_alloc<float2>(lpf_alloc((size_t)value));
            if (!d2 || !l2) return -1;
            (void)hipFree(b.d_dec);
            (void)hipFree(b.d_lpf);
            b.d_dec = d2;
            b.d_lpf = l2;
            b.dec_cap = (size_t)value;
            if (!b.owns_buffers) { p->d_dec = d2; p->d_lpf = l2; }
        }
        return 0;
    }
    return -1;
}

extern "C" int64_t irdm_get_stat(const irdm_pipeline_t *p, const char *key)
{
    if (!p || !key) return -1;
    if (!strcmp(key, "scan_fast_chunks")) return (int64_t)p->stat_fast_chunks;
    if (!strcmp(key, "scan_fallbacks")) return (int64_t)p->stat_fallbacks;
    if (!strncmp(key, "host_us_", 8) && key[8] >= '0' && key[8] <= '9') return (int64_t)p->host_us[key[8] - '0'];
    if (!strcmp(key, "band_chunks")) return (int64_t)p->stat_band_chunks;
    if (!strcmp(key, "band_extra")) return (int64_t)p->stat_band_extra;
    if (!strcmp(key, "scan_chained")) return (int64_t)p->stat_chained;
    if (!strcmp(key, "scan_chain_undone")) return (int64_t)p->stat_chain_undone;
    if (!strcmp(key, "k1_lists")) return (int64_t)p->stat_k1_lists;
    if (!strcmp(key, "band_rounds")) return (int64_t)p->stat_band_rounds;
    if (!strcmp(key, "band_retries")) return (int64_t)p->stat_band_retries;
    if (!strcmp(key, "band_aborts")) return (int64_t)p->stat_band_aborts;
    if (!strncmp(key, "tl_dur_", 7)) { const int i = atoi(key + 7); return i >= 0 && i < 32 ? (int64_t)p->stat_tl_dur[i] : -1; }
    if (!strncmp(key, "tl_gap_", 7)) { const int i = atoi(key + 7); return i >= 0 && i < 32 ? (int64_t)p->stat_tl_gap[i] : -1; }
    if (!strncmp(key, "tl_n_", 5)) { const int i = atoi(key + 5); return i >= 0 && i < 32 ? (int64_t)p->stat_tl_n[i] : -1; }
    if (!strncmp(key, "plan_tp_", 8)) {
        const int i = atoi(key + 8);
        return i >= 0 && i < 16 ? (int64_t)p->stat_plan_tp[i] : -1;
    }
    if (!strcmp(key, "rot_rows")) return (int64_t)p->rot_rows_used;
    if (!strcmp(key, "rot_prebuilt_runs")) return (int64_t)p->rot_pre_runs;
    if (!strcmp(key, "rot_rows_cap")) return (int64_t)(p->rot_blocks_cap / p->rot_runs);      // (in whole rows)
    if (!strcmp(key, "rot_blocks")) return (int64_t)p->rot_blocks_used;
    if (!strcmp(key, "rot_blocks_cap")) return (int64_t)p->rot_blocks_cap;
    if (!strcmp(key, "rot_grows")) return (int64_t)p->stat_rot_grows;
    if (!strcmp(key, "rot_builds")) return (int64_t)p->stat_rot_builds;
    if (!strcmp(key, "rot_runs")) return (int64_t)p->stat_rot_rows;
    if (!strcmp(key, "rot_ckpts")) return (int64_t)p->stat_rot_ckpts;
    if (!strcmp(key, "band_steps")) return (int64_t)p->stat_band_steps;
    if (!strcmp(key, "scratch_outputs")) return (int64_t)p->bc[0].dec_cap;
    if (!strcmp(key, "scratch_grows")) return (int64_t)p->stat_scratch_grows;
    if (!strcmp(key, "tiles_grows")) return (int64_t)p->stat_tiles_grows;
    if (!strcmp(key, "ring_waits")) return (int64_t)p->stat_ring_waits;
    if (!strcmp(key, "spec_passes")) return (int64_t)p->stat_spec_passes;
    if (!strcmp(key, "spec_scans")) return (int64_t)p->stat_spec_scans;
    if (!strcmp(key, "sum_restarts")) return (int64_t)p->stat_sum_restarts;
    if (!strcmp(key, "scratch_peak")) return (int64_t)p->stat_scratch_peak;
    if (!strcmp(key, "band_last_flags")) return (int64_t)p->last_band_flags;
    if (!strcmp(key, "scan_dense_frames")) return (int64_t)p->stat_dense_frames;
    return -1;
}

// Kernel clock (option "kernel_clock" 1): the device's own record of a kernel's launches -- first wavefront in to last
// wavefront out, s_memrealtime -- summed since the last reset.  which: 0 the register-resident decimator, 1 K1.
extern "C" int irdm_kernel_clock(irdm_pipeline_t *p, int which, double *sum_ms, uint64_t *launches, double *last_ms, int reset)
{
    if (!p || !p->d_kclk || which < 0 || which > 1) return -1;
    pipeline_enter(p);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    std::vector<unsigned long long> h((size_t)(6 + kMaxBc - 3) * kKClkWords);
    if (hipMemcpy(h.data(), p->d_kclk, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    unsigned long long ticks = 0, n = 0, last = 0;
    // (records 0..2 and 6..: the decimator per batch context; 3..5: K1 per feed slot)
    std::vector<int> recs;
    if (which == 1) recs = { 3, 4, 5 };
    else
        for (int c = 0; c < kMaxBc; c++) recs.push_back(c < 3 ? c : 3 + c);
    for (int r : recs) {
        ticks += h[(size_t)r * kKClkWords + 128];
        n += h[(size_t)r * kKClkWords + 129];
        if (h[(size_t)r * kKClkWords + 130] > last) last = h[(size_t)r * kKClkWords + 130];
    }
    if (sum_ms) *sum_ms = (double)ticks * 1e-5;          // 10 ns ticks
    if (launches) *launches = n;
    if (last_ms) *last_ms = (double)last * 1e-5;
    if (reset) {
        for (int r : recs) {
            unsigned long long z[3] = { 0, 0, 0 };
            if (hipMemcpy(p->d_kclk + (size_t)r * kKClkWords + 128, z, sizeof(z), hipMemcpyHostToDevice) != hipSuccess) return -1;
        }
    }
    return 0;
}

extern "C" int irdm_last_timings(const irdm_pipeline_t *p, float *ms_out, int n)
{
    if (!p || !ms_out) return -1;
    for (int i = 0; i < n && i < 6; i++) ms_out[i] = p->last_ms[i];
    return n < 6 ? n : 6;
}

// ===========================================================================
// 3. RAW line (frame_output.c:144-199)
// ===========================================================================
extern "C" int irdm_format_raw(const irdm_demod_t *f, const char *file_info, uint64_t *t0_io, char *buf,
                               size_t cap)
{
    if (!f || !t0_io || !buf) return -1;
    char auto_info[64];
    if (*t0_io == 0) *t0_io = (f->timestamp / 1000000000ULL) * 1000000000ULL;
    const uint64_t t0 = *t0_io;
    if (!file_info || !file_info[0]) {
        snprintf(auto_info, sizeof(auto_info), "i-%llu-t1", (unsigned long long)(t0 / 1000000000ULL));
        file_info = auto_info;
    }
    const double ts_ms = (double)(f->timestamp - t0) / 1000000.0;
    const int freq_hz = (int)(f->center_frequency + 0.5);
    const int payload = f->n_payload_symbols < 0 ? 0 : f->n_payload_symbols;
    int pos = snprintf(buf, cap, "RAW: %s %012.4f %010d N:%05.2f%+06.2f I:%011llu %3d%% %.5f %3d ", file_info,
                       ts_ms, freq_hz, f->magnitude, f->noise, (unsigned long long)f->id, f->confidence,
                       f->level, payload);
    if (pos < 0 || (size_t)pos + (size_t)f->n_bits + 2 > cap) return -1;
    for (int i = 0; i < f->n_bits; i++) buf[pos++] = (char)('0' + f->bits[i]);
    buf[pos++] = '\n';
    buf[pos] = 0;
    return pos;
}

// the same line from a compact record (option packed_records): bits 8 per byte, MSB first
extern "C" int irdm_format_raw_packed(const irdm_demod_packed_t *f, const char *file_info, uint64_t *t0_io, char *buf,
                                      size_t cap)
{
    if (!f || !t0_io || !buf) return -1;
    char auto_info[64];
    if (*t0_io == 0) *t0_io = (f->timestamp / 1000000000ULL) * 1000000000ULL;
    const uint64_t t0 = *t0_io;
    if (!file_info || !file_info[0]) {
        snprintf(auto_info, sizeof(auto_info), "i-%llu-t1", (unsigned long long)(t0 / 1000000000ULL));
        file_info = auto_info;
    }
    const double ts_ms = (double)(f->timestamp - t0) / 1000000.0;
    const int freq_hz = (int)(f->center_frequency + 0.5);
    const int payload = f->n_payload_symbols < 0 ? 0 : f->n_payload_symbols;
    int pos = snprintf(buf, cap, "RAW: %s %012.4f %010d N:%05.2f%+06.2f I:%011llu %3d%% %.5f %3d ", file_info,
                       ts_ms, freq_hz, f->magnitude, f->noise, (unsigned long long)f->id, f->confidence,
                       f->level, payload);
    const int nb = f->n_bits < 0 ? 0 : (f->n_bits > IRDM_MAX_BITS ? IRDM_MAX_BITS : f->n_bits);
    if (pos < 0 || (size_t)pos + (size_t)nb + 2 > cap) return -1;
    for (int i = 0; i < nb; i++) buf[pos++] = (char)('0' + ((f->bits[i >> 3] >> (7 - (i & 7))) & 1));
    buf[pos++] = '\n';
    buf[pos] = 0;
    return pos;
}

extern "C" long long irdm_format_raw_packed_batch(const irdm_demod_packed_t *f, int n, const char *file_info, uint64_t *t0_io,
                                                  char *buf, size_t cap)
{
    if (!f || n < 0 || !t0_io || !buf) return -1;
    size_t pos = 0;
    for (int i = 0; i < n; i++) {
        const int len = irdm_format_raw_packed(&f[i], file_info, t0_io, buf + pos, cap - pos);
        if (len < 0) return -1;
        pos += (size_t)len;
    }
    return (long long)pos;
}

// ===========================================================================
// 4. --save-bursts (qpsk_demod.c:339-389)
// ===========================================================================
extern "C" int irdm_save_burst(const irdm_frame_info_t *info, const float *samples, const char *dir)
{
    if (!info || !samples || !dir || info->drop_reason != 0 || info->num_samples <= 0) return -1;
    struct stat st;
    memset(&st, 0, sizeof(st));
    if (stat(dir, &st) == -1) {
        if (mkdir(dir, 0755) == -1 && errno != EEXIST) {
            fprintf(stderr, "Warning: failed to create burst save directory: %s\n", strerror(errno));
            return -1;
        }
    }
    const char *dir_str = info->demod_direction == 1 ? "DL" : info->demod_direction == 2 ? "UL" : "UN";
    char base[512];
    snprintf(base, sizeof(base), "%s/%020lu_%011.0f_%lu_%s", dir, (unsigned long)info->timestamp,
             info->center_frequency, (unsigned long)info->id, dir_str);
    char path[520];
    snprintf(path, sizeof(path), "%s.cf32", base);
    FILE *f = fopen(path, "wb");
    if (!f) {
        fprintf(stderr, "Warning: failed to save burst IQ: %s\n", strerror(errno));
        return -1;
    }
    fwrite(samples, 2 * sizeof(float), (size_t)info->num_samples, f);
    fclose(f);
    snprintf(path, sizeof(path), "%s.meta", base);
    f = fopen(path, "w");
    if (!f) return -1;
    fprintf(f, "burst_id: %lu\n", (unsigned long)info->id);
    fprintf(f, "timestamp_ns: %lu\n", (unsigned long)info->timestamp);
    fprintf(f, "center_freq_hz: %.0f\n", info->center_frequency);
    fprintf(f, "sample_rate_hz: %.0f\n", info->sample_rate);
    fprintf(f, "samples_per_symbol: %.2f\n", info->samples_per_symbol);
    fprintf(f, "direction: %s\n", dir_str);
    fprintf(f, "magnitude_db: %.2f\n", info->magnitude);
    fprintf(f, "noise_dbfs_hz: %.2f\n", info->noise);
    fprintf(f, "num_samples: %zu\n", (size_t)info->num_samples);
    fprintf(f, "uw_start_offset: %.2f\n", info->uw_start);
    fclose(f);
    return 0;
}

// many lines into one buffer: one write()/fwrite() per poll batch instead of the reference's fflush per line
// (frame_output.c:196-198), which is the sink bottleneck at >= 1e5 lines/s; the bytes are identical
extern "C" long long irdm_format_raw_batch(const irdm_demod_t *f, int n, const char *file_info, uint64_t *t0_io,
                                           char *buf, size_t cap)
{
    if (!f || n < 0 || !t0_io || !buf) return -1;
    size_t pos = 0;
    for (int i = 0; i < n; i++) {
        const int len = irdm_format_raw(&f[i], file_info, t0_io, buf + pos, cap - pos);
        if (len < 0) return -1;          // cap too small (IRDM_RAW_LINE_MAX bytes per frame always suffice)
        pos += (size_t)len;
    }
    return (long long)pos;
}

extern "C" const char *irdm_version(void) { return "irdm_hip 0.1 (gfx950)"; }

}  // namespace irdmh
