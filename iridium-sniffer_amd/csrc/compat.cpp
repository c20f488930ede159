// compat.cpp -- the reference's stage-level entry points (burst_detect.h:67-94, burst_downmix.h:64-73,
// qpsk_demod.h:42) as thin adapters over the batched C-ABI (include/irdm_hip.h).  See include/irdm_compat.h for the
// contract; the struct mirrors below have the reference's layouts with `float complex *` spelled `float *`.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "../../include/irdm_hip.h"

extern "C" {

typedef struct {
    uint64_t id, start, stop, last_active;
    int center_bin;
    float magnitude, noise;
} burst_info_t;

typedef struct {
    burst_info_t info;
    double center_frequency;
    int sample_rate;
    int fft_size;
    uint64_t start_time_ns;
    size_t num_samples;
    float *samples;             // float complex *
} burst_data_t;

typedef struct {
    double center_frequency;
    int sample_rate, fft_size, burst_pre_len, burst_post_len, burst_width, max_bursts, max_burst_len;
    float threshold;
    int history_size, use_gpu;
} burst_config_t;

typedef void (*burst_callback_t)(burst_data_t *burst, void *user);

typedef struct {
    uint64_t id, timestamp;
    double center_frequency;
    float sample_rate, samples_per_symbol;
    int direction;              // ir_direction_t
    float magnitude, noise, uw_start;
    size_t num_samples;
    float *samples;             // float complex *
} downmix_frame_t;

typedef struct {
    int output_sample_rate, search_depth, handle_multiple_frames;
} downmix_config_t;

typedef struct {
    uint64_t id, timestamp;
    double center_frequency;
    int direction;
    float magnitude, noise;
    int confidence;
    float level;
    int n_symbols, n_payload_symbols;
    uint8_t *bits;
    float *llr;
    int n_bits;
} demod_frame_t;

extern int use_gardner __attribute__((weak));     // main.c:143 when the host is the reference's main.c

}  // extern "C"

namespace {

constexpr int kBlock = 32768;                      // main.c:225: the file reader's block, the detector's feed granularity

int gardner_setting() { return &use_gardner ? use_gardner : 1; }

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// stage A
// ---------------------------------------------------------------------------------------------------------------
struct _burst_detector {
    burst_config_t cfg;
    irdm_pipeline_t *p;
    int fmt;                    // format of the first feed call (a detector is fed one format, main.c:228-262)
    std::vector<unsigned char> stage;   // input not yet making up a whole block
    uint64_t start_time_ns;
    uint64_t total;
    burst_callback_t last_cb;           // the callback of the last feed: the tail of the stream is delivered through it
    void *last_user;
    // what main.c's stats thread reads while the detector thread feeds (main.c:455-456).  The reference's getters are free
    // (plain reads of the detector's fields); here the figures live on the device, so they are fetched when a getter asks
    // and a feed has run since the last fetch -- not after every 32768-sample block (a device-wide wait and three
    // copies 305 times a second at 10 MHz).  `mu` keeps the getter's fetch and the detector thread's feed apart.
    std::recursive_mutex mu;        // (recursive: a burst callback may ask the getters)
    bool stats_dirty;
    int active;
    float noise_floor, peak_signal;
};

static void detector_refresh(_burst_detector *d)
{
    if (!d->p || !d->stats_dirty) return;
    irdm_detector_stats_t st;
    if (irdm_detector_stats(d->p, &st) == 0) {
        d->active = st.active_bursts;
        d->noise_floor = st.noise_floor_dbfs_hz;
        d->peak_signal = st.peak_signal_db;
    }
    d->stats_dirty = false;
}

extern "C" _burst_detector *burst_detector_create(burst_config_t *config)
{
    if (!config || config->sample_rate <= 0) return nullptr;
    const int n = 1 << (int)round(log2(config->sample_rate / 1000.0));
    // only the default geometry (burst_detect.c:180-226) is implemented
    if ((config->fft_size && config->fft_size != n) || (config->burst_pre_len && config->burst_pre_len != 2 * n) ||
        (config->burst_post_len && config->burst_post_len != (int)(config->sample_rate * 16e-3)) ||
        (config->burst_width && config->burst_width != 40000) ||
        (config->max_bursts && config->max_bursts != (int)((config->sample_rate / 40000.0f) * 0.8f)) ||
        (config->max_burst_len && config->max_burst_len != (int)(config->sample_rate * 0.09)) ||
        (config->history_size && config->history_size != 512)) {
        fprintf(stderr, "irdm_hip: burst_detector_create: only the default detector geometry is supported\n");
        return nullptr;
    }
    _burst_detector *d = new _burst_detector();
    d->cfg = *config;
    d->p = nullptr;
    d->fmt = -1;
    d->start_time_ns = 0;
    d->total = 0;
    d->last_cb = nullptr;
    d->last_user = nullptr;
    d->stats_dirty = false;
    d->active = 0;
    d->noise_floor = 0.0f;             // burst_detect.c:364-365: no baseline yet
    d->peak_signal = 0.0f;
    return d;
}

static int detector_open(_burst_detector *d, int fmt)
{
    irdm_config_t c;
    memset(&c, 0, sizeof(c));
    c.center_frequency = d->cfg.center_frequency;
    c.sample_rate = d->cfg.sample_rate;
    c.threshold_db = d->cfg.threshold;
    c.format = fmt;
    c.feed_block = kBlock;
    c.use_gardner = gardner_setting();
    c.start_time_ns = 0;                           // wall clock at the first samples, as burst_detect.c:849-853
    c.max_chunk_samples = 64 * (size_t)kBlock;
    c.max_bursts_per_chunk = 1024;
    c.pipeline_depth = 0;
    d->p = irdm_create(&c);
    if (!d->p) return -1;
    irdm_set_option(d->p, "detect_only", 1);
    d->fmt = fmt;
    return 0;
}

// the bursts a feed finished -> malloc'd burst_data_t through the callback (ownership passes, burst_detect.h:69-71)
static void detector_emit(_burst_detector *d, int emitted, burst_callback_t cb, void *user)
{
    std::vector<irdm_burst_t> recs((size_t)(emitted > 0 ? emitted : 0));
    const int got = emitted > 0 ? irdm_poll_bursts(d->p, recs.data(), emitted) : 0;
    for (int i = 0; i < got; i++) {
        const irdm_burst_t &r = recs[i];
        burst_data_t *b = static_cast<burst_data_t *>(malloc(sizeof(burst_data_t)));
        float *s = static_cast<float *>(malloc(sizeof(float) * 2 * (size_t)r.num_samples));
        if (!b || !s || irdm_burst_samples(d->p, i, s, (size_t)r.num_samples) != (int)r.num_samples) {
            free(b);
            free(s);
            continue;
        }
        b->info.id = r.id; b->info.start = r.start; b->info.stop = r.stop; b->info.last_active = r.last_active;
        b->info.center_bin = r.center_bin; b->info.magnitude = r.magnitude; b->info.noise = r.noise;
        b->center_frequency = d->cfg.center_frequency;
        b->sample_rate = d->cfg.sample_rate;
        b->fft_size = irdm_fft_size(d->p);
        b->start_time_ns = irdm_start_time_ns(d->p);
        b->num_samples = (size_t)r.num_samples;
        b->samples = s;
        d->total++;
        if (cb) cb(b, user);                   // ownership of b and b->samples passes to the callee
        else { free(s); free(b); }
    }
    d->stats_dirty = true;
}

static void detector_feed(_burst_detector *d, const void *iq, size_t num_samples, int fmt, burst_callback_t cb, void *user)
{
    if (!d || !iq || !num_samples) return;
    std::lock_guard<std::recursive_mutex> lk(d->mu);
    if (!d->p && detector_open(d, fmt) != 0) {
        fprintf(stderr, "irdm_hip: burst_detector_feed: no device context\n");
        return;
    }
    if (fmt != d->fmt) {
        fprintf(stderr, "irdm_hip: burst_detector_feed: a detector takes one sample format\n");
        return;
    }
    d->last_cb = cb;
    d->last_user = user;
    const size_t bps = fmt == IRDM_FMT_CF32 ? 8 : 2;
    const unsigned char *src = static_cast<const unsigned char *>(iq);
    d->stage.insert(d->stage.end(), src, src + num_samples * bps);
    const size_t whole = d->stage.size() / bps / kBlock * kBlock;
    size_t off = 0;
    while (off < whole) {
        const size_t n = whole - off < 64 * (size_t)kBlock ? whole - off : 64 * (size_t)kBlock;
        const int emitted = irdm_feed_host(d->p, d->stage.data() + off * bps, n);
        off += n;
        if (emitted < 0) {
            fprintf(stderr, "irdm_hip: burst_detector_feed: device path failed, samples dropped\n");
            break;
        }
        detector_emit(d, emitted, cb, user);
    }
    d->stage.erase(d->stage.begin(), d->stage.begin() + off * bps);
}

extern "C" void burst_detector_feed(_burst_detector *det, const int8_t *iq, size_t num_samples, burst_callback_t cb, void *user)
{
    detector_feed(det, iq, num_samples, IRDM_FMT_CI8, cb, user);
}

extern "C" void burst_detector_feed_cf32(_burst_detector *det, const float *iq, size_t num_samples, burst_callback_t cb,
                                         void *user)
{
    detector_feed(det, iq, num_samples, IRDM_FMT_CF32, cb, user);
}

extern "C" uint64_t burst_detector_total_count(_burst_detector *det) { return det ? det->total : 0; }

// burst_detect.c:355-395, as main.c's stats thread calls them (any thread; values as of the end of the last feed)
extern "C" int burst_detector_active_count(_burst_detector *det)
{
    if (!det) return 0;
    std::lock_guard<std::recursive_mutex> lk(det->mu);
    detector_refresh(det);
    return det->active;
}
extern "C" float burst_detector_noise_floor(_burst_detector *det)
{
    if (!det) return 0.0f;
    std::lock_guard<std::recursive_mutex> lk(det->mu);
    detector_refresh(det);
    return det->noise_floor;
}
extern "C" float burst_detector_peak_signal(_burst_detector *det)
{
    if (!det) return 0.0f;
    std::lock_guard<std::recursive_mutex> lk(det->mu);
    detector_refresh(det);
    return det->peak_signal;
}

extern "C" void burst_detector_destroy(_burst_detector *det)
{
    if (!det) return;
    det->mu.lock();
    if (det->p && !det->stage.empty()) {
        // The end of the stream: the samples that never made up a whole 32768-sample block.  The reference's reader
        // feeds its short last read like any other (main.c:223-271) and the detector processes every whole frame of it
        // (burst_detect.c:746-842), so bursts that expire there are emitted; destroy is called by the detector thread
        // right behind its feed loop (burst_detect.c:941-960), with the burst queue still open: they go through the
        // callback of the last feed.
        const size_t bps = det->fmt == IRDM_FMT_CF32 ? 8 : 2;
        const int emitted = irdm_feed_host(det->p, det->stage.data(), det->stage.size() / bps);
        if (emitted >= 0) detector_emit(det, emitted, det->last_cb, det->last_user);
        det->stage.clear();
    }
    if (det->p) {
        // burst_detect.c:350-351: the line the reference's test script greps
        fprintf(stderr, "burst_detect: tagged %llu bursts total\n", (unsigned long long)irdm_tagged_bursts(det->p));
        irdm_destroy(det->p);
    }
    det->mu.unlock();
    delete det;
}

// ---------------------------------------------------------------------------------------------------------------
// stages B and C.  The reference gives every downmix worker a context of its own because the context IS the scratch
// memory (burst_downmix.c:107-112) and qpsk_demod() takes none.  Here a handle is a few words: all workers and the
// demodulator share ONE device context per capture sample rate behind a mutex (a device context holds the rotator
// table, the history ring and the per-burst scratch: one of them instead of five for main.c's layout, main.c:175).
// ---------------------------------------------------------------------------------------------------------------
namespace {
std::mutex g_stage_mu;
struct StageCtx {
    irdm_pipeline_t *p;
    int sample_rate;
};
std::vector<StageCtx> g_stage;
std::vector<float> g_row;                          // one frame's samples, guarded by g_stage_mu
irdm_demod_t g_demod_rec;                          // 4.5 KB: kept off the stack, guarded by g_stage_mu

// the context for `sample_rate` (0: any, the demodulator works at the downmixer's 250 kHz whatever the capture rate);
// g_stage_mu held
irdm_pipeline_t *stage_ctx(int sample_rate, double center_frequency)
{
    for (const StageCtx &c : g_stage)
        if (!sample_rate || c.sample_rate == sample_rate) return c.p;
    irdm_config_t c;
    memset(&c, 0, sizeof(c));
    c.center_frequency = center_frequency;
    c.sample_rate = sample_rate ? sample_rate : 2000000;
    c.format = IRDM_FMT_CF32;
    c.use_gardner = gardner_setting();
    c.start_time_ns = 1;
    c.max_chunk_samples = kBlock;
    c.max_bursts_per_chunk = 16;
    irdm_pipeline_t *p = irdm_create(&c);
    if (!p) return nullptr;
    g_stage.push_back(StageCtx{ p, c.sample_rate });
    if (g_row.empty()) g_row.assign(2 * IRDM_MAX_FRAME_SAMPLES, 0.0f);
    return p;
}
}  // namespace

struct _burst_downmix {
    downmix_config_t cfg;
};

extern "C" _burst_downmix *burst_downmix_create(downmix_config_t *config)
{
    // burst_downmix.c:228-239: 0 selects the default; only the defaults are implemented (250 kHz = 10 samples per
    // symbol, search depth = the output rate; handle_multiple_frames is stored and never read by the reference either)
    if (config && ((config->output_sample_rate && config->output_sample_rate != 250000) ||
                   (config->search_depth && config->search_depth != 250000))) {
        fprintf(stderr, "irdm_hip: burst_downmix_create: only the default downmix configuration is supported\n");
        return nullptr;
    }
    _burst_downmix *dm = new _burst_downmix();
    if (config) dm->cfg = *config;
    else memset(&dm->cfg, 0, sizeof(dm->cfg));
    return dm;
}

extern "C" int burst_downmix_process(_burst_downmix *dm, burst_data_t *burst, downmix_frame_t **frames_out)
{
    if (frames_out) *frames_out = nullptr;
    if (!dm || !burst || !frames_out || !burst->samples || burst->sample_rate <= 0) return 0;
    irdm_burst_t info;
    memset(&info, 0, sizeof(info));
    info.id = burst->info.id; info.start = burst->info.start; info.stop = burst->info.stop;
    info.last_active = burst->info.last_active; info.center_bin = burst->info.center_bin;
    info.magnitude = burst->info.magnitude; info.noise = burst->info.noise;
    info.num_samples = burst->num_samples;
    irdm_frame_info_t fi;
    std::vector<float> frame;
    {
        std::lock_guard<std::mutex> lk(g_stage_mu);
        irdm_pipeline_t *p = stage_ctx(burst->sample_rate, burst->center_frequency);
        if (!p) return 0;
        // every burst carries its stream's centre frequency and start time (burst_detect.h:40-48, burst_downmix.c:659-671)
        if (irdm_set_stream_origin(p, burst->center_frequency, burst->start_time_ns ? burst->start_time_ns : 1) != 0) return 0;
        const int rc = irdm_downmix_burst(p, &info, burst->samples, burst->num_samples, &fi, g_row.data());
        if (rc != 1) return 0;
        frame.assign(g_row.begin(), g_row.begin() + 2 * (size_t)fi.num_samples);
    }
    downmix_frame_t *f = static_cast<downmix_frame_t *>(malloc(sizeof(downmix_frame_t)));
    float *s = static_cast<float *>(malloc(sizeof(float) * 2 * (size_t)fi.num_samples));
    if (!f || !s) {
        free(f);
        free(s);
        return 0;
    }
    memcpy(s, frame.data(), sizeof(float) * 2 * (size_t)fi.num_samples);
    f->id = fi.id; f->timestamp = fi.timestamp; f->center_frequency = fi.center_frequency;
    f->sample_rate = fi.sample_rate; f->samples_per_symbol = fi.samples_per_symbol; f->direction = fi.direction;
    f->magnitude = fi.magnitude; f->noise = fi.noise; f->uw_start = fi.uw_start;
    f->num_samples = (size_t)fi.num_samples;
    f->samples = s;
    *frames_out = f;
    return 1;
}

extern "C" void burst_downmix_destroy(_burst_downmix *dm)
{
    delete dm;                  // (the shared device context lives until irdm_compat_shutdown)
}

// ---------------------------------------------------------------------------------------------------------------
// stage C: qpsk_demod() has no context argument
// ---------------------------------------------------------------------------------------------------------------
extern "C" int qpsk_demod(downmix_frame_t *in, demod_frame_t **out)
{
    if (out) *out = nullptr;
    if (!in || !out || !in->samples || in->num_samples > IRDM_MAX_FRAME_SAMPLES) return 0;
    std::lock_guard<std::mutex> lk(g_stage_mu);
    irdm_pipeline_t *p = stage_ctx(0, 1622000000.0);
    if (!p) return 0;
    memcpy(g_row.data(), in->samples, sizeof(float) * 2 * in->num_samples);
    const int ns = (int)in->num_samples, dir = in->direction;
    irdm_demod_t &d = g_demod_rec;
    if (irdm_qpsk_demod_batch(p, g_row.data(), &ns, &dir, 1, &d) != 0 || !d.ok) return 0;
    demod_frame_t *f = static_cast<demod_frame_t *>(calloc(1, sizeof(demod_frame_t)));
    uint8_t *bits = static_cast<uint8_t *>(malloc(d.n_bits > 0 ? (size_t)d.n_bits : 1));
    float *llr = static_cast<float *>(malloc(sizeof(float) * (d.n_bits > 0 ? (size_t)d.n_bits : 1)));
    if (!f || !bits || !llr) {
        free(f);
        free(bits);
        free(llr);
        return 0;
    }
    in->direction = d.direction;                   // qpsk_demod.c:454-463
    memcpy(bits, d.bits, (size_t)d.n_bits);
    memcpy(llr, d.llr, sizeof(float) * (size_t)d.n_bits);
    f->id = in->id; f->timestamp = in->timestamp; f->direction = in->direction;
    f->magnitude = in->magnitude; f->noise = in->noise;
    f->confidence = d.confidence; f->level = d.level;
    f->n_symbols = d.n_symbols; f->n_payload_symbols = d.n_symbols - 12;
    f->bits = bits; f->llr = llr; f->n_bits = d.n_bits;
    if (d.n_symbols > 0) {                         // qpsk_demod.c:521-527
        const double duration = (double)d.n_symbols / 25000;
        f->center_frequency = in->center_frequency + d.total_phase / duration / M_PI / 2.0;
    } else {
        f->center_frequency = in->center_frequency;
    }
    *out = f;
    return 1;
}

extern "C" void irdm_compat_shutdown(void)
{
    std::lock_guard<std::mutex> lk(g_stage_mu);
    for (StageCtx &c : g_stage) irdm_destroy(c.p);
    g_stage.clear();
}
