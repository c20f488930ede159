// compat.cpp -- the reference's stage-level entry points (burst_detect.h:67-94, burst_downmix.h:64-73,
// qpsk_demod.h:42) as thin adapters over the batched C-ABI (include/irdm_hip.h).  See include/irdm_compat.h for the
// contract; the struct mirrors below have the reference's layouts with `float complex *` spelled `float *`.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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
};

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

static void detector_feed(_burst_detector *d, const void *iq, size_t num_samples, int fmt, burst_callback_t cb, void *user)
{
    if (!d || !iq || !num_samples) return;
    if (!d->p && detector_open(d, fmt) != 0) {
        fprintf(stderr, "irdm_hip: burst_detector_feed: no device context\n");
        return;
    }
    if (fmt != d->fmt) {
        fprintf(stderr, "irdm_hip: burst_detector_feed: a detector takes one sample format\n");
        return;
    }
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
        std::vector<irdm_burst_t> recs((size_t)emitted);
        const int got = emitted ? irdm_poll_bursts(d->p, recs.data(), emitted) : 0;
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

extern "C" void burst_detector_destroy(_burst_detector *det)
{
    if (!det) return;
    if (det->p) {
        // burst_detect.c:350-351: the line the reference's test script greps
        fprintf(stderr, "burst_detect: tagged %llu bursts total\n", (unsigned long long)irdm_tagged_bursts(det->p));
        irdm_destroy(det->p);
    }
    delete det;
}

// ---------------------------------------------------------------------------------------------------------------
// stage B: one context per worker thread, as in the reference (burst_downmix.c:107-112); the device context is created
// at the first burst (its sample rate is not known earlier)
// ---------------------------------------------------------------------------------------------------------------
struct _burst_downmix {
    downmix_config_t cfg;
    irdm_pipeline_t *p;
    int sample_rate;
    std::vector<float> frame;
};

extern "C" _burst_downmix *burst_downmix_create(downmix_config_t *config)
{
    _burst_downmix *dm = new _burst_downmix();
    if (config) dm->cfg = *config;
    else memset(&dm->cfg, 0, sizeof(dm->cfg));
    dm->p = nullptr;
    dm->sample_rate = 0;
    dm->frame.resize(2 * IRDM_MAX_FRAME_SAMPLES);
    return dm;
}

extern "C" int burst_downmix_process(_burst_downmix *dm, burst_data_t *burst, downmix_frame_t **frames_out)
{
    if (frames_out) *frames_out = nullptr;
    if (!dm || !burst || !frames_out || !burst->samples) return 0;
    if (!dm->p) {
        irdm_config_t c;
        memset(&c, 0, sizeof(c));
        c.center_frequency = burst->center_frequency;
        c.sample_rate = burst->sample_rate;
        c.format = IRDM_FMT_CF32;
        c.use_gardner = gardner_setting();
        c.start_time_ns = burst->start_time_ns ? burst->start_time_ns : 1;
        c.max_chunk_samples = kBlock;
        c.max_bursts_per_chunk = 16;
        dm->p = irdm_create(&c);
        dm->sample_rate = burst->sample_rate;
        if (!dm->p) return 0;
    }
    if (burst->sample_rate != dm->sample_rate) return 0;
    irdm_burst_t info;
    memset(&info, 0, sizeof(info));
    info.id = burst->info.id; info.start = burst->info.start; info.stop = burst->info.stop;
    info.last_active = burst->info.last_active; info.center_bin = burst->info.center_bin;
    info.magnitude = burst->info.magnitude; info.noise = burst->info.noise;
    info.num_samples = burst->num_samples;
    irdm_frame_info_t fi;
    const int rc = irdm_downmix_burst(dm->p, &info, burst->samples, burst->num_samples, &fi, dm->frame.data());
    if (rc != 1) return 0;
    downmix_frame_t *f = static_cast<downmix_frame_t *>(malloc(sizeof(downmix_frame_t)));
    float *s = static_cast<float *>(malloc(sizeof(float) * 2 * (size_t)fi.num_samples));
    if (!f || !s) {
        free(f);
        free(s);
        return 0;
    }
    memcpy(s, dm->frame.data(), sizeof(float) * 2 * (size_t)fi.num_samples);
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
    if (!dm) return;
    if (dm->p) irdm_destroy(dm->p);
    delete dm;
}

// ---------------------------------------------------------------------------------------------------------------
// stage C: qpsk_demod() has no context argument; one process-wide device context, created on first use
// ---------------------------------------------------------------------------------------------------------------
namespace {
std::mutex g_demod_mu;
irdm_pipeline_t *g_demod = nullptr;
std::vector<float> g_demod_row;
}  // namespace

extern "C" int qpsk_demod(downmix_frame_t *in, demod_frame_t **out)
{
    if (out) *out = nullptr;
    if (!in || !out || !in->samples || in->num_samples > IRDM_MAX_FRAME_SAMPLES) return 0;
    std::lock_guard<std::mutex> lk(g_demod_mu);
    if (!g_demod) {
        irdm_config_t c;
        memset(&c, 0, sizeof(c));
        c.center_frequency = 1622000000.0;
        c.sample_rate = 2000000;                   // stage C works at the downmixer's 250 kHz whatever the capture rate
        c.format = IRDM_FMT_CF32;
        c.use_gardner = gardner_setting();
        c.start_time_ns = 1;
        c.max_chunk_samples = kBlock;
        c.max_bursts_per_chunk = 16;
        g_demod = irdm_create(&c);
        if (!g_demod) return 0;
        g_demod_row.assign(2 * IRDM_MAX_FRAME_SAMPLES, 0.0f);
    }
    memcpy(g_demod_row.data(), in->samples, sizeof(float) * 2 * in->num_samples);
    const int ns = (int)in->num_samples, dir = in->direction;
    static irdm_demod_t d;                         // 4.5 KB: kept off the stack, guarded by g_demod_mu
    if (irdm_qpsk_demod_batch(g_demod, g_demod_row.data(), &ns, &dir, 1, &d) != 0 || !d.ok) return 0;
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
    std::lock_guard<std::mutex> lk(g_demod_mu);
    if (g_demod) irdm_destroy(g_demod);
    g_demod = nullptr;
}
