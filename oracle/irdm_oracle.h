/*
 * irdm_oracle.h -- CPU ORACLE for the Iridium burst-detect / downmix / DQPSK path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (iridium-sniffer_amd/)
 * may include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, as the checker / reported CPU baseline.
 *
 * This is a plain-C restatement of the reference's scalar (--no-simd --no-gpu)
 * algorithm; each function in irdm_oracle.c cites the reference file:line it
 * follows.  The one arithmetic substitution is the FFT: the reference calls
 * FFTW3 (not vendored, not installed, version unpinned -> "parity unpinned" at
 * that boundary); the oracle uses the repo's pinned radix-2 DIT float32 FFT
 * (definition in DESIGN.md section "Pinned FFT"), which is checked against
 * numpy.fft (double) in tests/test_oracle_fft.py.
 *
 * Pinning status: every non-FFT function is checked bit-for-bit against the
 * reference's own sources compiled in place (oracle/_ref, see oracle/Makefile)
 * by tests/test_oracle_vs_ref.py; whole stage C (qpsk_demod.c) likewise.
 */
#ifndef IRDM_ORACLE_H
#define IRDM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- records (flat, ctypes-friendly) ---- */

typedef struct {
    uint64_t id;
    uint64_t start;
    uint64_t stop;
    uint64_t last_active;
    int32_t center_bin;
    float magnitude;     /* dB, burst_detect.c:572 */
    float noise;         /* dBFS/Hz, burst_detect.c:583-586 */
    float peak_rel;      /* raw relative magnitude of the creating peak */
    float base_sum;      /* baseline_sum[center_bin] at creation */
    uint64_t num_samples;
    uint64_t avail_end;  /* sample_count when the burst was extracted */
} orc_burst_rec_t;

#define ORC_MAX_FRAME_SAMPLES 4440
#define ORC_MAX_SYMBOLS 448
#define ORC_MAX_BITS (2 * ORC_MAX_SYMBOLS)

typedef struct {
    uint64_t id;
    uint64_t timestamp;
    double center_frequency;
    float sample_rate;
    float samples_per_symbol;
    int32_t direction;
    float magnitude;
    float noise;
    float uw_start;          /* sub-sample correction */
    int32_t num_samples;
    /* stage probes (not in the reference struct; for stage-level parity tests) */
    int32_t dec_len;
    int32_t start;
    float center_offset;
    int32_t uw_start_idx;
    float corr_re, corr_im;
    int32_t drop_reason;     /* 0 = frame produced */
    float samples[2 * ORC_MAX_FRAME_SAMPLES];
} orc_frame_t;

typedef struct {
    uint64_t id;
    uint64_t timestamp;
    double center_frequency;
    int32_t direction;
    float magnitude;
    float noise;
    int32_t confidence;
    float level;
    int32_t n_symbols;
    int32_t n_payload_symbols;
    int32_t n_bits;
    int32_t ok;              /* 1 if UW check passed (frame emitted) */
    float total_phase;
    uint8_t bits[ORC_MAX_BITS];
    float llr[ORC_MAX_BITS];
} orc_demod_t;

/* ---- pinned FFT ---- */
/* In-place complex FFT of n (power of two) interleaved floats.
 * dir = -1 forward (e^{-j...}), +1 backward (unnormalised). */
void orc_fft(float *data, int n, int dir);

/* ---- DSP primitives restated (simd_generic.c, fir_filter.c, window_func.c, rotator.h) ---- */
void orc_fir_ccf(const float *taps, int ntaps, const float *in, float *out, int n);
void orc_fir_ccf_dec(const float *taps, int ntaps, const float *in, float *out,
                     int n_out, int decimation);
/* simd_avx2.c:62-108; selection of the form stage B uses (0 scalar = --no-simd, 1 AVX2 + FMA = the reference's default) */
void orc_fir_ccf_dec_avx2(const float *taps, int ntaps, const float *in, float *out, int n_out, int decimation);
void orc_fir_ccf_avx2(const float *taps, int ntaps, const float *in, float *out, int n);
void orc_fir_fff_avx2(const float *taps, int ntaps, const float *in, float *out, int n);
void orc_fftshift_mag_avx2(const float *fft_out, float *mag_shifted, int fft_size);
void orc_mag_squared_avx2(const float *in, float *out, int n);
void orc_set_fir_order(int order);
int orc_get_fir_order(void);
void orc_fir_fff(const float *taps, int ntaps, const float *in, float *out, int n);
void orc_window_cf(const float *samples, const float *window, float *out, int n);
void orc_fftshift_mag(const float *fft_out, float *mag_shifted, int fft_size);
void orc_baseline_update(float *sum, const float *old_hist, const float *new_mag, int n);
void orc_relative_mag(const float *mag, const float *baseline, float *out, int n);
void orc_convert_i8_cf(const int8_t *iq, float *out, size_t n);
void orc_mag_squared(const float *in, float *out, int n);
float orc_max_float(const float *in, int n);
void orc_csquare_window(const float *in, const float *window, float *out, int n);
void orc_blackman_window(float *w, int n);
int orc_lpf_taps(float *out, int cap, float gain, float sample_rate, float cutoff,
                 float transition);
int orc_rrc_taps(float *out, int cap, float gain, float sample_rate, float symbol_rate,
                 float alpha, int ntaps);
int orc_rc_taps(float *out, int cap, float sample_rate, float symbol_rate, float alpha,
                int ntaps);
int orc_box_taps(float *out, int cap, int length);
/* rotator: phase/incr are (re,im) pairs; phase updated in place incl. the final renorm */
void orc_rotator_rotate_n(float *phase, const float *incr, float *out, const float *in, int n);

/* ---- stage A: burst detector ---- */
typedef struct orc_detector orc_detector_t;
typedef void (*orc_burst_cb)(const orc_burst_rec_t *rec, const float *samples, void *user);

/* threshold_db <= 0 -> 16 dB default.  fft_size 0 -> auto. */
orc_detector_t *orc_detector_create(double center_frequency, int sample_rate,
                                    float threshold_db, int fft_size);
void orc_detector_destroy(orc_detector_t *d);
void orc_detector_feed_cf32(orc_detector_t *d, const float *iq, size_t n,
                            orc_burst_cb cb, void *user);
void orc_detector_feed_i8(orc_detector_t *d, const int8_t *iq, size_t n,
                          orc_burst_cb cb, void *user);
/* feed precomputed magnitude frames (the reference's GPU branch,
 * burst_detect.c:637-674): used to test gpu_burst_fft_* drop-in */
int orc_detector_fft_size(const orc_detector_t *d);
uint64_t orc_detector_tagged(const orc_detector_t *d);
/* probes for stage-level parity */
const float *orc_detector_baseline_sum(const orc_detector_t *d);
const float *orc_detector_last_magnitude(const orc_detector_t *d);
/* compute one magnitude frame with the detector's window + pinned FFT */
void orc_detector_magnitude_frame(orc_detector_t *d, const float *iq_frame, float *mag_out);
/* optional: record every magnitude frame (frames x fft_size floats) */
void orc_detector_set_mag_sink(orc_detector_t *d, float *sink, size_t max_frames);
size_t orc_detector_frames_done(const orc_detector_t *d);

/* ---- stage B: burst downmix ---- */
typedef struct orc_downmix orc_downmix_t;
orc_downmix_t *orc_downmix_create(void);
void orc_downmix_destroy(orc_downmix_t *dm);
/* returns 1 and fills *out if a frame was produced, else 0 (out->drop_reason set) */
int orc_downmix_process(orc_downmix_t *dm, const orc_burst_rec_t *rec, const float *samples,
                        double center_frequency, int sample_rate, int fft_size,
                        uint64_t start_time_ns, orc_frame_t *out);
/* design outputs for upload-parity checks */
const float *orc_downmix_taps(const orc_downmix_t *dm, int which, int *ntaps);
const float *orc_downmix_sync_fft(const orc_downmix_t *dm, int uplink, int *sync_len);
const float *orc_downmix_cfo_window(const orc_downmix_t *dm, int *n);

/* ---- stage C: qpsk demod ---- */
int orc_qpsk_demod(const orc_frame_t *in, int use_gardner, orc_demod_t *out);

/* ---- RAW line (frame_output.c:160-199) ---- */
/* t0 = 0 -> derive from this frame's timestamp as ensure_initialized does; returns
 * the t0 used via *t0_io.  Returns line length (incl. '\n'). */
int orc_format_raw(const orc_demod_t *f, const char *file_info, uint64_t *t0_io,
                   char *buf, size_t cap);

/* ---- post-demod bit layer: frame_decode() (frame_decode.c:414-598), SURVEY 8f row 3 ----
 * access code, de-interleave, BCH(31,21)/BCH(7,3) syndromes, Chase decoding on the LLRs, IRA / IBC field
 * extraction.  decoded_frame_t (frame_decode.h:26-60) flattened; bch_len = decoded data bits assembled. */
typedef struct {
    int32_t type;            /* frame_type_t: 0 unknown, 1 IRA, 2 IBC */
    int32_t sat_id, beam_id;
    int32_t pos_xyz[3];      /* IRA */
    int32_t alt;
    int32_t n_pages;
    double lat, lon;
    uint32_t page_tmsi[12];
    int32_t page_msc[12];
    int32_t timeslot, sv_blocking, bc_type;   /* IBC */
    uint32_t iri_time;
    int32_t bch_len;
    int32_t pad;
} orc_decoded_t;

/* bits: n_bits hard bits (access code first), llr: n_bits reliabilities or NULL.  Returns frame_decode()'s
 * return value (1 = IRA or IBC recognised). */
int orc_frame_decode(const uint8_t *bits, const float *llr, int n_bits, orc_decoded_t *out);

/* ida_decode() (ida_decode.c:543-665): LCW extraction (3 BCH codes behind a 46-bit permutation), payload descramble
 * (124-bit blocks -> 4 x BCH(31,20) with Chase decoding), IDA fields, CRC-CCITT, LCW header text.
 * ida_burst_t (ida_decode.h:31-56) flattened, minus the fields copied from the demod record. */
typedef struct {
    int32_t ok;                 /* ida_decode()'s return value */
    int32_t ft, lcw_ft, lcw_code, ec_lcw;
    uint32_t lcw3_val;
    int32_t da_ctr, da_len, cont, crc_ok;
    uint32_t stored_crc, computed_crc;
    int32_t fixederrs, payload_len, bch_len, pad;
    uint8_t payload[32];
    uint8_t bch_stream[256];
    char lcw_header[128];
} orc_ida_t;

/* direction: the demodulator's ir_direction_t (0 undefined -> rejected, ida_decode.c:551-552) */
int orc_ida_decode(const uint8_t *bits, const float *llr, int n_bits, int direction, orc_ida_t *out);
/* format_lcw_header (ida_decode.c:405-539): "LCW(ft,T:..,C:..,..)" padded to 110 characters + one space */
void orc_format_lcw_header(int ft, int lcw_ft, int lcw_code, uint32_t lcw3_val, char *out, int outsz);

/* ---- whole stream: detect -> downmix -> demod, reference file-mode plumbing ---- */
typedef struct {
    double center_frequency;
    int sample_rate;
    float threshold_db;   /* <=0: default */
    int format;           /* 0 ci8, 1 ci16, 2 cf32 */
    int block;            /* samples per feed call; 0 -> 32768 (main.c:225) */
    int use_gardner;      /* default 1 */
    uint64_t start_time_ns; /* replaces the wall clock (burst_detect.c:849-853); must be != 0 */
} orc_stream_cfg_t;

typedef struct {
    orc_burst_rec_t *bursts; size_t n_bursts, cap_bursts;
    orc_frame_t *frames;     size_t n_frames, cap_frames;   /* one per burst (drop_reason set) */
    orc_demod_t *demods;     size_t n_demods, cap_demods;   /* one per produced frame */
    uint64_t n_tagged;
    uint64_t n_samples;
} orc_stream_out_t;

/* caller provides the arrays (cap_*); returns 0, or -1 if a cap was exceeded */
int orc_run_stream(const void *iq, size_t n_samples, const orc_stream_cfg_t *cfg,
                   orc_stream_out_t *out);

#ifdef __cplusplus
}
#endif
#endif
