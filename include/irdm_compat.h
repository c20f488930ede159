/*
 * irdm_compat.h -- the reference's three stage-level entry points, with the reference's own names, argument types and
 * ownership rules, exported by libirdm_hip.so as thin adapters over the batched C-ABI of irdm_hip.h.  A host that is
 * written against the reference's stage API (main.c's worker threads, main.c:307-325, :371-373, :801-823) links against
 * these unchanged:
 *
 *   stage A   burst_detector_create / _feed / _feed_cf32 / _destroy, burst_callback_t      (burst_detect.h:67-94)
 *   stage B   burst_downmix_create / _process / _destroy                                   (burst_downmix.h:64-73)
 *   stage C   qpsk_demod                                                                   (qpsk_demod.h:42)
 *
 * The struct declarations below restate the reference's layouts (x86-64 SysV; field order and types of
 * burst_detect.h:29-65, burst_downmix.h:24-62, qpsk_demod.h:24-38) so that this header can stand in for the three
 * reference headers when the reference's CPU stages are not compiled in.  Do not include it together with them.
 *
 * Ownership, exactly as in the reference:
 *   - the burst callback receives a malloc'd burst_data_t and a malloc'd burst->samples; the callee frees both
 *     (burst_detect.h:69-71);
 *   - burst_downmix_process returns 0 or 1; on 1, *frames_out is a malloc'd downmix_frame_t whose samples are
 *     malloc'd, caller frees both; the burst is not freed (burst_downmix.c:812-823);
 *   - qpsk_demod returns 1 and a calloc'd demod_frame_t with malloc'd bits and llr (caller frees all three,
 *     main.c:371-373), or 0 and nothing allocated; in->direction is updated as qpsk_demod.c:444, :454-463 do.
 * Globals the reference's stage C reads: `use_gardner` (qpsk_demod.c:34) is picked up if the host defines it
 * (weak reference; default 1).  `save_bursts_dir` is not consulted here (irdm_save_burst in irdm_hip.h).
 *
 * Results equal the reference's when the detector is fed in the file reader's blocks of 32768 samples
 * (main.c:225): the adapter stages input to whole blocks of that size (irdm_config_t.feed_block).
 * Not provided: the *_thread functions (they are main.c's queue plumbing around these entry points) and non-default
 * burst_config_t geometry (fft_size, pre/post lengths, burst_width, max_bursts, max_burst_len, history_size other
 * than 0 / the defaults make burst_detector_create return NULL).
 */
#ifndef IRDM_COMPAT_H
#define IRDM_COMPAT_H

#include <complex.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct _burst_detector;
typedef struct _burst_detector burst_detector_t;

typedef struct {                /* burst_detect.h:29-37 */
    uint64_t id;
    uint64_t start;
    uint64_t stop;
    uint64_t last_active;
    int center_bin;
    float magnitude;
    float noise;
} burst_info_t;

typedef struct {                /* burst_detect.h:40-48 */
    burst_info_t info;
    double center_frequency;
    int sample_rate;
    int fft_size;
    uint64_t start_time_ns;
    size_t num_samples;
    float complex *samples;
} burst_data_t;

typedef struct {                /* burst_detect.h:51-63 */
    double center_frequency;
    int sample_rate;
    int fft_size;
    int burst_pre_len;
    int burst_post_len;
    int burst_width;
    int max_bursts;
    int max_burst_len;
    float threshold;
    int history_size;
    int use_gpu;
} burst_config_t;

typedef void (*burst_callback_t)(burst_data_t *burst, void *user);

burst_detector_t *burst_detector_create(burst_config_t *config);
void burst_detector_feed(burst_detector_t *det, const int8_t *iq, size_t num_samples, burst_callback_t cb, void *user);
void burst_detector_feed_cf32(burst_detector_t *det, const float *iq, size_t num_samples, burst_callback_t cb, void *user);
uint64_t burst_detector_total_count(burst_detector_t *det);
int burst_detector_active_count(burst_detector_t *det);       /* burst_detect.h:82; values as of the end of the last feed */
float burst_detector_noise_floor(burst_detector_t *det);      /* burst_detect.h:88 */
float burst_detector_peak_signal(burst_detector_t *det);      /* burst_detect.h:91 */
void burst_detector_destroy(burst_detector_t *det);

typedef enum {                  /* burst_downmix.h:24-28 */
    DIR_UNDEF = 0,
    DIR_DOWNLINK = 1,
    DIR_UPLINK = 2,
} ir_direction_t;

typedef struct {                /* burst_downmix.h:31-43 */
    uint64_t id;
    uint64_t timestamp;
    double center_frequency;
    float sample_rate;
    float samples_per_symbol;
    ir_direction_t direction;
    float magnitude;
    float noise;
    float uw_start;
    size_t num_samples;
    float complex *samples;
} downmix_frame_t;

typedef struct _burst_downmix burst_downmix_t;

typedef struct {                /* burst_downmix.h:49-53 */
    int output_sample_rate;
    int search_depth;
    int handle_multiple_frames;
} downmix_config_t;

burst_downmix_t *burst_downmix_create(downmix_config_t *config);
int burst_downmix_process(burst_downmix_t *dm, burst_data_t *burst, downmix_frame_t **frames_out);
void burst_downmix_destroy(burst_downmix_t *dm);

typedef struct {                /* qpsk_demod.h:24-38 */
    uint64_t id;
    uint64_t timestamp;
    double center_frequency;
    ir_direction_t direction;
    float magnitude;
    float noise;
    int confidence;
    float level;
    int n_symbols;
    int n_payload_symbols;
    uint8_t *bits;
    float *llr;
    int n_bits;
} demod_frame_t;

int qpsk_demod(downmix_frame_t *in, demod_frame_t **out);

/* release the process-wide stage-C context qpsk_demod creates on first use (optional; not in the reference) */
void irdm_compat_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif
