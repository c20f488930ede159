/*
 * compat_main.c -- TEST PROGRAM: drives libirdm_hip.so through the reference's stage-level API only
 * (include/irdm_compat.h), the way the reference's worker threads do (main.c:223-284 file blocks of 32768 samples ->
 * burst_detector_feed[_cf32]; burst callback -> burst_downmix_process (burst_downmix.c:812-823); frame -> qpsk_demod;
 * then the frees of main.c:371-373), single-threaded.  Prints one line per burst / frame / demodulated frame for
 * tests/test_gpu_compat.py to compare with the oracle.
 *
 *   compat_main FILE RATE ci8|cf32
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "irdm_compat.h"

int use_gardner = 1;            /* the global qpsk_demod.c:34 reads (main.c:143) */

static burst_downmix_t *g_dm;
static int n_bursts, n_frames, n_demods;

static void on_burst(burst_data_t *b, void *user)
{
    (void)user;
    n_bursts++;
    printf("B %llu %llu %llu %llu %d %.9g %.9g %zu\n", (unsigned long long)b->info.id, (unsigned long long)b->info.start,
           (unsigned long long)b->info.stop, (unsigned long long)b->info.last_active, b->info.center_bin,
           b->info.magnitude, b->info.noise, b->num_samples);
    downmix_frame_t *frame = NULL;
    if (burst_downmix_process(g_dm, b, &frame) > 0 && frame) {
        n_frames++;
        demod_frame_t *d = NULL;
        const int dir_in = frame->direction;
        if (qpsk_demod(frame, &d)) {
            n_demods++;
            printf("D %llu %llu %.3f %d %d %.9g %d %d %d ", (unsigned long long)d->id, (unsigned long long)d->timestamp,
                   d->center_frequency, d->direction, d->confidence, d->level, d->n_symbols, d->n_payload_symbols, d->n_bits);
            for (int i = 0; i < d->n_bits; i++) putchar('0' + d->bits[i]);
            printf(" %.9g %.9g\n", d->n_bits ? d->llr[0] : 0.0f, d->n_bits ? d->llr[d->n_bits - 1] : 0.0f);
            free(d->bits);
            free(d->llr);
            free(d);
        }
        printf("F %llu %llu %.3f %d %d %zu %.9g\n", (unsigned long long)frame->id, (unsigned long long)frame->timestamp,
               frame->center_frequency, dir_in, frame->direction, frame->num_samples, frame->uw_start);
        free(frame->samples);
        free(frame);
    }
    free(b->samples);
    free(b);
}

int main(int argc, char **argv)
{
    if (argc != 4) {
        fprintf(stderr, "usage: %s FILE RATE ci8|cf32\n", argv[0]);
        return 2;
    }
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    const int rate = atoi(argv[2]);
    const int cf32 = strcmp(argv[3], "cf32") == 0;
    burst_config_t cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.center_frequency = 1622000000.0;
    cfg.sample_rate = rate;
    cfg.use_gpu = 1;
    burst_detector_t *det = burst_detector_create(&cfg);
    downmix_config_t dc;
    memset(&dc, 0, sizeof(dc));
    g_dm = burst_downmix_create(&dc);
    if (!det || !g_dm) return 1;
    const size_t bps = cf32 ? 8 : 2;
    void *blk = malloc(32768 * bps);
    size_t n;
    while ((n = fread(blk, bps, 32768, f)) > 0) {
        if (cf32) burst_detector_feed_cf32(det, (const float *)blk, n, on_burst, NULL);
        else burst_detector_feed(det, (const int8_t *)blk, n, on_burst, NULL);
    }
    /* what main.c's stats thread reads (main.c:455-456) */
    printf("S %d %.9g %.9g\n", burst_detector_active_count(det), burst_detector_noise_floor(det), burst_detector_peak_signal(det));
    const unsigned long long before = (unsigned long long)burst_detector_total_count(det);
    burst_detector_destroy(det);        /* delivers the bursts of the stream's last, short block */
    printf("T %d %d %d %d %llu\n", n_bursts, n_bursts, n_frames, n_demods, before);
    burst_downmix_destroy(g_dm);
    irdm_compat_shutdown();
    free(blk);
    fclose(f);
    return 0;
}
