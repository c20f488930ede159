/*
 * ref_glue.c -- thin ctypes-facing shims over the REFERENCE's own sources.
 *
 * TEST INFRASTRUCTURE.  This file contains no algorithm: it defines the two
 * globals qpsk_demod.c expects from main.c (qpsk_demod.c:34-35), and flattens
 * struct/pointer interfaces into plain arrays so tests can call the reference
 * through ctypes.  It is compiled together with the reference sources where
 * they lie under /root/reference (see oracle/Makefile, target _ref); the
 * result goes to oracle/_ref/ only and is never part of the product.
 */
#include <complex.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <stdio.h>
#include <unistd.h>

#include "qpsk_demod.h"   /* reference header (burst_detect.h neutralised by the recipe's -D flags) */
#include "frame_output.h" /* reference header */
#include "frame_decode.h" /* reference header */
#include "ida_decode.h"   /* reference header */
#include "rotator.h"      /* reference header */
#include "simd_kernels.h" /* reference header */

char *save_bursts_dir = NULL;   /* main.c global read by qpsk_demod.c:34 */
int use_gardner = 1;            /* main.c:143 global read by qpsk_demod.c:35 */

int diagnostic_mode = 0;        /* main.c globals read by frame_output.c:36-38 */
int parsed_mode = 0;
int acars_enabled = 0;

void ref_set_use_gardner(int v) { use_gardner = v; }

/* --save-bursts (main.c sets the global from options.c): qpsk_demod then calls its static save_burst_iq (qpsk_demod.c:339-389,
 * :441-444, :468-470) */
void ref_set_save_bursts_dir(const char *dir)
{
    free(save_bursts_dir);
    save_bursts_dir = dir ? strdup(dir) : NULL;
}

/* qpsk_demod with every downmix_frame_t field the .meta file prints (qpsk_demod.c:376-385); returns its return value and
 * the direction it left in the input frame (what the file name and the "direction:" line carry) */
int ref_qpsk_demod_save(const float *samples, int num_samples, float sample_rate, float samples_per_symbol, int direction,
                        double center_frequency, uint64_t id, uint64_t timestamp, float magnitude, float noise, float uw_start,
                        int *direction_left)
{
    downmix_frame_t in;
    memset(&in, 0, sizeof(in));
    in.id = id;
    in.timestamp = timestamp;
    in.center_frequency = center_frequency;
    in.sample_rate = sample_rate;
    in.samples_per_symbol = samples_per_symbol;
    in.direction = (ir_direction_t)direction;
    in.magnitude = magnitude;
    in.noise = noise;
    in.uw_start = uw_start;
    in.num_samples = (size_t)num_samples;
    in.samples = malloc(sizeof(float complex) * (size_t)num_samples);
    memcpy(in.samples, samples, sizeof(float complex) * (size_t)num_samples);
    demod_frame_t *out = NULL;
    const int r = qpsk_demod(&in, &out);
    *direction_left = (int)in.direction;
    free(in.samples);
    if (r) {
        free(out->bits);
        free(out->llr);
        free(out);
    }
    return r;
}

/* sizeof / offsetof of the reference's stage-level structs as its own headers declare them (burst_downmix.h:31-53,
 * qpsk_demod.h:24-38), for tests/test_ref_pins.py: name, value pairs */
#include <stddef.h>
#define REF_LAY(T, F) { #T "." #F, (long)offsetof(T, F) }
#define REF_SZ(T) { "sizeof " #T, (long)sizeof(T) }
static const struct { const char *name; long value; } ref_layout_tab[] = {
    REF_SZ(downmix_frame_t), REF_LAY(downmix_frame_t, id), REF_LAY(downmix_frame_t, timestamp),
    REF_LAY(downmix_frame_t, center_frequency), REF_LAY(downmix_frame_t, sample_rate),
    REF_LAY(downmix_frame_t, samples_per_symbol), REF_LAY(downmix_frame_t, direction), REF_LAY(downmix_frame_t, magnitude),
    REF_LAY(downmix_frame_t, noise), REF_LAY(downmix_frame_t, uw_start), REF_LAY(downmix_frame_t, num_samples),
    REF_LAY(downmix_frame_t, samples),
    REF_SZ(downmix_config_t), REF_LAY(downmix_config_t, output_sample_rate), REF_LAY(downmix_config_t, search_depth),
    REF_LAY(downmix_config_t, handle_multiple_frames),
    REF_SZ(demod_frame_t), REF_LAY(demod_frame_t, id), REF_LAY(demod_frame_t, timestamp),
    REF_LAY(demod_frame_t, center_frequency), REF_LAY(demod_frame_t, direction), REF_LAY(demod_frame_t, magnitude),
    REF_LAY(demod_frame_t, noise), REF_LAY(demod_frame_t, confidence), REF_LAY(demod_frame_t, level),
    REF_LAY(demod_frame_t, n_symbols), REF_LAY(demod_frame_t, n_payload_symbols), REF_LAY(demod_frame_t, bits),
    REF_LAY(demod_frame_t, llr), REF_LAY(demod_frame_t, n_bits),
    REF_SZ(ir_direction_t), { "DIR_UNDEF", DIR_UNDEF }, { "DIR_DOWNLINK", DIR_DOWNLINK }, { "DIR_UPLINK", DIR_UPLINK },
};
int ref_layout(int i, const char **name, long *value)
{
    if (i < 0 || i >= (int)(sizeof(ref_layout_tab) / sizeof(ref_layout_tab[0]))) return 0;
    *name = ref_layout_tab[i].name;
    *value = ref_layout_tab[i].value;
    return 1;
}

/* frame_output_print (frame_output.c:160-199) writes the RAW line to stdout; this shim hands it the fields through a
 * demod_frame_t and returns what it printed (fd 1 is pointed at a temporary file for the duration of the call).
 * frame_output.c keeps file_info and t0 in statics set by the FIRST frame of the process: call frame_output_init
 * through ref_frame_output_init before the first line, in a fresh process per stream. */
void ref_frame_output_init(const char *file_info) { frame_output_init(file_info); }

int ref_frame_output_line(uint64_t id, uint64_t timestamp, double center_frequency, float magnitude, float noise,
                          int confidence, float level, int n_payload_symbols, int n_bits, const uint8_t *bits,
                          char *out, int cap)
{
    demod_frame_t f;
    memset(&f, 0, sizeof(f));
    f.id = id;
    f.timestamp = timestamp;
    f.center_frequency = center_frequency;
    f.magnitude = magnitude;
    f.noise = noise;
    f.confidence = confidence;
    f.level = level;
    f.n_payload_symbols = n_payload_symbols;
    f.n_bits = n_bits;
    f.bits = (uint8_t *)bits;
    FILE *tmp = tmpfile();
    if (!tmp) return -1;
    fflush(stdout);
    const int saved = dup(1);
    dup2(fileno(tmp), 1);
    frame_output_print(&f);
    fflush(stdout);
    dup2(saved, 1);
    close(saved);
    rewind(tmp);
    const int n = (int)fread(out, 1, (size_t)cap - 1, tmp);
    out[n > 0 ? n : 0] = 0;
    fclose(tmp);
    return n;
}

/* rotator.h:36-46 (static inline in the reference) */
void ref_rotator_rotate_n(float *phase, const float *incr, float *out, const float *in, int n)
{
    rotator_t r;
    rotator_init(&r);
    rotator_set_phase(&r, phase[0] + phase[1] * I);
    rotator_set_phase_incr(&r, incr[0] + incr[1] * I);
    rotator_rotate_n(&r, (float complex *)out, (const float complex *)in, n);
    phase[0] = crealf(r.phase);
    phase[1] = cimagf(r.phase);
}

/* qpsk_demod.c:393 through a flat interface.  Returns the reference's return value. */
int ref_qpsk_demod(const float *samples, int num_samples, float samples_per_symbol,
                   int direction, double center_frequency, uint64_t id, uint64_t timestamp,
                   float magnitude, float noise,
                   int *direction_out, int *confidence, float *level, int *n_symbols,
                   int *n_payload, int *n_bits, uint8_t *bits, float *llr,
                   double *center_frequency_out)
{
    downmix_frame_t in;
    memset(&in, 0, sizeof(in));
    in.id = id;
    in.timestamp = timestamp;
    in.center_frequency = center_frequency;
    in.sample_rate = 250000.0f;
    in.samples_per_symbol = samples_per_symbol;
    in.direction = (ir_direction_t)direction;
    in.magnitude = magnitude;
    in.noise = noise;
    in.num_samples = (size_t)num_samples;
    in.samples = malloc(sizeof(float complex) * (size_t)num_samples);
    memcpy(in.samples, samples, sizeof(float complex) * (size_t)num_samples);
    demod_frame_t *out = NULL;
    int r = qpsk_demod(&in, &out);
    free(in.samples);
    if (!r)
        return 0;
    *direction_out = (int)out->direction;
    *confidence = out->confidence;
    *level = out->level;
    *n_symbols = out->n_symbols;
    *n_payload = out->n_payload_symbols;
    *n_bits = out->n_bits;
    memcpy(bits, out->bits, (size_t)out->n_bits);
    if (out->llr)
        memcpy(llr, out->llr, sizeof(float) * (size_t)out->n_bits);
    *center_frequency_out = out->center_frequency;
    free(out->bits);
    free(out->llr);
    free(out);
    return 1;
}

/* frame_decode() (frame_decode.c:414) through a flat interface; the layout of `out` is orc_decoded_t's (irdm_oracle.h) */
typedef struct {
    int32_t type, sat_id, beam_id, pos_xyz[3], alt, n_pages;
    double lat, lon;
    uint32_t page_tmsi[12];
    int32_t page_msc[12];
    int32_t timeslot, sv_blocking, bc_type;
    uint32_t iri_time;
    int32_t bch_len, pad;
} ref_decoded_t;

int ref_frame_decode(const uint8_t *bits, const float *llr, int n_bits, ref_decoded_t *out)
{
    static int init;
    if (!init) { frame_decode_init(); init = 1; }
    demod_frame_t f;
    memset(&f, 0, sizeof(f));
    f.bits = (uint8_t *)bits;
    f.llr = (float *)llr;
    f.n_bits = n_bits;
    decoded_frame_t d;
    const int r = frame_decode(&f, &d);
    memset(out, 0, sizeof(*out));
    out->type = (int32_t)d.type;
    out->bch_len = -1;                       /* not exposed by the reference */
    if (d.type == FRAME_IRA) {
        out->sat_id = d.ira.sat_id;
        out->beam_id = d.ira.beam_id;
        out->lat = d.ira.lat;
        out->lon = d.ira.lon;
        out->alt = d.ira.alt;
        for (int i = 0; i < 3; i++) out->pos_xyz[i] = d.ira.pos_xyz[i];
        out->n_pages = d.ira.n_pages;
        for (int i = 0; i < d.ira.n_pages && i < 12; i++) {
            out->page_tmsi[i] = d.ira.pages[i].tmsi;
            out->page_msc[i] = d.ira.pages[i].msc_id;
        }
    } else if (d.type == FRAME_IBC) {
        out->sat_id = d.ibc.sat_id;
        out->beam_id = d.ibc.beam_id;
        out->timeslot = d.ibc.timeslot;
        out->sv_blocking = d.ibc.sv_blocking;
        out->bc_type = d.ibc.bc_type;
        out->iri_time = d.ibc.iri_time;
    }
    return r;
}

/* ida_decode() (ida_decode.c:543) through a flat interface; the layout of `out` is orc_ida_t's (irdm_oracle.h) */
typedef struct {
    int32_t ok, ft, lcw_ft, lcw_code, ec_lcw;
    uint32_t lcw3_val;
    int32_t da_ctr, da_len, cont, crc_ok;
    uint32_t stored_crc, computed_crc;
    int32_t fixederrs, payload_len, bch_len, pad;
    uint8_t payload[32];
    uint8_t bch_stream[256];
    char lcw_header[128];
} ref_ida_t;

int ref_ida_decode(const uint8_t *bits, const float *llr, int n_bits, int direction, ref_ida_t *out)
{
    static int init;
    if (!init) { frame_decode_init(); ida_decode_init(); init = 1; }
    demod_frame_t f;
    memset(&f, 0, sizeof(f));
    f.bits = (uint8_t *)bits;
    f.llr = (float *)llr;
    f.n_bits = n_bits;
    f.direction = (ir_direction_t)direction;
    ida_burst_t b;
    const int r = ida_decode(&f, &b);
    memset(out, 0, sizeof(*out));
    out->ok = r;
    if (!r) return 0;
    out->ft = b.lcw.ft;
    out->lcw_ft = b.lcw.lcw_ft;
    out->lcw_code = b.lcw.lcw_code;
    out->ec_lcw = b.lcw.ec_lcw;
    out->lcw3_val = b.lcw.lcw3_val;
    out->da_ctr = b.da_ctr;
    out->da_len = b.da_len;
    out->cont = b.cont;
    out->crc_ok = b.crc_ok;
    out->stored_crc = b.stored_crc;
    out->computed_crc = b.computed_crc;
    out->fixederrs = b.fixederrs;
    out->payload_len = b.payload_len;
    out->bch_len = b.bch_len;
    memcpy(out->payload, b.payload, sizeof(out->payload));
    memcpy(out->bch_stream, b.bch_stream, sizeof(out->bch_stream));
    memcpy(out->lcw_header, b.lcw_header, sizeof(out->lcw_header));
    return 1;
}

