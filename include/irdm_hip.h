/*
 * irdm_hip.h -- C-ABI of the MI355X (gfx950) Iridium hot path.
 *
 * Plain C, no HIP or torch types: a C99 host (the reference's main.c /
 * burst_detect.c) links this library and nothing else.  Every entry point
 * names the reference interface it replaces (file:line into the reference).
 *
 * Layers
 *   1. gpu_burst_fft_*      -- the reference's one accelerator plug point
 *                              (opencl/burst_fft.h:35-47), same names and
 *                              conventions, so burst_detect.c's USE_GPU branch
 *                              (burst_detect.c:304-319, :655-674) links unchanged.
 *   2. irdm_*               -- batched superset: detect -> downmix -> demod for
 *                              whole chunks of the IQ stream, device-resident or
 *                              host buffers.  Same conventions as layer 1: opaque
 *                              context, NULL / -1 on error, caller-owned buffers,
 *                              one thread per context.
 *   3. irdm_format_raw      -- frame_output.c:160-199 RAW line, host C.
 */
#ifndef IRDM_HIP_H
#define IRDM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* 1. Reference plug point (opencl/burst_fft.h)                        */
/* ------------------------------------------------------------------ */

typedef struct gpu_burst_fft gpu_burst_fft_t;

/* opencl/burst_fft.h:35-36.  fft_size: power of two (2048..16384 use the
 * LDS-resident kernel).  window: fft_size floats, already /0.42
 * (burst_detect.c:247-250), copied.  NULL on failure => the caller falls back
 * to its CPU path (burst_detect.c:316-318). */
gpu_burst_fft_t *gpu_burst_fft_create(int fft_size, int batch_size, const float *window);

/* opencl/burst_fft.h:39.  NULL-safe. */
void gpu_burst_fft_destroy(gpu_burst_fft_t *g);

/* opencl/burst_fft.h:46-47.  input: batch_count*fft_size interleaved (re,im)
 * host floats; output: batch_count*fft_size host floats, |X|^2 DC-shifted.
 * 0 ok, -1 error (batch_count <= 0 or > batch_size, opencl/burst_fft.c:325-326)
 * => the caller discards the batch (burst_detect.c:659-665). */
int gpu_burst_fft_process(gpu_burst_fft_t *g, const float *input, float *output,
                          int batch_count);

/* Same computation on device-resident buffers (no H2D/D2H); stream is a
 * hipStream_t passed as void* (NULL = default stream).  Asynchronous. */
int gpu_burst_fft_process_device(gpu_burst_fft_t *g, const void *d_input, void *d_output,
                                 int batch_count, void *stream);

/* ------------------------------------------------------------------ */
/* 2. Batched pipeline                                                  */
/* ------------------------------------------------------------------ */

#define IRDM_FMT_CI8  0   /* options.c FMT_CI8  */
#define IRDM_FMT_CI16 1   /* options.c FMT_CI16: raw int16 pairs; narrowed to (int8)(v >> 8) (main.c:245-246) in the
                             kernels' load stage, on the device */
#define IRDM_FMT_CF32 2   /* options.c FMT_CF32 */

typedef struct {
    double center_frequency;   /* -c, burst_config_t.center_frequency (burst_detect.h:52) */
    int sample_rate;           /* -r, burst_config_t.sample_rate */
    float threshold_db;        /* -d; <= 0 -> 16 dB (iridium.h:37) */
    int format;                /* IRDM_FMT_* */
    int feed_block;            /* samples per reference feed call; 0 -> 32768 (main.c:225).
                                  Results equal the reference fed in blocks of this size. */
    int use_gardner;           /* main.c:143 default 1; 0 = --no-gardner */
    uint64_t start_time_ns;    /* replaces the wall clock read at burst_detect.c:849-853; 0 -> now */
    int device;                /* HIP device ordinal */
    size_t max_chunk_samples;  /* largest chunk passed to irdm_feed_*; 0 -> 64 Mi */
    int max_bursts_per_chunk;  /* sizing hint for the burst-record buffers, 0 -> 4096; a chunk with more finished bursts
                                  is redone with larger buffers (costs one dense scan), never dropped */
    int pipeline_depth;        /* 0: irdm_feed_* returns with the chunk's results pollable.
                                  1 .. 5: throughput mode.  irdm_feed_*(k) returns once chunk k is ingested (FFT done,
                                     samples in the history ring) and its detector scan is launched; the scan stays in
                                     flight while the caller produces chunk k+1.  The bursts of chunk k enter their
                                     per-burst chain (decimator .. demodulator, on a stream of its own) during
                                     irdm_feed_*(k+1); pipeline_depth + 1 chains are in flight and their records
                                     become pollable when their context is needed again, i.e. during
                                     irdm_feed_*(k+1+pipeline_depth), or at irdm_flush -- identical records, same order.
                                     A chain is 2.5-3 ms of dependent launches: the pipeline's period is at least that
                                     latency / (pipeline_depth + 1).  Every context holds ~0.5 GB of per-burst scratch and
                                     the history ring one more chunk.  Values above 5 are treated as 5. */
} irdm_config_t;

/* burst_info_t (burst_detect.h:29-37) + what emit_gone_bursts adds (burst_detect.c:703-742) */
typedef struct {
    uint64_t id;
    uint64_t start;
    uint64_t stop;
    uint64_t last_active;
    int32_t center_bin;
    float magnitude;           /* dB */
    float noise;               /* dBFS/Hz */
    float peak_rel;            /* raw relative magnitude of the creating peak */
    float base_sum;            /* baseline_sum[center_bin] at creation */
    uint64_t num_samples;      /* burst_data_t.num_samples */
    uint64_t avail_end;        /* sample_count at extraction time */
} irdm_burst_t;

#define IRDM_MAX_FRAME_SAMPLES 4440     /* IR_MAX_FRAME_LENGTH_SIMPLEX * 10 sps */
#define IRDM_MAX_BITS 896

/* downmix_frame_t (burst_downmix.h:32-51) without the sample pointer, plus stage probes */
typedef struct {
    uint64_t id;
    uint64_t timestamp;
    double center_frequency;
    float sample_rate;
    float samples_per_symbol;
    int32_t direction;         /* ir_direction_t */
    float magnitude;
    float noise;
    float uw_start;
    int32_t num_samples;
    int32_t dec_len;
    int32_t start;
    float center_offset;
    int32_t uw_start_idx;
    float corr_re, corr_im;
    int32_t drop_reason;       /* 0 = frame produced; 1..5 = the reference's five early returns
                                  (burst_downmix.c:645, :677, :702, :744, :773) */
    int32_t demod_ok;          /* qpsk_demod()'s return value for this frame (0 when no frame was produced) */
    int32_t demod_direction;   /* in->direction as qpsk_demod leaves it (qpsk_demod.c:444, :454-463): DIR_UNDEF (0)
                                  when the unique word was rejected, else the verified direction */
} irdm_frame_info_t;

/* demod_frame_t (qpsk_demod.h:24-38), bits/llr inline */
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
    int32_t ok;
    float total_phase;
    uint8_t bits[IRDM_MAX_BITS];
    float llr[IRDM_MAX_BITS];
} irdm_demod_t;

/* The same frame without the soft outputs, hard bits 8 per byte (MSB first: bit i of the frame is
 * (bits[i / 8] >> (7 - i % 8)) & 1): everything frame_output_print reads (frame_output.c:168-197) in 176 bytes.  With
 * option "packed_records" 1 a context queues ONLY these (and the burst records): irdm_poll_demods / irdm_poll_frames
 * then return nothing, and 136 bytes per burst cross PCIe instead of 4.5 KB. */
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
    int32_t ok;
    float total_phase;
    uint8_t bits[IRDM_MAX_BITS / 8];
} irdm_demod_packed_t;

/* decoded_frame_t (frame_decode.h:26-60), flattened: the post-demod bit layer's result for one demodulated frame */
typedef struct {
    int32_t type;              /* frame_type_t: 0 FRAME_UNKNOWN, 1 FRAME_IRA, 2 FRAME_IBC */
    int32_t sat_id, beam_id;
    int32_t pos_xyz[3];        /* ira_data_t */
    int32_t alt;
    int32_t n_pages;
    double lat, lon;
    uint32_t page_tmsi[12];
    int32_t page_msc[12];
    int32_t timeslot, sv_blocking, bc_type;    /* ibc_data_t */
    uint32_t iri_time;
    int32_t bch_len;           /* decoded data bits assembled (probe; not in the reference's struct) */
    int32_t pad;
    uint64_t id;               /* the frame's burst id */
    uint64_t timestamp;        /* decoded_frame_t.timestamp */
    double frequency;          /* decoded_frame_t.frequency = the refined centre frequency */
} irdm_decoded_t;

/* ida_burst_t (ida_decode.h:31-56), flattened: one IDA burst after LCW + BCH decoding, before multi-burst reassembly */
typedef struct {
    int32_t ok;                /* ida_decode()'s return value */
    int32_t ft, lcw_ft, lcw_code, ec_lcw;      /* lcw_t */
    uint32_t lcw3_val;
    int32_t da_ctr, da_len, cont, crc_ok;
    uint32_t stored_crc, computed_crc;
    int32_t fixederrs, payload_len, bch_len;
    int32_t direction;         /* ir_direction_t of the frame */
    uint8_t payload[32];
    uint8_t bch_stream[256];
    char lcw_header[128];      /* "LCW(...)" padded to 110 characters + one space (format_lcw_header) */
    uint64_t id;               /* the frame's burst id */
    uint64_t timestamp;
    double frequency;
    float magnitude, noise, level;
    int32_t confidence, n_symbols;             /* n_symbols = the frame's payload symbols (ida_decode.c:648) */
    int32_t pad;
} irdm_ida_t;

/* option "chunk_marks" 1: one mark per batch of records a context pushes to its queues -- the chunk they belong to (chunks
 * counted from 0 in the order fed) and how many records each queue received, in queue order.  A chunk without bursts
 * leaves no mark; a chunk with more bursts than a batch holds leaves several, one after the other. */
typedef struct {
    uint64_t chunk;
    uint32_t n_bursts, n_frames, n_demods, n_packed, n_decoded, n_ida;
} irdm_chunk_mark_t;

typedef struct irdm_pipeline irdm_pipeline_t;

/* burst_detector_create + burst_downmix_create (burst_detect.c:174, burst_downmix.c:223):
 * derives every constant the reference derives, designs the filters on the host with the
 * host libm, uploads them.  NULL on failure (no device, bad config, allocation). */
irdm_pipeline_t *irdm_create(const irdm_config_t *cfg);
void irdm_destroy(irdm_pipeline_t *p);

/* burst_detector_feed / _feed_cf32 (burst_detect.h:74-79) for a whole chunk.
 * n_samples must be a multiple of feed_block except for the last chunk of the stream.
 * _device: d_iq is a device pointer to raw samples in the configured format (cf32 pairs, int16 pairs or int8
 * pairs), stream = the hipStream_t that produced them, or NULL when the data is already complete (no ordering is
 * established then; NULL does not mean the legacy default stream).  The buffer may be reused when the call returns.
 * pipeline_depth 0: returns after the chunk is fully processed (results pollable); pipeline_depth 1: see above.
 * _host: the same for a host buffer; the H2D copy is asynchronous DMA when the buffer is pinned
 * (irdm_host_alloc, hipHostMalloc, hipHostRegister) and overlaps the previous chunk's detector scan.
 * Returns the number of bursts whose records became pollable, or -1 on error. */
int irdm_feed_device(irdm_pipeline_t *p, const void *d_iq, size_t n_samples, void *stream);
/* irdm_feed_device in two halves: _begin = what does not depend on the detector state (K1 of the chunk, its copy into
 * the history ring), _end = detector scan + per-burst work.  A time-sharded rank calls _begin, receives the previous
 * rank's state (irdm_import_state_device), then calls _end.
 * pipeline_depth >= 1: one chunk of look-ahead -- irdm_feed_begin(k+1) may be called before irdm_feed_end(k) (the calls
 * alternate after that; a third pending begin returns -1), which puts K1 of the next chunk on the GPU before the host
 * waits for the detector scan of chunk k-1, and lets the library enqueue the speculation pass and the scan of chunk k+1 with
 * chunk k's.  (A second chunk begun ahead was measured in rounds 5 and 6 -- 70.6 against 74.1-74.4 Gsamples/s -- and removed.)
 * The buffer handed to _begin may be reused when the matching _end has returned.  A detector state may only be imported
 * (irdm_import_state*) while no chunk but the one begun last is pending. */
int irdm_feed_begin(irdm_pipeline_t *p, const void *d_iq, size_t n_samples, void *stream);
int irdm_feed_end(irdm_pipeline_t *p);
/* pipeline_depth >= 1: where the producer of the next chunk (an H2D copy, a conversion kernel) may write it so that the
 * context does not have to copy it into its history ring (8 B/sample read + written for cf32): the ring slot of the
 * stream position the next irdm_feed_begin starts at.  Pass the pointer to irdm_feed_begin / irdm_feed_device as
 * d_iq; nothing waits for the chunk to be "released" then.  NULL at pipeline_depth 0 or if the chunk would straddle the
 * end of the ring -- it never does when every chunk but the last has max_chunk_samples.  The slot belongs to the
 * producer until it is fed; it is reused one ring length (irdm_ring_ptr) later. */
void *irdm_ingest_ptr(irdm_pipeline_t *p, size_t n_samples);
/* the history ring (device memory, configured sample format) and its length in samples: sample i of the stream lives
 * at index i % length */
void *irdm_ring_ptr(irdm_pipeline_t *p, uint64_t *len_samples);
int irdm_feed_host(irdm_pipeline_t *p, const void *h_iq, size_t n_samples);
/* pipeline_depth >= 1: finish the detector scan in flight and every per-burst chain, oldest first; all records of the
 * chunks fed so far are pollable afterwards.  Returns bursts processed or -1 (also when a chunk handed over with
 * irdm_feed_begin still waits for its irdm_feed_end). */
int irdm_flush(irdm_pipeline_t *p);
/* irdm_flush without the waiting (pipeline_depth >= 1): settles the detector scan in flight, enqueues the per-burst work of
 * its bursts and returns; records of batches that have already finished become pollable.  For callers that interleave
 * other work -- a time-sharded rank's next super-step -- with the chain.  Returns the number of bursts whose records were
 * emitted, -1 on error.  (No reference counterpart: burst_downmix / qpsk_demod run on their own threads there,
 * main.c:667-694.) */
int irdm_advance(irdm_pipeline_t *p);
/* Pinned (page-locked) host memory for feed buffers, for hosts without HIP headers.  NULL on failure. */
void *irdm_host_alloc(size_t bytes);
void irdm_host_free(void *ptr);

/*
 * Device buffers for hosts without HIP headers (the C99 binary, tests): an IQ chunk that lives in HBM before it is
 * fed with irdm_feed_device (the case the headline metric is quoted on).  irdm_device_upload is synchronous.
 */
void *irdm_device_alloc(int device, size_t bytes);
void irdm_device_free(void *dptr);
int irdm_device_upload(void *dptr, const void *host, size_t bytes);
/* synchronous device-to-device copy on the current device (e.g. a resident chunk into its irdm_ingest_ptr slot) */
int irdm_device_copy(void *dst, const void *src, size_t bytes);

/* Results of all chunks fed so far, in burst-emission order; each call drains up to max
 * entries.  bursts: one per emitted burst (burst_callback_t payload minus samples).
 * frames: one per emitted burst (drop_reason says whether a frame was produced).
 * demods: one per frame that passed the unique-word check (frame_output_print input). */
int irdm_poll_bursts(irdm_pipeline_t *p, irdm_burst_t *out, int max);
int irdm_poll_frames(irdm_pipeline_t *p, irdm_frame_info_t *out, float *samples_out /* max*2*4440 or NULL */, int max);
int irdm_poll_demods(irdm_pipeline_t *p, irdm_demod_t *out, int max);
int irdm_poll_demods_packed(irdm_pipeline_t *p, irdm_demod_packed_t *out, int max);   /* option "packed_records" 1 */

/* "tagged N bursts total" (burst_detect.c:350-351) and stat_sample_count (main.c:199) */
uint64_t irdm_tagged_bursts(const irdm_pipeline_t *p);
uint64_t irdm_sample_count(const irdm_pipeline_t *p);
size_t irdm_max_chunk_samples(const irdm_pipeline_t *p);      /* the configured value with its default resolved */
size_t irdm_bytes_per_sample(const irdm_pipeline_t *p);       /* of the configured input format */
/* samples in front of a stream position that a context taking over there must be given (irdm_seed_history*): the
 * reference's ring (stale-slot reads reach one ring length back, burst_detect.c:292-296, :401-422) + the longest burst window */
size_t irdm_required_overlap(const irdm_pipeline_t *p);
/* host wait until K1 and the history-ring copy of every chunk handed over so far have read their input buffers */
int irdm_wait_ingest(irdm_pipeline_t *p);
int irdm_fft_size(const irdm_pipeline_t *p);
int irdm_set_stream_origin(irdm_pipeline_t *p, double center_frequency, uint64_t start_time_ns);  /* for the stage-level calls */
uint64_t irdm_start_time_ns(const irdm_pipeline_t *p);   /* burst_data_t.start_time_ns (burst_detect.c:849-853) */

/* Stage probes (parity tests): magnitudes of the last chunk (frames x fft_size floats,
 * device -> host copy), current baseline sum. */
/* What the reference's stats thread asks the detector (main.c:455-456): burst_detector_active_count / _noise_floor /
 * _peak_signal (burst_detect.c:355-395).  Settles the scan in flight and reads the detector state back (a few tens of
 * KB): call it from the feeding thread, at most once per feed.  peak_signal_db covers the bursts finished or still
 * active; bursts a squelch dumped (burst_detect.c:594-631) were seen by the reference's running maximum only. */
typedef struct {
    int32_t active_bursts;
    int32_t primed;
    float noise_floor_dbfs_hz;
    float peak_signal_db;
} irdm_detector_stats_t;
int irdm_detector_stats(irdm_pipeline_t *p, irdm_detector_stats_t *out);

int irdm_last_magnitudes(irdm_pipeline_t *p, float *out, size_t max_frames);
int irdm_baseline_sum(irdm_pipeline_t *p, float *out);
/* burst_data_t.samples of the i-th burst emitted by the LAST chunk (re-gathered) */
int irdm_burst_samples(irdm_pipeline_t *p, int burst_in_chunk, float *out, size_t max_samples);

/* Stage B alone for one burst: burst_downmix_process() (burst_downmix.h:70).  `info` carries the
 * burst_info_t fields (id, start, center_bin, magnitude, noise are used), `samples` the burst_data_t IQ
 * (host, num_samples complex floats).  Fills *frame (drop_reason 0 = frame produced, 1..5 = the reference's
 * early returns) and, when a frame was produced, frame_samples (2 * IRDM_MAX_FRAME_SAMPLES floats).
 * cf32 contexts only.  Returns 1 (frame), 0 (dropped) or -1 (error). */
int irdm_downmix_burst(irdm_pipeline_t *p, const irdm_burst_t *info, const float *samples, size_t num_samples,
                       irdm_frame_info_t *frame, float *frame_samples);

/* Stage C alone, batched: qpsk_demod() (qpsk_demod.h:42) for n downmixed frames.
 * samples: n rows of 2*IRDM_MAX_FRAME_SAMPLES floats (re,im interleaved, row-padded);
 * num_samples[i] <= IRDM_MAX_FRAME_SAMPLES; direction[i] = the downmixer's ir_direction_t.
 * out[i].ok = the reference's return value (1 = unique word accepted); metadata fields other
 * than direction, confidence, level, the symbol counts, bits, llr and total_phase are left zero.  Returns 0 or -1. */
int irdm_qpsk_demod_batch(irdm_pipeline_t *p, const float *samples, const int *num_samples,
                          const int *direction, int n, irdm_demod_t *out);

/* Post-demod bit layer alone, batched: frame_decode() (frame_decode.h:63) for n demodulated frames -- access code,
 * de-interleave, BCH(31,21)/(7,3) syndromes, Chase decoding on the LLRs (use_llr = 0: hard decisions only, as
 * frame->llr == NULL), IRA / IBC fields.  in[i].bits / llr / n_bits / id / timestamp / center_frequency are read.
 * out[i].type is 0 when frame_decode() returns 0.  Returns 0 or -1.
 * With the option "decode_frames" = 1 the pipeline runs the same kernel behind the demodulator and
 * irdm_poll_decoded returns one record per irdm_poll_demods record, in the same order. */
int irdm_frame_decode_batch(irdm_pipeline_t *p, const irdm_demod_t *in, int n, int use_llr, irdm_decoded_t *out);
int irdm_poll_decoded(irdm_pipeline_t *p, irdm_decoded_t *out, int max);
/* ida_decode() (ida_decode.h) the same way: LCW extraction, payload descramble, BCH(31,20) + Chase, CRC-CCITT.
 * in[i].direction is the demodulator's direction.  Option "decode_ida" = 1 runs it in the pipeline;
 * irdm_poll_ida returns one record per irdm_poll_demods record (ok = 0 when ida_decode() returns 0).
 * Multi-burst reassembly (ida_reassemble) stays on the host and consumes these records. */
int irdm_ida_decode_batch(irdm_pipeline_t *p, const irdm_demod_t *in, int n, int use_llr, irdm_ida_t *out);
int irdm_poll_ida(irdm_pipeline_t *p, irdm_ida_t *out, int max);

/* ---- time-chunk sharding of ONE stream across GPUs ----
 * The detector is sequential across frames (noise-floor ring, active bursts, ids); exact
 * sharding hands its state from the rank that scanned chunk k to the rank that scans chunk k+1.
 * irdm_state_bytes: size of the blob (DetState + baseline sum + 512-frame history).
 * irdm_export_state / irdm_import_state: blob to / from a HOST buffer (the caller moves it
 * between ranks, e.g. RCCL send/recv of the same bytes).  Returns bytes written / 0, or -1.
 * irdm_seed_history: tells a fresh context that its stream position is abs_start and gives it
 * the n_samples of IQ (host buffer, configured format) that precede that position, so burst
 * windows reaching back across the chunk boundary read real samples (the chunk overlap). */
size_t irdm_state_bytes(const irdm_pipeline_t *p);
long long irdm_export_state(irdm_pipeline_t *p, void *buf, size_t cap);
int irdm_import_state(irdm_pipeline_t *p, const void *buf, size_t n);
int irdm_seed_history(irdm_pipeline_t *p, const void *h_iq, size_t n_samples, uint64_t abs_start);
/* the same three with DEVICE buffers (a blob RCCL moves GPU to GPU; a chunk overlap received from the previous rank):
 * no host bounce, and only the detector is waited for -- K1 of the next chunk and per-burst work in flight go on */
long long irdm_export_state_device(irdm_pipeline_t *p, void *d_buf, size_t cap);
int irdm_import_state_device(irdm_pipeline_t *p, const void *d_buf, size_t n);
/* The same hand-off in two parts, so that the history (512 x N floats: 16-32 MiB) can FOLLOW the part a scan needs first:
 * head = the first irdm_state_head_bytes() bytes of the blob (header, detector state, baseline sums), history = the rest.
 * Round 0 of the band scan reads no history; a rank imports the head, asks irdm_expect_history(p, buf) -- 1: the scan
 * that the next irdm_feed_end enqueues waits ON THE DEVICE (a one-lane kernel polling a word the host writes) until
 * irdm_import_state_history_device(p, buf, n) says the history has arrived in buf, copies it in and goes on; the caller
 * must make that call before it settles the scan (irdm_export_state_device, irdm_flush, the next irdm_feed_end); 0: this
 * scan cannot wait (pipeline_depth 0, an unprimed detector, a sequential scan mode): import the history first -- calls
 * irdm_feed_end, receives the history while K1 and round 0 run, and imports it.  0 ok, -1 error. */
size_t irdm_state_head_bytes(const irdm_pipeline_t *p);
int irdm_import_state_head_device(irdm_pipeline_t *p, const void *d_buf, size_t n);
int irdm_expect_history(irdm_pipeline_t *p, const void *d_hist_buf);   /* d_hist_buf: where the history will arrive (device memory) */
int irdm_import_state_history_device(irdm_pipeline_t *p, const void *d_hist_buf, size_t n);
int irdm_seed_history_device(irdm_pipeline_t *p, const void *d_iq, size_t n_samples, uint64_t abs_start);

/* ---- a group: ONE stream across several GPUs of ONE process (replaces main.c:667-694's thread layout
 * for N > 1) ----
 * irdm_group_create builds one context per device (cfg->device is ignored; devices == NULL: 0 .. n_gpus-1; pipeline_depth
 * at least 1), two RCCL communicator sets over them (ncclCommInitAll: one for IQ samples, one for the detector state, so
 * that a 16-32 MiB state hop never queues behind a 0.5 GB slice) and per member two landing buffers of
 * [overlap | cfg->max_chunk_samples] samples.  The stream is cut into chunks of max_chunk_samples; chunk k goes to member
 * k mod N.  librccl is loaded when the first group is created (dlopen: a process that never makes a group never loads it;
 * IRDM_RCCL_LIB names another file); a group of one member needs no RCCL at all unless "group_loopback" is set.
 *
 * irdm_group_stage_host / _device: the samples of the next feed start moving -- from host memory (pinned for speed: each
 *   member's slice over its own PCIe link) or from device memory of member 0 (an RCCL scatter: grouped ncclSend / ncclRecv
 *   from member 0 to every other member) -- into the landing buffers the current feed does not use, together with each
 *   chunk's overlap (the samples in front of it that burst windows and the reference's stale ring reads reach back to:
 *   the tail of the previous member's landing buffer, GPU to GPU).  Returns at once; n_samples <= n_gpus *
 *   max_chunk_samples, chunks of max_chunk_samples except the last of the stream.  Calling it before the previous
 *   irdm_group_feed puts the scatter under that feed's compute (two super-steps may be staged at most: the one being fed
 *   and the one behind it).  The buffer must be complete when the call is made and stay unchanged until the
 *   irdm_group_feed_* that consumes it has returned.
 * irdm_group_feed_host / _device: stages (unless exactly this buffer was staged) and runs the super-step: K1 of every
 *   chunk at once, then the detector's chain member by member -- ncclRecv of the previous member's state head, import, scan
 *   (round 0 while the 512-frame history is still arriving), export, ncclSend to the next member -- and every chunk's
 *   per-burst chain enqueued behind its scan (irdm_advance): the chains overlap the other members' scans and the next
 *   super-step.  Returns the number of chunks fed, -1 on error.
 * irdm_group_flush: everything in flight completes.
 * irdm_group_poll_*: the members' records merged in STREAM order (chunk by chunk, as one context would have queued them);
 *   a chunk's records come out once every earlier chunk is complete.
 * irdm_group_set_option: an option of every member ("group_loopback" 1: a group of one member runs the whole protocol --
 *   overlap seed, state export, ncclSend / ncclRecv to itself, import -- for tests on one GPU).
 * irdm_group_get_stat: "hops", "hop_bytes", "scatter_bytes", "overlap_bytes", "overlap_samples", "chunks", "late_history"
 *   (scans that took their history behind round 0), "tagged" (the stream's burst count: it travels with the detector
 *   state), any other key: the sum of irdm_get_stat over the members.
 * One thread drives a group.  Feeding or polling a member directly (irdm_group_member) is not allowed; options, statistics
 * and kernel clocks of a member are. */
typedef struct irdm_group irdm_group_t;
int irdm_device_count(void);                           /* GPUs this process sees (0: none, or no HIP runtime) */
/* EXPERIMENTAL for n_gpus > 1 (a warning says so once): run on emulated devices and as ranks sharing one GPU only.
 * While the RCCL communicators are made (in irdm_group_create for n_gpus > 1, and when "group_loopback" is first set on a
 * group of one) the process's file descriptor 1 is pointed at its stderr and restored afterwards: librccl prints a version
 * banner on stdout, the stream a host prints its RAW lines to.  A host with other threads writing to stdout in that window
 * sees their output on stderr; IRDM_GROUP_KEEP_STDOUT=1 in the environment leaves the descriptor alone (banner included). */
irdm_group_t *irdm_group_create(const irdm_config_t *cfg, int n_gpus, const int *devices);
void irdm_group_destroy(irdm_group_t *g);
int irdm_group_size(const irdm_group_t *g);
irdm_pipeline_t *irdm_group_member(irdm_group_t *g, int i);
int irdm_group_set_option(irdm_group_t *g, const char *key, int value);
int64_t irdm_group_get_stat(const irdm_group_t *g, const char *key);
int irdm_group_stage_host(irdm_group_t *g, const void *h_iq, size_t n_samples);
int irdm_group_stage_device(irdm_group_t *g, const void *d_iq, size_t n_samples);
int irdm_group_feed_host(irdm_group_t *g, const void *h_iq, size_t n_samples);
int irdm_group_feed_device(irdm_group_t *g, const void *d_iq, size_t n_samples);
int irdm_group_flush(irdm_group_t *g);
int irdm_group_poll_bursts(irdm_group_t *g, irdm_burst_t *out, int max);
int irdm_group_poll_frames(irdm_group_t *g, irdm_frame_info_t *out, float *samples_out /* max*2*4440 or NULL */, int max);
int irdm_group_poll_demods(irdm_group_t *g, irdm_demod_t *out, int max);
int irdm_group_poll_demods_packed(irdm_group_t *g, irdm_demod_packed_t *out, int max);
int irdm_group_poll_decoded(irdm_group_t *g, irdm_decoded_t *out, int max);
int irdm_group_poll_ida(irdm_group_t *g, irdm_ida_t *out, int max);
/* what the group rests on in a single context: the marks (option "chunk_marks") and the number of chunks whose records
 * are all in the queues */
int irdm_poll_chunk_marks(irdm_pipeline_t *p, irdm_chunk_mark_t *out, int max);
uint64_t irdm_chunks_complete(const irdm_pipeline_t *p);

/* Options (irdm_set_option; every one a field of THIS context -- two contexts of a process may differ in all of them; set them
 * before the first feed unless noted).  Twenty keys:
 *
 *   what a caller chooses
 *   "keep_frame_samples"  0/1, default 0: irdm_poll_frames returns metadata only
 *   "packed_records"      0/1, default 0: 1 = only burst records and compact frame records (irdm_poll_demods_packed: what
 *                         frame_output_print reads, hard bits 8 per byte, no LLRs), written to pinned memory by the chain's
 *                         last kernel
 *   "chunk_marks"         0/1, default 0: see irdm_chunk_mark_t (what a group merges its members' records with)
 *   "decode_frames" / "decode_ida"   0/1, default 0: the post-demod bit layer, see irdm_poll_decoded / irdm_poll_ida
 *   "detect_only"         0/1, default 0: 1 = stage A alone (burst_detector_feed's role): burst records only
 *   "fir_order" (alias "simd_order")   default 1 = the arithmetic of the reference's AVX2 kernels, simd_avx2.c -- what
 *                         simd_init() (simd_generic.c:33-57) selects on x86: fir_ccf_dec :62-108, fir_ccf :28-55, fir_fff
 *                         :115-138, fftshift_mag :177-221, mag_squared :304-323; 0 = simd_generic.c, what --no-simd and every
 *                         non-x86 build run.  (The other six dispatched kernels are the same operations in both files.)
 *   "host_cfo"            0/1, default 0: 1 = the fine-CFO libm step (cexpf) on a host helper thread instead of the device's
 *                         restatement of glibc's sincosf (forced when irdm_create finds that restatement differs from THIS
 *                         host's libm)
 *   "scan_mode"           0 = band-parallel speculative scan where the FFT size supports it, the sequential scans as its
 *                         exact fallback (default); 1 = dense sequential scan only; 2 / 3 = the sparse leader scan on one CU /
 *                         with updater workgroups, dense fallback; 4 = band scan with the dense scan as its only fallback
 *   "rot_prebuild"        0/1, default 1 where pipeline_depth >= 1: every centre bin's rotator-checkpoint row, as far as a burst
 *                         of ordinary length needs it, built by one background launch behind create (n bins x runs x 16 KB:
 *                         0.07 / 1.3 / 3.2 GB at 2 / 10 / 12 MHz) instead of by the chains that first meet the bin; only
 *                         before the first burst
 *   "kernel_clock"        0/1, default 0: see irdm_kernel_clock
 *
 *   diagnostic
 *   "band_timeline"       0/1: the band scan's passes stamp a device timeline (stats "tl_dur_i" / "tl_gap_i" / "tl_n_i")
 *
 *   test hooks (paths a default run at the standard rates takes rarely or never)
 *   "fir_generic"         1 = the any-M decimator (what 2 / 4 MHz streams take) at every rate
 *   "post_generic"        1 = the runtime-tap-count instances of the per-burst filters
 *   "k1_lists"            default 1: the FFT kernel writes the band scan's candidate lists; 0 = a prefilter pass does (what a
 *                         stale-list retry and an unprimed detector take)
 *   "band_first"          default 0 = as many band-scan rounds up front as the previous chunk needed; n = always n
 *   "band_spec"           default 1: round 0 of a chained scan is a speculation pass beside the previous scan; 0 = classical
 *   "band_selfcheck"      bit mask, see BandParams::selfcheck (csrc/band_core.hpp): both boundary tests compared, records
 *                         spoilt, the walk without look-ahead, the plan pass without its LDS, the guess spoilt in one frame
 *   "rot_pool_rows"       n = an empty on-demand checkpoint arena of n whole rows (a prebuilt one is given up)
 *   "scratch_outputs"     n = the decimated / low-passed scratch of every batch context with room for n outputs to begin with
 *
 * Stats (irdm_get_stat): "scan_fast_chunks", "scan_fallbacks", "scan_dense_frames", "band_chunks", "band_rounds",
 * "band_retries", "band_aborts", "band_extra", "band_last_flags", "k1_lists", "scan_chained", "scan_chain_undone", "spec_passes",
 * "spec_scans", "sum_restarts", "host_us_0".."host_us_9", "rot_rows", "rot_prebuilt_runs",
 * "rot_rows_cap", "rot_blocks", "rot_blocks_cap", "rot_builds", "rot_runs", "rot_ckpts", "rot_grows" (rotator checkpoints:
 * centre bins with a row / runs per bin prebuilt / the arena in whole rows / blocks of 2048 checkpoints in use / allocated /
 * build launches on chains / runs built / checkpoints built / times the arena doubled), "band_steps" (update steps the scans'
 * last rounds walked), "scratch_outputs", "scratch_grows", "scratch_peak" (decimated samples a batch context holds / times it
 * doubled / most a batch needed). */
int irdm_set_option(irdm_pipeline_t *p, const char *key, int value);
int64_t irdm_get_stat(const irdm_pipeline_t *p, const char *key);

/* per-stage device time of the last chunk in milliseconds (hipEvent):
 * [0] fft+mag  [1] detector scan  [2] rotate+FIR decimate  [3] downmix post  [4] demod  [5] total */
int irdm_last_timings(const irdm_pipeline_t *p, float *ms_out, int n);
/* Option "kernel_clock" 1: the chip-filling kernels stamp the 100 MHz device clock when their first wavefront starts and
 * when their last one ends (s_memrealtime, per launch); this returns the spans summed over the launches since the last
 * reset -- the kernel's own duration on the device, free of the dispatch wait a host-side event bracket includes (what
 * bench.py's roofline divides the algorithmic bytes by).  which: 0 = the register-resident decimator
 * (fir_decimate_kernel_f / _r), 1 = K1 (fft_mag_p32_kernel / fft_mag_r16_kernel).  Waits for the device.  0 ok, -1 error. */
int irdm_kernel_clock(irdm_pipeline_t *p, int which, double *sum_ms, uint64_t *launches, double *last_ms, int reset);

/* ------------------------------------------------------------------ */
/* 3. RAW line (frame_output.c:160-199)                                 */
/* ------------------------------------------------------------------ */
/* t0_io: 0 on first call -> set as ensure_initialized does (frame_output.c:144-158).
 * file_info NULL/"" -> "i-<t0 s>-t1".  Returns the line length incl. '\n', or -1. */
int irdm_format_raw(const irdm_demod_t *f, const char *file_info, uint64_t *t0_io,
                    char *buf, size_t cap);
/* The same for n frames, lines concatenated in buf (NUL-terminated): one write per batch instead of the
 * reference's fflush per line (frame_output.c:196-198).  cap >= n * IRDM_RAW_LINE_MAX always suffices.
 * Returns the total length or -1. */
#define IRDM_RAW_LINE_MAX 1280          /* 256 B of prefix (file_info <= 128 chars) + IRDM_MAX_BITS + newline, rounded up */
long long irdm_format_raw_batch(const irdm_demod_t *f, int n, const char *file_info, uint64_t *t0_io,
                                char *buf, size_t cap);
/* the same lines from compact records (option "packed_records") */
int irdm_format_raw_packed(const irdm_demod_packed_t *f, const char *file_info, uint64_t *t0_io, char *buf, size_t cap);
long long irdm_format_raw_packed_batch(const irdm_demod_packed_t *f, int n, const char *file_info, uint64_t *t0_io,
                                       char *buf, size_t cap);

/* --save-bursts (qpsk_demod.c:339-389): writes <dir>/<timestamp>_<freq>_<id>_<DL|UL|UN>.cf32 (the frame's cf32 samples at
 * 250 kHz) and the matching .meta text file for one downmixed frame (info->drop_reason == 0), creating dir if needed.
 * samples: 2 * info->num_samples floats as returned by irdm_poll_frames with "keep_frame_samples" = 1.
 * Returns 0, or -1 (no frame / I/O error, message on stderr as the reference prints). */
int irdm_save_burst(const irdm_frame_info_t *info, const float *samples, const char *dir);

/* The fine-CFO step's cexpf(i x) (burst_downmix.c:716-717) as the device evaluates it -- glibc's sincosf restated,
 * csrc/libm_port.hpp -- for n arbitrary arguments (test / audit surface: tests/test_gpu_libm.py,
 * tools/check_sincosf_gpu.c compare it with the host's libm).  0 ok, -1 error; NaN where |x| >= 120. */
int irdm_sincosf_probe(int device, const float *x, size_t n, float *re, float *im);

const char *irdm_version(void);

#ifdef __cplusplus
}
#endif
#endif
