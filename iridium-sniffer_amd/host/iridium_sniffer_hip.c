/*
 * iridium_sniffer_hip.c -- file-mode command line over the MI355X hot path, plain C99.
 *
 * Mirrors the reference's file-mode surface (options.c:186-551, main.c:223-284, frame_output.c:160-199):
 *     iridium-sniffer-hip -f FILE -r RATE [-c FREQ] [--format ci8|ci16|cf32] [-d DB]
 *                         [--file-info STR] [--no-gardner] [--no-simd] [--chunk SAMPLES] [-v]
 * IQ file in, iridium-toolkit "RAW:" lines on stdout, "burst_detect: tagged N bursts total" on stderr
 * (burst_detect.c:350-351, the line test-configurations.sh:140 greps).  Everything between the file
 * read and the line printer runs on the GPU through the C-ABI in include/irdm_hip.h; there is no CPU
 * path here (the reference's own --no-gpu binary is the CPU path).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <pthread.h>
#include <semaphore.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include "irdm_hip.h"

static const char *ext_of(const char *p)
{
    const char *d = strrchr(p, '.');
    return d ? d + 1 : "";
}

/* print every finished frame (frame_output_print, frame_output.c:160-199) and discard the other record queues */
static const char *g_save_dir;

/* file reader thread: fills the two pinned buffers alternately.  A regular file is read by `n_slices` helper threads,
 * each with pread() on its own slice of the chunk (one thread copying out of the page cache moves 5-7 GB/s; the H2D copy
 * behind it runs at 50 GB/s); a pipe (stdin) is read with fread() as before. */
#define MAX_SLICES 16
typedef struct {
    int fd;
    char *dst;
    off_t off;
    size_t len, got;
    sem_t go, done;
    volatile int quit;
    pthread_t th;
} slice_t;

typedef struct {
    FILE *f;
    size_t bps, chunk;
    void *buf[2];
    size_t n[2];
    sem_t filled, empty;
    volatile int stop;
    int n_slices;               /* 0: fread */
    off_t pos, size;
    slice_t sl[MAX_SLICES];
} reader_t;

static void *slice_main(void *arg)
{
    slice_t *s = arg;
    for (;;) {
        sem_wait(&s->go);
        if (s->quit) break;
        size_t g = 0;
        while (g < s->len) {
            const ssize_t r = pread(s->fd, s->dst + g, s->len - g, s->off + (off_t)g);
            if (r <= 0) break;
            g += (size_t)r;
        }
        s->got = g;
        sem_post(&s->done);
    }
    return NULL;
}

/* the next chunk of a regular file into dst: the slices in parallel; returns the samples read */
static size_t read_slices(reader_t *r, void *dst)
{
    size_t want = r->chunk * r->bps;
    if (r->pos >= r->size) return 0;
    if ((off_t)want > r->size - r->pos) want = (size_t)(r->size - r->pos);
    const int T = r->n_slices;
    /* ceil(want / T) rounded up to a page: T slices of `per` bytes always cover `want` (floor here handed out a
     * T + 1-th slice that has no thread when want / T was a multiple of 4096 and want % T != 0); the last slice used
     * takes what is left */
    const size_t per = ((want + (size_t)T - 1) / (size_t)T + 4095) & ~(size_t)4095;
    int used = 0;
    for (size_t o = 0; o < want && used < T; o += per, used++) {
        slice_t *s = &r->sl[used];
        s->dst = (char *)dst + o;
        s->off = r->pos + (off_t)o;
        s->len = (want - o < per || used == T - 1) ? want - o : per;
        sem_post(&s->go);
    }
    size_t got = 0;
    int shortfall = 0;
    for (int i = 0; i < used; i++) {
        sem_wait(&r->sl[i].done);
        if (!shortfall) got += r->sl[i].got;
        if (r->sl[i].got < r->sl[i].len) shortfall = 1;       /* (a file that shrank: what lies before the gap counts) */
    }
    r->pos += (off_t)got;
    return got / r->bps;
}

static void *reader_main(void *arg)
{
    reader_t *r = arg;
    for (int k = 0;; k ^= 1) {
        sem_wait(&r->empty);
        if (r->stop) break;
        r->n[k] = r->n_slices ? read_slices(r, r->buf[k]) : fread(r->buf[k], r->bps, r->chunk, r->f);
        sem_post(&r->filled);
        if (r->n[k] < r->chunk) {                   /* short read: after it an explicit end marker */
            if (r->n[k] != 0) {
                sem_wait(&r->empty);
                if (!r->stop) { r->n[k ^ 1] = 0; sem_post(&r->filled); }
            }
            break;
        }
    }
    return NULL;
}

/* --gpus N: the stream goes through a group (irdm_group_*: chunk k on GPU k mod N, records merged in stream order by the
 * library); the polls below then read the group's queues */
static irdm_group_t *g_group;
#define irdm_poll_demods_packed(p, o, m) (g_group ? irdm_group_poll_demods_packed(g_group, o, m) : irdm_poll_demods_packed(p, o, m))
#define irdm_poll_demods(p, o, m) (g_group ? irdm_group_poll_demods(g_group, o, m) : irdm_poll_demods(p, o, m))
#define irdm_poll_bursts(p, o, m) (g_group ? irdm_group_poll_bursts(g_group, o, m) : irdm_poll_bursts(p, o, m))
#define irdm_poll_frames(p, o, s, m) (g_group ? irdm_group_poll_frames(g_group, o, s, m) : irdm_poll_frames(p, o, s, m))

static void drain(irdm_pipeline_t *p, irdm_demod_t *d, const char *file_info, uint64_t *t0, char *line, size_t cap)
{
    int n;
    if (!g_save_dir) {
        /* RAW lines need no LLRs: compact records (hard bits 8 per byte), 176 bytes per frame instead of 4.5 KB */
        static irdm_demod_packed_t dp[256];
        while ((n = irdm_poll_demods_packed(p, dp, 256)) > 0) {
            const long long len = irdm_format_raw_packed_batch(dp, n, file_info, t0, line, cap);   /* one write per batch */
            if (len > 0) fwrite(line, 1, (size_t)len, stdout);
        }
    }
    while ((n = irdm_poll_demods(p, d, 256)) > 0) {
        const long long len = irdm_format_raw_batch(d, n, file_info, t0, line, cap);     /* one write per batch */
        if (len > 0) fwrite(line, 1, (size_t)len, stdout);
    }
    irdm_burst_t tmp[256];
    while (irdm_poll_bursts(p, tmp, 256) > 0) {}
    if (g_save_dir) {
        /* every frame handed to the demodulator is saved, accepted or not (qpsk_demod.c:443-445, :468-470) */
        static irdm_frame_info_t fi[16];
        static float fs[16 * 2 * IRDM_MAX_FRAME_SAMPLES];
        while ((n = irdm_poll_frames(p, fi, fs, 16)) > 0)
            for (int i = 0; i < n; i++)
                if (fi[i].drop_reason == 0) irdm_save_burst(&fi[i], fs + (size_t)i * 2 * IRDM_MAX_FRAME_SAMPLES, g_save_dir);
    } else {
        irdm_frame_info_t fi[256];
        while (irdm_poll_frames(p, fi, NULL, 256) > 0) {}
    }
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char **argv)
{
    const double t_main = now_s();
    int timing = 0;
    unsigned long long fed = 0;
    const char *file = NULL, *file_info = NULL, *format = NULL;
    double rate = 0, freq = 1622000000.0, db = 0;
    int gardner = 1, verbose = 0, no_simd = 0;
    size_t chunk = (size_t)16 << 20;
    int depth = 1;
    int read_threads = 6;       /* pread() helpers per chunk of a regular file (0: one fread thread) */
    int gpus = 0;               /* --gpus N: one stream across N GPUs of this process (0: one context on device 0) */
    int chunk_given = 0, loopback = 0;
    const char *save_dir = NULL;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
#define NEXT() (i + 1 < argc ? argv[++i] : (fprintf(stderr, "missing value for %s\n", a), exit(2), ""))
        if (!strcmp(a, "-f") || !strcmp(a, "--file")) file = NEXT();
        else if (!strcmp(a, "-r") || !strcmp(a, "--sample-rate")) rate = atof(NEXT());
        else if (!strcmp(a, "-c") || !strcmp(a, "--center-freq")) freq = atof(NEXT());
        else if (!strcmp(a, "-d") || !strcmp(a, "--threshold")) db = atof(NEXT());
        else if (!strcmp(a, "--format")) format = NEXT();
        else if (!strcmp(a, "--file-info")) file_info = NEXT();
        else if (!strcmp(a, "--chunk")) { chunk = (size_t)atoll(NEXT()); chunk_given = 1; }
        else if (!strcmp(a, "--gpus")) gpus = atoi(NEXT());         /* the reference's thread layout for N > 1 (main.c:667-694) */
        else if (!strcmp(a, "--group-loopback")) loopback = 1;      /* test aid: --gpus 1 hands the detector state to itself over RCCL */
        else if (!strcmp(a, "--no-gardner")) gardner = 0;
        else if (!strcmp(a, "--save-bursts")) save_dir = NEXT();   /* options.c --save-bursts: IQ + .meta per downmixed frame */
        else if (!strcmp(a, "--read-threads")) read_threads = atoi(NEXT());
        else if (!strcmp(a, "--depth")) depth = atoi(NEXT());       /* 0: per-chunk latency, 1: throughput (default) */
        else if (!strcmp(a, "-v") || !strcmp(a, "--verbose")) verbose = 1;
        else if (!strcmp(a, "--timing")) timing = 1;                /* start-up and streaming time on stderr */
        else if (!strcmp(a, "--no-simd")) no_simd = 1;              /* options.c:240, :351; main.c:567 simd_init(no_simd) */
        else if (!strcmp(a, "--no-gpu")) {
            fprintf(stderr, "%s: this binary is the GPU path; use the reference binary for the CPU path\n", a);
            return 2;
        } else {
            fprintf(stderr, "unknown option %s\n", a);
            return 2;
        }
    }
    if (!file || rate <= 0) {
        fprintf(stderr, "usage: %s -f FILE -r RATE [-c FREQ] [--format ci8|ci16|cf32] [-d DB] [--file-info STR] [--no-simd] [--save-bursts DIR] [--gpus N]\n", argv[0]);
        return 2;
    }
    if (!format) format = ext_of(file);              /* autodetect by extension, options.c:533-544 */
    int fmt = IRDM_FMT_CI8;
    size_t bps = 2;
    if (!strcmp(format, "cf32") || !strcmp(format, "fc32") || !strcmp(format, "cfile")) { fmt = IRDM_FMT_CF32; bps = 8; }
    else if (!strcmp(format, "ci16") || !strcmp(format, "cs16")) { fmt = IRDM_FMT_CI16; bps = 4; }

    irdm_config_t c;
    memset(&c, 0, sizeof(c));
    c.center_frequency = freq;
    c.sample_rate = (int)rate;
    c.threshold_db = (float)db;
    c.format = fmt;
    c.feed_block = 32768;
    c.use_gardner = gardner;
    /* (several GPUs: a chunk must hold the samples a member is given from in front of its chunk -- 2 s of signal, the
     * reference's ring -- so the default grows with the rate) */
    if (gpus > 1 && !chunk_given) chunk = (size_t)(2.75 * rate);
    chunk = chunk / 32768 * 32768;
    if (chunk == 0) chunk = 32768;
    c.max_chunk_samples = chunk;
    c.pipeline_depth = depth;
    if (gpus < 0 || gpus > 64) { fprintf(stderr, "--gpus %d\n", gpus); return 2; }
    if (gpus > irdm_device_count()) {
        fprintf(stderr, "--gpus %d: this host has %d GPU%s\n", gpus, irdm_device_count(), irdm_device_count() == 1 ? "" : "s");
        return 2;
    }
    irdm_pipeline_t *p;
    if (gpus > 0) {
        g_group = irdm_group_create(&c, gpus, NULL);
        if (!g_group) {
            fprintf(stderr, "irdm_group_create failed (%d MI355X asked for / RCCL / bad parameters)\n", gpus);
            return 1;
        }
        if (loopback && irdm_group_set_option(g_group, "group_loopback", 1) != 0) {
            fprintf(stderr, "--group-loopback: refused (chunk smaller than the overlap, or no RCCL)\n");
            return 1;
        }
        p = irdm_group_member(g_group, 0);
    } else {
        p = irdm_create(&c);
        if (!p) {
            fprintf(stderr, "irdm_create failed (no MI355X / bad parameters)\n");
            return 1;
        }
    }
#define SET_OPTION(key, v) (g_group ? irdm_group_set_option(g_group, key, v) : irdm_set_option(p, key, v))
    /* --no-simd: the reference points its eleven dispatched kernels at simd_generic.c instead of simd_avx2.c
     * (simd_generic.c:33-57); here the same switch selects the kernels that follow the generic file's operation order */
    if (no_simd && SET_OPTION("fir_order", 0) != 0) {
        fprintf(stderr, "--no-simd: the library refused fir_order 0\n");
        return 1;
    }
    if (save_dir) SET_OPTION("keep_frame_samples", 1);
    else SET_OPTION("packed_records", 1);
    g_save_dir = save_dir;
    if (verbose) fprintf(stderr, "%s: fft_size=%d chunk=%zu samples, %d GPU%s\n", irdm_version(), irdm_fft_size(p), chunk,
                         gpus > 0 ? gpus : 1, gpus > 1 ? "s" : "");
    /* a group is fed a super-step at a time: one chunk per member */
    const size_t step = chunk * (size_t)(gpus > 0 ? gpus : 1);

    FILE *f = strcmp(file, "-") ? fopen(file, "rb") : stdin;
    if (!f) { perror(file); return 1; }
    /* Two pinned read buffers and a reader thread (the reference's spewer thread, main.c:223-284): the file read of
     * chunk k+1 overlaps the H2D copy and GPU work of chunk k; the H2D copy itself is DMA that overlaps chunk k-1's
     * detector scan. */
    reader_t rd;
    memset(&rd, 0, sizeof(rd));
    rd.f = f;
    rd.bps = bps;
    rd.chunk = step;
    for (int i = 0; i < 2; i++) {
        rd.buf[i] = irdm_host_alloc(step * bps);
        if (!rd.buf[i]) { fprintf(stderr, "irdm_host_alloc failed\n"); return 1; }
    }
    sem_init(&rd.filled, 0, 0);
    sem_init(&rd.empty, 0, 2);
    {
        struct stat sb;
        if (f != stdin && read_threads > 0 && fstat(fileno(f), &sb) == 0 && S_ISREG(sb.st_mode)) {
            rd.n_slices = read_threads > MAX_SLICES ? MAX_SLICES : read_threads;
            rd.size = sb.st_size - sb.st_size % (off_t)bps;
            for (int i = 0; i < rd.n_slices; i++) {
                rd.sl[i].fd = fileno(f);
                sem_init(&rd.sl[i].go, 0, 0);
                sem_init(&rd.sl[i].done, 0, 0);
                if (pthread_create(&rd.sl[i].th, NULL, slice_main, &rd.sl[i]) != 0) { rd.n_slices = i; break; }
            }
        }
    }
    pthread_t th;
    if (pthread_create(&th, NULL, reader_main, &rd) != 0) { fprintf(stderr, "pthread_create failed\n"); return 1; }
    const double t_ready = now_s();
    irdm_demod_t *d = malloc(sizeof(*d) * 256);
    static char line[256 * IRDM_RAW_LINE_MAX];
    uint64_t t0 = 0;
    int rc = 0;
    for (int k = 0;; k ^= 1) {
        sem_wait(&rd.filled);
        const size_t r = rd.n[k];
        if (r == 0) break;                          /* end of file */
        if (rc == 0 && (g_group ? irdm_group_feed_host(g_group, rd.buf[k], r) : irdm_feed_host(p, rd.buf[k], r)) < 0) {
            fprintf(stderr, "burst_detect: GPU processing failed\n");
            rc = 1;
        }
        fed += r;
        sem_post(&rd.empty);                        /* irdm_feed_host has consumed the buffer when it returns */
        if (rc == 0) drain(p, d, file_info, &t0, line, sizeof line);
        if (r < step) { rd.stop = 1; sem_post(&rd.empty); break; }    /* ragged last chunk = end of stream */
    }
    rd.stop = 1;
    sem_post(&rd.empty);
    pthread_join(th, NULL);
    for (int i = 0; i < rd.n_slices; i++) {
        rd.sl[i].quit = 1;
        sem_post(&rd.sl[i].go);
        pthread_join(rd.sl[i].th, NULL);
    }
    if (rc == 0 && (g_group ? irdm_group_flush(g_group) : irdm_flush(p)) < 0) { fprintf(stderr, "burst_detect: GPU processing failed\n"); rc = 1; }
    drain(p, d, file_info, &t0, line, sizeof line);
    fflush(stdout);
    if (timing) {
        const double t_done = now_s();
        fprintf(stderr, "irdm timing: startup %.3f s (HIP initialisation + device context), stream %.3f s for %llu samples = %.1f Msamples/s\n",
                t_ready - t_main, t_done - t_ready, fed, t_done > t_ready ? fed / (t_done - t_ready) / 1e6 : 0.0);
    }
    fprintf(stderr, "burst_detect: tagged %lu bursts total\n",
            (unsigned long)(g_group ? (uint64_t)irdm_group_get_stat(g_group, "tagged") : irdm_tagged_bursts(p)));
    /* Everything is printed and flushed.  Giving 5-9 GB of device memory, the pinned buffers and the HIP runtime back piece
     * by piece took 0.2 s of a 0.77 s run; the process is about to end and the kernel reclaims all of it at once, so the
     * binary leaves here unless IRDM_CLEAN_EXIT=1 asks for the orderly teardown (leak checkers, embedding tests). */
    const char *ce = getenv("IRDM_CLEAN_EXIT");
    if (!(ce && ce[0] == '1')) {
        fflush(stderr);
        _exit(rc);
    }
    const double t_down = now_s();
    if (g_group) irdm_group_destroy(g_group);
    else irdm_destroy(p);
    irdm_host_free(rd.buf[0]);
    irdm_host_free(rd.buf[1]);
    free(d);
    if (f != stdin) fclose(f);
    if (timing) fprintf(stderr, "irdm timing: teardown %.3f s\n", now_s() - t_down);
    return rc;
}
