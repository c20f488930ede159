// pipeline.hpp -- INTERNAL header of the host side above the C-ABI (include/irdm_hip.h): the context's state and the functions
// its source files share.  One context = one stream of IQ on one GPU, driven by one host thread (the reference's convention for
// gpu_burst_fft_t / burst_downmix_t contexts, burst_downmix.c:107-112).
//
//   plug.cpp        gpu_burst_fft_*: the reference's accelerator plug point (opencl/burst_fft.h:35-47)
//   create.cpp      irdm_create / irdm_destroy, the small getters
//   chain.cpp       the per-burst chain of a chunk (K4 .. K7): rotator checkpoint arena, enqueue, records
//   scan_host.cpp   the detector scan of a chunk from the host's side: launch, chaining, speculation pass, settle, fallbacks
//   feed.cpp        irdm_feed_* / irdm_flush / irdm_advance, the polls, buffers for hosts without HIP headers
//   state.cpp       detector-state export / import (time-chunk sharding), the stage-level batch calls
//   api.cpp         options, statistics, kernel clock, RAW line formatting, --save-bursts
#pragma once
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <errno.h>
#include <sys/stat.h>
#include <algorithm>
#include <deque>
#include <new>
#include <vector>
#include <thread>
#include <mutex>
#include <condition_variable>
#include "../../include/irdm_hip.h"
#include "common.hpp"
#include "host_design.hpp"
#include "kernels.hpp"
#include "libm_port.hpp"
#include "band_core.hpp"
#include "types.hpp"

using namespace irdm;

namespace irdmh {

template <typename T>
inline T *dev_alloc(size_t count)
{
    void *p = nullptr;
    if (hipMalloc(&p, count * sizeof(T)) != hipSuccess) return nullptr;
    return static_cast<T *>(p);
}

template <typename T>
inline T *dev_upload(const T *src, size_t count)
{
    T *p = dev_alloc<T>(count);
    if (!p) return nullptr;
    if (hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(p);
        return nullptr;
    }
    return p;
}

inline int ilog2(int n)
{
    int l = 0;
    while ((1 << l) < n) l++;
    return (1 << l) == n ? l : -1;
}

}  // namespace irdmh


// ===========================================================================
// 2. batched pipeline
// ===========================================================================
struct irdm_pipeline;

// Batch contexts a pipeline may hold (pipeline_depth + 1 of them are used).  A chunk's per-burst chain is a string of
// dependent launches, several of them a handful of wavefronts long (the phase recurrence, the Gardner / PLL loop): 2.5-3 ms
// from first to last in run, whatever the chip could do beside it.  With three contexts the pipeline's period was that
// latency divided by three -- the feeding thread spent 0.6 ms of every 1.05 ms step waiting for the oldest chain
// (profiles/r5_spec_ab.json, host_us "wait_older_chain") -- and nothing done to the scan or the decimator moved it.
constexpr int kMaxBc = 6;
// the low-passed scratch of a batch context holds `cap` float2 outputs and, behind them, `cap` floats: the start filter's
// outputs (post_tiles_kernel -> post_cfo_kernel)
static inline size_t lpf_alloc(size_t cap) { return cap + cap / 2 + 8; }
static inline float *box_of(float2 *lpf, size_t cap) { return reinterpret_cast<float *>(lpf + cap); }
// Feed slots: chunks that may be between irdm_feed_begin and the settling of their scan -- the one being scanned, the one
// whose scan is chained behind it, and TWO begun ahead (round 5: one more than before, so that K1 of chunk k + 2 is on the GPU
// a period early and runs in the stretches where the chains in flight are in their lane-per-burst kernels instead of in
// front of the decimator of every period, kernel trace in profiles/r5_kernel_trace_gantt.txt).  A slot owns a magnitude
// buffer, K1's candidate lists with the levels they were built against, and the events of its K1 and ring copy.
constexpr int kFeedSlots = 3;
constexpr unsigned kLookAhead = 1;

// One batch of finished bursts on its way through the per-burst stages K4..K7.  pipeline_depth 0 uses one context on the
// detector's stream; pipeline_depth >= 1 alternates between two, each on a stream of its own, so that the FIR of one
// chunk's bursts overlaps the latency-bound tail (sync correlation, demodulator, result copies) of the previous one's.
struct BatchCtx {
    irdm_pipeline *owner;
    hipStream_t stream;
    hipEvent_t ev[4];            // FIR begin / FIR end / post end / demod end
    hipEvent_t ev_cfo;           // work records are in the mapped buffer: the helper thread may do the host step
    hipEvent_t ev_rot;           // behind this context's latest build of rotator checkpoints (rot_rows_prepare)
    BurstWork *d_work;
    FirTile *d_tiles;
    size_t tiles_cap;
    float2 *d_dec, *d_lpf, *d_rrc_ws, *d_frames, *d_demod_ws;
    size_t dec_cap;              // outputs (float2) d_dec and d_lpf hold each: a batch's rows lie end to end by actual length
    DemodOut *d_demod;
    DecodedOut *d_decoded;
    IdaOut *d_ida;
    BurstWork *hp_work, *hp_work_dev;   // host / device view of the same mapped pinned buffer
    FirTile *hp_tiles;
    DemodOut *hp_demod;
    DemodPacked *hp_packed;                 // packed_records: the demodulator's result without LLRs, bits 8 per byte (pinned; written by demod_par_kernel)
    uint32_t *hp_flag, *hp_flag_dev;    // [0] sequence number the helper publishes, [1] time-out flag of the waiting kernel
    int4 *hp_rot_new, *hp_rot_new_dev, *d_rot_new;   // (bin, row, from, to) of the checkpoint runs this batch has to build: mapped pinned / device
    uint32_t cfo_seq;
    bool packed;                 // this batch came back as DemodPacked records
    bool cfo_on_device;          // this batch's libm step ran on the device: h_cfreq is filled from the returned records
    std::vector<double> h_cfreq;
    std::vector<irdm_burst_t> recs;
    int n;                       // bursts in flight (0: idle)
    uint64_t chunk_no;           // the chunk they come from
    uint64_t ring_lo = 0, ring_hi = 0;   // absolute sample range this batch's decimator may read from the history ring (incl. the
                                 // stale slots one reference ring length back); it reads nothing behind ev[1]
    bool owns_buffers;           // context 1 allocates its own device scratch; context 0 aliases the pipeline's
    float ms[3];                 // fir, post, demod of the last finished batch
};

struct irdm_pipeline {
    irdm_config_t cfg;
    DetParams P;
    int dev_fmt;                // device sample format == cfg.format: 0 ci8, 1 ci16 (narrowed in the load stage,
                                // main.c:245-246), 2 cf32
    size_t bps;                 // bytes per device sample
    int feed_block, decim, out_rate;
    float peak_signal_db;    // burst_detector_peak_signal over the finished bursts (starts at 0 like the reference's calloc)
    bool dev_cfo;            // the fine-CFO libm step runs on the device (the port reproduces this host's cexpf)
    bool dev_cfo_ok;         // irdm_create's self-check: libm_port.hpp reproduces THIS host's cexpf (option host_cfo cannot override a failed check)
    float sps;
    uint64_t ref_ring, ring_len;
    size_t l_cap;
    int n_ckpt, dec_stride, burst_cap, gone_cap;
    size_t max_chunk;
    int search_depth, pre_start;
    int in_ntaps, noise_ntaps, start_ntaps, rrc_ntaps, dl_len, ul_len;

    hipStream_t stream;      // detector (K1, prefilter, K2)
    hipStream_t bstream;     // per-burst stages + history ring (== stream unless pipeline_depth 1)
    hipStream_t stream2;
    int bstream_prio;        // priority of the per-burst streams
    hipStream_t sstream;     // detector scan kernels; pipeline_depth 1: a stream with CUs of its own (CU mask), so that the
                             // sequential leader wavefront is not slowed down by the per-burst kernels running beside it
    hipEvent_t ev_scan_in, ev_scan_out;
    // band_spec (scan_band.hip): round 0 of chunk k + 1 as a speculation pass on a second workspace and stream, beside chunk
    // k's scan; the scan of chunk k + 1 then opens with round 1
    int band_spec_opt = 1;                  // option band_spec
    void *d_band_spec = nullptr;
    BandWork band_spec = {};
    DetState *d_state_spec = nullptr;       // the carried bursts a speculation pass starts from (the previous pass's survivors)
    hipStream_t stream_spec = nullptr;
    hipEvent_t ev_sums1 = nullptr;          // behind the first sums pass of the latest band-scan launch (its sum_new: the pass's sums)
    hipEvent_t ev_spec_done = nullptr;      // behind the latest speculation pass
    uint64_t spec_for_no = ~0ull;           // the chunk the speculation workspace holds a pass for (~0: none)
    int spec_frames = 0;                    // ... and its frames
    uint64_t stat_spec_passes = 0, stat_spec_scans = 0, stat_sum_restarts = 0;
    uint32_t seq_counter = 0;               // scans numbered so far (HistJob::seq; never 0)
    uint32_t fl_seq = 0;                    // number of the scan in flight
    uint32_t chain_seq = 0;                 // ... of the chained launch (scan_chain_try), taken over by scan_launch
    uint64_t chain_no = 0;                  // the chunk the chained launch in flight (chain_pending) scans
    int fir_order = 1;       // option fir_order / simd_order: 1 simd_avx2.c's operation order, 0 simd_generic.c's (--no-simd); per pipeline
    int fir_generic = 0;     // test hook fir_generic: 1 = always the any-M decimator (what 2 / 4 MHz streams take)
    int post_generic = 0;    // test hook post_generic: 1 = the runtime-tap-count instances of post_tiles / post_cfo / post2
    irdm::BandTune band_tune;   // options band_selfcheck / band_timeline / band_sum_restart
    hipEvent_t ev_sk[2];     // bracket the scan kernel itself on sstream (last_timings[1], bench.py's roofline)
    hipEvent_t ev[10];   // 0 start,1 fft,2 scan,3 pre-fir,4 fir,5 post,6 demod,7 end,8 caller sync

    float *d_window, *d_hist, *d_sum, *d_mag;
    float2 *d_tw, *d_tw4096, *d_tw2048, *d_dl_fft, *d_ul_fft, *d_rot_incr, *d_rot_table;
    DetState *d_state;
    GoneBurst *d_gone;
    PeakCand *d_cand_a, *d_cand_b;
    void *d_ring, *d_stage;
    float *d_in_taps, *d_noise_taps, *d_start_taps, *d_rrc_taps, *d_cfo_window;
    int *d_fir_off;
    BurstWork *d_work;
    FirTile *d_tiles;
    size_t tiles_cap;
    float2 *d_dec, *d_lpf, *d_rrc_ws, *d_frames, *d_demod_ws, *d_probe;
    DemodOut *d_demod;
    DecodedOut *d_decoded;      // post-demod bit layer (bitlayer.hip)
    int2 *d_syn_ra, *d_syn_hdr; // BCH syndrome -> (error count, locator) tables (frame_decode.c:95-135)
    int *d_nbits;
    int decode_frames, decode_ida, detect_only;
    IdaOut *d_ida;
    int2 *d_syn_da, *d_syn_l1, *d_syn_l2, *d_syn_l3;
    int *d_dirs;
    std::vector<IdaOut> h_ida;
    std::deque<irdm_ida_t> q_ida;
    std::vector<DecodedOut> h_decoded;
    std::deque<irdm_decoded_t> q_decoded;
    // sparse scan (scan_fast.hip): prefilter lists, status word, pre-chunk snapshot for the dense fallback
    unsigned *d_counts, *d_goff;
    ListEntry *d_entries, *d_compact;
    float *d_pre, *d_sum_bak, *d_hist_bak;
    DetState *d_state_bak;
    int *d_status;
    unsigned long long *d_mc_ops;   // multi-CU sparse scan: operation words leader -> updaters, completion counters back
    unsigned *d_mc_done;
    int mc_ops_cap, mc_updaters, mc_auto;
    int scan_cus;               // CUs the scan stream may use (its CU mask, or the whole device)
    int scan_mode;              // 0 auto (sparse multi-CU where the device has the CUs, dense fallback), 1 dense only,
                                // 2 sparse on one CU, 3 sparse multi-CU
    uint64_t stat_fast_chunks, stat_fallbacks, stat_dense_frames;
    int host_primed, host_hist_idx;
    // band-parallel speculative scan (scan_band.hip): the default where band_scan_supported()
    void *d_band;               // one allocation, carved into `band`
    BandWork band;
    float *d_smin;              // smallest sum every bin went through in the last band scan (stale-list retry)
    bool band_ok;
    int fl_mode;                // scan in flight: 0 dense, 1 sparse (leader/updaters), 2 band
    int fl_done;                // frames the dense scan primed before the in-flight scan proper
    uint64_t stat_plan_tp[16] = {};
    uint64_t stat_tl_dur[32] = {}, stat_tl_gap[32] = {}, stat_tl_n[32] = {};
    uint64_t stat_band_chunks, stat_band_rounds, stat_band_retries, stat_band_aborts, stat_band_extra, stat_chain_undone, stat_chained;
    uint32_t last_band_flags;

    std::vector<GoneBurst> h_gone;
    // pipeline_depth 1: bursts of the last fed chunk, processed during the next feed / irdm_flush
    std::vector<GoneBurst> pend_gone;
    bool has_pending;
    uint64_t pend_c1, pend_no, fl_no;
    int depth;
    int *h_pin;              // pinned host words: [0..63] scan status, [64..65] n_gone/overflow, [66..67] hist_idx/primed.
                             // (a D2H copy into pageable memory blocks the host until the stream drains -- that would
                             // serialise pipeline_depth 1's deferred work behind the detector scan)
    int deferred_emitted;
    bool caller_ordered;     // the current chunk was handed over on a stream (ev[8] recorded)
    int k1_first;            // per-burst chains start behind K1 (1) / K1 + ring copy (2) of the chunk just fed
    // detector scan in flight (scan_launch .. scan_finish)
    bool fl_active, fl_sparse;
    bool fl_band_ran;        // the band scan's control block on the device belongs to the scan in flight
    const float *fl_mag, *d_mag_last;
    int fl_frames;
    uint64_t fl_c1, fl_c0;
    hipStream_t fstream;     // K1 (== stream unless pipeline_depth 1)
    float *d_mag2;           // pipeline_depth 1: second magnitude buffer
    std::vector<BurstWork> h_work;
    // the per-burst chains (bursts_enqueue / bursts_finish) and the helper thread that does their host step
    BatchCtx bc[kMaxBc];
    int n_bc;
    std::thread cfo_thread;
    std::mutex cfo_mu;
    std::condition_variable cfo_cv;
    std::deque<BatchCtx *> cfo_jobs;
    bool cfo_quit;
    GoneBurst *hp_gone;         // pinned copy of the finished-burst records of a scan
    int hp_gone_cap;
    // Two sets of the scan's export targets (pinned words, pinned records, timing events): a band scan launched BEHIND
    // the one still in flight (scan_chain) exports into the other set.  h_pin / hp_gone / ev_sk / ev_end alias the set of
    // the scan that scan_finish settles next.
    int *h_pin_set[2];
    GoneBurst *hp_gone_set[2];
    hipEvent_t ev_sk_set[2][2], ev_end_set[2], ev_end;
    int out_sel;
    bool chain_pending;         // feed_end has enqueued this chunk's scan behind the previous one
    int chain_sel, chain_band_first;
    bool settle_clean;          // the scan settled last committed on its own (no continuation, retry or fallback)
    hipEvent_t ev_ring;         // pipeline_depth >= 1: the history-ring copy of the last fed chunk
    uint64_t chunk_no;          // chunks fed so far
    // chunks between irdm_feed_begin and irdm_feed_end: at most one at pipeline_depth 0, two (one chunk of look-ahead:
    // K1 of chunk N+1 is on the GPU before the host waits for the scan of chunk N-1) otherwise
    struct FeedSlot {
        const void *iq;
        uint64_t c0, c1;
        float *mag;
        int frames;
        bool in_ring;           // the caller wrote the chunk where irdm_ingest_ptr() said: no copy into the ring
        bool lists;             // K1 wrote the band scan's candidate lists of the chunk (k1_pre / k1_counts / k1_entries)
        hipEvent_t ev_start, ev_k1, ev_copy;
    } fs[kFeedSlots];
    // candidate lists written by K1 (fft_mag_r16_kernel<.., LISTS>), one set per feed slot: the reference levels the
    // lists were built against, the per-frame counts and entries
    float *k1_pre[kFeedSlots];
    unsigned *k1_counts[kFeedSlots];
    ListEntry *k1_entries[kFeedSlots];
    int k1_lists;               // option: 1 = let K1 build the lists where it can
    int band_first;             // band-scan rounds enqueued up front: 0 = as many as the previous chunk needed (at least
                                // 2, kBandFirst to begin with), n = always n (test hook)
    int band_auto, fl_band_first;
    const FeedSlot *fl_feed;    // feed slot of the scan in flight
    uint64_t stat_k1_lists;
    uint64_t begin_no, end_no;  // feeds begun / ended; slot = number % kFeedSlots
    uint64_t begun_samples;     // absolute index the next irdm_feed_begin starts at
    float *d_mag3;
    double host_us[10];         // pipeline_depth >= 1, accumulated host time: K1+ring enqueue, settle, chain enqueue, scan enqueue, wait for the older chain, final sync
    std::vector<FirTile> h_tiles;
    std::vector<DemodOut> h_demod;
    std::vector<float> h_frames;

    // result queues
    std::deque<irdm_burst_t> q_bursts;
    std::deque<irdm_frame_info_t> q_frames;
    std::deque<std::vector<float>> q_frame_samples;
    std::deque<irdm_demod_t> q_demods;
    std::deque<irdm_demod_packed_t> q_packed;
    int packed_records;         // option: queue irdm_demod_packed_t records only
    // option "chunk_marks": one mark per batch of records pushed to the queues above -- which chunk (in the order fed)
    // they belong to and how many records went to each queue -- for a caller that merges the records of several contexts
    // in stream order (group.cpp)
    std::deque<irdm_chunk_mark_t> q_marks;
    int chunk_marks;

    uint64_t total_samples, tagged, start_time_ns;
    bool stream_closed;
    // last chunk (probes)
    int last_frames;
    const void *last_chunk;
    uint64_t last_chunk_start, last_chunk_end;
    std::vector<irdm_burst_t> last_bursts;
    float last_ms[6];
    int keep_frame_samples;
    // rotator checkpoint rows on demand (rot_rows_prepare)
    int *d_rot_slot = nullptr;              // [n][rot_runs] centre bin, run -> block of d_rot_table, -1: none yet (written by the kernel that builds the run)
    int rot_runs = 0;                       // runs of kRotRun checkpoints per bin
    int rot_rows_used = 0;                  // bins that have a row
    int rot_blocks_used = 0, rot_blocks_cap = 0;
    std::vector<int> rot_len_h, rot_want, rot_touched;   // per bin: checkpoints built (or being built) / wanted by the batch at hand
    std::vector<int> rot_build_ctx;         // per bin: the batch context whose chain built (or is building) its latest run, -1: none
    std::vector<uint32_t> rot_build_gen;    // ... and which of that context's builds it was
    uint32_t rot_gen[kMaxBc] = {}, rot_done_gen[kMaxBc] = {};   // per context: builds enqueued / known to be complete (its stream was waited for)
    std::vector<float2 *> rot_retired;      // outgrown pools
    uint64_t stat_rot_builds = 0, stat_rot_rows = 0, stat_rot_ckpts = 0, stat_rot_grows = 0, stat_band_steps = 0;
    // rot_prebuild (default where pipeline_depth >= 1): every centre bin's row, as far as an ordinary burst needs it, built by
    // ONE background launch behind create instead of by the chains that first meet the bin (a stream's first chunks bring
    // hundreds of new bins each: 2.4-3.9 ms of checkpoint recurrence in front of a chain, 11.5 ms at 12 MHz dense)
    hipStream_t stream_rot_pre = nullptr;
    hipEvent_t ev_rot_pre = nullptr;
    int4 *d_rot_pre_news = nullptr;
    bool rot_pre_pending = false;           // chains wait for ev_rot_pre until the host has seen it complete
    int rot_pre_runs = 0;                   // runs of kRotRun checkpoints every bin's row was prebuilt with (0: none)
    size_t scratch_init = 0;                 // outputs the decimated / low-passed scratch of a context holds to begin with
    std::vector<float2 *> scratch_retired;   // outgrown scratch (freed when the context is closed, like the rotator pools)
    std::vector<void *> tiles_retired, tiles_host_retired;   // outgrown strip lists (device / pinned host): likewise
    uint64_t stat_scratch_grows = 0, stat_scratch_peak = 0, stat_tiles_grows = 0, stat_ring_waits = 0;
    // time-chunk sharding: the previous chunk's 512-frame history may arrive AFTER this chunk's scan has been enqueued
    // (irdm_expect_history / irdm_import_state_history_device): [0] sequence number the import publishes, [1] time-out
    // flag of the waiting kernel, in mapped pinned memory; the import's copies run on gstream
    uint32_t *hp_gate = nullptr, *hp_gate_dev = nullptr;
    uint32_t gate_seq = 0;
    bool gate_armed = false;        // the next band scan enqueued from frame 0 waits for gate_seq before its first history read
    bool gate_open_pending = false; // a scan in flight waits for the history to arrive in gate_src
    const void *gate_src = nullptr; // the caller's receive buffer (device memory) the scan copies the history from
    // kernel clock (option "kernel_clock", common.hpp): records 0..2 the decimator of bc[0..2], 3..5 K1 of feed slot 0..2
    unsigned long long *d_kclk = nullptr;
    int kernel_clock = 0;
    unsigned long long *kclk_rec(int i) const { return kernel_clock && d_kclk ? d_kclk + (size_t)i * kKClkWords : nullptr; }
    // (the decimator of batch context c: records 0..2, and 6.. for the contexts beyond the third)
    unsigned long long *kclk_fir(int c) const { return kclk_rec(c < 3 ? c : 3 + c); }
};

// The rotator checkpoints (rotator.h:36-46 restated: the phase of the float recurrence every 16 samples -- a whole row for
// a centre bin would be l_cap / 16 of them, 0.55 MB at 10 MHz) are kept for the centre bins bursts have actually appeared
// on and as far as those bursts have needed them: an arena of blocks of kRotRun checkpoints, a bin's row = its list of
// blocks (d_rot_slot[bin][run]), built by one lane per bin on the chain that needs them (the recurrence is sequential:
// 9 ns a sample) and continued when a longer burst comes.  The whole table -- a row for every FFT bin, 4.5 GB at 10 MHz
// and 10.9 GB at 12 MHz, built at create in rounds 1-3 -- is the arena's upper bound: it grows by doubling; an arena that
// has been outgrown stays allocated until the context is closed (chains in flight still read it).

// Every C-ABI entry that enqueues work for a pipeline: the calling thread is put on the pipeline's device.  (Every switch a
// launch helper reads -- the arithmetic order, the test hooks -- is a field of the pipeline and travels as an argument: two
// contexts of one process may differ in all of them.)
static inline void pipeline_enter(const irdm_pipeline *p) { (void)hipSetDevice(p->cfg.device); }

namespace irdmh {

template <typename T>
static int drain(std::deque<T> &q, T *out, int max)
{
    int n = 0;
    while (n < max && !q.empty()) {
        out[n++] = q.front();
        q.pop_front();
    }
    return n;
}

// create.cpp
void pipeline_free(irdm_pipeline *p);
bool chain_stream_create(irdm_pipeline *, hipStream_t *out, int prio);

// chain.cpp
SampleSource make_source(const irdm_pipeline *p, const void *chunk, uint64_t c0, uint64_t c1);
int ring_guard(irdm_pipeline *p, uint64_t a0, uint64_t a1, hipStream_t st);
int ring_update(irdm_pipeline *p, const void *d_iq, uint64_t c0, uint64_t c1, hipStream_t st);
irdm_decoded_t finish_decoded(const DecodedOut &d, uint64_t id, uint64_t timestamp, double frequency);
int lcw_field(const char *b, int from, int to);
void format_lcw_header(int ft, int lcw_ft, int lcw_code, uint32_t lcw3_val, char *out, size_t outsz);
irdm_ida_t finish_ida(const IdaOut &d, const irdm_demod_t &f);
void cfreq_from_records(BatchCtx &b);
void fine_cfo_host(BatchCtx &b);
void cfo_helper_main(irdm_pipeline *p);
int rot_arena_reset(irdm_pipeline *p, long long blocks);
int rot_prebuild(irdm_pipeline *p);
int rot_rows_prepare(irdm_pipeline *p, BatchCtx &b, int nb, hipStream_t st);
int bursts_enqueue(irdm_pipeline *p, BatchCtx &b, const SampleSource &src, const GoneBurst *gone_list, int nb);
int bursts_finish(irdm_pipeline *p, BatchCtx &b);
int bursts_finish_records(irdm_pipeline *p, BatchCtx &b);
int process_bursts(irdm_pipeline *p, BatchCtx &b, const SampleSource &src, const GoneBurst *gone_list, int n_gone);

// scan_host.cpp
int hist_fence(irdm_pipeline *);
uint32_t next_scan_seq(irdm_pipeline *p);
int scan_hop_in(irdm_pipeline *p);
int scan_hop_out(irdm_pipeline *p);
int scan_dense(irdm_pipeline *p, const float *mag, int n_frames, bool timed);
int scan_pick(const irdm_pipeline *p);
int scan_snapshot(irdm_pipeline *p);
int scan_restore(irdm_pipeline *p);
int scan_band_enqueue_at(irdm_pipeline *p, const float *mag, int n_frames, int done, int retry, bool more_rounds,
                                uint64_t c0, const irdm_pipeline::FeedSlot *feed, int sel, int first, int chained,
                                uint32_t seq, uint64_t chunk_no, bool use_spec = false);
int scan_band_enqueue(irdm_pipeline *p, const float *mag, int n_frames, int done, int retry, bool more_rounds = false);
int scan_legacy_enqueue(irdm_pipeline *p, const float *mag, int n_frames, int done, bool sparse);
int scan_export(irdm_pipeline *p);
void scan_select_outputs(irdm_pipeline *p, int sel);
int scan_chain_try(irdm_pipeline *p, irdm_pipeline::FeedSlot &f, uint64_t no);
int spec_enqueue(irdm_pipeline *p, irdm_pipeline::FeedSlot &nx, uint64_t no);
int scan_launch(irdm_pipeline *p, const float *mag, int n_frames, uint64_t c1);
int scan_finish(irdm_pipeline *p, int *n_gone_out);
int settle(irdm_pipeline *p);
int quiesce(irdm_pipeline *p);
int deferred_enqueue(irdm_pipeline *p);
int deferred_finish(irdm_pipeline *p, BatchCtx &b);

}  // namespace irdmh
