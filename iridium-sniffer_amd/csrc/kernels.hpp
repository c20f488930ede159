// kernels.hpp -- host-callable launchers of the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include "types.hpp"

namespace irdm {

// detect.hip
// kclk (here and below): the kernel's clock record (common.hpp, KClk), or nullptr
int launch_fft_mag_lists(int log_n, int fmt, const void *iq, const float *window, const float2 *tw, float *mag,
                         int n_frames, const float *pre, unsigned *counts, ListEntry *entries, int cap,
                         hipStream_t stream, unsigned long long *kclk = nullptr);
int launch_prefilter_threshold(const float *sum, float thr, float *pre, int n, hipStream_t stream);
int launch_fft_mag(int log_n, int fmt, const void *iq, const float *window, const float2 *tw,
                   float *mag, int n_frames, hipStream_t stream, unsigned long long *kclk = nullptr);
int launch_detect_scan(const DetParams &P, DetState *st, float *sum, float *hist, const float *mag,
                       int n_frames, GoneBurst *gone, int gone_cap, PeakCand *cand_a,
                       PeakCand *cand_b, hipStream_t stream);

// scan_fast.hip
int launch_prefilter(const float *sum, float thr, float *pre, const float *mag, int n,
                     unsigned *counts, ListEntry *entries, unsigned *goff, ListEntry *compact,
                     int n_frames, hipStream_t stream);
int launch_detect_scan_fast(const DetParams &P, DetState *st, float *sum, float *hist, const float *mag,
                            int n_frames, const unsigned *counts, const unsigned *goff,
                            const ListEntry *compact, const float *pre, GoneBurst *gone, int gone_cap,
                            int *status, unsigned long long *mc_ops, int mc_ops_cap, unsigned *mc_done,
                            int mc_updaters, hipStream_t stream);

// scan_band.hip (band-parallel speculative scan; band_core.hpp holds the per-band state machine)
constexpr int kBandRounds = 5;           // speculation rounds before giving up (2-3 on the benchmark scenes)
constexpr int kBandFirst = 3;            // rounds enqueued up front; the rest only if the verdict is still open
struct BandRec;
struct BandParams;
struct BandCtl {                         // control block on the device, copied to the host after the scan
    int32_t status;                      // 0 running, 1 accepted (and committed), 2 aborted (carried state untouched)
    int32_t rounds;
    uint32_t flags;                      // BAND_F_* abort reasons
    int32_t n_upd, n_snap, agree_fail, mismatch, first_mismatch;
    int32_t n_gone, n_total, committed, h0;
    int32_t k_restart, restart_slot;     // the sums pass of this round starts at update step k_restart (a multiple of 64) from the
                                         // snapshot restart_slot the previous round stored: every step before is unchanged (0: from the start)
    int32_t n_restarts, pad1;            // (statistics: rounds of this scan whose sums pass was restarted)
    uint32_t tp[16];                     // plan pass phase stamps (10 ns ticks since the pass began): [0..7] round 1, [8..15] the last verdict
};
struct SumStep {                         // one update step of the sums pass, as the byte offsets its buffer accesses take
    uint32_t nw_off;                     // the magnitude row that enters the sum (offset into the chunk's magnitudes)
    uint32_t ol_off;                     // the row it replaces: steps < kHistory a row of the carried history ring, later ones
                                         // the magnitude row that entered 512 steps earlier
    uint32_t snap_off;                   // offset into BandWork::snap of the snapshot after the step, ~0u: none
    uint32_t pad;
};
// What the history pass needs of an accepted scan, written by its commit (band_tail, below): the pass runs on a side
// stream beside the NEXT chunk's round 0, whose first pass resets the control block and whose plan passes reuse the
// update-step arrays -- so the last <= 512 update frames are listed here.
struct HistJob {
    uint32_t seq;                        // BandParams::seq of the scan that wrote it (a history launch of another scan does nothing)
    int32_t n;                           // rows to copy: min(n_upd, kHistory)
    int32_t h0, n_upd;                   // ring position before the chunk, update steps of the chunk
    int32_t frame[kHistory];             // frame[i]: the magnitude row of update step n_upd - 1 - i
};
struct BandWork {                        // device workspace, carved out of one allocation (band_work_carve)
    BandCtl *ctl;
    uint8_t *uq, *uf;                    // speculated per-frame updates: frame ends quiet / forces an update
    int32_t *cnt_before, *tmp, *upd_frame, *old_row, *snap_after, *need, *snap_slot, *slot_pre, *slot_post;
    SumStep *steps;                      // per update step, padded (the padding reads row 0 and is not applied)
    uint64_t *cross;
    float *relq, *snap;
    int snap_cap;
    uint64_t *occ, *busy, *forced;
    uint32_t *conc;
    BandRec *recs;
    uint32_t *rec_count;
    float *sum_new;
    uint32_t *tot;
    uint64_t *ids;
    uint32_t *flags;
    uint32_t *rank;                      // commit: record -> place in creation order
    unsigned long long *tl;              // [2 halves][2][32] pass timeline (BandParams::tl_sel): per slot earliest start | latest end, 10 ns ticks
    unsigned *walk_host;                 // HOST counter: walk workgroups launched with BandParams::ahead so far (what bar[8] will reach)
    unsigned *bar;                       // [8] walk workgroups done (never reset; BandParams::ahead); [0..2] grid barrier of the cooperative kernel: arrive count, generation, abort (zero
                                         // when idle); [4] the last scan committed; [5] serial number of a void launch;
                                         // band_tail: [9] / [10] workgroups of the crossing / walk pass that have left (back to zero by
                                         // the last one), [11] next entry of the pair list to hand out, [12] entries, [13] of them heavy
    uint32_t *pairs;                     // band_tail: the (band, 64-frame block) pairs with a segment start, listed by the crossing
                                         // pass's last workgroup: heavy ones from the front, the others from the back
    HistJob *hist_job;                   // band_tail: see HistJob
};
extern int g_band_tail;                  // 1: fewer launches per scan (default 0: measured slower) -- the walk pass's last workgroup runs the next plan
                                         // pass (verdict, commit, export), the walk hands out its pairs from a list the crossing
                                         // pass's last workgroup made, the history copy runs on a side stream
extern std::atomic<unsigned long long> g_band_tail_launches;
extern int g_band_sum_restart;           // 1 (default): later rounds' sums passes restart at the last stored state in front of the first changed frame
extern int g_band_hist_side;             // 1: a launch per pass, but the history copy on the side stream (HistJob) and the export in the last plan pass
extern int g_band_tail_threads;          // threads per workgroup of the walk pass with the tail (256 / 512 / 1024)
extern int g_band_sum_bins;              // bins per wavefront of the sums pass: 64 (default), 32 or 16
extern int g_band_plan_ahead;            // 1: plan passes launched ahead on the side stream (-1 until band_resolve_env(): IRDM_PLAN_AHEAD or the default)
constexpr int kBandPlanAheadDefault = 0;
void band_resolve_env();
extern int g_band_fuse_commit;           // 1 (default): the accepting plan pass runs the commit itself
extern int g_band_plan_threads;          // threads of the plan pass's one workgroup: 256, 512 or 1024 (default)
extern int g_band_walk_wave;             // 1 (default): the walk pass with a wavefront per band and segment; 0: a lane per band
extern int g_band_timeline;              // 1: the band scan's passes record a device timeline (read back by the pipeline per chunk)
extern int g_band_selfcheck;             // test hook: BandParams::selfcheck of the launches that follow
extern int g_band_fold_sums0;            // 1 (default): round 0's sums pass inside its plan pass
extern int g_band_cross_groups;          // workgroups of the crossing pass (default 256 = 1024 wavefronts)
extern int g_band_cross_wave;            // 1 (default): crossing pass = fixed grid of frame-walking wavefronts; 0: a workgroup per frame
extern int g_band_coop;                  // 1: the rounds of a band scan as one cooperative launch; 0 (default): a launch per pass
int band_list_cap(int n);                // entries per frame the band scan's lists hold
int band_scan_supported(const DetParams &D, BandParams *out, int n_frames, uint64_t idx0);
size_t band_work_bytes(int n, size_t max_chunk, bool spec = false);
int band_work_carve(BandWork *W, void *base, int n, size_t max_chunk, bool spec = false);
int launch_band_spec(const DetParams &D, BandWork S, DetState *st_spec, const float *sum_src, int n_frames, uint64_t idx0,
                     const unsigned *counts, const ListEntry *entries, int have_prev, hipStream_t stream);
int launch_band_scan(const DetParams &D, BandWork W, DetState *st, float *sum, float *hist, const float *mag,
                     int n_frames, uint64_t idx0, const unsigned *counts, const ListEntry *entries, const float *pre,
                     float *smin, GoneBurst *gone, int gone_cap, int round_begin, int round_end, GoneBurst *hp_gone,
                     uint32_t *hp_hdr, void *hp_ctl, int hp_cap, int chained, int tl_sel, hipStream_t stream,
                     hipStream_t side = nullptr, hipEvent_t *plan_ev = nullptr, const uint32_t *gate_flag = nullptr,
                     uint32_t gate_seq = 0, uint32_t *gate_err = nullptr, const void *gate_src = nullptr,
                     size_t gate_bytes = 0,                                   // gate: see irdm_expect_history
                     uint32_t scan_seq = 0, hipEvent_t hist_wait = nullptr, hipEvent_t hist_done = nullptr,
                     hipEvent_t hist_hop = nullptr, const BandWork *spec = nullptr, hipEvent_t sums_done = nullptr);
// band_tail (scan_seq != 0, side, hist_done and hist_hop given): scan_seq numbers the SCAN (the same for its first launch,
// a continuation and a retry); hist_wait: the previous scan's history copy (on `side`) -- waited for before the first pass
// that may read or overwrite what that copy uses; hist_done: recorded on `side` behind this scan's history copy;
// hist_hop: scratch event that carries the order from `stream` to `side`.
constexpr int kBandTlSlots = 32;         // plan / sums / cross / walk of round r: 4 r + 0..3; commit 24; history 25
// smin != nullptr: keep `pre` where it is lower and cap it by 0.45 * thr * smin (retry after a stale list)
int launch_prefilter_lists(const float *sum, float thr, float *pre, const float *smin, const float *mag, int n,
                           unsigned *counts, ListEntry *entries, int n_frames, int cap, hipStream_t stream);

// where a burst window's samples live: the chunk being fed, or the history ring
struct SampleSource {
    const void *chunk;        // device pointer, configured format
    uint64_t chunk_start;     // absolute index of chunk[0]
    uint64_t chunk_end;
    const void *ring;         // history ring, same format, indexed by absolute index % ring_len
    uint64_t ring_len;
    uint64_t ref_ring;        // the REFERENCE's ring size (burst_detect.c:292-296): stale slot = a - ref_ring
    int fmt;                  // 0 ci8, 2 cf32
};

// downmix.hip
// rows of the checkpoint pool for centre bins that have none yet: news[i] = (bin, row), slot[bin] = row when done
int launch_rotator_rows(const float2 *incr, float2 *table, int n_runs, const int4 *news, int n_new, int *runs, hipStream_t stream);
int fir_tile_row(int decim);
extern int g_fir_force_generic;   // 1: always the runtime-M decimator kernel
extern int g_fir_layout;          // 2 (default): persistent column-major kernel, 1: column-major tile, 0: polyphase rows
extern int g_fir_prof;            // 1: time the persistent decimator's phases (debug)
extern int g_fir_budget;          // persistent kernel: tiles per workgroup before it retires
extern int g_fir_reserve_cus;     // persistent kernel: CUs left free for the other streams
// fir_reg.hip: the register-resident decimator (M = 40, 48)
extern int g_fir_strip;           // double blocks of 128 columns per strip
extern int g_chain_cus;           // CUs the chains' streams may use (IRDM_CHAIN_CU_RESERVE; 0: all)
extern int g_fir_grid;            // workgroups of the register-resident decimator (0: one per strip)
extern int g_fir_slice;           // strips per launch of the register-resident decimator (0: all in one launch)
extern thread_local int g_fir_order;           // 1 (default): the decimating FIR in the order of the reference's AVX2 kernel (simd_avx2.c:62-108),
                                  // 0: of its scalar kernel (simd_generic.c:86-96, --no-simd)
int fir_fma_tile_out(int decim);
int fir_mfma_tile_out(int decim);          // fir_layout 4: the decimator on the matrix cores (fir_reg.hip, fir_decimate_kernel_x)
int launch_fir_mfma(const SampleSource &src, const FirGeom *geom, int n_tiles, int decim, const float *taps,
                    const float2 *rot_table, float2 *dec, hipStream_t stream, unsigned long long *kclk = nullptr);
int launch_fir_fma(const SampleSource &src, const FirGeom *geom, int n_tiles, int decim, const float *taps,
                   const float2 *rot_table, float2 *dec, hipStream_t stream,
                   unsigned long long *kclk = nullptr, unsigned *next_tile = nullptr);
extern int g_fir_claim;           // 1: the resident decimator grid claims its strips from a counter (default 0: measured slower)   // fir_decimate_kernel_f; 0 ok, -1 error, 1 not applicable
int fir_reg_supported(int decim);
int fir_reg_tile_out(int decim);
int launch_fir_reg(const SampleSource &src, const FirGeom *geom, int n_tiles, int decim, const float *taps,
                   const float2 *rot_table, float2 *dec, hipStream_t stream,
                   unsigned long long *kclk = nullptr);   // 0 ok, -1 error, 1 not applicable
int fir_tile_out(int decim, int aligned);   // outputs per FirTile of the kernel launch_fir_decimate() picks (aligned:
                                  // ring_len and ref_ring are multiples of 8 samples)
extern int g_fft_force_radix2;    // 1: always the radix-2 LDS FFT kernel
extern int g_fft_kernel;          // 1 (default): K1 = fft_mag_p32_kernel at N = 8192 / 16384; 0: fft_mag_r16_kernel
int fir_needs_tile_list(int decim, int aligned);
int launch_fir_decimate(const SampleSource &src, const BurstWork *work, int n_bursts, FirTile *tiles, size_t tiles_cap,
                        int n_tiles, int decim, const float *taps, const int *tap_off, const float2 *rot_incr,
                        const float2 *rot_table, int n_ckpt, float2 *dec,
                        hipStream_t stream, unsigned long long *kclk = nullptr,    // kclk: the register-resident kernel's clock record
                        const int *rot_slot = nullptr);                            // rot_slot[bin] = the bin's row in rot_table
// folds a kernel-clock record's slots into its sums and re-arms them (common.hpp); enqueue behind the kernel
int launch_kclk_fold(unsigned long long *kclk, hipStream_t stream);
int launch_gone_export(const DetState *st, const GoneBurst *gone, int cap, GoneBurst *hp_gone, uint32_t *hp_hdr,
                       const void *ctl, void *hp_ctl, int ctl_bytes, hipStream_t stream);
int launch_copy_to_host(void *dst, const void *src, size_t bytes, hipStream_t stream);
int launch_copy2_to_host(void *dst_a, const void *src_a, size_t bytes_a, void *dst_b, const void *src_b, size_t bytes_b,
                         hipStream_t stream);
int launch_copy_words(void *dst, const void *src, size_t bytes, hipStream_t stream);   // bytes % 4 == 0
int launch_copy_wide(void *dst, const void *src, size_t bytes, hipStream_t stream);    // device to device, chunk-sized: DMA (default) or kernel
extern int g_copy_wide;           // 1: by kernel instead of hipMemcpyAsync (A/B)
int launch_wait_host_flag(const uint32_t *flag, uint32_t seq, uint32_t *err, hipStream_t stream);
int launch_gather_burst(const SampleSource &src, uint64_t start, uint64_t avail_end, int n,
                        float2 *out, hipStream_t stream);
extern int g_small_wg;            // threads per workgroup of the little copy / threshold kernels (256; option small_wg 64)
extern int g_post_generic;        // 1: runtime-tap-count instances of post1 / post2 (test hook)
extern int g_rot_store;           // rot_phase: 1 (default) rows through LDS, 0 a row per lane
extern int g_post_split;          // post1 as post_tiles_kernel + post_cfo_kernel (1, default) or one workgroup per burst (0)
int launch_downmix_post1(BurstWork *work, int n_bursts, int max_dec_len, float2 *dec,
                         float2 *lpf, float *box, const float *noise_taps, int noise_ntaps,
                         const float *start_taps, int start_ntaps, int search_depth, int pre_start,
                         const float *cfo_window, const float2 *tw4096, BurstWork *hp_work, hipStream_t stream,
                         unsigned long long *kclk);      // kclk: the decimator's kernel-clock record, folded here (nullptr: none)
// the fine-CFO libm step of the per-burst chain (rot_phase_kernel): on the device, or taken from the host's records
struct CfoStep {
    int on_device;
    int n_fft, sample_rate, out_rate;
    double center_frequency;
};
int launch_downmix_post2(BurstWork *work, int n_bursts, const float2 *lpf,
                         const float *rrc_taps, int rrc_ntaps, const float2 *tw2048,
                         const float2 *dl_fft, const float2 *ul_fft, int dl_len, int ul_len,
                         float sps, float2 *rrc_ws, float2 *frames, const BurstWork *hp_work, const CfoStep &cfo,
                         hipStream_t stream);
int launch_sincosf_probe(const float *x, size_t n, float *re, float *im, hipStream_t stream);

// demod.hip
int launch_ida_decode(const DemodOut *frames, int n_frames, const int2 *syn_da, const int2 *syn_l1, const int2 *syn_l2,
                      const int2 *syn_l3, int use_llr, const int *n_bits, const int *direction, IdaOut *out,
                      hipStream_t stream);
int launch_frame_decode(const DemodOut *frames, int n_frames, const int2 *syn_ra, const int2 *syn_hdr, int use_llr,
                        const int *n_bits, DecodedOut *out, hipStream_t stream);
int launch_demod_pack(const DemodOut *in, int n_bursts, DemodPacked *out, hipStream_t stream);
// hp_packed / hp_work != nullptr (packed_records): demod_par_kernel writes the DemodPacked and work records straight into
// pinned host memory -- the chain's last launch
int launch_demod(const BurstWork *work, int n_bursts, const float2 *frames, int use_gardner,
                 float sps, float2 *ws, DemodOut *out, hipStream_t stream, DemodPacked *hp_packed = nullptr,
                 BurstWork *hp_work = nullptr);

}  // namespace irdm
