// kernels.hpp -- host-callable launchers of the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "types.hpp"

namespace irdm {

// detect.hip
// kclk (here and below): the kernel's clock record (common.hpp, KClk), or nullptr
// order (here and below): the reference's AVX2 forms of its dispatched kernels (1, simd_avx2.c) or the generic ones (0,
// simd_generic.c, --no-simd); per pipeline (option fir_order)
int launch_fft_mag_lists(int log_n, int fmt, const void *iq, const float *window, const float2 *tw, float *mag,
                         int n_frames, const float *pre, unsigned *counts, ListEntry *entries, int cap,
                         hipStream_t stream, unsigned long long *kclk, int order);
int launch_prefilter_threshold(const float *sum, float thr, float *pre, int n, hipStream_t stream);
int launch_fft_mag(int log_n, int fmt, const void *iq, const float *window, const float2 *tw,
                   float *mag, int n_frames, hipStream_t stream, unsigned long long *kclk, int order);
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
    unsigned *bar;                       // [4] the last scan committed; [5] serial number of a void launch
};
// switches of the band scan a pipeline carries (per pipeline; options band_selfcheck / band_timeline / band_sum_restart)
struct BandTune {
    int selfcheck = 0;                   // test hook: BandParams::selfcheck of the launches
    int timeline = 0;                    // 1: the passes record a device timeline (read back by the pipeline per chunk)
    int sum_restart = 1;                 // 1 (default): later rounds' sums passes restart at the last stored state in front of the first changed frame
};
int band_list_cap(int n);                // entries per frame the band scan's lists hold
int band_scan_supported(const DetParams &D, BandParams *out, int n_frames, uint64_t idx0);
size_t band_work_bytes(int n, size_t max_chunk, bool spec = false);
int band_work_carve(BandWork *W, void *base, int n, size_t max_chunk, bool spec = false);
int launch_band_spec(const DetParams &D, BandWork S, DetState *st_spec, const float *sum_src, int n_frames, uint64_t idx0,
                     const unsigned *counts, const ListEntry *entries, int have_prev, hipStream_t stream, const BandTune &tune);
int launch_band_scan(const DetParams &D, BandWork W, DetState *st, float *sum, float *hist, const float *mag,
                     int n_frames, uint64_t idx0, const unsigned *counts, const ListEntry *entries, const float *pre,
                     float *smin, GoneBurst *gone, int gone_cap, int round_begin, int round_end, GoneBurst *hp_gone,
                     uint32_t *hp_hdr, void *hp_ctl, int hp_cap, int chained, int tl_sel, hipStream_t stream,
                     const BandTune &tune, const uint32_t *gate_flag = nullptr, uint32_t gate_seq = 0, uint32_t *gate_err = nullptr,
                     const void *gate_src = nullptr, size_t gate_bytes = 0,              // gate: see irdm_expect_history
                     const BandWork *spec = nullptr, hipEvent_t sums_done = nullptr);
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
// fir_reg.hip: the register-resident decimators (M = 40, 48): fir_decimate_kernel_f (AVX2 order) / _r (scalar order);
// 0 ok, -1 error, 1 not applicable
int fir_reg_supported(int decim);
int fir_fma_tile_out(int decim);
int fir_reg_tile_out(int decim);
int launch_fir_fma(const SampleSource &src, const FirGeom *geom, int n_tiles, int decim, const float *taps,
                   const float2 *rot_table, float2 *dec, hipStream_t stream, unsigned long long *kclk = nullptr);
int launch_fir_reg(const SampleSource &src, const FirGeom *geom, int n_tiles, int decim, const float *taps,
                   const float2 *rot_table, float2 *dec, hipStream_t stream, unsigned long long *kclk = nullptr);
// the decimator a batch takes (downmix.hip): aligned = ring_len and ref_ring are multiples of 8 samples; generic = the test
// hook fir_generic (always the any-M kernel)
int fir_tile_out(int decim, int aligned, int generic, int order);   // outputs per FirTile of the kernel launch_fir_decimate() picks
int fir_needs_tile_list(int decim, int aligned, int generic);
int launch_fir_decimate(const SampleSource &src, const BurstWork *work, int n_bursts, FirTile *tiles, size_t tiles_cap,
                        int n_tiles, int decim, const float *taps, const int *tap_off, const float2 *rot_incr,
                        const float2 *rot_table, int n_ckpt, float2 *dec,
                        hipStream_t stream, unsigned long long *kclk,     // kclk: the register-resident kernel's clock record
                        const int *rot_slot, int order, int generic);    // rot_slot[bin][run] = the block of rot_table
// folds a kernel-clock record's slots into its sums and re-arms them (common.hpp); enqueue behind the kernel
int launch_kclk_fold(unsigned long long *kclk, hipStream_t stream);
int launch_gone_export(const DetState *st, const GoneBurst *gone, int cap, GoneBurst *hp_gone, uint32_t *hp_hdr,
                       const void *ctl, void *hp_ctl, int ctl_bytes, hipStream_t stream);
int launch_copy_to_host(void *dst, const void *src, size_t bytes, hipStream_t stream);
int launch_copy2_to_host(void *dst_a, const void *src_a, size_t bytes_a, void *dst_b, const void *src_b, size_t bytes_b,
                         hipStream_t stream);
int launch_copy_words(void *dst, const void *src, size_t bytes, hipStream_t stream);   // bytes % 4 == 0
int launch_wait_host_flag(const uint32_t *flag, uint32_t seq, uint32_t *err, hipStream_t stream);
int launch_gather_burst(const SampleSource &src, uint64_t start, uint64_t avail_end, int n,
                        float2 *out, hipStream_t stream);
int launch_downmix_post1(BurstWork *work, int n_bursts, int max_dec_len, float2 *dec,
                         float2 *lpf, float *box, const float *noise_taps, int noise_ntaps,
                         const float *start_taps, int start_ntaps, int search_depth, int pre_start,
                         const float *cfo_window, const float2 *tw4096, BurstWork *hp_work, hipStream_t stream,
                         unsigned long long *kclk,       // kclk: the decimator's kernel-clock record, folded here (nullptr: none)
                         int order, int generic);        // generic: the runtime-tap-count instances (test hook post_generic)
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
                         hipStream_t stream, int order, int generic);
int launch_sincosf_probe(const float *x, size_t n, float *re, float *im, hipStream_t stream);

// demod.hip
int launch_ida_decode(const DemodOut *frames, int n_frames, const int2 *syn_da, const int2 *syn_l1, const int2 *syn_l2,
                      const int2 *syn_l3, int use_llr, const int *n_bits, const int *direction, IdaOut *out,
                      hipStream_t stream);
int launch_frame_decode(const DemodOut *frames, int n_frames, const int2 *syn_ra, const int2 *syn_hdr, int use_llr,
                        const int *n_bits, DecodedOut *out, hipStream_t stream);
// hp_packed / hp_work != nullptr (packed_records): demod_par_kernel writes the DemodPacked and work records straight into
// pinned host memory -- the chain's last launch
int launch_demod(const BurstWork *work, int n_bursts, const float2 *frames, int use_gardner,
                 float sps, float2 *ws, DemodOut *out, hipStream_t stream, DemodPacked *hp_packed = nullptr,
                 BurstWork *hp_work = nullptr);

}  // namespace irdm
