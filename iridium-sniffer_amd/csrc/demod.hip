// demod.hip -- stage C on gfx950 (qpsk_demod.c:393-535): Gardner-timed DQPSK demodulation.  The sequential
// per-symbol recurrences (<= 448 steps) run one frame per lane, the per-symbol independent work one symbol per lane.
//
// Numerics: float32 in the reference's operation order.  cabsf is reproduced
// exactly (double sqrt); cargf/atan2f, cosf, sinf come from the device libm and
// may differ from glibc in the last ulp -> soft outputs (level, LLR, refined
// frequency) are compared with tolerance 1e-4, hard bits exactly.
#include "common.hpp"
#include "types.hpp"
#include "kernels.hpp"

namespace irdm {

#define IRDM_PI_F 3.14159265358979323846f
constexpr float kSqrt1_2 = 0.70710678118654752f;

__device__ __forceinline__ float2 cscale(float a, float2 v) { return make_float2(a * v.x, a * v.y); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// qpsk_demod.c:56-81 Catmull-Rom, operation order of the C expressions
__device__ float2 cubic_interp(const float2 *in, int n, float pos)
{
    int idx = (int)pos;
    const float mu = pos - (float)idx;
    if (idx < 1) idx = 1;
    if (idx >= n - 2) idx = n - 3;
    const float2 s0 = in[idx - 1], s1 = in[idx], s2 = in[idx + 1], s3 = in[idx + 2];
    const float mu2 = mu * mu;
    const float mu3 = mu2 * mu;
    // a = -0.5f*s0 + 1.5f*s1 - 1.5f*s2 + 0.5f*s3
    const float2 a = cadd(csub(cadd(cscale(-0.5f, s0), cscale(1.5f, s1)), cscale(1.5f, s2)), cscale(0.5f, s3));
    // b = s0 - 2.5f*s1 + 2.0f*s2 - 0.5f*s3
    const float2 b = csub(cadd(csub(s0, cscale(2.5f, s1)), cscale(2.0f, s2)), cscale(0.5f, s3));
    // c = -0.5f*s0 + 0.5f*s2
    const float2 c = cadd(cscale(-0.5f, s0), cscale(0.5f, s2));
    // a*mu3 + b*mu2 + c*mu + d
    return cadd(cadd(cadd(cscale(mu3, a), cscale(mu2, b)), cscale(mu, c)), s1);
}

// Two kernels.
//
// demod_seq_kernel -- the true recurrences, ONE LANE PER FRAME, 32 frames per workgroup of TWO wavefronts:
//   * wavefront 0, the timing loop (qpsk_demod.c:85-130, or decimate_simple :134-141): position n+1 depends on the timing
//     error of position n -- four samples at pos and four at pos - sps/2, two interpolations, the error: ~130 instructions per
//     symbol;
//   * wavefront 1, the PLL (:145-195): phi_{i+1} depends on phi_i through cabsf / two divisions / atan2f / sincosf / cabsf /
//     two divisions: ~240 instructions per symbol, almost all of them on the chain.
//   The two recurrences do not feed each other (the reference runs them one after the other over the whole frame), and a
//   lone wavefront issues an instruction every 6-7 cycles whether or not it depends on the one before (profiles/
//   r3_valu_issue.txt): with both loops in one wavefront (rounds 3-5: one loop body, skewed by a symbol so that the scheduler
//   could interleave them) a symbol cost the SUM of their instructions.  Here the timing wavefront hands its symbols over in
//   blocks of kSeqBlock through LDS, a workgroup barrier per block, and runs one block ahead of the PLL wavefront.
//   The samples reach the timing loop through an LDS window, not from memory: its loads depend on the position the
//   previous symbol's error has just set, and beside the decimator and K1 of the other chunks a dependent load took
//   ~1.2 us -- the timing loop ALONE kept the launch at 0.62 ms in run (0.29 alone; the PLL alone: 0.35 / 0.26;
//   profiles/r6_demod_split.json).  The positions advance by sps +- 0.5 a symbol, so what the loop will read is known
//   blocks ahead: every block the wavefront requests the next kSeqChunk samples of each frame (plain 16-byte loads, in
//   flight while the block's symbols are computed), drops them into a ring of kSeqRing samples per frame at the block's
//   end, and the interpolations read the ring (64-byte LDS reads).  A frame whose position has drifted out of the ring
//   (more than +28 / -58 samples off sps per symbol: the loop's own limits allow it, a signal does not do it) reads memory
//   for that symbol as before -- the ring is a cache, the arithmetic is cubic_interp's either way.
//   (One wavefront per frame -- the first layout: lane 0 ran the chains out of 48 KB of LDS -- kept 667 wavefronts and
//   125 KB of every CU's LDS busy for as long as the longest chain.)
// demod_par_kernel -- everything that is per-symbol independent, one wavefront per frame, short:
//   slicer, per-symbol magnitudes, confidence flags, unique-word angles, |.| for the LLR scale: one symbol per lane;
//   only the float sums whose order matters (level :247, LLR scale :489-497) and the end-of-frame rule (:210-225, a
//   running maximum) are walked in order by lane 0, over values already computed.
constexpr int kSeqFrames = 32;          // frames per workgroup (lanes 0..31 of either wavefront)
constexpr int kSeqBlock = 4;            // symbols per hand-over block
constexpr int kSeqChunk = 40;           // samples per frame requested per block (kSeqBlock symbols at 10 samples per symbol)
constexpr int kSeqRing = 128;           // samples per frame in the LDS window
constexpr int kSeqLead = 64;            // samples in the window before the first symbol
constexpr int kSeqPitch = kSeqRing + 3; // + the first three slots once more behind the last (a 4-sample read never wraps); odd
                                        //   pitch: the 32 rows start on different banks
static_assert(kSeqChunk % 8 == 0 && kSeqLead % 32 == 0 && kSeqRing % 4 == 0 && kMaxFrameSamples % 4 == 0, "whole 4-sample pieces, half of them per half wavefront");

// qpsk_demod.c:56-81 on four samples already fetched
__device__ __forceinline__ float2 cubic_interp4(float2 s0, float2 s1, float2 s2, float2 s3, float mu)
{
    const float mu2 = mu * mu;
    const float mu3 = mu2 * mu;
    const float2 a = cadd(csub(cadd(cscale(-0.5f, s0), cscale(1.5f, s1)), cscale(1.5f, s2)), cscale(0.5f, s3));
    const float2 b = csub(cadd(csub(s0, cscale(2.5f, s1)), cscale(2.0f, s2)), cscale(0.5f, s3));
    const float2 c = cadd(cscale(-0.5f, s0), cscale(0.5f, s2));
    return cadd(cadd(cadd(cscale(mu3, a), cscale(mu2, b)), cscale(mu, c)), s1);
}

__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4))) void demod_seq_kernel(const BurstWork *__restrict__ work, int n_bursts,
                                                        const float2 *__restrict__ frames, int use_gardner, float sps,
                                                        float2 *__restrict__ ws, DemodOut *__restrict__ out)
{
    // the frames' sample windows (dynamic: with the 33.5 KB in the kernel's static size the compiler takes the workgroups
    // an LDS can hold for the occupancy and allocates 132 registers; two wavefronts of 128 fit the one 256-register slot
    // the decimator's resident grid leaves on a SIMD)
    extern __shared__ __attribute__((aligned(16))) unsigned char seq_smem[];
    float2 *const s_win = reinterpret_cast<float2 *>(seq_smem);                       // kSeqFrames * kSeqPitch
    __shared__ float2 s_sym[2][kSeqBlock][kSeqFrames];      // [block parity][symbol of the block][frame]
    __shared__ int s_cnt[2][kSeqFrames];                    // symbols of the block, per frame
    __shared__ int s_more[2];                               // a frame of the workgroup goes on behind the block
    // a handful of wavefronts whose dependent chains decide when the chunk's records are complete, sharing SIMDs with
    // the decimator's wavefronts of the next chunk: they go first when they have an instruction ready
    __builtin_amdgcn_s_setprio(3);
    const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;   // (wavefront-uniform: scalar branches)
    // lane l < 32 runs frame l's chain; lane l + 32 only carries the other half of that frame's sample requests
    const int b = blockIdx.x * kSeqFrames + (lane & (kSeqFrames - 1));
    const bool frame_ok = b < n_bursts && work[b].drop_reason == 0;
    const bool active = frame_ok && lane < kSeqFrames;
    const int upper = lane >> 5;
    const int n_samples = frame_ok ? work[b].num_samples : 0;
    const float2 *fr = frames + (size_t)(frame_ok ? b : 0) * kMaxFrameSamples;
    float2 *po = ws + (size_t)(frame_ok ? b : 0) * 2 * kMaxSymbols + kMaxSymbols;
    const int row = (lane & (kSeqFrames - 1)) * kSeqPitch;

    // ---- wavefront 0: the symbols ----
    float pos = 0.0f, toff = 0.0f;
    float2 prev = make_float2(0.0f, 0.0f);
    int made = 0;
    const float lim = (float)(n_samples - 3);
    const int step = (int)sps;
    int n_simple = step > 0 ? (n_samples + step - 1) / step : 0;
    if (n_simple > kMaxSymbols) n_simple = kMaxSymbols;
    int filled = 0;                                 // the window holds the samples [filled - kSeqRing, filled) of every frame
    // four samples [i, i + 4) of the frame out of memory (zeros behind the frame buffer's end: never interpolated)
    auto request = [&](int i, float4 &lo, float4 &hi) {
        lo = hi = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (frame_ok && i + 4 <= kMaxFrameSamples) {
            const float4 *g = reinterpret_cast<const float4 *>(fr + i);
            lo = g[0];
            hi = g[1];
        }
    };
    // ... into their ring slots (i % 4 == 0: the piece does not wrap), slots 0..2 once more behind the ring's end
    auto deposit = [&](int i, const float4 &lo, const float4 &hi) {
        const int slot = i & (kSeqRing - 1);
        float2 *w = s_win + row + slot;
        w[0] = make_float2(lo.x, lo.y);
        w[1] = make_float2(lo.z, lo.w);
        w[2] = make_float2(hi.x, hi.y);
        w[3] = make_float2(hi.z, hi.w);
        if (slot == 0) {
            w[kSeqRing] = make_float2(lo.x, lo.y);
            w[kSeqRing + 1] = make_float2(lo.z, lo.w);
            w[kSeqRing + 2] = make_float2(hi.x, hi.y);
        }
    };
    // cubic_interp (qpsk_demod.c:56-81) with the four samples from the window where it holds them
    auto interp = [&](float p) -> float2 {
        int idx = (int)p;
        const float mu = p - (float)idx;
        if (idx < 1) idx = 1;
        if (idx >= n_samples - 2) idx = n_samples - 3;
        const bool in_win = idx - 1 >= filled - kSeqRing && idx + 2 < filled;
        const float2 *w = s_win + row + ((idx - 1) & (kSeqRing - 1));
        float2 r = cubic_interp4(w[0], w[1], w[2], w[3], mu);
        // (the whole interpolation under the branch: a wait for these loads is a wait for the block's requests in front of
        // them too -- loads return in order -- and with only the loads under the branch the wait stood behind it, in every
        // symbol's way)
        if (!in_win) r = cubic_interp4(fr[idx - 1], fr[idx], fr[idx + 1], fr[idx + 2], mu);
        return r;
    };
    auto produce_block = [&](int buf) {
        int cnt = 0;
        if (use_gardner) {
            // the next kSeqChunk samples of every frame: requested now (lanes l and l + 32 half of frame l's each), in the
            // window from the next block on
            constexpr int NP = kSeqChunk / 8;
            float4 lo[NP], hi[NP];
            const int piece0 = filled + 4 * NP * upper;
#pragma unroll
            for (int i = 0; i < NP; i++) request(piece0 + 4 * i, lo[i], hi[i]);
            // decimate_gardner: symbol `made` at the loop's position, then the timing update (the position of the next one)
#pragma unroll 1
            for (int k = 0; k < kSeqBlock; k++) {
                const bool live = active && pos < lim && made < kMaxSymbols;
                if (__builtin_amdgcn_ballot_w64(live) == 0) break;
                if (live) {
                    const float mid_pos = pos - sps * 0.5f;
                    // the on-time sample and the mid-point sample (used only where the reference computes it) are independent
                    const float2 on = interp(pos);
                    const float2 mid = interp(mid_pos >= 1.0f ? mid_pos : 1.0f);
                    s_sym[buf][k][lane] = on;
                    const bool upd = made > 0 && mid_pos >= 1.0f;
                    const float2 diff = csub(prev, on);
                    // crealf(diff * conjf(mid)) = diff.x*mid.x - diff.y*(-mid.y)
                    const float p0 = diff.x * mid.x, p1 = diff.y * (-mid.y);
                    float err = p0 - p1;
                    err = err > 1.0f ? 1.0f : err;
                    err = err < -1.0f ? -1.0f : err;
                    const float t1 = toff + 0.0002f * err;
                    float adj = 0.02f * err + t1;
                    adj = adj > 0.5f ? 0.5f : adj;
                    adj = adj < -0.5f ? -0.5f : adj;
                    toff = upd ? t1 : toff;
                    pos = upd ? pos + adj : pos;
                    prev = on;
                    pos += sps;
                    made++;
                    cnt++;
                }
            }
            // (behind the block's last interpolation in every lane: the wavefront runs in lockstep -- an ordering point for
            // the compiler and for the CPU emulation, whose lanes do not; no instruction)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < NP; i++) deposit(piece0 + 4 * i, lo[i], hi[i]);
            filled += kSeqChunk;
        } else {
            // decimate_simple: every (int)sps-th sample
#pragma unroll
            for (int k = 0; k < kSeqBlock; k++) {
                if (active && made < n_simple) {
                    s_sym[buf][k][lane] = fr[(size_t)made * step];
                    made++;
                    cnt++;
                }
            }
        }
        if (lane < kSeqFrames) s_cnt[buf][lane] = cnt;
        const bool more = use_gardner ? (active && pos < lim && made < kMaxSymbols) : (active && made < n_simple);
        const unsigned long long any = __builtin_amdgcn_ballot_w64(more);
        if (lane == 0) s_more[buf] = any != 0;
    };

    // ---- wavefront 1: qpsk_pll, alpha = 0.2 ----
    int n = 0;
    float2 phi = make_float2(1.0f, 0.0f);
    float total_phase = 0.0f;
    auto pll = [&](int i, float2 sym) {
        const float2 v = cmul(sym, phi);
        po[i] = v;
        float2 xh;
        xh.x = v.x >= 0 ? kSqrt1_2 : -kSqrt1_2;
        // (:160-167: ++, +-, -- else -+: the imaginary sign follows v.y >= 0 in the right half plane, !(v.y < 0) in the left)
        xh.y = v.x >= 0 ? (v.y >= 0 ? kSqrt1_2 : -kSqrt1_2) : (v.y < 0 ? -kSqrt1_2 : kSqrt1_2);
        const float2 er = cmul(make_float2(xh.x, -xh.y), v);
        const float em = cabs_f(er);
        const bool go = !(em < 1e-10f);               // (:174: `continue` below that)
        const float2 unit = make_float2(er.x / em, er.y / em);
        const float ang = atan2f(unit.y, unit.x);
        const float sa = 0.2f * ang;
        // (glibc's own polynomial for this range -- libm_port.hpp, 14 double-precision operations, bit for bit the host's
        // cosf / sinf -- was tried in place of the device libm's single-precision routine: exact, and 0.03 ms slower)
        float sn, cs;
        sincosf(sa, &sn, &cs);          // one shared argument reduction for cosf(sa), sinf(sa) (:184)
        const float2 corr = make_float2(cs, sn);
        const float tp = total_phase + sa;
        const float2 p1 = cmul(make_float2(corr.x, -corr.y), phi);
        const float pm = cabs_f(p1);
        const float2 p2 = make_float2(p1.x / pm, p1.y / pm);
        const float2 p3 = pm > 0 ? p2 : p1;
        total_phase = go ? tp : total_phase;
        phi = go ? p3 : phi;
    };
    auto consume_block = [&](int buf) {
        const int cnt = lane < kSeqFrames ? s_cnt[buf][lane] : 0;
        // (the next symbol's LDS read is in flight while this one's chain runs)
        float2 nxt = s_sym[buf][0][lane & (kSeqFrames - 1)];
#pragma unroll 1
        for (int k = 0; k < kSeqBlock; k++) {
            if (__builtin_amdgcn_ballot_w64(k < cnt) == 0) break;
            const float2 cur = nxt;
            nxt = s_sym[buf][(k + 1) & (kSeqBlock - 1)][lane & (kSeqFrames - 1)];        // (stale where k + 1 >= cnt: not used)
            if (k < cnt) pll(n++, cur);
        }
    };

    if (role == 0 && use_gardner) {
        // the first kSeqLead samples of every frame
        constexpr int NL = kSeqLead / 8;
#pragma unroll 1
        for (int h = 0; h < NL; h += 4) {
            float4 lo[4], hi[4];
#pragma unroll
            for (int u = 0; u < 4; u++) request(4 * NL * upper + 4 * (h + u), lo[u], hi[u]);
#pragma unroll
            for (int u = 0; u < 4; u++) deposit(4 * NL * upper + 4 * (h + u), lo[u], hi[u]);
        }
        filled = kSeqLead;
    }
    // block j is produced in phase j and consumed in phase j + 1
    bool produced_all = false;
    for (int j = 0;; j++) {
        if (role == 0) {
            if (!produced_all) produce_block(j & 1);
        } else if (j > 0) {
            consume_block((j - 1) & 1);
        }
        __syncthreads();
        if (produced_all) break;
        produced_all = s_more[j & 1] == 0;
    }
    if (role == 1 && active) {
        out[b].n_symbols = n;               // (demod_par_kernel replaces it by the frame's symbol count)
        out[b].total_phase = total_phase;
    }
}

// Export (packed_records): with hp_packed != nullptr the wavefront writes its burst's DemodPacked record -- the six scalars
// and the hard bits 8 per byte, MSB first, exactly what demod_pack_kernel makes of the DemodOut -- and the burst's work record
// straight into pinned host memory (34 + 22 words, one store instruction each), and leaves the DemodOut's bits and LLRs
// unwritten: the chain ends with this kernel instead of demod_pack_kernel and a copy kernel behind it.
__device__ __forceinline__ void demod_export(const BurstWork *__restrict__ w, int lane, int ok, int direction, int confidence, int ns, float level,
                                             float total_phase, const int *s_sym, DemodPacked *__restrict__ hp_packed,
                                             BurstWork *__restrict__ hp_work, int b)
{
    uint32_t *dst = reinterpret_cast<uint32_t *>(hp_packed + b);
    static_assert(sizeof(DemodPacked) == 4 * 34 && kMaxBits / 8 == 4 * 28, "six scalars and 28 words of bits");
    uint32_t v = 0;
    if (lane == 0) v = (uint32_t)ok;
    else if (lane == 1) v = (uint32_t)direction;
    else if (lane == 2) v = (uint32_t)confidence;
    else if (lane == 3) v = (uint32_t)ns;
    else if (lane == 4) v = __float_as_uint(level);
    else if (lane == 5) v = __float_as_uint(total_phase);
    else if (lane < 34 && ok) {
        // word lane - 6 = bytes 4 (lane - 6) .. + 3 = symbols 16 (lane - 6) .. + 15, two bits each, MSB first within a byte
        const int dq[4] = { 0, 2, 3, 1 };
        const int s0 = 16 * (lane - 6);
#pragma unroll
        for (int byte = 0; byte < 4; byte++) {
            uint32_t bv = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int i = s0 + 4 * byte + k;
                if (i < ns) {
                    const int sq = s_sym[i] & 3, old = i > 0 ? (s_sym[i - 1] & 3) : 0;
                    bv |= (uint32_t)dq[(sq - old + 4) % 4] << (6 - 2 * k);
                }
            }
            v |= bv << (8 * byte);
        }
    }
    if (lane < 34) dst[lane] = v;
    static_assert(sizeof(BurstWork) == 4 * 22, "the work record as 22 words");
    if (lane < 22) reinterpret_cast<uint32_t *>(hp_work + b)[lane] = reinterpret_cast<const uint32_t *>(w)[lane];
    __threadfence_system();
}

__global__ __launch_bounds__(64) void demod_par_kernel(const BurstWork *__restrict__ work, int n_bursts,
                                                       const float2 *__restrict__ ws, DemodOut *__restrict__ out,
                                                       DemodPacked *__restrict__ hp_packed, BurstWork *__restrict__ hp_work)
{
    __shared__ float2 s_po[kMaxSymbols];
    __shared__ float s_mag[kMaxSymbols];      // sqrtf(re^2 + im^2) of the PLL output (:205, :231)
    __shared__ float s_abs[kMaxSymbols];      // cabsf of the PLL output (:493)
    __shared__ int s_sym[kMaxSymbols];        // quadrant | confidence flag << 2
    __shared__ int s_res[4];          // ok, direction, ns, confidence
    __shared__ float s_resf[3];       // level, total_phase, llr scale
    __builtin_amdgcn_s_setprio(2);
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    if (b >= n_bursts) return;
    DemodOut &o = out[b];
    const BurstWork w = work[b];
    if (w.drop_reason != 0) {
        if (lane == 0) o.ok = 0;
        if (hp_packed) demod_export(work + b, lane, 0, 0, 0, 0, 0.0f, 0.0f, s_sym, hp_packed, hp_work, b);
        return;
    }
    const int n = o.n_symbols;
    if (lane == 0) s_resf[1] = o.total_phase;
    const float2 *po = ws + (size_t)b * 2 * kMaxSymbols + kMaxSymbols;
    for (int i = lane; i < n; i += 64) s_po[i] = po[i];
    __syncthreads();

    // step 3a: per-symbol values of demod_qpsk (qpsk_demod.c:199-260), one symbol per lane
    for (int i = lane; i < n; i += 64) {
        const float re = s_po[i].x, im = s_po[i].y;
        const float a = re * re, bq = im * im;
        s_mag[i] = sqrtf(a + bq);
        s_abs[i] = cabs_f(s_po[i]);
        int sq;
        if (re >= 0 && im >= 0) sq = 0;
        else if (re < 0 && im >= 0) sq = 1;
        else if (re < 0) sq = 2;
        else sq = 3;
        const float phase = (atan2f(im, re) + IRDM_PI_F) * 180.0f / IRDM_PI_F;
        const float offs = 45.0f - fmodf(phase, 90.0f);
        s_sym[i] = sq | ((fabsf(offs) <= 22.0f) ? 4 : 0);
    }
    __syncthreads();

    if (lane == 0) {
        // step 3b: the end-of-frame rule fixes ns, then level and confidence are accumulated over [0, ns) in order
        float max_mag = 0.0f, sum = 0.0f;
        int low = 0, ns = 0, n_ok = 0;
        for (int i = 0; i < n; i++) {
            const float mag = s_mag[i];
            if (mag > max_mag) max_mag = mag;
            ns++;
            if (mag < max_mag / 8.0f) {
                if (++low >= 3) { ns -= 3; break; }
            } else {
                low = 0;
            }
        }
        for (int i = 0; i < ns; i++) {
            sum += s_mag[i];
            n_ok += s_sym[i] >> 2;
        }
        const float level = ns > 0 ? sum / (float)ns : 0.0f;
        const int confidence = ns > 0 ? (100 * n_ok) / ns : 0;

        // step 4: unique word (qpsk_demod.c:277-325, :429-465)
        const int UW_DL[12] = { 0, 2, 2, 2, 2, 0, 0, 0, 2, 0, 0, 2 };
        const int UW_UL[12] = { 2, 2, 0, 0, 0, 2, 0, 0, 2, 0, 2, 2 };
        int direction = w.direction;
        int ok = 1;
        int dl_ok = 0, ul_ok = 0;
        if (ns >= 12) {
            int dd = 0, du = 0;
            for (int i = 0; i < 12; i++) {
                const int sq = s_sym[i] & 3;
                int a = abs(sq - UW_DL[i]); if (a == 3) a = 1; dd += a;
                int c = abs(sq - UW_UL[i]); if (c == 3) c = 1; du += c;
            }
            dl_ok = dd <= 2;
            ul_ok = du <= 2;
        }
        if (!dl_ok && !ul_ok) {
            float de = 999.0f, ue = 999.0f;
            if (ns >= 12) {
                de = 0.0f; ue = 0.0f;
                for (int i = 0; i < 12; i++) {
                    float actual = atan2f(s_po[i].y, s_po[i].x);
                    if (actual < 0) actual += 2.0f * IRDM_PI_F;
                    {
                        const float expect = IRDM_PI_F * 0.25f + (float)UW_DL[i] * IRDM_PI_F * 0.5f;
                        float df = actual - expect;
                        if (df > IRDM_PI_F) df -= 2.0f * IRDM_PI_F;
                        if (df < -IRDM_PI_F) df += 2.0f * IRDM_PI_F;
                        de += fabsf(df) * (float)(2.0 / 3.14159265358979323846);
                    }
                    {
                        const float expect = IRDM_PI_F * 0.25f + (float)UW_UL[i] * IRDM_PI_F * 0.5f;
                        float df = actual - expect;
                        if (df > IRDM_PI_F) df -= 2.0f * IRDM_PI_F;
                        if (df < -IRDM_PI_F) df += 2.0f * IRDM_PI_F;
                        ue += fabsf(df) * (float)(2.0 / 3.14159265358979323846);
                    }
                }
            }
            const float mn = de < ue ? de : ue;
            if (mn > 3.0f) ok = 0;                                     // UW failed: no frame
            direction = ue < de ? 2 : 1;
        } else {
            if (ul_ok && !dl_ok) direction = 2;
            else if (dl_ok && !ul_ok) direction = 1;
        }
        // LLR normalisation (:489-497): sum of |.| in symbol order
        float sm = 0.0f;
        if (ok)
            for (int i = 0; i < ns; i++) sm += s_abs[i];
        s_res[0] = ok; s_res[1] = direction; s_res[2] = ns; s_res[3] = confidence;
        s_resf[0] = level;
        s_resf[2] = (ns > 0 && sm > 0) ? (kSqrt1_2 / (sm / (float)ns)) : 1.0f;
    }
    __syncthreads();
    const int ok = s_res[0], ns = s_res[2];
    if (lane == 0) {
        o.ok = ok;
        o.direction = s_res[1];
        o.confidence = s_res[3];
        o.n_symbols = ns;
        o.level = s_resf[0];
        o.total_phase = s_resf[1];
    }
    if (hp_packed) {
        demod_export(work + b, lane, ok, s_res[1], s_res[3], ns, s_resf[0], s_resf[1], s_sym, hp_packed, hp_work, b);
        return;
    }
    if (!ok) return;
    // steps 5-7: decode_dqpsk (:264-273), bits MSB first (:329-335), LLR (:498-503): per-symbol independent
    const float scale = s_resf[2];
    for (int i = lane; i < ns; i += 64) {
        const int dq[4] = { 0, 2, 3, 1 };
        const int sq = s_sym[i] & 3, old = i > 0 ? (s_sym[i - 1] & 3) : 0;
        const int v = dq[(sq - old + 4) % 4];
        o.bits[2 * i] = (uint8_t)((v >> 1) & 1);
        o.bits[2 * i + 1] = (uint8_t)(v & 1);
        o.llr[2 * i] = fabsf(s_po[i].x) * scale;
        o.llr[2 * i + 1] = fabsf(s_po[i].y) * scale;
    }
}

int launch_demod(const BurstWork *work, int n_bursts, const float2 *frames, int use_gardner,
                 float sps, float2 *ws, DemodOut *out, hipStream_t stream, DemodPacked *hp_packed, BurstWork *hp_work)
{
    if (n_bursts <= 0) return 0;
    hipLaunchKernelGGL(demod_seq_kernel, dim3((n_bursts + kSeqFrames - 1) / kSeqFrames), dim3(128),
                       sizeof(float2) * kSeqFrames * kSeqPitch, stream, work, n_bursts,
                       frames, use_gardner, sps, ws, out);
    hipLaunchKernelGGL(demod_par_kernel, dim3(n_bursts), dim3(64), 0, stream, work, n_bursts, ws, out, hp_packed, hp_work);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace irdm
