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
// demod_seq_kernel -- the true recurrences, ONE LANE PER FRAME (64 frames per wavefront):
//   * Gardner loop (qpsk_demod.c:85-130): position n+1 depends on the timing error of position n;
//   * PLL (:145-195): phi_{i+1} depends on phi_i through cabsf / atan2f / cosf / sinf.
//   Both are chains of a few hundred dependent instructions per symbol (a dependent VALU instruction issues ~20 cycles
//   after its producer, tools/ubench/valu_chain.hip), <= 448 symbols: ~0.5 ms however they are laid out.  One
//   wavefront per frame (the earlier layout: lane 0 ran the chains out of 48 KB of LDS) kept 667 wavefronts and 125 KB
//   of every CU's LDS busy for that long -- the decimator (50 KB per workgroup) and K1 (66 KB) of the other chunks in
//   flight could not be placed until they retired.  A lane per frame needs 11 wavefronts and no LDS; the samples come
//   from HBM/L2 (4 neighbours per interpolation, 32 contiguous bytes per lane).
// demod_par_kernel -- everything that is per-symbol independent, one wavefront per frame, short:
//   slicer, per-symbol magnitudes, confidence flags, unique-word angles, |.| for the LLR scale: one symbol per lane;
//   only the float sums whose order matters (level :247, LLR scale :489-497) and the end-of-frame rule (:210-225, a
//   running maximum) are walked in order by lane 0, over values already computed.
__global__ __launch_bounds__(64) void demod_seq_kernel(const BurstWork *__restrict__ work, int n_bursts,
                                                       const float2 *__restrict__ frames, int use_gardner, float sps,
                                                       float2 *__restrict__ ws, DemodOut *__restrict__ out)
{
    // a handful of wavefronts whose dependent chains decide when the chunk's records are complete, sharing SIMDs with
    // the decimator's wavefronts of the next chunk: they go first when they have an instruction ready
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= n_bursts) return;
    if (work[b].drop_reason != 0) return;
    const int n_samples = work[b].num_samples;
    const float2 *fr = frames + (size_t)b * kMaxFrameSamples;
    float2 *dec = ws + (size_t)b * 2 * kMaxSymbols;
    float2 *po = dec + kMaxSymbols;

    // Steps 1 and 2 in ONE loop: decimate_gardner (qpsk_demod.c:85-130) / decimate_simple (:134-141) produces the
    // symbols, qpsk_pll (:145-195, alpha = 0.2) consumes them.  The two recurrences -- the timing loop's position and the
    // PLL's phase -- do not feed each other (the reference runs them one after the other over the whole frame), and a
    // wavefront issues in order: what the instruction stream holds between a load and its first use is all that hides
    // the load.  So the loop is skewed by one symbol: an iteration produces symbol i + 1 (four-sample loads at pos and
    // pos - sps/2 out of L2, the two interpolations, the timing error: ~0.6 k cycles of which the loads are most) and
    // runs the PLL on symbol i (cabsf, two divisions, atan2f, sincosf, cabsf, two divisions: ~130 dependent
    // instructions, ~2.2 k cycles), and both are written without branches (the reference's `if`s as selects of values
    // computed either way) so that the two chains sit in ONE basic block and the scheduler interleaves them: an
    // iteration costs the PLL chain, not the sum.  Same operations on the same operands in the same order per chain.
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
    if (use_gardner) {
        float pos = 0.0f, toff = 0.0f;
        float2 prev = make_float2(0.0f, 0.0f);
        int made = 0;
        // symbol `made` at the loop's position, then the timing update (the position of the next one).  `live` false (the
        // frame has ended: its last iteration only runs the PLL): the same instructions on clamped positions, nothing
        // kept -- the store rewrites the last symbol with itself -- so that the loop body stays one basic block.
        auto produce = [&](bool live, float2 last) -> float2 {
            const float mid_pos = pos - sps * 0.5f;
            // the on-time sample and the mid-point sample (used only where the reference computes it) are independent
            const float2 on = cubic_interp(fr, n_samples, pos);
            const float2 mid = cubic_interp(fr, n_samples, mid_pos >= 1.0f ? mid_pos : 1.0f);
            dec[live ? made : made - 1] = live ? on : last;
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
            return on;
        };
        const float lim = (float)(n_samples - 3);
        bool have = pos < lim && made < kMaxSymbols;
        float2 cur = make_float2(0.0f, 0.0f);
        if (have) cur = produce(true, cur);
        while (have) {
            const bool more = pos < lim && made < kMaxSymbols;
            const float2 nxt = produce(more, cur);
            pll(n, cur);
            n++;
            cur = nxt;
            have = more;
        }
    } else {
        const int step = (int)sps;
        n = (n_samples + step - 1) / step;
        if (n > kMaxSymbols) n = kMaxSymbols;
        float2 sym = n > 0 ? fr[0] : make_float2(0.0f, 0.0f);
        for (int i = 0; i < n; i++) {
            const float2 nxt = i + 1 < n ? fr[(size_t)(i + 1) * step] : sym;
            dec[i] = sym;
            pll(i, sym);
            sym = nxt;
        }
    }
    out[b].n_symbols = n;               // (demod_par_kernel replaces it by the frame's symbol count)
    out[b].total_phase = total_phase;
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
    hipLaunchKernelGGL(demod_seq_kernel, dim3((n_bursts + 63) / 64), dim3(64), 0, stream, work, n_bursts,
                       frames, use_gardner, sps, ws, out);
    hipLaunchKernelGGL(demod_par_kernel, dim3(n_bursts), dim3(64), 0, stream, work, n_bursts, ws, out, hp_packed, hp_work);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace irdm
