// mfma_fir.hip -- can the 801-tap decimating FIR in the AVX2 kernel's order (option fir_order 1: avx2_fir_ccf_dec,
// simd_avx2.c:62-108) leave the VALU?  (round 4's verdict, item 3; DESIGN.md section 5 "The decimator".)
//
// The product's fir_decimate_kernel_f issues one v_pk_fma_f32 per tap and output pair and is bound by that instruction's
// issue rate (8.4 cycles with two 256-register wavefronts per SIMD: 42 TFLOP/s useful at 53 % of the HBM roofline).
// MI355X_MICROARCH.md says v_mfma_f32_16x16x4_f32 runs at 155 TFLOP/s and is "exact f32 (== an fmaf chain, bitwise)" --
// which is the arithmetic fir_order 1 specifies: out = ((a0 + a2) + (a1 + a3)) + t[800] * y[800] with
// a_j = fma(t[4m + j], y[qM + 4m + j], a_j), m ascending, a_j(0) = +0.
//
// The product as a matrix product, for ONE polyphase j:  D[i][n] += sum_k A[i][k] * B[k][n]
//   rows i   16 consecutive outputs q0 + i of a strip,
//   cols n   16 real streams = (re, im) of 8 strips that are filtered side by side,
//   k        four consecutive positions u = 4 s + kk of the polyphase sequence y_j[u] = y[q0 M + 4 u + j],
//   A[i][k] = t[4 (u - (M/4) i) + j], zero outside the 200 taps (the Toeplitz band: fma(0, y, acc) == acc),
//   B[k][n] = y_j[u] of stream n,
// s ascending over the 88 steps a row group's 350 positions take: every accumulator sees its taps in ascending order.
// 57 % of the multiply-adds the instruction performs are the filter's (16 rows share a 350-position window, 200 of
// which are a row's own).  A wavefront owns one j (the four accumulators of the AVX2 kernel = the four wavefronts of a
// workgroup); its 88 A operands are the same for every row group and stay in registers; B comes from LDS, where the
// window of (already rotated) samples lies de-interleaved by polyphase and stream, one ds_read_b32 per MFMA.
//
// What this measures:
//   exact   the whole path (window from global memory per row group, combine, tail tap) against the same loop on the
//           host (the oracle's orc_fir_ccf_dec_avx2 restated; this file is test tooling, it links nothing), bit for bit
//           on every output, and a probe of the instruction itself: k order, zero taps, signed zeros
//   core    the MFMA chain with its LDS reads alone (the window loaded once, row groups repeated): the rate the matrix
//           pipe sustains with this operand traffic -- TFLOP/s executed and useful
//   full    window loads + chain + combine, one workgroup per CU at a time (no loader / consumer split yet)
// RG = 1: four wavefronts per workgroup, a dependent MFMA every 40 cycles per SIMD (the guide: 32 issue, 40 dependent);
// RG = 2: eight wavefronts, two row groups per window (132 KB of LDS), two chains per SIMD.
//
// Build:  hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 tools/ubench/mfma_fir.hip -o tools/ubench/mfma_fir
// Usage:  mfma_fir [decim=40] [strips=2048] [outputs_per_strip=512] [reps=5]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)

constexpr int kTaps = 801;
constexpr int kRows = 16;             // outputs per row group (the MFMA's M)
constexpr int kStreams = 16;          // real streams per workgroup (the MFMA's N): 8 complex strips
constexpr int kStrips = kStreams / 2;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// positions of one polyphase a window of RG row groups covers, rounded up to whole MFMA steps
template <int M, int RG>
struct Geo {
    static constexpr int q = M / 4;                                  // polyphase positions per output
    static constexpr int span = (RG * kRows - 1) * q + 200;          // positions a window needs
    static constexpr int steps1 = ((kRows - 1) * q + 200 + 3) / 4;   // MFMA steps of one row group
    static constexpr int pos_chain = (RG - 1) * kRows * q + 4 * steps1;     // the last steps read a little past `span`
    static constexpr int pos_tail = (RG * kRows - 1) * q + 201;             // the tail tap's sample (tap 800) of the last row
    static constexpr int pos = ((pos_chain > pos_tail ? pos_chain : pos_tail) + 3) / 4 * 4;   // positions held
    static constexpr int pitch = ((pos + 63 - 4) / 64) * 64 + 4;     // words per stream: == 4 (mod 64), >= pos: lane (kk, n) -> bank 4 n + kk
    static constexpr size_t lds_words = (size_t)4 * kStreams * pitch + (size_t)RG * 4 * kRows * kStreams;
};

// MODE 0: full (window from global memory per row group, results stored), 1: core (window loaded once, `repeat` row groups
// on it, one result stored at the end)
template <int M, int RG, int MODE>
__global__ __launch_bounds__(256 * RG) void mfma_fir_kernel(const float2 *__restrict__ y, size_t strip_len, int n_out,
                                                            const float *__restrict__ taps, float2 *__restrict__ out, int repeat)
{
    using G = Geo<M, RG>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *S = lds;                                      // [4 j][16 n][pitch]
    float *Cx = lds + (size_t)4 * kStreams * G::pitch;   // [RG][4 j][16 rows][16 cols]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = wave & 3, rg = wave >> 2;
    const int i_row = lane & 15, kk = lane >> 4;
    const int strip0 = blockIdx.x * kStrips;

    // this wavefront's 88 A operands: lane (kk, i) of step s holds t[4 (4 s + kk - q i) + j], zero outside the taps.  (The
    // vector body of the AVX2 kernel covers taps 0 .. 799; tap 800 is its scalar tail.)
    float a_reg[G::steps1];
#pragma unroll
    for (int s = 0; s < G::steps1; s++) {
        const int u = 4 * s + kk - G::q * i_row;
        a_reg[s] = (u >= 0 && u < 200) ? taps[4 * u + j] : 0.0f;
    }
    const float t_last = taps[kTaps - 1];

    const int n_groups = n_out / (kRows * RG);
    f32x4 acc = { 0.0f, 0.0f, 0.0f, 0.0f };
    for (int g = 0; g < (MODE == 1 ? 1 : n_groups); g++) {
        const int q0 = g * kRows * RG;
        // ---- the window: samples [q0 M, q0 M + 4 pos) of the 8 strips, de-interleaved by polyphase and stream ----
        // (a thread takes 16 bytes = two complex samples; zeros past the strip's end)
        const size_t x0 = (size_t)q0 * M;
        for (int e = tid; e < kStrips * (4 * G::pos / 2); e += 256 * RG) {
            const int st = e / (4 * G::pos / 2), xp = (e % (4 * G::pos / 2)) * 2;
            float4 v = { 0.0f, 0.0f, 0.0f, 0.0f };
            if (x0 + xp + 1 < strip_len) v = *reinterpret_cast<const float4 *>(y + (size_t)(strip0 + st) * strip_len + x0 + xp);
            // sample x0 + xp: polyphase xp & 3, position xp >> 2; the next one likewise
            float *p0 = S + ((size_t)((xp & 3) * kStreams + 2 * st)) * G::pitch + (xp >> 2);
            p0[0] = v.x;
            p0[G::pitch] = v.y;
            float *p1 = S + ((size_t)(((xp + 1) & 3) * kStreams + 2 * st)) * G::pitch + ((xp + 1) >> 2);
            p1[0] = v.z;
            p1[G::pitch] = v.w;
        }
        __syncthreads();
        // ---- the chain: 88 dependent MFMAs, B from LDS ----
        const float *bp = S + ((size_t)(j * kStreams + (lane & 15))) * G::pitch + rg * kRows * G::q + kk;
        for (int r = 0; r < (MODE == 1 ? repeat : 1); r++) {
            acc = f32x4{ 0.0f, 0.0f, 0.0f, 0.0f };
            if (MODE == 1) asm volatile("" : "+v"(acc));       // (every repetition is computed: the chain starts from a value the compiler cannot see through)
#pragma unroll
            for (int s = 0; s < G::steps1; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_reg[s], bp[4 * s], acc, 0, 0, 0);
            if (MODE == 1) asm volatile("" : "+v"(acc));
        }
        // ---- combine the four accumulators (one per wavefront), add the tail tap, store ----
        // D: lane l, register r holds row 4 (l >> 4) + r, column l & 15
#pragma unroll
        for (int r = 0; r < 4; r++) Cx[(((size_t)rg * 4 + j) * kRows + 4 * kk + r) * kStreams + (lane & 15)] = acc[r];
        __syncthreads();
        for (int e = tid; e < RG * kRows * kStreams; e += 256 * RG) {
            const int g2 = e / (kRows * kStreams), row = (e / kStreams) % kRows, col = e % kStreams;
            const float *c = Cx + ((size_t)g2 * 4 * kRows + row) * kStreams + col;
            const float a0 = c[0], a1 = c[kRows * kStreams], a2 = c[2 * kRows * kStreams], a3 = c[3 * kRows * kStreams];
            // the tail: sample (q0 + 16 g2 + row) M + 800 -> polyphase 0, position (16 g2 + row) q + 200
            const float yt = S[(size_t)col * G::pitch + (g2 * kRows + row) * G::q + 200];
            const float v = ((a0 + a2) + (a1 + a3)) + t_last * yt;
            float *o = reinterpret_cast<float *>(out + (size_t)(strip0 + col / 2) * n_out + q0 + g2 * kRows + row);
            o[col & 1] = v;
        }
        __syncthreads();
    }
}

// the instruction alone: D = A B + C for one 16x16x4 product with chosen operands (k order, zero taps, signed zeros)
__global__ void mfma_probe_kernel(const float *a, const float *b, const float *c, float *d)
{
    const int lane = threadIdx.x;
    f32x4 acc = { c[4 * lane + 0], c[4 * lane + 1], c[4 * lane + 2], c[4 * lane + 3] };
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[lane], b[lane], acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) d[4 * lane + r] = acc[r];
}

// avx2_fir_ccf_dec (simd_avx2.c:62-108) as the oracle restates it (oracle/irdm_oracle.c orc_fir_ccf_dec_avx2); this
// translation unit is compiled -ffp-contract=off: fmaf is the fused operation, nothing else is
static void fir_ref(const float *taps, int ntaps, const float *in, float *out, int n_out, int decimation)
{
    for (int i = 0; i < n_out; i++) {
        const float *p = in + 2 * (size_t)i * (size_t)decimation;
        float re[4] = { 0, 0, 0, 0 }, im[4] = { 0, 0, 0, 0 };
        int k = 0;
        for (; k + 3 < ntaps; k += 4)
            for (int j = 0; j < 4; j++) {
                re[j] = fmaf(taps[k + j], p[2 * (k + j)], re[j]);
                im[j] = fmaf(taps[k + j], p[2 * (k + j) + 1], im[j]);
            }
        float ar = (re[0] + re[2]) + (re[1] + re[3]);
        float ai = (im[0] + im[2]) + (im[1] + im[3]);
        for (; k < ntaps; k++) {
            ar += taps[k] * p[2 * k];
            ai += taps[k] * p[2 * k + 1];
        }
        out[2 * (size_t)i] = ar;
        out[2 * (size_t)i + 1] = ai;
    }
}

static uint32_t lcg(uint32_t &s) { return s = s * 1664525u + 1013904223u; }
static float urand(uint32_t &s) { return (float)((lcg(s) >> 8) & 0xffff) / 32768.0f - 1.0f; }

template <int M, int RG>
static int run(int strips, int n_out, int reps)
{
    using G = Geo<M, RG>;
    const size_t strip_len = ((size_t)(n_out - 1) * M + kTaps + 7) & ~(size_t)7;
    printf("decim %d, RG %d (%d wavefronts per workgroup): %d steps per row group, window %d positions x 4 polyphases x 16 streams, LDS %zu bytes\n",
           M, RG, 4 * RG, G::steps1, G::pos, G::lds_words * 4);
    std::vector<float> taps(kTaps);
    uint32_t seed = 12345;
    for (int k = 0; k < kTaps; k++) taps[k] = 0.02f * urand(seed) * (1.0f + 0.5f * sinf(0.01f * k));
    std::vector<float> y((size_t)strips * strip_len * 2);
    for (auto &v : y) v = urand(seed) * 0.05f;
    // (denormal-free but wide dynamic range in a few places: large and tiny samples next to each other)
    for (size_t i = 0; i < y.size(); i += 997) y[i] *= 1e4f;
    for (size_t i = 3; i < y.size(); i += 1499) y[i] *= 1e-6f;
    float *d_taps;
    float2 *d_y, *d_out;
    CK(hipMalloc(&d_taps, sizeof(float) * kTaps));
    CK(hipMalloc(&d_y, sizeof(float) * y.size()));
    CK(hipMalloc(&d_out, sizeof(float2) * (size_t)strips * n_out));
    CK(hipMemcpy(d_taps, taps.data(), sizeof(float) * kTaps, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_y, y.data(), sizeof(float) * y.size(), hipMemcpyHostToDevice));
    CK(hipMemset(d_out, 0xff, sizeof(float2) * (size_t)strips * n_out));
    const size_t lds = G::lds_words * 4;
    CK(hipFuncSetAttribute((const void *)mfma_fir_kernel<M, RG, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)mfma_fir_kernel<M, RG, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 grid(strips / kStrips), block(256 * RG);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // ---- exactness ----
    hipLaunchKernelGGL((mfma_fir_kernel<M, RG, 0>), grid, block, lds, 0, d_y, strip_len, n_out, d_taps, d_out, 1);
    CK(hipDeviceSynchronize());
    std::vector<float> got((size_t)strips * n_out * 2), want((size_t)n_out * 2);
    CK(hipMemcpy(got.data(), d_out, sizeof(float) * got.size(), hipMemcpyDeviceToHost));
    size_t bad = 0, checked = 0;
    double worst = 0;
    const int check_strips = strips < 256 ? strips : 256;           // (256 strips x 512 outputs x 2 = 262144 values on the host)
    for (int st = 0; st < check_strips; st++) {
        const int sidx = (int)((size_t)st * (strips - 1) / (check_strips > 1 ? check_strips - 1 : 1));
        fir_ref(taps.data(), kTaps, y.data() + (size_t)sidx * strip_len * 2, want.data(), n_out, M);
        for (int i = 0; i < 2 * n_out; i++) {
            const float a = got[(size_t)sidx * n_out * 2 + i], b = want[i];
            checked++;
            if (memcmp(&a, &b, 4) != 0) {
                if (bad < 5) printf("  MISMATCH strip %d output %d %s: mfma %.9g (0x%08x)  fmaf chain %.9g (0x%08x)\n", sidx, i / 2, i & 1 ? "im" : "re",
                                    a, *(const uint32_t *)&a, b, *(const uint32_t *)&b);
                bad++;
                worst = fmax(worst, fabs((double)a - b) / fmax(fabs((double)b), 1e-30));
            }
        }
    }
    printf("  exact: %zu of %zu values differ from avx2_fir_ccf_dec's arithmetic%s", bad, checked, bad ? "" : " -- bit for bit\n");
    if (bad) printf(" (worst relative difference %.3g)\n", worst);
    // ---- rates ----
    const double flop_exec_group = (double)G::steps1 * 2.0 * 16 * 16 * 4 * 4;        // four wavefronts (j) per row group
    const double flop_useful_group = 2.0 * 16 * 16 * 800;
    {
        const int repeat = 64;
        hipLaunchKernelGGL((mfma_fir_kernel<M, RG, 1>), grid, block, lds, 0, d_y, strip_len, n_out, d_taps, d_out, repeat);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; r++)
            hipLaunchKernelGGL((mfma_fir_kernel<M, RG, 1>), grid, block, lds, 0, d_y, strip_len, n_out, d_taps, d_out, repeat);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        const double groups = (double)grid.x * RG * repeat;
        printf("  core : %.3f ms for %.0f row groups: %.1f TFLOP/s executed, %.1f TFLOP/s useful (the chain + its LDS reads; window load and combine once per launch)\n",
               ms, groups, groups * flop_exec_group / ms * 1e-9, groups * flop_useful_group / ms * 1e-9);
    }
    {
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; r++)
            hipLaunchKernelGGL((mfma_fir_kernel<M, RG, 0>), grid, block, lds, 0, d_y, strip_len, n_out, d_taps, d_out, 1);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        const double groups = (double)grid.x * (n_out / (kRows * RG)) * RG;
        const double bytes = (double)strips * strip_len * 8 + (double)strips * n_out * 8;
        printf("  full : %.3f ms for %d strips x %d outputs: %.1f TFLOP/s useful, %.2f TB/s of algorithmic bytes (%.3f GB: every sample once + the outputs); window re-read factor %.2f\n",
               ms, strips, n_out, groups * flop_useful_group / ms * 1e-9, bytes / ms * 1e-9, bytes * 1e-9,
               (double)(4 * G::pos) / (RG * kRows * M));
    }
    CK(hipFree(d_taps));
    CK(hipFree(d_y));
    CK(hipFree(d_out));
    return bad ? 1 : 0;
}

static int probe()
{
    // A[i][k] = lane (k * 16 + i), B[k][n] = lane (k * 16 + n); C/D: lane l register r = row 4 (l >> 4) + r, column l & 15
    std::vector<float> a(64), b(64), c(256), d(256);
    uint32_t seed = 7;
    int bad = 0;
    for (int trial = 0; trial < 2000; trial++) {
        for (int l = 0; l < 64; l++) {
            a[l] = urand(seed) * (trial % 3 == 0 ? 1e3f : 1.0f);
            b[l] = urand(seed);
            if (trial % 5 == 1 && (l >> 4) != 2) a[l] = 0.0f;              // the Toeplitz band's zeros
            if (trial % 7 == 2 && (l & 1)) a[l] = -0.0f;
        }
        for (int e = 0; e < 256; e++) c[e] = trial % 4 == 3 ? 0.0f : urand(seed) * 10.0f;
        float *da, *db, *dc, *dd;
        CK(hipMalloc(&da, 256));
        CK(hipMalloc(&db, 256));
        CK(hipMalloc(&dc, 1024));
        CK(hipMalloc(&dd, 1024));
        CK(hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice));
        CK(hipMemcpy(dc, c.data(), 1024, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        CK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
        for (int l = 0; l < 64; l++)
            for (int r = 0; r < 4; r++) {
                const int row = 4 * (l >> 4) + r, col = l & 15;
                float acc = c[4 * l + r];
                for (int k = 0; k < 4; k++) acc = fmaf(a[k * 16 + row], b[k * 16 + col], acc);
                if (memcmp(&acc, &d[4 * l + r], 4) != 0) {
                    if (bad < 5) printf("  probe trial %d row %d col %d: mfma %.9g fmaf chain (k ascending) %.9g\n", trial, row, col, d[4 * l + r], acc);
                    bad++;
                }
            }
        CK(hipFree(da));
        CK(hipFree(db));
        CK(hipFree(dc));
        CK(hipFree(dd));
    }
    printf("v_mfma_f32_16x16x4_f32 vs fma(a[k], b[k], acc) for k = 0..3 in order: %d of %d values differ%s\n", bad, 2000 * 256,
           bad ? "" : " -- bit for bit (zeros in A, -0, large and small operands)");
    return bad;
}

int main(int argc, char **argv)
{
    const int decim = argc > 1 ? atoi(argv[1]) : 40;
    int strips = argc > 2 ? atoi(argv[2]) : 2048;
    int n_out = argc > 3 ? atoi(argv[3]) : 512;
    const int reps = argc > 4 ? atoi(argv[4]) : 5;
    strips = strips / kStrips * kStrips;
    n_out = n_out / 32 * 32;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("%s, %d CUs, %d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
    int rc = probe();
    if (decim == 48) {
        rc |= run<48, 1>(strips, n_out, reps);
        rc |= run<48, 2>(strips, n_out, reps);
    } else {
        rc |= run<40, 1>(strips, n_out, reps);
        rc |= run<40, 2>(strips, n_out, reps);
    }
    return rc ? 1 : 0;
}
