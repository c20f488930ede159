// sum_bench.hip -- what a step of the band scan's sums pass costs a lone wavefront ("Where the
// scan's time goes now"; profiles/r4_sched_experiments.txt).  One wavefront per 64 bins walks `steps` dependent updates
// s = (s - old[k]) + new[k]; nothing else runs on its SIMD, so every instruction -- vector, scalar, branch, wait -- takes an
// issue slot of its own.  Variants:
//   0  the chain alone: operands from registers (32 steps per block, reloaded from LDS-free constants): issue cost only
//   1  rows from memory: two buffer loads per step, row offsets by v_readlane from a descriptor register (64 steps per
//      vector load), two batches of 32 steps in flight -- the product's loop without snapshots
//   2  variant 1 + a snapshot store behind every step (offset by v_readlane, five wait states, store)
//   3  variant 1 with rows 32 KB apart replaced by ONE row read again and again (L2 hits): memory latency taken out
//   4  variant 2 with the store in two of the 64 lanes only (a snapshot for the bins a frame's list names)
//   5  variant 2 with a store every eighth step
// Prints ns per step and the implied cycles at the clock the device reports.
// Build:  hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench/sum_bench.hip -o tools/ubench/sum_bench
// Usage:  sum_bench [steps=5000] [n_bins=8192] [reps=10]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)

constexpr int kDepth = 32;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)(bytes > 0x7fffffffu ? 0x7fffffffu : bytes), 0x00020000);
}

struct Step {
    unsigned nw_off, ol_off, snap_off, pad;
};

template <int VARIANT>
__global__ __launch_bounds__(64) void sum_kernel(const float *__restrict__ mag, size_t mag_bytes, const Step *__restrict__ steps, int n,
                                                 float *__restrict__ snap, size_t snap_bytes, float *__restrict__ out)
{
    const int lane = threadIdx.x, b = blockIdx.x * 64 + lane, boff = b * 4;
    float s = 1.0f + 1e-3f * lane, smin = s;
    if (VARIANT == 0) {
        float a[kDepth], o[kDepth];
#pragma unroll
        for (int j = 0; j < kDepth; j++) {
            a[j] = 1e-3f * (j + lane);
            o[j] = 1e-3f * (j + 1);
        }
        for (int k0 = 0; k0 < n; k0 += kDepth) {
#pragma unroll
            for (int j = 0; j < kDepth; j++) {
                const float d = s - o[j];
                s = d + a[j];
                smin = __builtin_fminf(smin, s);
            }
            // (keep the compiler from hoisting the chain: the operands change a little every block)
#pragma unroll
            for (int j = 0; j < kDepth; j += 8) a[j] += 1e-7f;
        }
    } else {
        const __amdgpu_buffer_rsrc_t r_mag = rsrc(mag, mag_bytes), r_snap = rsrc(snap, snap_bytes);
        const uint4 *steps4 = reinterpret_cast<const uint4 *>(steps);
        float nwA[kDepth], olA[kDepth], nwB[kDepth], olB[kDepth];
#define LOAD(NWv, OLv, D, half)                                                                                      \
    _Pragma("unroll") for (int j = 0; j < kDepth; j++) {                                                             \
        const int o_nw = __builtin_amdgcn_readlane((int)D.x, (half) * kDepth + j);                                   \
        const int o_ol = __builtin_amdgcn_readlane((int)D.y, (half) * kDepth + j);                                   \
        NWv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_mag, boff, o_nw, 0));              \
        OLv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_mag, boff, o_ol, 0));              \
    }
#define CONSUME(NWv, OLv, D, half)                                                                                   \
    _Pragma("unroll") for (int j = 0; j < kDepth; j++) {                                                             \
        const float dd = s - OLv[j];                                                                                 \
        s = dd + NWv[j];                                                                                             \
        smin = __builtin_fminf(smin, s);                                                                             \
        if (VARIANT == 2 || (VARIANT == 5 && j % 8 == 0) || (VARIANT == 4 && ((lane + j) & 63) < 2))                 \
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, s), r_snap, boff,                          \
                                                  __builtin_amdgcn_readlane((int)D.z, (half) * kDepth + j), 0);      \
    }
        uint4 d = steps4[lane], dn = steps4[2 * kDepth + lane];
        LOAD(nwA, olA, d, 0)
        for (int k0 = 0; k0 < n; k0 += 2 * kDepth) {
            const uint4 dnn = steps4[k0 + 4 * kDepth + lane];
            LOAD(nwB, olB, d, 1)
            CONSUME(nwA, olA, d, 0)
            if (k0 + 2 * kDepth < n) { LOAD(nwA, olA, dn, 0) }
            CONSUME(nwB, olB, d, 1)
            d = dn;
            dn = dnn;
        }
#undef LOAD
#undef CONSUME
    }
    out[b] = s + smin;
}

int main(int argc, char **argv)
{
    const int steps = argc > 1 ? atoi(argv[1]) : 5000;
    const int n_bins = argc > 2 ? atoi(argv[2]) : 8192;
    const int reps = argc > 3 ? atoi(argv[3]) : 10;
    const int n = (steps + 63) / 64 * 64, rows = n + 512;
    const size_t row = (size_t)n_bins * 4;
    float *mag, *snap, *out;
    Step *d_steps;
    CK(hipMalloc(&mag, (size_t)rows * row));
    CK(hipMalloc(&snap, (size_t)n * row));
    CK(hipMalloc(&out, row));
    CK(hipMalloc(&d_steps, sizeof(Step) * (n + 512)));
    CK(hipMemset(mag, 0, (size_t)rows * row));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const double mhz = prop.clockRate / 1e3;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("# %d steps, %d bins (%d wavefronts), %d reps; device clock %.0f MHz\n", n, n_bins, n_bins / 64, reps, mhz);
    for (int variant = 0; variant < 6; variant++) {
        std::vector<Step> h(n + 512);
        for (int k = 0; k < n + 512; k++) {
            const bool one_row = variant == 3;
            h[k].nw_off = (unsigned)((one_row ? 512 : (size_t)(k + 512)) * row);      // the row that enters
            h[k].ol_off = (unsigned)((one_row ? 0 : (size_t)k) * row);                // the row that entered 512 steps earlier
            h[k].snap_off = (unsigned)((size_t)(k < n ? k : 0) * row);
            h[k].pad = 0;
        }
        CK(hipMemcpy(d_steps, h.data(), sizeof(Step) * h.size(), hipMemcpyHostToDevice));
        auto go = [&]() {
            const dim3 grid(n_bins / 64), block(64);
            switch (variant) {
            case 0: hipLaunchKernelGGL(sum_kernel<0>, grid, block, 0, st, mag, (size_t)rows * row, d_steps, n, snap, (size_t)n * row, out); break;
            case 1: hipLaunchKernelGGL(sum_kernel<1>, grid, block, 0, st, mag, (size_t)rows * row, d_steps, n, snap, (size_t)n * row, out); break;
            case 2: hipLaunchKernelGGL(sum_kernel<2>, grid, block, 0, st, mag, (size_t)rows * row, d_steps, n, snap, (size_t)n * row, out); break;
            case 4: hipLaunchKernelGGL(sum_kernel<4>, grid, block, 0, st, mag, (size_t)rows * row, d_steps, n, snap, (size_t)n * row, out); break;
            case 5: hipLaunchKernelGGL(sum_kernel<5>, grid, block, 0, st, mag, (size_t)rows * row, d_steps, n, snap, (size_t)n * row, out); break;
            default: hipLaunchKernelGGL(sum_kernel<1>, grid, block, 0, st, mag, (size_t)rows * row, d_steps, n, snap, (size_t)n * row, out); break;
            }
        };
        go();
        CK(hipStreamSynchronize(st));
        float best = 1e30f, total = 0;
        for (int i = 0; i < reps; i++) {
            CK(hipEventRecord(e0, st));
            go();
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
            total += ms;
        }
        static const char *names[] = { "chain alone (registers)", "rows from memory, readlane offsets", "... + a snapshot store per step",
                                       "rows from one cached row (no HBM latency)", "... + a store in 2 of 64 lanes per step",
                                       "... + a snapshot store every 8th step" };
        const double ns = best * 1e6 / n;
        printf("variant %d  %-44s best %.1f us  mean %.1f us  %.1f ns/step = %.0f cycles\n", variant, names[variant], best * 1e3,
               total / reps * 1e3, ns, ns * mhz / 1e3);
    }
    return 0;
}
