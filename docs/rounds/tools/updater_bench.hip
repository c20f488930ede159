// tools/updater_bench.hip -- measurement aid (not part of the product): throughput of the detector scan's UPDATER
// workgroups (scan_fast.hip, multi-CU form) in isolation.  The host pre-publishes a synthetic operation list (runs of R
// quiet frames), so the updaters never wait for a leader: what is measured is their own pace -- microseconds per quiet
// frame -- for the variants a round-2 change would pick from:
//   V0  one operation at a time, groups of 8 / 2 / 1 frames, sums + barrier + counter after every operation (today)
//   V1  V0 with 16-frame groups
//   V2  V0, but sums / counter only when no further operation is already published (look-ahead batching)
//   V3  V2 with 16-frame groups
// Context (bench.py --density 2, 134 bursts / 7600 quiet frames per chunk): the scan takes 5.35 ms = 0.7 us per quiet frame
// with 7 updaters, the stand-alone sweep of tools/quiet_sweep.hip reached 0.17-0.64 us.  Every spin is bounded.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/updater_bench tools/updater_bench.hip ; run: build/updater_bench [W] [R]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int kN = 8192, kHist = 512, kThreads = 512;
constexpr unsigned long long kValid = 1ull << 63;
constexpr int kExit = 8;

__host__ __device__ inline unsigned long long pack(int f0, int run, int hidx, int flags)
{
    return kValid | ((unsigned long long)(unsigned)f0 << 32) | ((unsigned long long)(unsigned)run << 20) |
           ((unsigned long long)(unsigned)hidx << 8) | (unsigned long long)(unsigned)flags;
}

template <int G>
__device__ __forceinline__ void group(const float *__restrict__ mag, float *__restrict__ hist, int f0, int &hidx,
                                      float4 &s4, int g4, float thr, bool &bad)
{
    float4 m[G], old[G];
    int row[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
        row[g] = hidx;
        m[g] = reinterpret_cast<const float4 *>(mag + (size_t)(f0 + g) * kN)[g4];
        old[g] = reinterpret_cast<const float4 *>(hist + (size_t)hidx * kN)[g4];
        if (++hidx == kHist) hidx = 0;
    }
    float s[4] = { s4.x, s4.y, s4.z, s4.w };
#pragma unroll
    for (int g = 0; g < G; g++) {
        const float mv[4] = { m[g].x, m[g].y, m[g].z, m[g].w }, ov[4] = { old[g].x, old[g].y, old[g].z, old[g].w };
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (mv[u] > 0.99f * thr * s[u] && mv[u] / s[u] > thr) bad = true;
            const float d = s[u] - ov[u];
            s[u] = d + mv[u];
        }
        reinterpret_cast<float4 *>(hist + (size_t)row[g] * kN)[g4] = m[g];
    }
    s4 = make_float4(s[0], s[1], s[2], s[3]);
}

template <int GMAX, bool LOOKAHEAD>
__global__ __launch_bounds__(kThreads) void updaters(const float *__restrict__ mag, float *__restrict__ hist,
                                                     float *__restrict__ sum_g, const unsigned long long *ops, int n_ops,
                                                     unsigned *done, int *flags_out, long long *ticks)
{
    const int w = blockIdx.x, W = gridDim.x, tid = threadIdx.x;
    const int per = (kN / 4 + W - 1) / W;
    const int g_lo = w * per, g_hi = g_lo + per < kN / 4 ? g_lo + per : kN / 4;
    const int cnt = g_hi > g_lo ? g_hi - g_lo : 0;
    __shared__ float4 s_loc[kThreads];
    __shared__ unsigned long long s_op, s_next;
    if (tid < cnt) s_loc[tid] = reinterpret_cast<const float4 *>(sum_g)[g_lo + tid];
    __syncthreads();
    const long long t0 = wall_clock64();
    unsigned long long known = 0ull;            // LOOKAHEAD: the next operation word if it was already there
    for (unsigned seen = 0;; seen++) {
        if (tid == 0) {
            unsigned long long op = known;
            if (op == 0ull && (int)seen < n_ops) {
                int spins = 0;
                while ((op = __hip_atomic_load(&ops[seen], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0ull) {
                    if (++spins > 2000000) break;
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (op == 0ull) { atomicOr(flags_out, 256); op = pack(0, 0, 0, kExit); }
            s_op = op;
            // peek at the following word while this one is processed
            s_next = (LOOKAHEAD && (int)seen + 1 < n_ops)
                         ? __hip_atomic_load(&ops[seen + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
        __syncthreads();
        const unsigned long long op = s_op;
        known = s_next;
        if ((int)(op & 0xffu) & kExit) break;
        const int f0 = (int)((op >> 32) & 0x7fffffffu), run = (int)((op >> 20) & 0xfffu);
        bool bad = false;
        if (tid < cnt) {
            float4 s4 = s_loc[tid];
            int hidx = (int)((op >> 8) & 0xfffu), k0 = 0;
            if (GMAX >= 16)
                for (; run - k0 >= 16; k0 += 16) group<16>(mag, hist, f0 + k0, hidx, s4, g_lo + tid, 0.0452066f, bad);
            for (; run - k0 >= 8; k0 += 8) group<8>(mag, hist, f0 + k0, hidx, s4, g_lo + tid, 0.0452066f, bad);
            for (; run - k0 >= 2; k0 += 2) group<2>(mag, hist, f0 + k0, hidx, s4, g_lo + tid, 0.0452066f, bad);
            for (; k0 < run; k0++) group<1>(mag, hist, f0 + k0, hidx, s4, g_lo + tid, 0.0452066f, bad);
            s_loc[tid] = s4;
            const bool publish = !LOOKAHEAD || known == 0ull || ((int)(known & 0xffu) & kExit);
            if (publish) {
                unsigned *dst = reinterpret_cast<unsigned *>(sum_g) + 4 * (size_t)(g_lo + tid);
                __hip_atomic_store(dst + 0, __float_as_uint(s4.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dst + 1, __float_as_uint(s4.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dst + 2, __float_as_uint(s4.z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dst + 3, __float_as_uint(s4.w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (bad) atomicOr(flags_out, 1);
        const bool publish = !LOOKAHEAD || known == 0ull || ((int)(known & 0xffu) & kExit);
        if (publish) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0)
                __hip_atomic_store(&done[w * 16], seen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid == 0) ticks[w] = wall_clock64() - t0;
}

int main(int argc, char **argv)
{
    const int W = argc > 1 ? atoi(argv[1]) : 7, R = argc > 2 ? atoi(argv[2]) : 32;
    const int F = 4096;
    if (W < 4 || W > 64 || R < 1 || R > 512) { fprintf(stderr, "W in 4..64 (a workgroup holds <= 512 float4 groups), R in 1..512\n"); return 1; }
    float *mag, *hist, *sum;
    unsigned long long *ops;
    unsigned *done;
    int *flags;
    long long *ticks;
    (void)hipMalloc(&mag, sizeof(float) * (size_t)F * kN);
    (void)hipMalloc(&hist, sizeof(float) * (size_t)kHist * kN);
    (void)hipMalloc(&sum, sizeof(float) * kN);
    (void)hipMalloc(&done, sizeof(unsigned) * 16 * 64);
    (void)hipMalloc(&flags, 4);
    (void)hipMalloc(&ticks, 8 * 64);
    std::vector<float> h((size_t)F * kN);
    for (size_t i = 0; i < h.size(); i++) h[i] = 1.0f + (float)(rand() % 1000) * 1e-3f;
    (void)hipMemcpy(mag, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> s0(kN, 768.0f);
    std::vector<unsigned long long> hops;
    int hidx = 0;
    for (int f = 0; f + R <= F; f += R) {
        hops.push_back(pack(f, R, hidx, 0));
        hidx = (hidx + R) % kHist;
    }
    hops.push_back(pack(0, 0, 0, kExit));
    const int n_ops = (int)hops.size(), frames = (n_ops - 1) * R;
    (void)hipMalloc(&ops, sizeof(unsigned long long) * n_ops);
    (void)hipMemcpy(ops, hops.data(), sizeof(unsigned long long) * n_ops, hipMemcpyHostToDevice);
#define RUN(NAME, GMAX, LA)                                                                                       \
    do {                                                                                                          \
        (void)hipMemcpy(hist, h.data(), sizeof(float) * (size_t)kHist * kN, hipMemcpyHostToDevice);              \
        (void)hipMemcpy(sum, s0.data(), sizeof(float) * kN, hipMemcpyHostToDevice);                               \
        (void)hipMemset(done, 0, sizeof(unsigned) * 16 * 64);                                                     \
        (void)hipMemset(flags, 0, 4);                                                                             \
        hipLaunchKernelGGL((updaters<GMAX, LA>), dim3(W), dim3(kThreads), 0, 0, mag, hist, sum, ops, n_ops, done, \
                           flags, ticks);                                                                         \
        (void)hipDeviceSynchronize();                                                                             \
        long long t[64]; int fl;                                                                                  \
        (void)hipMemcpy(t, ticks, 8 * W, hipMemcpyDeviceToHost); (void)hipMemcpy(&fl, flags, 4, hipMemcpyDeviceToHost); \
        long long tm = 0; for (int i = 0; i < W; i++) tm = t[i] > tm ? t[i] : tm;                                 \
        printf("%s: %d updaters, runs of %d: %.3f us per quiet frame (%d frames)%s\n", NAME, W, R,               \
               (double)tm * 0.01 / frames, frames, fl ? "  [flags set]" : "");                                    \
    } while (0)
    RUN("V0 today          ", 8, false);
    RUN("V1 16-frame groups", 16, false);
    RUN("V2 look-ahead     ", 8, true);
    RUN("V3 both           ", 16, true);
    return 0;
}
