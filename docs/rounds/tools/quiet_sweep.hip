// tools/quiet_sweep.hip -- measurement aid (not part of the product): how fast can the noise-floor update of QUIET frames
// (sum = (sum - oldest) + mag for every bin, history row replaced, exact crossing check) run when the bins are spread
// over W workgroups that agree, once per block of frames, on "no bin crossed" through a counter in device memory?
// Input for the round-2 design of the detector scan: today one CU does it at ~1.3 us per frame.
// Every spin is bounded; on a timeout the kernel sets a flag and all workgroups leave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int kN = 8192, kHist = 512, kBlock = 32;

template <int B>   // bins per thread
__global__ __launch_bounds__(256) void sweep(const float *__restrict__ mag, float *__restrict__ hist, float *__restrict__ sum,
                                             int n_frames, float thr, unsigned *counters, int *crossed, int *fail,
                                             long long *ticks)
{
    const int W = gridDim.x, tid = threadIdx.x;
    const int bin0 = (blockIdx.x * 256 + tid) * B;
    __shared__ int s_fail, s_cross;
    if (tid == 0) { s_fail = 0; s_cross = 0; }
    __syncthreads();
    float s[B];
    for (int b = 0; b < B; b++) s[b] = sum[bin0 + b];
    long long t0 = 0;
    if (blockIdx.x == 0 && tid == 0) t0 = wall_clock64();
    int hidx = 0;
    for (int f0 = 0; f0 < n_frames; f0 += kBlock) {
        float m[kBlock][B], o[kBlock][B];
        // every load of the block in flight at once
#pragma unroll
        for (int g = 0; g < kBlock; g++)
#pragma unroll
            for (int b = 0; b < B; b++) {
                m[g][b] = mag[(size_t)(f0 + g) * kN + bin0 + b];
                o[g][b] = hist[(size_t)((hidx + g) % kHist) * kN + bin0 + b];
            }
        bool cross = false;
#pragma unroll
        for (int g = 0; g < kBlock; g++)
#pragma unroll
            for (int b = 0; b < B; b++) {
                if (m[g][b] > 0.99f * thr * s[b] && m[g][b] / s[b] > thr) cross = true;
                s[b] = (s[b] - o[g][b]) + m[g][b];
                hist[(size_t)((hidx + g) % kHist) * kN + bin0 + b] = m[g][b];
            }
        hidx = (hidx + kBlock) % kHist;
        if (cross) s_cross = 1;
        __syncthreads();
        // agreement: all W workgroups arrive, then read the verdict
        if (tid == 0) {
            if (s_cross) atomicOr(crossed, 1);
            __threadfence();
            __hip_atomic_fetch_add(&counters[f0 / kBlock], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(&counters[f0 / kBlock], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)W) {
                if (++spins > 400000) { *fail = 1; s_fail = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) s_fail = 1;
        }
        __syncthreads();
        if (s_fail) break;
    }
    for (int b = 0; b < B; b++) sum[bin0 + b] = s[b];
    if (blockIdx.x == 0 && tid == 0) *ticks = wall_clock64() - t0;
}

int main()
{
    const int F = 4096;
    float *mag, *hist, *sum;
    unsigned *counters;
    int *crossed, *fail;
    long long *ticks;
    (void)hipMalloc(&mag, sizeof(float) * (size_t)F * kN);
    (void)hipMalloc(&hist, sizeof(float) * (size_t)kHist * kN);
    (void)hipMalloc(&sum, sizeof(float) * kN);
    (void)hipMalloc(&counters, sizeof(unsigned) * (F / kBlock));
    (void)hipMalloc(&crossed, 4); (void)hipMalloc(&fail, 4); (void)hipMalloc(&ticks, 8);
    std::vector<float> h((size_t)F * kN);
    for (size_t i = 0; i < h.size(); i++) h[i] = 1.0f + (float)(rand() % 1000) * 1e-3f;
    (void)hipMemcpy(mag, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(hist, h.data(), sizeof(float) * (size_t)kHist * kN, hipMemcpyHostToDevice);
    std::vector<float> s0(kN, 768.0f);
#define RUN(B)                                                                                                    \
    do {                                                                                                          \
        (void)hipMemcpy(sum, s0.data(), sizeof(float) * kN, hipMemcpyHostToDevice);                               \
        (void)hipMemset(counters, 0, sizeof(unsigned) * (F / kBlock));                                            \
        (void)hipMemset(crossed, 0, 4); (void)hipMemset(fail, 0, 4);                                              \
        const int W = kN / (256 * B);                                                                             \
        hipLaunchKernelGGL(sweep<B>, dim3(W), dim3(256), 0, 0, mag, hist, sum, F, 0.0452066f, counters, crossed,   \
                           fail, ticks);                                                                          \
        (void)hipDeviceSynchronize();                                                                             \
        long long t; int fl;                                                                                      \
        (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&fl, fail, 4, hipMemcpyDeviceToHost);\
        printf("%d bins/thread, %2d workgroups: %.3f us per quiet frame (%d frames, agreement every %d)%s\n", B, W,\
               (double)t * 0.01 / F, F, kBlock, fl ? "  [TIMEOUT]" : "");                                         \
    } while (0)
    RUN(4);
    RUN(2);
    RUN(1);
    return 0;
}
