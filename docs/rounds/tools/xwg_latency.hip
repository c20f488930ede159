// tools/xwg_latency.hip -- measurement aid (not part of the product): round-trip latency of a flag hand-shake between two
// workgroups through device memory on gfx950, with and without a 32 KB payload published under an agent-scope fence.
// Input for the round-2 design decision "baseline commands on several CUs" (detector scan).  Every spin is
// bounded, so the kernel always terminates.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(256) void pingpong(unsigned *flag_a, unsigned *flag_b, float *payload, int rounds, int bytes,
                                               long long *ticks, int *fail)
{
    const int wg = blockIdx.x, tid = threadIdx.x;
    const int n = bytes / 4;
    __shared__ int s_fail;
    if (tid == 0) s_fail = 0;
    __syncthreads();
    long long t0 = 0;
    if (wg == 0 && tid == 0) t0 = wall_clock64();
    for (int r = 1; r <= rounds; r++) {
        if (wg == 0) {
            for (int i = tid; i < n; i += 256) payload[i] = (float)(r + i);          // producer data
            __syncthreads();
            if (tid == 0) {
                __threadfence();                                                       // agent-scope release
                __hip_atomic_store(flag_a, (unsigned)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (__hip_atomic_load(flag_b, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)r) {
                    if (++spins > 200000) { *fail = 1; s_fail = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
        } else {
            if (tid == 0) {
                int spins = 0;
                while (__hip_atomic_load(flag_a, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)r) {
                    if (++spins > 200000) { *fail = 1; s_fail = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                __threadfence();                                                       // agent-scope acquire
            }
            __syncthreads();
            float acc = 0.0f;
            for (int i = tid; i < n; i += 256) acc += payload[i];                      // consumer reads the data
            if (acc == -1.0f) payload[0] = acc;                                        // keep the loads
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flag_b, (unsigned)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (s_fail) break;                     // uniform exit: every thread of the workgroup sees the same LDS flag
    }
    if (wg == 0 && tid == 0) *ticks = wall_clock64() - t0;
}

int main()
{
    unsigned *fa, *fb;
    float *payload;
    long long *ticks;
    int *fail;
    (void)hipMalloc(&fa, 4); (void)hipMalloc(&fb, 4); (void)hipMalloc(&payload, 1 << 20); (void)hipMalloc(&ticks, 8);
    (void)hipMalloc(&fail, 4);
    const int rounds = 2000;
    for (int bytes : { 0, 4096, 32768, 262144 }) {
        (void)hipMemset(fa, 0, 4); (void)hipMemset(fb, 0, 4); (void)hipMemset(fail, 0, 4);
        hipLaunchKernelGGL(pingpong, dim3(2), dim3(256), 0, 0, fa, fb, payload, rounds, bytes, ticks, fail);
        (void)hipDeviceSynchronize();
        long long t; int f;
        (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
        printf("payload %7d B: %.2f us per round trip (2 hand-shakes)%s\n", bytes, (double)t * 0.01 / rounds, f ? "  [TIMEOUT]" : "");
    }
    return 0;
}
