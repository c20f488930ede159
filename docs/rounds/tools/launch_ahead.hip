// What does it cost a dependent single-workgroup pass to start behind a many-workgroup pass, and does launching it AHEAD on a
// second high-priority stream (resident, waiting on a counter the first pass's workgroups bump) hide that cost?
// The question behind DESIGN.md section 8 "plan / commit launched ahead": in the detector scan the plan pass (1 workgroup,
// 1024 threads, 82 KB of LDS) waits 20-400 us behind the walk pass when the per-burst chains fill the chip.
//
//   worker   256 workgroups x 256 threads, each busy for `work_us`, then release + counter++
//   waiter   1 workgroup x 1024 threads, 80 KB LDS: stamps its start, waits (bounded) for counter >= target, stamps again
//   filler   (optional, low-priority stream) a chip-filling kernel like K1: 8192 workgroups x 512 threads, 66 KB LDS, ~30 us each
//
// A: worker ; waiter on ONE stream (the scan as it is).   B: worker on stream 1, waiter on stream 2 enqueued right after
// it (launch ahead).  Printed per variant, alone and beside the filler: when the waiter started relative to the worker's
// first start / last end, and when it saw the counter.  All waits are bounded (20 ms).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Stamps {
    unsigned long long worker_first, worker_last, waiter_start, waiter_seen, waiter_end;
    unsigned counter, timed_out;
};

__global__ __launch_bounds__(256) void worker(Stamps *s, unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) atomicMin(&s->worker_first, t0);
    float a = threadIdx.x;
    while (wall_clock64() - t0 < ticks) a = a * 1.0001f + 0.5f;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (a == 12345.0f) s->timed_out = 2;          // (keeps the loop)
        atomicMax(&s->worker_last, wall_clock64());
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(&s->counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(1024) void waiter(Stamps *s, unsigned target)
{
    extern __shared__ unsigned char lds[];
    __shared__ int ok;
    if (threadIdx.x == 0) {
        s->waiter_start = wall_clock64();
        const unsigned long long t0 = s->waiter_start;
        ok = 1;
        while (__hip_atomic_load(&s->counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > 2000000ull) {       // 20 ms
                ok = 0;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        s->waiter_seen = wall_clock64();
        if (!ok) s->timed_out = 1;
    }
    __syncthreads();
    lds[threadIdx.x] = (unsigned char)threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) s->waiter_end = wall_clock64() + lds[5] - 5;
}

__global__ __launch_bounds__(512) void filler(float *sink, unsigned long long ticks)
{
    extern __shared__ unsigned char lds[];
    const unsigned long long t0 = wall_clock64();
    float a = threadIdx.x;
    lds[threadIdx.x] = 1;
    while (wall_clock64() - t0 < ticks) a = a * 1.0001f + lds[threadIdx.x & 255];
    if (a == 12345.0f) sink[0] = a;
}

int main()
{
    int lo = 0, hi = 0;
    CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t s1, s2, sf;
    CHECK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi));
    CHECK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi));
    CHECK(hipStreamCreateWithPriority(&sf, hipStreamNonBlocking, lo));
    Stamps *d = nullptr, h;
    float *sink = nullptr;
    CHECK(hipMalloc(&d, sizeof(Stamps)));
    CHECK(hipMalloc(&sink, 256));
    CHECK(hipFuncSetAttribute((const void *)waiter, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CHECK(hipFuncSetAttribute((const void *)filler, hipFuncAttributeMaxDynamicSharedMemorySize, 66 * 1024));
    hipEvent_t ev;
    CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const unsigned n_workers = 256;
    const unsigned long long work = 15000;            // 150 us in 10 ns ticks
    printf("variant        load    waiter start - worker first   waiter start - worker last   seen - worker last   (us)\n");
    for (int load = 0; load < 2; load++) {
        for (int variant = 0; variant < 3; variant++) {
            double acc[3] = { 0, 0, 0 };
            int n = 0, timeouts = 0;
            for (int rep = 0; rep < 12; rep++) {
                Stamps init = { ~0ull, 0, 0, 0, 0, 0, 0 };
                CHECK(hipMemcpy(d, &init, sizeof(init), hipMemcpyHostToDevice));
                if (load) hipLaunchKernelGGL(filler, dim3(8192), dim3(512), 66 * 1024, sf, sink, 3000ull);
                hipLaunchKernelGGL(worker, dim3(n_workers), dim3(256), 0, s1, d, work);
                if (variant == 0) {
                    hipLaunchKernelGGL(waiter, dim3(1), dim3(1024), 80 * 1024, s1, d, n_workers);       // in stream
                } else if (variant == 1) {
                    hipLaunchKernelGGL(waiter, dim3(1), dim3(1024), 80 * 1024, s2, d, n_workers);       // ahead, other stream
                } else {
                    CHECK(hipEventRecord(ev, s1));                                                      // other stream behind an event
                    CHECK(hipStreamWaitEvent(s2, ev, 0));
                    hipLaunchKernelGGL(waiter, dim3(1), dim3(1024), 80 * 1024, s2, d, n_workers);
                }
                CHECK(hipDeviceSynchronize());
                CHECK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
                if (h.timed_out == 1) timeouts++;
                if (rep >= 2) {
                    acc[0] += ((double)h.waiter_start - (double)h.worker_first) / 100.0;
                    acc[1] += ((double)h.waiter_start - (double)h.worker_last) / 100.0;
                    acc[2] += ((double)h.waiter_seen - (double)h.worker_last) / 100.0;
                    n++;
                }
            }
            const char *names[3] = { "in stream    ", "ahead        ", "behind event " };
            printf("%s  %s  %10.1f  %28.1f  %22.1f   timeouts %d\n", names[variant], load ? "filler" : "alone ", acc[0] / n,
                   acc[1] / n, acc[2] / n, timeouts);
        }
    }
    return 0;
}
