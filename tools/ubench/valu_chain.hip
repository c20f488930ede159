// micro-benchmark: issue cost of dependent / independent packed-fp32 chains for ONE wavefront per SIMD (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#define R16(x) x x x x x x x x x x x x x x x x
__global__ void k(unsigned long long *out, float2 *sink, int iters, int mode)
{
    float2 a = make_float2(threadIdx.x, 1.0f), b = make_float2(0.5f, 0.25f), c = make_float2(1.0f, 2.0f), d = a, e = b, f = c;
    float2 m0, m1;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (mode == 0) {          // dependent pk_add chain
            asm volatile(R16("v_pk_add_f32 %0, %0, %1\n") : "+v"(a) : "v"(b));
        } else if (mode == 1) {   // two independent pk_add chains
            asm volatile(R16("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n") : "+v"(a), "+v"(d) : "v"(b));
        } else if (mode == 2) {   // FIR pattern: pk_mul (independent) -> pk_add (chain)
            asm volatile(R16("v_pk_mul_f32 %1, %2, %3\n v_pk_add_f32 %0, %0, %1\n") : "+v"(a), "=&v"(m0) : "v"(b), "v"(c));
        } else if (mode == 3) {   // FIR pattern, two outputs
            asm volatile(R16("v_pk_mul_f32 %2, %4, %5\n v_pk_mul_f32 %3, %4, %6\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %3\n") : "+v"(a), "+v"(d), "=&v"(m0), "=&v"(m1) : "v"(b), "v"(c), "v"(f));
        } else if (mode == 4) {   // scalar fp32: mul, mul, add, add (re / im chains separately)
            asm volatile(R16("v_mul_f32 %2, %4, %5\n v_mul_f32 %3, %4, %6\n v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3\n") : "+v"(a.x), "+v"(a.y), "=&v"(m0.x), "=&v"(m0.y) : "v"(b.x), "v"(c.x), "v"(c.y));
        } else if (mode == 5) {   // dependent v_add_f32 chain
            asm volatile(R16("v_add_f32 %0, %0, %1\n") : "+v"(a.x) : "v"(b.x));
        } else if (mode == 6) {   // independent pk_mul only
            asm volatile(R16("v_pk_mul_f32 %0, %2, %3\n v_pk_mul_f32 %1, %2, %3\n") : "=&v"(m0), "=&v"(m1) : "v"(b), "v"(c));
        } else if (mode == 7) {   // FIR pattern with the multiply of the NEXT tap ahead of the add (software skew)
            asm volatile(R16("v_pk_mul_f32 %2, %3, %4\n v_pk_add_f32 %0, %0, %1\n v_pk_mul_f32 %1, %3, %5\n v_pk_add_f32 %0, %0, %2\n") : "+v"(a), "+v"(m0), "=&v"(m1) : "v"(b), "v"(c), "v"(f));
        } else if (mode == 9) {   // 4 independent pk_add chains
            asm volatile(R16("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n") : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b));
        } else if (mode == 10) {  // 8 independent v_add_f32 chains
            asm volatile(R16("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
                         : "+v"(a.x), "+v"(a.y), "+v"(d.x), "+v"(d.y), "+v"(e.x), "+v"(e.y), "+v"(f.x), "+v"(f.y) : "v"(b.x));
        } else if (mode == 8) {   // FIR pattern with an SGPR tap + s_mov per second tap
            asm volatile(R16("v_pk_mul_f32 %1, %2, %3\n s_mov_b32 s20, s21\n v_pk_add_f32 %0, %0, %1\n") : "+v"(a), "=&v"(m0) : "v"(b), "v"(c) : "s20");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = make_float2(a.x + d.x + m0.x + m1.x, a.y + d.y + e.x + f.y);
}
int main()
{
    unsigned long long *out; float2 *sink;
    hipMalloc(&out, 8 * 4096); hipMalloc(&sink, 8 * 4096 * 256);
    const char *names[] = { "pk_add chain", "2 indep pk_add chains (2 instr)", "pk_mul+pk_add chain (2 instr)", "2 outputs: 2 mul + 2 add (4 instr)",
                            "scalar mul,mul,add,add (4 instr)", "v_add_f32 chain", "2 indep pk_mul (2 instr)", "skewed mul/add x2 (4 instr)", "pk_mul, s_mov, pk_add (3 instr)", "4 indep pk_add chains (4 instr)", "8 indep v_add chains (8 instr)" };
    for (int grid = 1; grid <= 256; grid *= 256)
    for (int wg = 256; wg <= 512; wg *= 2)          // 256: one wave per SIMD; 512: two per SIMD
        for (int mode = 0; mode < 11; mode++) {
            if (mode != 2 && mode != 3 && mode != 4 && mode != 7 && mode != 9 && mode != 10) continue;
            const int iters = 200000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k, dim3(grid), dim3(wg), 0, 0, out, sink, 1000, mode);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(wg), 0, 0, out, sink, iters, mode);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c; hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
            printf("grid %3d wg %3d  %-40s %.2f ticks per group, %.2f ns per group\n", grid, wg, names[mode], (double)c / (iters * 16.0), ms * 1e6 / (iters * 16.0));
        }
    return 0;
}
