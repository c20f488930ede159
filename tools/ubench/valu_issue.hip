// micro-benchmark: VALU issue rate of fp32 multiply / add streams on gfx950 as a function of wavefronts per SIMD.
// The decimator's floor (DESIGN.md "The decimator") rests on these figures.  Every mode is an unrolled block of
// independent-or-chained instructions; the table reports, per mode and per occupancy,
//   ticks/instr/wave : s_memtime ticks one wavefront needs per instruction (latency + arbitration seen by the wave)
//   cyc/instr/SIMD   : wall time x clock / (instructions issued on one SIMD): the issue cost that bounds throughput
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_issue valu_issue.hip ; run: ./valu_issue [clock_MHz]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define R8(x) x x x x x x x x
#define R16(x) R8(x) R8(x)

enum { M_ADD8, M_MUL8, M_FMA8, M_PKADD4, M_PKMUL4, M_PKFMA4, M_FIR_PK1, M_FIR_PK2, M_FIR_PK4, M_FIR_S1, M_FIR_S2, M_FIR_PK2_SGPR,
       M_FIR_S2_SGPR, M_FIR_PK1_SKEW, M_COUNT };
static const char *kNames[M_COUNT] = {
    "v_add_f32 x8 independent chains",      "v_mul_f32 x8 independent",          "v_fma_f32 x8 independent chains",
    "v_pk_add_f32 x4 independent chains",   "v_pk_mul_f32 x4 independent",       "v_pk_fma_f32 x4 independent chains",
    "FIR pk, 1 output/lane (mul,add)",      "FIR pk, 2 outputs/lane (2mul,2add)", "FIR pk, 4 outputs/lane (4mul,4add)",
    "FIR plain, 1 output/lane (2mul,2add)", "FIR plain, 2 outputs/lane (4mul,4add)", "FIR pk, 2 outputs, tap from SGPR",
    "FIR plain, 2 outputs, tap from SGPR",  "FIR pk, 1 output, product one tap ahead" };
static const int kInstr[M_COUNT] = { 8, 8, 8, 4, 4, 4, 2, 4, 8, 4, 8, 4, 8, 4 };       // instructions per group
static const int kTaps[M_COUNT] = { 0, 0, 0, 0, 0, 0, 1, 2, 4, 1, 2, 2, 2, 2 };        // complex tap-outputs per group and lane

__global__ void k(unsigned long long *out, float *sink, int iters, int mode)
{
    float a0 = threadIdx.x, a1 = 1.0f, a2 = 2.0f, a3 = 3.0f, a4 = 4.0f, a5 = 5.0f, a6 = 6.0f, a7 = 7.0f;
    float2 p0 = make_float2(a0, 1.0f), p1 = make_float2(2.0f, a0), p2 = make_float2(3.0f, 1.5f), p3 = make_float2(a0, a0);
    float2 m0, m1, m2, m3;
    float s0, s1, s2, s3;
    const float b = 1.0000001f, c = 0.999f;
    const float2 b2 = make_float2(b, b), c2 = make_float2(c, 1.001f), d2 = make_float2(0.5f, 0.25f), e2 = make_float2(0.3f, 0.7f), f2 = make_float2(0.1f, 0.9f);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        switch (mode) {
        case M_ADD8:
            asm volatile(R16("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            break;
        case M_MUL8:
            asm volatile(R16("v_mul_f32 %0, %8, %9\n v_mul_f32 %1, %8, %9\n v_mul_f32 %2, %8, %9\n v_mul_f32 %3, %8, %9\n"
                             "v_mul_f32 %4, %8, %9\n v_mul_f32 %5, %8, %9\n v_mul_f32 %6, %8, %9\n v_mul_f32 %7, %8, %9\n")
                         : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(b), "v"(c));
            break;
        case M_FMA8:
            asm volatile(R16("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            break;
        case M_PKADD4:
            asm volatile(R16("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(b2));
            break;
        case M_PKMUL4:
            asm volatile(R16("v_pk_mul_f32 %0, %4, %5\n v_pk_mul_f32 %1, %4, %5\n v_pk_mul_f32 %2, %4, %5\n v_pk_mul_f32 %3, %4, %5\n")
                         : "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3) : "v"(b2), "v"(c2));
            break;
        case M_PKFMA4:
            asm volatile(R16("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(b2), "v"(c2));
            break;
        case M_FIR_PK1:
            asm volatile(R16("v_pk_mul_f32 %1, %2, %3\n v_pk_add_f32 %0, %0, %1\n") : "+v"(p0), "=&v"(m0) : "v"(b2), "v"(c2));
            break;
        case M_FIR_PK2:
            asm volatile(R16("v_pk_mul_f32 %2, %4, %5\n v_pk_mul_f32 %3, %4, %6\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %3\n")
                         : "+v"(p0), "+v"(p1), "=&v"(m0), "=&v"(m1) : "v"(b2), "v"(c2), "v"(d2));
            break;
        case M_FIR_PK4:
            asm volatile(R16("v_pk_mul_f32 %4, %8, %9\n v_pk_mul_f32 %5, %8, %10\n v_pk_mul_f32 %6, %8, %11\n v_pk_mul_f32 %7, %8, %12\n"
                             "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_add_f32 %2, %2, %6\n v_pk_add_f32 %3, %3, %7\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3)
                         : "v"(b2), "v"(c2), "v"(d2), "v"(e2), "v"(f2));
            break;
        case M_FIR_S1:
            asm volatile(R16("v_mul_f32 %2, %4, %5\n v_mul_f32 %3, %4, %6\n v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3\n")
                         : "+v"(a0), "+v"(a1), "=&v"(s0), "=&v"(s1) : "v"(b), "v"(c), "v"(a7));
            break;
        case M_FIR_S2:
            asm volatile(R16("v_mul_f32 %4, %8, %9\n v_mul_f32 %5, %8, %10\n v_mul_f32 %6, %8, %11\n v_mul_f32 %7, %8, %12\n"
                             "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %6\n v_add_f32 %3, %3, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3)
                         : "v"(b), "v"(c), "v"(a7), "v"(a6), "v"(a5));
            break;
        case M_FIR_PK2_SGPR:
            asm volatile(R16("v_pk_mul_f32 %2, %4, s[20:21] op_sel_hi:[1,0]\n v_pk_mul_f32 %3, %5, s[20:21] op_sel_hi:[1,0]\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %3\n")
                         : "+v"(p0), "+v"(p1), "=&v"(m0), "=&v"(m1) : "v"(c2), "v"(d2) : "s20", "s21");
            break;
        case M_FIR_S2_SGPR:
            asm volatile(R16("v_mul_f32 %4, s20, %8\n v_mul_f32 %5, s20, %9\n v_mul_f32 %6, s20, %10\n v_mul_f32 %7, s20, %11\n"
                             "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %6\n v_add_f32 %3, %3, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3)
                         : "v"(c), "v"(a7), "v"(a6), "v"(a5) : "s20");
            break;
        case M_FIR_PK1_SKEW:
            asm volatile(R16("v_pk_mul_f32 %2, %3, %4\n v_pk_add_f32 %0, %0, %1\n v_pk_mul_f32 %1, %3, %5\n v_pk_add_f32 %0, %0, %2\n")
                         : "+v"(p0), "+v"(m0), "=&v"(m1) : "v"(b2), "v"(c2), "v"(d2));
            break;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y +
                                                  m0.x + m1.y + m2.x + m3.y + s0 + s1 + s2 + s3;
}

int main(int argc, char **argv)
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const double mhz = argc > 1 ? atof(argv[1]) : prop.clockRate / 1000.0;
    const int n_cu = prop.multiProcessorCount;
    unsigned long long *out;
    float *sink;
    hipMalloc(&out, 8 * 65536);
    hipMalloc(&sink, 4 * 2048 * 2048);
    printf("# %s, %d CUs, clock %.0f MHz (cyc/instr/SIMD assumes this clock for the whole run)\n", prop.name, n_cu, mhz);
    printf("# %-42s %5s %18s %16s %22s\n", "mode", "w/SIMD", "ticks/instr/wave", "cyc/instr/SIMD", "ns per complex tap*64");
    const int iters = 20000;
    for (int mode = 0; mode < M_COUNT; mode++) {
        for (int wps = 1; wps <= 8; wps = wps < 4 ? wps + 1 : wps * 2) {
            // wps wavefronts per SIMD: workgroups of 256 lanes (one wavefront per SIMD), wps workgroups per CU
            const int wg = 256, grid = n_cu * wps;
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipLaunchKernelGGL(k, dim3(grid), dim3(wg), 0, 0, out, sink, 100, mode);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(wg), 0, 0, out, sink, iters, mode);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            static unsigned long long h[65536];
            hipMemcpy(h, out, 8 * grid * 4, hipMemcpyDeviceToHost);
            double tick = 0;
            for (int i = 0; i < grid * 4; i++) tick += (double)h[i];
            tick /= grid * 4.0;
            const double instr_per_wave = (double)iters * 16.0 * kInstr[mode];
            const double cyc_simd = ms * 1e-3 * mhz * 1e6 / (instr_per_wave * wps);
            printf("%-44s %5d %18.2f %16.2f", kNames[mode], wps, tick / instr_per_wave, cyc_simd);
            if (kTaps[mode]) printf(" %22.3f", ms * 1e6 / ((double)iters * 16.0 * kTaps[mode] * wps));
            printf("\n");
            hipEventDestroy(e0);
            hipEventDestroy(e1);
        }
    }
    return 0;
}
