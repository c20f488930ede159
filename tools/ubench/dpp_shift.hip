// does gfx950 execute the wavefront-wide DPP shifts (wave_shr:1 / wave_ror:1), and at what cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#define R16(x) x x x x x x x x x x x x x x x x
__global__ void probe(int *out)
{
    const int lane = threadIdx.x;
    int shr = -1, ror = -1, shr0 = -1;
    shr = __builtin_amdgcn_update_dpp(shr, lane * 10, 0x138, 0xf, 0xf, false);     // wave_shr:1, lane 0 keeps `old`
    shr0 = __builtin_amdgcn_update_dpp(shr0, lane * 10, 0x138, 0xf, 0xf, true);    // bound_ctrl: lane 0 reads 0
    ror = __builtin_amdgcn_update_dpp(ror, lane * 10, 0x13c, 0xf, 0xf, false);     // wave_ror:1
    out[lane] = shr;
    out[64 + lane] = shr0;
    out[128 + lane] = ror;
}
__global__ void timing(unsigned long long *t, float *sink, int iters, int mode)
{
    float a = threadIdx.x, b = 1.0f, c = 2.0f, d = 3.0f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (mode == 0) asm volatile(R16("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                                        "v_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n")
                                    : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        else if (mode == 1) asm volatile(R16("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        else asm volatile(R16("v_readlane_b32 s20, %0, 63\n v_readlane_b32 s21, %1, 63\n s_nop 3\n v_writelane_b32 %2, s20, 0\n v_writelane_b32 %3, s21, 0\n")
                          : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "s20", "s21");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
int main()
{
    int *out, h[192];
    hipMalloc(&out, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out);
    hipError_t e = hipDeviceSynchronize();
    printf("probe: %s\n", hipGetErrorString(e));
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("wave_shr:1            lanes 0,1,2,15,16,17,31,32,33,63: %d %d %d %d %d %d %d %d %d %d\n", h[0], h[1], h[2], h[15], h[16], h[17], h[31], h[32], h[33], h[63]);
    printf("wave_shr:1 bound_ctrl lanes 0,1,2,15,16,17,31,32,33,63: %d %d %d %d %d %d %d %d %d %d\n", h[64], h[65], h[66], h[79], h[80], h[81], h[95], h[96], h[97], h[127]);
    printf("wave_ror:1            lanes 0,1,2,15,16,17,31,32,33,63: %d %d %d %d %d %d %d %d %d %d\n", h[128], h[129], h[130], h[143], h[144], h[145], h[159], h[160], h[161], h[191]);
    unsigned long long *t; float *sink;
    hipMalloc(&t, 8 * 1024); hipMalloc(&sink, 4 * 1024 * 256);
    const char *names[] = { "v_mov_b32_dpp wave_shr:1 (4 per group)", "v_mov_b32 (4 per group)", "2 readlane + s_nop 3 + 2 writelane" };
    for (int mode = 0; mode < 3; mode++) {
        hipLaunchKernelGGL(timing, dim3(256), dim3(256), 0, 0, t, sink, 20000, mode);
        hipDeviceSynchronize();
        unsigned long long c; hipMemcpy(&c, t, 8, hipMemcpyDeviceToHost);
        printf("%-44s %.2f ticks per group (one wavefront per SIMD)\n", names[mode], (double)c / (20000 * 16.0));
    }
    return 0;
}
