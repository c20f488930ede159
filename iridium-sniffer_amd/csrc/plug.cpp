// plug.cpp -- the reference's accelerator plug point, gpu_burst_fft_* (opencl/burst_fft.h:35-47), on gfx950.
#include "pipeline.hpp"

using namespace irdmh;

// ===========================================================================
// 1. gpu_burst_fft_* : the reference's plug point (opencl/burst_fft.h:35-47)
// ===========================================================================
struct gpu_burst_fft {
    int n, log_n, batch;
    int order;                  // fftshift_mag in the reference's AVX2 form (1: what an x86 host with AVX2 computes) or its generic form (0)
    float *d_window;
    float2 *d_tw;
    float2 *d_in;
    float *d_out;
    hipStream_t stream;
};

extern "C" int gpu_burst_fft_process(gpu_burst_fft_t *g, const float *input, float *output, int batch_count);

extern "C" gpu_burst_fft_t *gpu_burst_fft_create(int fft_size, int batch_size, const float *window)
{
    const int lg = ilog2(fft_size);
    if (lg < 8 || lg > 14 || batch_size <= 0 || !window) return nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        fprintf(stderr, "irdm_hip: no HIP device\n");
        return nullptr;
    }
    gpu_burst_fft *g = new (std::nothrow) gpu_burst_fft();
    if (!g) return nullptr;
    g->n = fft_size;
    g->log_n = lg;
    g->batch = batch_size;
    g->order = getenv("IRDM_NO_SIMD") ? 0 : 1;          // (the plug point has no option call: the reference's --no-simd as an environment switch)
    std::vector<cfloat> tw = design_twiddles(fft_size);
    g->d_window = dev_upload(window, (size_t)fft_size);
    g->d_tw = reinterpret_cast<float2 *>(dev_upload(tw.data(), tw.size()));
    g->d_in = dev_alloc<float2>((size_t)fft_size * batch_size);
    g->d_out = dev_alloc<float>((size_t)fft_size * batch_size);
    g->stream = nullptr;
    if (!g->d_window || !g->d_tw || !g->d_in || !g->d_out ||
        hipStreamCreate(&g->stream) != hipSuccess) {
        gpu_burst_fft_destroy(g);
        return nullptr;
    }
    // Does the device actually compute?  One DC frame through the context's own path, as the reference's Vulkan back
    // end does at init (vulkan/burst_fft.c:324-394: DC in, all the energy in bin 0, within a factor of two).  The window
    // is fused here, so the DC bin (index N/2 after the fftshift) holds (sum of the window)^2; the check is tighter than
    // the reference's because the arithmetic is pinned.  A context that fails is not handed out: NULL sends the caller
    // to its CPU path (burst_detect.c:316-318).
    {
        std::vector<float> in((size_t)2 * fft_size), out((size_t)fft_size);
        for (int i = 0; i < fft_size; i++) {
            in[2 * i] = 1.0f;
            in[2 * i + 1] = 0.0f;
        }
        double wsum = 0;
        for (int i = 0; i < fft_size; i++) wsum += window[i];
        const double expected = wsum * wsum;
        bool good = gpu_burst_fft_process(g, in.data(), out.data(), 1) == 0;
        if (good) {
            const double dc = out[fft_size / 2], far = out[0];
            good = expected > 0 && fabs(dc - expected) <= 1e-3 * expected && far <= 1e-3 * expected;
            if (!good)
                fprintf(stderr, "irdm_hip: gpu_burst_fft_create: DC self-test failed (expected %.6g in bin N/2, got %.6g; bin 0 %.6g)\n",
                        expected, dc, far);
        } else {
            fprintf(stderr, "irdm_hip: gpu_burst_fft_create: DC self-test could not run\n");
        }
        if (!good) {
            gpu_burst_fft_destroy(g);
            return nullptr;
        }
    }
    return g;
}

extern "C" void gpu_burst_fft_destroy(gpu_burst_fft_t *g)
{
    if (!g) return;
    if (g->stream) (void)hipStreamDestroy(g->stream);
    (void)hipFree(g->d_window);
    (void)hipFree(g->d_tw);
    (void)hipFree(g->d_in);
    (void)hipFree(g->d_out);
    delete g;
}

extern "C" int gpu_burst_fft_process_device(gpu_burst_fft_t *g, const void *d_input, void *d_output,
                                            int batch_count, void *stream)
{
    if (!g || !d_input || !d_output || batch_count <= 0) return -1;
    return launch_fft_mag(g->log_n, 2, d_input, g->d_window, g->d_tw, static_cast<float *>(d_output),
                          batch_count, static_cast<hipStream_t>(stream), nullptr, g->order);
}

extern "C" int gpu_burst_fft_process(gpu_burst_fft_t *g, const float *input, float *output,
                                     int batch_count)
{
    if (!g || !input || !output) return -1;
    if (batch_count <= 0 || batch_count > g->batch) return -1;       // opencl/burst_fft.c:325-326
    const size_t ns = (size_t)g->n * batch_count;
    IRDM_HIP_CHECK(hipMemcpyAsync(g->d_in, input, ns * sizeof(float2), hipMemcpyHostToDevice, g->stream));
    if (launch_fft_mag(g->log_n, 2, g->d_in, g->d_window, g->d_tw, g->d_out, batch_count, g->stream, nullptr, g->order) != 0)
        return -1;
    IRDM_HIP_CHECK(hipMemcpyAsync(output, g->d_out, ns * sizeof(float), hipMemcpyDeviceToHost, g->stream));
    IRDM_HIP_CHECK(hipStreamSynchronize(g->stream));
    return 0;
}
