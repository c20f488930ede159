/* check_sincosf_gpu.c -- the device's cexpf(i x) (irdm_sincosf_probe: csrc/libm_port.hpp on gfx950) against the host's
 * libm for EVERY float with |x| <= lim (default 2: the fine-CFO step stays within pi/2), in slabs of 32 Mi values.
 *   gcc -O2 -std=gnu99 -pthread -Iinclude -o check_sincosf_gpu tools/check_sincosf_gpu.c -Liridium-sniffer_amd -lirdm_hip -lm
 * Plain C over the C-ABI: what a maintainer of the reference would run to audit the step. */
#include <complex.h>
#include <gnu/libc-version.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "irdm_hip.h"

enum { SLAB = 32 * 1024 * 1024, THREADS = 16 };
static float *g_x, *g_re, *g_im;
static size_t g_n;
static unsigned long long g_bad[THREADS];

static void *worker(void *arg)
{
    const int t = (int)(intptr_t)arg;
    unsigned long long bad = 0;
    for (size_t i = (size_t)t; i < g_n; i += THREADS) {
        const float complex z = cexpf(g_x[i] * I);
        const float hr = crealf(z), hi = cimagf(z);
        if (memcmp(&hr, &g_re[i], 4) != 0 || memcmp(&hi, &g_im[i], 4) != 0) {
            if (!bad) fprintf(stderr, "first mismatch of thread %d: x=%a host=(%a,%a) device=(%a,%a)\n", t, g_x[i], hr, hi, g_re[i], g_im[i]);
            bad++;
        }
    }
    g_bad[t] = bad;
    return NULL;
}

int main(int argc, char **argv)
{
    const float lim = argc > 1 ? (float)atof(argv[1]) : 2.0f;
    uint32_t top;
    memcpy(&top, &lim, 4);
    g_x = malloc(sizeof(float) * SLAB);
    g_re = malloc(sizeof(float) * SLAB);
    g_im = malloc(sizeof(float) * SLAB);
    unsigned long long total = 0, bad = 0;
    for (int sgn = 0; sgn < 2; sgn++)
        for (uint64_t u0 = 0; u0 <= top; u0 += SLAB) {
            g_n = (size_t)((uint64_t)top + 1 - u0 < SLAB ? (uint64_t)top + 1 - u0 : SLAB);
            for (size_t i = 0; i < g_n; i++) {
                const uint32_t w = (uint32_t)(u0 + i) | (sgn ? 0x80000000u : 0u);
                memcpy(&g_x[i], &w, 4);
            }
            if (irdm_sincosf_probe(0, g_x, g_n, g_re, g_im) != 0) {
                fprintf(stderr, "irdm_sincosf_probe failed\n");
                return 2;
            }
            pthread_t th[THREADS];
            for (int t = 0; t < THREADS; t++) pthread_create(&th[t], NULL, worker, (void *)(intptr_t)t);
            for (int t = 0; t < THREADS; t++) {
                pthread_join(th[t], NULL);
                bad += g_bad[t];
            }
            total += g_n;
        }
    printf("device cexpf(i x) vs host libm (%s), every float with |x| <= %g: %llu values, %llu mismatches\n",
           gnu_get_libc_version(), (double)lim, total, bad);
    return bad ? 1 : 0;
}
