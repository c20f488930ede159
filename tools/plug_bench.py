#!/usr/bin/env python3
"""Time the reference's literal plug point, gpu_burst_fft_process (opencl/burst_fft.h:46-47), the way
burst_detect.c:637-674 drives it: synchronous calls of batch_size frames from pageable host memory.
Prints us/call and Msamples/s for each (fft_size, batch)."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
irdm = importlib.import_module("iridium-sniffer_amd.irdm")


def main():
    rng = np.random.default_rng(1)
    for n, batch in ((8192, 16), (16384, 16), (8192, 64), (8192, 256)):
        w = (np.blackman(n) / 0.42).astype(np.float32)
        g = irdm.GpuBurstFFT(n, batch, w)
        x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex64)
        out = np.zeros((batch, n), np.float32)          # caller-owned buffers, reused like the detector's (burst_detect.c:301-305)
        xp, op = irdm._fp(x.view(np.float32)), irdm._fp(out)
        ref = g.process(x)
        for _ in range(5):
            assert g.L.gpu_burst_fft_process(g.h, xp, op, batch) == 0
        assert np.array_equal(out, ref)
        reps = 200 if batch <= 64 else 50
        t0 = time.perf_counter()
        for _ in range(reps):
            g.L.gpu_burst_fft_process(g.h, xp, op, batch)
        dt = (time.perf_counter() - t0) / reps
        print("fft_size %5d batch %3d: %8.1f us/call  %8.1f Msamples/s" % (n, batch, dt * 1e6, n * batch / dt / 1e6))
        g.close()


if __name__ == "__main__":
    main()
