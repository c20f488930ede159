#!/usr/bin/env python3
"""Wider differential run of the wavefront walk (iridium-sniffer_amd/csrc/band_wave.hpp on the emulated wavefront of
tests/wave_emul.hpp, inside tests/band_host.cpp's sequential restatement of the other band-scan passes) against the
oracle's detector: random scenes (tests/scenes.py: random_scene), whole stream and cut into chunks that split bursts,
with and without the 64-frame look-ahead.  No GPU.
Usage: python tools/fuzz_wave_walk.py [first_seed last_seed]   (the -m "not gpu" suite runs seeds 0-2)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "iridium-sniffer_amd"))
import scenes                      # noqa: E402
import test_band_host as T         # noqa: E402


def load():
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libbandhost.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so,
                           os.path.join(ROOT, "tests", "band_host.cpp")])
    L = C.CDLL(so)
    L.band_host_set_walker.argtypes = [C.c_int]
    L.band_host_scan.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_float, C.c_int, C.c_int, C.c_int, C.POINTER(T.Gone), C.c_int,
                                 C.POINTER(C.c_float), C.POINTER(C.c_int)]
    return L


def main():
    lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 60)
    L = load()
    ok = declined = bad = 0
    for seed in range(lo, hi):
        fs, iq = scenes.random_scene(seed)
        mag, ref, ref_sums = T.oracle_detect(iq, fs)
        for walker in (1, 2):
            for cf in (1 << 20, 53):
                L.band_host_set_walker(walker)
                rc, got, sums, _ = T.band_scan(L, mag, fs, cf)
                L.band_host_set_walker(0)
                if rc < 0:
                    declined += 1
                elif got != ref or not np.array_equal(sums.view(np.uint32), ref_sums.view(np.uint32)):
                    bad += 1
                    print("MISMATCH seed %d walker %d chunk_frames %d: %d records vs %d" % (seed, walker, cf, len(got), len(ref)))
                else:
                    ok += 1
    print("seeds %d..%d: %d equal to the oracle, %d declined, %d different" % (lo, hi - 1, ok, declined, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
