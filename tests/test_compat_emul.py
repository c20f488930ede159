"""The reference's stage-level entry points (include/irdm_compat.h: burst_detector_*, burst_downmix_*, qpsk_demod with the
reference's signatures, struct layouts and malloc ownership; csrc/compat.cpp) without a GPU: tests/compat_main.c -- the C
program that drives them the way main.c does -- linked against the emulated build of the product (tests/emul_build.py) and
checked by the very functions of tests/test_gpu_compat.py against the oracle."""
import os
import subprocess

import pytest

import emul_build
import test_gpu_compat as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def compat_exe_emul(tmp_path_factory):
    lib = emul_build.build()
    out = str(tmp_path_factory.mktemp("compat_emul") / "compat_main")
    libdir = os.path.dirname(lib)
    subprocess.check_call(["gcc", "-O1", "-std=gnu99", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", out,
                           os.path.join(ROOT, "tests", "compat_main.c"), "-L", libdir, "-l:" + os.path.basename(lib),
                           "-Wl,-rpath," + libdir, "-lm", "-lstdc++", "-pthread"])
    return out


@pytest.mark.parametrize("fmt", ["cf32", "ci8"])
def test_reference_stage_api_on_the_emulation(compat_exe_emul, tmp_path, fmt):
    G.test_reference_stage_api_matches_the_oracle(compat_exe_emul, tmp_path, fmt)


def test_stream_tail_and_getters_on_the_emulation(compat_exe_emul, tmp_path):
    G.test_stream_tail_and_stats_getters(compat_exe_emul, tmp_path)
