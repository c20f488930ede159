"""-m gpu: the reference's stage-level entry points (include/irdm_compat.h: burst_detector_*, burst_downmix_*,
qpsk_demod with the reference's signatures and malloc ownership) driven by a plain C program the way main.c drives
them, against the oracle."""
import os
import subprocess

import numpy as np
import pytest

import irdm
import orc
import siggen

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def compat_exe(tmp_path_factory):
    irdm.build()
    out = str(tmp_path_factory.mktemp("compat") / "compat_main")
    libdir = os.path.join(ROOT, "iridium-sniffer_amd")
    subprocess.check_call(["gcc", "-O1", "-std=gnu99", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", out,
                           os.path.join(ROOT, "tests", "compat_main.c"), "-L", libdir, "-lirdm_hip",
                           "-Wl,-rpath," + libdir, "-lm"])
    return out


def _run(exe, path, fs, fmt):
    r = subprocess.run([exe, path, str(fs), fmt], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = r.stdout.decode().splitlines()
    return ([l.split() for l in lines if l.startswith("B ")], [l.split() for l in lines if l.startswith("F ")],
            [l.split() for l in lines if l.startswith("D ")], [l.split() for l in lines if l.startswith("T ")][0],
            r.stderr.decode())


@pytest.mark.parametrize("fmt", ["cf32", "ci8"])
def test_reference_stage_api_matches_the_oracle(compat_exe, tmp_path, fmt):
    fs = 2_000_000
    iq, _ = siggen.standard_scene(fs, int(2.0 * fs) // 32768 * 32768, 8, seed=61, uplink_every=3, amp=0.03)
    if fmt == "cf32":
        data, ofmt = iq, 2
    else:
        data, ofmt = siggen.to_ci8(iq * 8), 0
    path = str(tmp_path / ("in." + fmt))
    data.tofile(path)
    ref = orc.run_stream(data, fs, fmt=ofmt)
    B, F, D, T, err = _run(compat_exe, path, fs, fmt)
    assert "burst_detect: tagged %d bursts total" % ref.n_tagged in err      # burst_detect.c:350-351
    assert int(T[1]) == ref.n_tagged == len(B)
    for b, r in zip(B, ref.bursts):
        assert [int(x) for x in b[1:6]] == [r.id, r.start, r.stop, r.last_active, r.center_bin]
        assert np.float32(b[6]) == np.float32(r.magnitude) and np.float32(b[7]) == np.float32(r.noise)
        assert int(b[8]) == r.num_samples
    ok_frames = [f for f in ref.frames if f.drop_reason == 0]
    assert len(F) == len(ok_frames) >= 4
    # the wall clock enters the timestamps (burst_detect.c:849-853): compare them relative to the first frame
    t0_got, t0_ref = int(F[0][2]), ok_frames[0].timestamp
    for f, r in zip(F, ok_frames):
        assert int(f[1]) == r.id and int(f[2]) - t0_got == r.timestamp - t0_ref
        assert abs(float(f[3]) - r.center_frequency) < 2e-3 and int(f[4]) == r.direction and int(f[6]) == r.num_samples
        assert np.float32(f[7]) == np.float32(r.uw_start)
    assert len(D) == len(ref.demods) >= 3
    for d, r in zip(D, ref.demods):
        assert int(d[1]) == r.id and int(d[4]) == r.direction and int(d[5]) == r.confidence
        assert (int(d[7]), int(d[8]), int(d[9])) == (r.n_symbols, r.n_payload_symbols, r.n_bits)
        assert d[10] == "".join(str(x) for x in r.bits[:r.n_bits])                       # hard bits identical
        assert abs(float(d[6]) - r.level) <= 1e-4 and abs(float(d[3]) - r.center_frequency) <= 0.05
        assert abs(float(d[11]) - r.llr[0]) <= 1e-4 and abs(float(d[12]) - r.llr[r.n_bits - 1]) <= 1e-4
    # qpsk_demod leaves the verified direction in the frame it was given (qpsk_demod.c:454-463)
    demod_ids = {int(d[1]): int(d[4]) for d in D}
    for f in F:
        if int(f[1]) in demod_ids:
            assert int(f[5]) == demod_ids[int(f[1])]


def test_non_default_geometry_is_refused(compat_exe):
    import ctypes as C
    L = irdm.lib()

    class Cfg(C.Structure):
        _fields_ = [("center_frequency", C.c_double), ("sample_rate", C.c_int), ("fft_size", C.c_int),
                    ("burst_pre_len", C.c_int), ("burst_post_len", C.c_int), ("burst_width", C.c_int),
                    ("max_bursts", C.c_int), ("max_burst_len", C.c_int), ("threshold", C.c_float),
                    ("history_size", C.c_int), ("use_gpu", C.c_int)]
    L.burst_detector_create.restype = C.c_void_p
    L.burst_detector_create.argtypes = [C.POINTER(Cfg)]
    c = Cfg(1.622e9, 2_000_000, 4096, 0, 0, 0, 0, 0, 0.0, 0, 1)
    assert not L.burst_detector_create(C.byref(c))
