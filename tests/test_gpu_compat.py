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


def test_stream_tail_and_stats_getters(compat_exe, tmp_path):
    """A file that is not a multiple of the reader's 32768-sample block, with a burst that expires inside the short
    last block: the reference feeds that block like any other (main.c:223-271, burst_detect.c:746-842), so the burst is
    emitted and counted.  And the getters main.c's stats thread calls (burst_detect.c:355-395)."""
    fs = 2_000_000
    n = 40 * 32768 + 30000
    # the last burst ends ~45 k samples before the end of the file: it expires (16 ms = 32 k samples after its last
    # active frame) inside the final, short block
    iq, _ = siggen.make_stream(fs, n, [
        dict(start=520 * 2048 + 3000, freq_hz=siggen.channel_freq(4), payload=list(np.random.default_rng(1).integers(0, 4, 150))),
        dict(start=n - 45000 - 17000, freq_hz=siggen.channel_freq(-9), payload=list(np.random.default_rng(2).integers(0, 4, 140)))],
        seed=9)
    path = str(tmp_path / "tail.cf32")
    iq.tofile(path)
    ref = orc.run_stream(iq, fs)
    whole = orc.run_stream(iq[:40 * 32768], fs)
    assert ref.n_tagged > whole.n_tagged, "the scene does not exercise the tail"
    B, F, D, T, err = _run(compat_exe, path, fs, "cf32")
    assert "burst_detect: tagged %d bursts total" % ref.n_tagged in err
    assert int(T[1]) == ref.n_tagged == len(B) and int(T[5]) == whole.n_tagged
    assert [int(b[1]) for b in B] == [r.id for r in ref.bursts]
    assert len(D) == len(ref.demods)
    S = [l.split() for l in subprocess.run([compat_exe, path, str(fs), "cf32"], stdout=subprocess.PIPE).stdout.decode().splitlines()
         if l.startswith("S ")][0]
    active, noise, peak = int(S[1]), float(S[2]), float(S[3])
    assert active >= 0 and -140.0 < noise < 0.0
    # (the burst of the tail is still active when the driver asks: the running maximum covers it, burst_detect.c:575-576)
    assert abs(peak - max(np.float32(r.magnitude) for r in ref.bursts)) < 1e-4


def test_detector_stats_formula():
    """irdm_detector_stats against burst_detect.c:363-380 evaluated on the baseline sums read back"""
    import ctypes as C
    fs = 2_000_000
    iq, _ = siggen.standard_scene(fs, 48 * 32768, 3, seed=12)
    p = irdm.Pipeline(fs, max_chunk_samples=len(iq), max_bursts_per_chunk=256)
    p.feed_host(iq)

    class St(C.Structure):
        _fields_ = [("active", C.c_int32), ("primed", C.c_int32), ("noise", C.c_float), ("peak", C.c_float)]
    st = St()
    L = irdm.lib()
    L.irdm_detector_stats.argtypes = [C.c_void_p, C.POINTER(St)]
    assert L.irdm_detector_stats(p.h, C.byref(st)) == 0
    base = p.baseline_sum().astype(np.float64)
    avg = np.float32(base.sum() / (2048 * 512))
    want = np.float32(10.0) * np.log10(avg / np.float32(fs / 2048), dtype=np.float32)
    assert st.primed == 1 and abs(st.noise - float(want)) < 1e-4
    bursts = p.poll_bursts()
    assert bursts and abs(st.peak - max(b.magnitude for b in bursts)) < 1e-6
    p.close()


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

    class Dm(C.Structure):
        _fields_ = [("output_sample_rate", C.c_int), ("search_depth", C.c_int), ("handle_multiple_frames", C.c_int)]
    L.burst_downmix_create.restype = C.c_void_p
    L.burst_downmix_create.argtypes = [C.POINTER(Dm)]
    assert not L.burst_downmix_create(C.byref(Dm(500000, 0, 0)))            # burst_downmix.c:228-239: not the default
    h = L.burst_downmix_create(C.byref(Dm(250000, 250000, 0)))
    assert h
    L.burst_downmix_destroy.argtypes = [C.c_void_p]
    L.burst_downmix_destroy(h)
