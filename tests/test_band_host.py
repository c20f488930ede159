"""The band-parallel speculative detector scan (csrc/band_core.hpp + the passes of csrc/scan_band.hip restated in
tests/band_host.cpp) against the oracle's sequential detector, on the CPU.

What is checked is the ALGORITHM the gfx950 kernels implement -- speculated update vector, exact sums, per-band
walks over activity segments, boundary agreement, id assignment by a global sort, chunk hand-over -- with the very
source file the walk kernel compiles.  The kernels themselves are checked on the GPU (tests/test_gpu_scenes.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import orc
import scenes
import siggen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Gone(C.Structure):
    _fields_ = [("id", C.c_uint64), ("start", C.c_uint64), ("stop", C.c_uint64), ("last_active", C.c_uint64),
                ("center_bin", C.c_int32), ("peak_rel", C.c_float), ("base_sum", C.c_float), ("pad", C.c_int32)]


@pytest.fixture(scope="module")
def bandlib():
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libbandhost.so")
    src = os.path.join(ROOT, "tests", "band_host.cpp")
    deps = [src, os.path.join(ROOT, "tests", "wave_emul.hpp")] + [os.path.join(ROOT, "iridium-sniffer_amd", "csrc", h)
                                                                   for h in ("band_core.hpp", "band_wave.hpp", "types.hpp")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, src])
    L = C.CDLL(so)
    L.band_host_set_walker.argtypes = [C.c_int]
    L.band_host_scan.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_float, C.c_int, C.c_int, C.c_int, C.POINTER(Gone), C.c_int,
                                 C.POINTER(C.c_float), C.POINTER(C.c_int)]
    return L


def det_params(fs):
    n = 1 << int(round(np.log2(fs / 1000.0)))
    thr = np.float32(np.float32(10.0) ** np.float32(16.0 / 10.0)) / np.float32(512) / np.float32(1.72)
    return dict(n=n, pre=2 * n, post=int(fs * 16e-3), width=40000 // (fs // n),
                max_bursts=int(np.float32(fs / np.float32(40000.0)) * np.float32(0.8)), max_len=int(fs * 0.09),
                thr=float(thr))


def oracle_detect(iq, fs):
    """The oracle's detector on the stream: magnitude plane, burst records in emission order, final sums."""
    L = orc.lib()
    d = L.orc_detector_create(1622000000.0, int(fs), 0.0, 0)
    n = L.orc_detector_fft_size(d)
    frames = len(iq) // n
    mag = np.zeros((frames, n), np.float32)
    L.orc_detector_set_mag_sink(d, orc.fptr(mag), frames)
    recs = []

    def cb(rec, samples, user):
        r = rec.contents
        recs.append((r.id, r.start, r.stop, r.last_active, r.center_bin, r.peak_rel, r.base_sum))

    cbf = orc.BURST_CB(cb)
    x = np.ascontiguousarray(iq).view(np.float32)
    for o in range(0, len(iq), 32768):
        blk = x[2 * o:2 * min(len(iq), o + 32768)]
        L.orc_detector_feed_cf32(d, orc.fptr(blk), len(blk) // 2, cbf, None)
    done = L.orc_detector_frames_done(d)
    sums = np.ctypeslib.as_array(L.orc_detector_baseline_sum(d), shape=(n,)).copy()
    L.orc_detector_destroy(d)
    return mag[:done], recs, sums


def band_scan(L, mag, fs, chunk_frames, band_w=0, max_rounds=8):
    p = det_params(fs)
    out = (Gone * 8192)()
    sums = np.zeros(p["n"], np.float32)
    stats = (C.c_int * 4)()
    rc = L.band_host_scan(orc.fptr(mag), mag.shape[0], p["n"], p["pre"], p["post"], p["width"], p["max_bursts"],
                          p["max_len"], p["thr"], chunk_frames, max_rounds, band_w, out, 8192, orc.fptr(sums), stats)
    recs = [(g.id, g.start, g.stop, g.last_active, g.center_bin, g.peak_rel, g.base_sum) for g in out[:max(rc, 0)]]
    return rc, recs, sums, list(stats)


def check(L, iq, fs, chunks, band_w=0):
    mag, ref, ref_sums = oracle_detect(iq, fs)
    assert len(ref) > 0
    for cf in chunks:
        rc, got, sums, stats = band_scan(L, mag, fs, cf, band_w)
        assert rc >= 0, "band scan aborted: flags 0x%x" % (-rc - 1000)
        assert got == ref, "chunk_frames %d: records differ" % cf
        assert np.array_equal(sums.view(np.uint32), ref_sums.view(np.uint32)), "chunk_frames %d: sums differ" % cf
    return stats


def test_detector_params_match_the_oracle():
    for fs in (2_000_000, 10_000_000, 12_000_000):
        p = det_params(fs)
        assert (p["n"], p["width"], p["max_bursts"]) == {2_000_000: (2048, 40, 40), 10_000_000: (8192, 32, 200),
                                                         12_000_000: (16384, 54, 240)}[fs]


@pytest.mark.parametrize("name", ["too_long", "dc_and_edges", "strong_simultaneous", "junk", "cfo_spread", "frame_lengths"])
def test_band_scan_scene_zoo(bandlib, name):
    """Scene zoo without squelch: whole stream in one chunk, and cut into chunks that split bursts."""
    fs, iq = scenes.ALL[name]()
    stats = check(bandlib, iq, fs, chunks=(1 << 20, 160, 37))
    assert stats[0] >= 1


@pytest.mark.parametrize("seed", range(6))
def test_band_scan_random_scenes(bandlib, seed):
    fs, iq = scenes.random_scene(seed)
    mag, ref, ref_sums = oracle_detect(iq, fs)
    rc, got, sums, stats = band_scan(bandlib, mag, fs, 1 << 20)
    if rc < 0:
        # only a possible squelch may make the scan decline (it then leaves the state to the sequential kernels)
        assert (-rc - 1000) & 16, "flags 0x%x" % (-rc - 1000)
        return
    assert got == ref
    assert np.array_equal(sums.view(np.uint32), ref_sums.view(np.uint32))


def test_band_scan_declines_squelch(bandlib):
    fs, iq = scenes.squelch()
    mag, ref, _ = oracle_detect(iq, fs)
    rc, _, _, _ = band_scan(bandlib, mag, fs, 1 << 20)
    # 44 carriers at once: either the bound on simultaneously active bursts (16) or an overflowing list (4) stops it
    assert rc < 0 and ((-rc - 1000) & (16 | 4))


def test_band_scan_narrow_bands_agree_or_decline(bandlib):
    """12 MHz with the bands of the 10 MHz configuration (128 own bins, 64-bin halo, where a mask is +-27 bins and a
    strong burst's skirt reaches further): whatever the halo no longer covers the boundary check has to catch -- the
    result is either exact or declined, never wrong."""
    fs = 12_000_000
    n = (520 * 16384 + 3 * 1024 * 1024) // 32768 * 32768
    rng = np.random.default_rng(16)
    first = 520 * 16384
    starts = np.sort(rng.integers(first, n - int(0.03 * fs), 90))
    # carriers every 3 channels around the band boundaries, strong: skirts and secondary detections across boundaries
    bursts = [dict(start=int(s), freq_hz=float(rng.uniform(-5.5e6, 5.5e6)),
                   payload=rng.integers(0, 4, int(rng.integers(119, 180))).tolist(), amp=float(rng.choice([0.05, 0.5, 1.0])))
              for s in starts]
    iq, _ = siggen.make_stream(fs, n, bursts, seed=16)
    mag, ref, ref_sums = oracle_detect(iq, fs)
    rc, got, sums, _ = band_scan(bandlib, mag, fs, 1 << 20, band_w=128)
    if rc >= 0:
        assert got == ref
        assert np.array_equal(sums.view(np.uint32), ref_sums.view(np.uint32))
    else:
        assert (-rc - 1000) & 32
    rc, got, sums, _ = band_scan(bandlib, mag, fs, 1 << 20)
    assert rc >= 0 and got == ref


def test_band_scan_10mhz_dense(bandlib):
    """10 MHz, 8192-point frames, 40 bursts per Msample (BASELINE config 5 density) over 4 Mi samples after priming."""
    fs = 10_000_000
    n = (520 * 8192 + 4 * 1024 * 1024) // 32768 * 32768
    rng = np.random.default_rng(5)
    first = 520 * 8192
    nb = 160
    starts = np.sort(rng.integers(first, n - int(0.03 * fs), nb))
    bursts = [dict(start=int(s), freq_hz=siggen.channel_freq(int(rng.integers(-110, 111)) or 1),
                   payload=rng.integers(0, 4, int(rng.integers(119, 180))).tolist()) for s in starts]
    iq, _ = siggen.make_stream(fs, n, bursts, seed=5)
    check(bandlib, iq, fs, chunks=(1 << 20, 128))


def test_band_scan_12mhz(bandlib):
    fs = 12_000_000
    n = (520 * 16384 + 3 * 1024 * 1024) // 32768 * 32768
    rng = np.random.default_rng(6)
    first = 520 * 16384
    starts = np.sort(rng.integers(first, n - int(0.03 * fs), 60))
    bursts = [dict(start=int(s), freq_hz=siggen.channel_freq(int(rng.integers(-130, 131)) or 1),
                   payload=rng.integers(0, 4, int(rng.integers(119, 180))).tolist(), amp=float(rng.choice([0.05, 0.3])))
              for s in starts]
    iq, _ = siggen.make_stream(fs, n, bursts, seed=6)
    check(bandlib, iq, fs, chunks=(1 << 20, 64))


WALKERS = {1: "a wavefront per band and segment (csrc/band_wave.hpp) on the emulated wavefront of tests/wave_emul.hpp",
           2: "the same without its 64-frame look-ahead (skim)"}


@pytest.mark.parametrize("walker", sorted(WALKERS))
@pytest.mark.parametrize("name", ["too_long", "dc_and_edges", "strong_simultaneous", "cfo_spread", "many_active_10m", "junk",
                                  "frame_lengths"])
def test_wavefront_walk_scene_zoo(bandlib, name, walker):
    """The walk pass in the form the GPU runs -- lane = burst slot, ballots, readlanes, DPP reductions, the look-ahead over
    64 frames -- executed lane by lane on the CPU (64 user-space contexts in lock step) inside the same sequential
    restatement of the other passes: records and final sums must equal the oracle's, whole stream and cut into chunks
    that split bursts (burst_detect.c:426-632)."""
    fs, iq = scenes.ALL[name]()
    bandlib.band_host_set_walker(walker)
    try:
        check(bandlib, iq, fs, chunks=(1 << 20, 37))
    finally:
        bandlib.band_host_set_walker(0)


@pytest.mark.parametrize("seed", range(3))
def test_wavefront_walk_random_scenes(bandlib, seed):
    fs, iq = scenes.random_scene(seed)
    mag, ref, ref_sums = oracle_detect(iq, fs)
    bandlib.band_host_set_walker(1)
    try:
        rc, got, sums, stats = band_scan(bandlib, mag, fs, 1 << 20)
    finally:
        bandlib.band_host_set_walker(0)
    if rc < 0:
        assert (-rc - 1000) & 16, "flags 0x%x" % (-rc - 1000)
        return
    assert got == ref
    assert np.array_equal(sums.view(np.uint32), ref_sums.view(np.uint32))


def test_wavefront_walk_12mhz_dense(bandlib):
    """16384-point frames, bands of 256 bins (eight crossing words per band), a dense scene"""
    fs = 12_000_000
    n = (520 * 16384 + 2 * 1024 * 1024) // 32768 * 32768
    rng = np.random.default_rng(12)
    first = 520 * 16384
    starts = np.sort(rng.integers(first, n - int(0.03 * fs), 70))
    bursts = [dict(start=int(s), freq_hz=siggen.channel_freq(int(rng.integers(-110, 111)) or 1),
                   payload=rng.integers(0, 4, int(rng.integers(119, 180))).tolist()) for s in starts]
    iq, _ = siggen.make_stream(fs, n, bursts, seed=12)
    bandlib.band_host_set_walker(1)
    try:
        check(bandlib, iq, fs, chunks=(1 << 20, 50))
    finally:
        bandlib.band_host_set_walker(0)
