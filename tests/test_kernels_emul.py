"""The band scan's GPU code on the CPU: iridium-sniffer_amd/csrc/scan_band.hip -- every kernel as the gfx950 build compiles
it (plan pass with its LDS path and the all-boundaries test, sums pass with buffer loads, frame-walking crossing pass,
wavefront walk with the 64-frame look-ahead, commit fused into the accepting plan pass, history) -- compiled with g++
against the HIP emulation of tests/hip_emul/hip/hip_runtime.h (a workgroup = user-space contexts in lock step, a launch
= its workgroups one after the other) and driven chunk by chunk through launch_band_scan() the way csrc/scan_host.cpp does
(tests/scan_emul.cpp), against the oracle's sequential detector (burst_detect.c:426-632, :689-698): burst records in
emission order, ids, and the final baseline sums, bit for bit.  No GPU."""
import ctypes as C
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

import orc
import scenes
import siggen
from test_band_host import Gone, det_params, oracle_detect

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "iridium-sniffer_amd", "csrc")


@pytest.fixture(scope="module")
def emul():
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libscanemul.so")
    inc = os.path.join(out_dir, "scan_band_emul.inc")
    src = os.path.join(ROOT, "tests", "scan_emul.cpp")
    deps = [src, os.path.join(ROOT, "tests", "hip_emul", "hip", "hip_runtime.h")] + [
        os.path.join(CSRC, h) for h in ("scan_band.hip", "band_core.hpp", "band_wave.hpp", "types.hpp", "kernels.hpp", "common.hpp")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        text = open(os.path.join(CSRC, "scan_band.hip")).read()
        # the one change: dynamic LDS arrays become pointers into the emulation's LDS buffer
        text, n = re.subn(r"extern __shared__ __attribute__\(\(aligned\(16\)\)\) unsigned char (\w+)\[\];",
                          r"unsigned char *\1 = hip_emul::dyn_lds();", text)
        assert n >= 2
        open(inc, "w").write(text)
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               "-I" + os.path.join(ROOT, "tests", "hip_emul"), "-I" + out_dir, "-I" + CSRC, "-o", so, src])
    L = C.CDLL(so)
    L.scan_emul_option.argtypes = [C.c_char_p, C.c_int]
    L.scan_emul_run.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_float, C.c_int, C.c_int, C.POINTER(Gone), C.c_int, C.POINTER(C.c_float),
                                C.POINTER(C.c_int)]
    return L


def run(L, mag, fs, chunk_frames, first_rounds=0):
    p = det_params(fs)
    out = (Gone * 8192)()
    sums = np.zeros(p["n"], np.float32)
    stats = (C.c_int * 8)()
    rc = L.scan_emul_run(orc.fptr(mag), mag.shape[0], p["n"], p["pre"], p["post"], p["width"], p["max_bursts"],
                         p["max_len"], p["thr"], chunk_frames, first_rounds, out, 8192, orc.fptr(sums), stats)
    recs = [(g.id, g.start, g.stop, g.last_active, g.center_bin, g.peak_rel, g.base_sum) for g in out[:max(rc, 0)]]
    return rc, recs, sums, list(stats)


_TINY = []


def _tiny_scene():
    """(mag, fs, chunk_frames) of a short scene: a run of it reads the emulation's counters"""
    if not _TINY:
        fs, iq = scenes.ALL["junk"]()
        _TINY.append((oracle_detect(iq, fs)[0], fs, 1 << 20))
    return _TINY[0]


def check(L, iq, fs, chunks, **kw):
    mag, ref, ref_sums = oracle_detect(iq, fs)
    assert len(ref) > 0
    stats = None
    for cf in chunks:
        rc, got, sums, stats = run(L, mag, fs, cf, **kw)
        assert rc >= 0, "the scan declined a chunk: flags 0x%x" % (-rc - 1000)
        assert got == ref, "chunk_frames %d: records differ" % cf
        assert np.array_equal(sums.view(np.uint32), ref_sums.view(np.uint32)), "chunk_frames %d: sums differ" % cf
    return stats


@pytest.mark.parametrize("name", ["too_long", "dc_and_edges", "strong_simultaneous", "cfo_spread", "junk"])
def test_scan_kernels_scene_zoo(emul, name):
    """whole stream in one chunk and cut into chunks that split bursts (carried bursts, history ring across chunks)"""
    fs, iq = scenes.ALL[name]()
    stats = check(emul, iq, fs, chunks=(1 << 20, 97))
    assert stats[1] >= 1
    try:
        # round 0 as a speculation pass on a second workspace (band_spec): with the carried bursts the previous pass left,
        # and with that guess withheld (2) -- a wrong guess costs a round, never a result
        for mode in (1, 2):
            emul.scan_emul_option(b"band_spec", mode)
            stats = check(emul, iq, fs, chunks=(97, 33))
            assert stats[6] >= stats[1] - 1 >= 2, "no speculation passes: %r" % stats
        emul.scan_emul_option(b"band_spec", 0)
        check(emul, iq, fs, chunks=(1 << 20,), first_rounds=1)
    finally:
        emul.scan_emul_option(b"band_spec", 0)


def test_scan_kernels_continuation_and_options(emul):
    """two rounds enqueued up front where the scene needs more (the continuation launch); the test hooks of the scan: the walk
    without look-ahead (band_selfcheck 8), both forms of the boundary test compared on the fly (1), the plan pass without its
    LDS and the commit as a launch of its own (16)"""
    fs, iq = scenes.ALL["too_long"]()
    stats = check(emul, iq, fs, chunks=(1 << 20,), first_rounds=1)
    assert stats[4] >= 1, "no continuation launch was needed: %r" % stats
    try:
        check(emul, iq, fs, chunks=(131, 1 << 20))
        for value in (8, 1, 16):
            emul.scan_emul_option(b"band_selfcheck", value)
            check(emul, iq, fs, chunks=(131,))
            check(emul, iq, fs, chunks=(1 << 20,), first_rounds=1)
    finally:
        emul.scan_emul_option(b"band_selfcheck", 0)


def test_scan_kernels_sums_pass_restart(emul):
    """A later round's sums pass restarts at the last stored state (one in 64 update steps) in front of the first frame
    whose update flag changed: a long quiet stream, the speculation pass's guess spoilt in ONE late frame (band_selfcheck
    32) so that a third round is needed whose prefix is unchanged -- same records, and the final sums bit for bit (the
    restarted recurrence IS the sequential one, simd_generic.c:129-135); with the restart off, the same."""
    fs = 2_000_000
    n = int(3.0 * fs) // 32768 * 32768
    iq = siggen.standard_scene(fs, n, 9, seed=101, first_start=int(0.7 * fs))[0]
    mag, ref, ref_sums = oracle_detect(iq, fs)
    try:
        emul.scan_emul_option(b"band_spec", 1)
        emul.scan_emul_option(b"band_selfcheck", 32)
        for restart, chunk in ((1, 800), (1, 500), (0, 800)):
            emul.scan_emul_option(b"band_sum_restart", restart)
            rc, got, sums, stats = run(emul, mag, fs, chunk)
            assert rc >= 0 and got == ref
            assert np.array_equal(sums.view(np.uint32), ref_sums.view(np.uint32))
            assert stats[0] > 2 * stats[1], stats                      # (the spoilt guess cost rounds)
            assert (stats[7] >= 2) if restart else (stats[7] == 0), stats
    finally:
        emul.scan_emul_option(b"band_spec", 0)
        emul.scan_emul_option(b"band_selfcheck", 0)
        emul.scan_emul_option(b"band_sum_restart", 1)


@pytest.mark.parametrize("seed", range(3))
def test_scan_kernels_random_scenes(emul, seed):
    fs, iq = scenes.random_scene(seed)
    mag, ref, ref_sums = oracle_detect(iq, fs)
    rc, got, sums, stats = run(emul, mag, fs, 1 << 20)
    if rc < 0:
        assert (-rc - 1000) & 16, "flags 0x%x" % (-rc - 1000)       # only a possible squelch may make it decline
        return
    assert got == ref
    assert np.array_equal(sums.view(np.uint32), ref_sums.view(np.uint32))


def test_scan_kernels_decline_squelch(emul):
    fs, iq = scenes.squelch()
    mag, ref, _ = oracle_detect(iq, fs)
    rc, _, _, _ = run(emul, mag, fs, 1 << 20)
    assert rc < 0 and ((-rc - 1000) & (16 | 4))


def _dense(fs, n_fft, samples, nb, seed):
    n = (520 * n_fft + samples) // 32768 * 32768
    rng = np.random.default_rng(seed)
    first = 520 * n_fft
    starts = np.sort(rng.integers(first, n - int(0.03 * fs), nb))
    bursts = [dict(start=int(s), freq_hz=siggen.channel_freq(int(rng.integers(-110, 111)) or 1),
                   payload=rng.integers(0, 4, int(rng.integers(119, 180))).tolist()) for s in starts]
    iq, _ = siggen.make_stream(fs, n, bursts, seed=seed)
    return iq


def test_scan_kernels_10mhz_dense(emul):
    """8192-point frames, 40 bursts per Msample (BASELINE config 5's density) over 4 Mi samples after priming: 64 bands of
    128 bins, four crossing words per band, the plan's LDS path at its frame capacity class"""
    fs = 10_000_000
    check(emul, _dense(fs, 8192, 4 * 1024 * 1024, 160, 5), fs, chunks=(1 << 20, 128))


def test_scan_kernels_12mhz(emul):
    """16384-point frames: 64 bands of 256 bins, eight crossing words per band (the <8> instantiations)"""
    fs = 12_000_000
    check(emul, _dense(fs, 16384, 2 * 1024 * 1024, 70, 12), fs, chunks=(1 << 20, 50))


# ---- csrc/detect.hip on the same emulation: K1 and the dense sequential scan ----

class Entry(C.Structure):
    _fields_ = [("bin", C.c_int32), ("mag", C.c_float)]


@pytest.fixture(scope="module")
def detect_emul():
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libdetectemul.so")
    inc = os.path.join(out_dir, "detect_emul.inc")
    src = os.path.join(ROOT, "tests", "detect_emul.cpp")
    deps = [src, os.path.join(ROOT, "tests", "hip_emul", "hip", "hip_runtime.h")] + [
        os.path.join(CSRC, h) for h in ("detect.hip", "host_design.cpp", "types.hpp", "kernels.hpp", "common.hpp")] + [
        os.path.join(ROOT, "tests", "hip_emul", "fft_bfly.inc")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        text = open(os.path.join(CSRC, "detect.hip")).read()
        text, n = re.subn(r"extern __shared__ __attribute__\(\(aligned\(16\)\)\) unsigned char (\w+)\[\];",
                          r"unsigned char *\1 = hip_emul::dyn_lds();", text)
        assert n >= 2
        open(inc, "w").write(text)
        # (csrc/fft_bfly.inc is inline gfx950 assembly; tests/hip_emul/fft_bfly.inc restates it in C++)
        shutil.copy(os.path.join(ROOT, "tests", "hip_emul", "fft_bfly.inc"), os.path.join(out_dir, "fft_bfly.inc"))
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               "-I" + os.path.join(ROOT, "tests", "hip_emul"), "-I" + out_dir, "-I" + CSRC, "-o", so, src])
    L = C.CDLL(so)
    L.detect_emul_k1.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                 C.POINTER(C.c_uint), C.POINTER(Entry), C.c_int]
    L.detect_emul_scan.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_float, C.c_int, C.POINTER(Gone), C.c_int, C.POINTER(C.c_float)]
    return L


def _k1(L, x, fmt, n, frames, variant, pre=None, cap=4096):
    mag = np.zeros((frames, n), np.float32)
    counts = np.zeros(frames, np.uint32)
    entries = (Entry * (frames * cap))()
    rc = L.detect_emul_k1(x.ctypes.data_as(C.c_void_p), fmt, n, frames, variant, orc.fptr(mag),
                          orc.fptr(pre) if pre is not None else None, counts.ctypes.data_as(C.POINTER(C.c_uint)), entries, cap)
    return rc, mag, counts, entries


@pytest.mark.parametrize("fs", [2_000_000, 4_000_000, 10_000_000, 12_000_000])
def test_k1_kernels_match_the_oracle_fft(detect_emul, fs):
    """window . N-point FFT . fftshift . |.|^2 (burst_detect.c:679-687) by the kernel each frame size takes, and by the same
    kernel writing the band scan's candidate lists, for cf32 / ci16 / ci8 input: magnitudes bit for
    bit those of the oracle's pinned FFT; the lists hold exactly the bins above the prefilter level"""
    n = det_params(fs)["n"]
    frames = 24 if n > 2048 else 64
    iq, _ = siggen.standard_scene(fs, (frames + 4) * n // 32768 * 32768 + 32768, 3, seed=fs // 1_000_000)
    for fmt in (2, 1, 0):
        if fmt == 2:
            x, as_cf = iq, iq
        elif fmt == 1:
            x = siggen.to_ci16(iq)
            as_cf = ((x.astype(np.int16) >> 8).astype(np.float32) / np.float32(128.0)).view(np.complex64)
        else:
            x = siggen.to_ci8(iq)
            as_cf = (x.astype(np.float32) / np.float32(128.0)).view(np.complex64)
        ref_mag, _, _ = oracle_detect(np.ascontiguousarray(as_cf), fs)
        assert ref_mag.shape[0] >= frames
        ref_mag = ref_mag[:frames]
        # variant 0: the kernel the size takes (32 points per lane at 8192 / 16384 points, radix 16 at 4096, the radix-2 LDS
        # kernel at 2048); 1: the same with the candidate lists
        rc, mag, _, _ = _k1(detect_emul, np.ascontiguousarray(x), fmt, n, frames, 0)
        assert rc == 0
        assert np.array_equal(mag.view(np.uint32), ref_mag.view(np.uint32)), (fs, fmt)
        if n >= 4096:
            pre = (np.float32(0.5) * np.percentile(ref_mag, 99.0, axis=0)).astype(np.float32)
            for variant in (1,):
                rc, mag, counts, entries = _k1(detect_emul, np.ascontiguousarray(x), fmt, n, frames, variant, pre=pre)
                assert rc == 0 and np.array_equal(mag.view(np.uint32), ref_mag.view(np.uint32))
                for f in range(frames):
                    want = sorted((int(b), float(ref_mag[f, b])) for b in np.nonzero(ref_mag[f] > pre)[0])
                    got = sorted((entries[f * 4096 + i].bin, entries[f * 4096 + i].mag) for i in range(int(counts[f])))
                    assert got == want, (fs, fmt, variant, f)


@pytest.mark.parametrize("name", ["too_long", "squelch", "dc_and_edges", "strong_simultaneous"])
def test_dense_scan_kernel_matches_the_oracle(detect_emul, name):
    """detect_scan_kernel (one workgroup walking the frames: the exact fallback of every faster scan, squelch included)
    from the first frame of the stream, whole and in chunks"""
    fs, iq = scenes.ALL[name]()
    mag, ref, ref_sums = oracle_detect(iq, fs)
    p = det_params(fs)
    for cf in (1 << 20, 61):
        out = (Gone * 8192)()
        sums = np.zeros(p["n"], np.float32)
        rc = detect_emul.detect_emul_scan(orc.fptr(mag), mag.shape[0], p["n"], p["pre"], p["post"], p["width"], p["max_bursts"],
                                          p["max_len"], p["thr"], cf, out, 8192, orc.fptr(sums))
        assert rc >= 0, rc
        got = [(g.id, g.start, g.stop, g.last_active, g.center_bin, g.peak_rel, g.base_sum) for g in out[:rc]]
        assert got == ref, "chunk_frames %d" % cf
        assert np.array_equal(sums.view(np.uint32), ref_sums.view(np.uint32))


# ---- csrc/demod.hip on the same emulation: stage C against the reference's own vectors ----

@pytest.fixture(scope="module")
def demod_emul():
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libdemodemul.so")
    inc = os.path.join(out_dir, "demod_emul.inc")
    src = os.path.join(ROOT, "tests", "demod_emul.cpp")
    deps = [src, os.path.join(ROOT, "tests", "hip_emul", "hip", "hip_runtime.h")] + [
        os.path.join(CSRC, h) for h in ("demod.hip", "types.hpp", "kernels.hpp", "common.hpp")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        text = open(os.path.join(CSRC, "demod.hip")).read()
        text = re.sub(r"extern __shared__ __attribute__\(\(aligned\(16\)\)\) unsigned char (\w+)\[\];",
                      r"unsigned char *\1 = hip_emul::dyn_lds();", text)
        open(inc, "w").write(text)
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               "-I" + os.path.join(ROOT, "tests", "hip_emul"), "-I" + out_dir, "-I" + CSRC, "-o", so, src])
    L = C.CDLL(so)
    L.demod_emul_run.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_float,
                                 C.c_void_p, C.c_void_p]
    L.demod_emul_sizes.argtypes = [C.POINTER(C.c_int)] * 4
    return L


def test_stage_c_kernels_match_the_reference_vectors(demod_emul):
    """demod_seq_kernel + demod_par_kernel (qpsk_demod.c:393-535) on the 18 frames of tests/golden/ref_stage_c.npz, whose
    expected outputs were produced by the reference's own qpsk_demod.c: verdict, direction, confidence, symbol count, hard
    bits and level exactly, LLRs within 1e-4 of the reference's (the tolerance of the GPU test); the packed record
    (demod_pack_kernel) carries the same hard bits 8 per byte, MSB first"""
    import json
    z = np.load(os.path.join(ROOT, "tests", "golden", "ref_stage_c.npz"))
    samples, ns, dirs, exp = z["samples"], z["num_samples"], z["direction"], json.loads(str(z["expected"]))
    sz = [C.c_int() for _ in range(4)]
    demod_emul.demod_emul_sizes(*[C.byref(s) for s in sz])
    out_bytes, packed_bytes, max_frame, max_bits = [s.value for s in sz]
    n = len(exp)
    buf = np.zeros((n, 2 * max_frame), np.float32)
    for i in range(n):
        buf[i, :2 * ns[i]] = samples[i, :2 * ns[i]]
    out = np.zeros(n * out_bytes, np.uint8)
    packed = np.zeros(n * packed_bytes, np.uint8)
    nsi = np.ascontiguousarray(ns, np.int32)
    dri = np.ascontiguousarray(dirs, np.int32)
    rc = demod_emul.demod_emul_run(orc.fptr(buf), nsi.ctypes.data_as(C.POINTER(C.c_int)), dri.ctypes.data_as(C.POINTER(C.c_int)),
                                   n, 1, 10.0, out.ctypes.data_as(C.c_void_p), packed.ctypes.data_as(C.c_void_p))
    assert rc == 0
    checked = 0
    for i, e in enumerate(exp):
        rec = out[i * out_bytes:(i + 1) * out_bytes]
        ok, direction, confidence, n_sym = rec[:16].view(np.int32)
        level = rec[16:20].view(np.float32)[0]
        assert int(ok) == e["ok"], i
        if not e["ok"]:
            continue
        assert (int(direction), int(confidence), int(n_sym), 2 * int(n_sym)) == (e["direction"], e["confidence"], e["n_symbols"],
                                                                             e["n_bits"]), i
        assert np.float32(level) == np.float32(e["level"]), i
        nb = e["n_bits"]
        bits = rec[24:24 + nb]
        assert bytes(bits).hex() == e["bits"], i
        llr = rec[24 + max_bits:24 + max_bits + 4 * nb].view(np.float32)
        ref_llr = np.frombuffer(bytes.fromhex(e["llr"]), np.float32)
        assert np.max(np.abs(llr - ref_llr)) <= 1e-4, i
        prec = packed[i * packed_bytes:(i + 1) * packed_bytes]
        assert tuple(prec[:16].view(np.int32)) == (ok, direction, confidence, n_sym)
        want = np.packbits(bits)               # MSB first
        assert np.array_equal(prec[24:24 + len(want)], want), i
        checked += 1
    assert checked >= 6


def test_stage_c_kernels_long_frames_full_workgroups_and_positions_outside_the_window(demod_emul):
    """demod_seq_kernel's structure (round 6: a timing wavefront and a PLL wavefront per 32 frames, the samples through a
    ring of 128 per frame in LDS that is filled 40 samples per block of 4 symbols) against the oracle's stage C (pinned to
    the reference's qpsk_demod.c, tests/test_oracle_vs_ref.py) on what the reference's 18 vectors do not reach: 75 frames
    (two full workgroups and a ragged third), lengths from 40 samples to the 4440 of a simplex frame (the ring wraps 34
    times), and samples-per-symbol values for which the timing loop's position runs ahead of / falls behind the window
    (10.4: +1.6 samples a block, outside after ~18 blocks; 9.25: -3 a block; the oracle sizes its symbol buffers from
    the rounded value as the reference does, which bounds the choice, and has no 448-symbol cap: frames of at most 4000
    samples at 9.25) so that the interpolations read memory instead
    -- same host libm on both sides here, so everything is compared bit for bit, LLRs included; with and without the
    Gardner loop."""
    sz = [C.c_int() for _ in range(4)]
    demod_emul.demod_emul_sizes(*[C.byref(s_) for s_ in sz])
    out_bytes, packed_bytes, max_frame, max_bits = [s_.value for s_ in sz]
    n = 75
    oracle = orc.lib()
    checked = okd = 0
    for gardner, sps in ((1, 10.0), (1, 10.4), (1, 9.25), (0, 10.0)):
        buf, lens, dirs = scenes.stage_c_frames(n, sps, max_frame)
        nsi = np.ascontiguousarray(np.minimum(lens, 4000) if sps < 10 else lens, np.int32)
        out = np.zeros(n * out_bytes, np.uint8)
        rc = demod_emul.demod_emul_run(orc.fptr(buf), nsi.ctypes.data_as(C.POINTER(C.c_int)), dirs.ctypes.data_as(C.POINTER(C.c_int)),
                                       n, gardner, sps, out.ctypes.data_as(C.c_void_p), None)
        assert rc == 0
        for i in range(n):
            fr = orc.Frame()
            fr.samples_per_symbol = sps
            fr.sample_rate = 250000.0
            fr.direction = int(dirs[i])
            fr.num_samples = int(nsi[i])
            np.ctypeslib.as_array(fr.samples)[:] = buf[i]
            d = orc.Demod()
            r_o = oracle.orc_qpsk_demod(C.byref(fr), gardner, C.byref(d))
            rec = out[i * out_bytes:(i + 1) * out_bytes]
            ok, direction, confidence, n_sym = [int(v) for v in rec[:16].view(np.int32)]
            assert ok == r_o, (sps, i)
            checked += 1
            if not ok:
                continue
            okd += 1
            assert (direction, confidence, n_sym) == (d.direction, d.confidence, d.n_symbols), (sps, i)
            assert rec[16:20].view(np.uint32)[0] == np.float32(d.level).view(np.uint32), (sps, i)
            nb = 2 * n_sym
            assert bytes(rec[24:24 + nb]) == bytes(d.bits[:nb]), (sps, i)
            llr = rec[24 + max_bits:24 + max_bits + 4 * nb].view(np.uint32)
            assert np.array_equal(llr, np.array(d.llr[:nb], np.float32).view(np.uint32)), (sps, i)
    assert checked == 4 * n and okd >= 3 * n, okd


# ---- csrc/bitlayer.hip on the same emulation: frame_decode against the oracle (pinned to the reference's object code) ----

class DecodedOut(C.Structure):
    _fields_ = [("type", C.c_int32), ("sat_id", C.c_int32), ("beam_id", C.c_int32), ("pos_xyz", C.c_int32 * 3),
                ("n_pages", C.c_int32), ("page_tmsi", C.c_uint32 * 12), ("page_msc", C.c_int32 * 12), ("timeslot", C.c_int32),
                ("sv_blocking", C.c_int32), ("bc_type", C.c_int32), ("iri_time", C.c_uint32), ("bch_len", C.c_int32)]


@pytest.fixture(scope="module")
def bitlayer_emul():
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libbitlayeremul.so")
    inc = os.path.join(out_dir, "bitlayer_emul.inc")
    src = os.path.join(ROOT, "tests", "bitlayer_emul.cpp")
    deps = [src, os.path.join(ROOT, "tests", "hip_emul", "hip", "hip_runtime.h")] + [
        os.path.join(CSRC, h) for h in ("bitlayer.hip", "types.hpp", "kernels.hpp", "common.hpp")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        open(inc, "w").write(open(os.path.join(CSRC, "bitlayer.hip")).read())
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               "-I" + os.path.join(ROOT, "tests", "hip_emul"), "-I" + out_dir, "-I" + CSRC, "-o", so, src])
    L = C.CDLL(so)
    L.bitlayer_emul_frame_decode.argtypes = [C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_int, C.c_int,
                                             C.POINTER(DecodedOut)]
    L.bitlayer_emul_sizes.argtypes = [C.POINTER(C.c_int)] * 2
    return L


@pytest.mark.parametrize("seed", (0, 1))
def test_frame_decode_kernel_matches_the_oracle(bitlayer_emul, seed):
    """IRA / IBC frames with injected bit errors (hard-decision and Chase paths, uplink and downlink access codes, junk):
    every field the kernel produces equals the oracle's frame_decode (frame_decode.c:414-598)"""
    from test_oracle_bitlayer import decode_with, make_cases
    L = orc.lib()
    L.orc_frame_decode.restype = C.c_int
    a, b = C.c_int(), C.c_int()
    bitlayer_emul.bitlayer_emul_sizes(C.byref(a), C.byref(b))
    assert a.value == C.sizeof(DecodedOut)
    max_bits = b.value
    cases = [c for c in make_cases(seed, n=180) if len(c[0]) <= max_bits]
    for use_llr in (True, False):
        sel = [(bits, llr) for bits, llr in cases if (llr is not None) == use_llr]
        n = len(sel)
        hb = np.zeros((n, max_bits), np.uint8)
        sl = np.zeros((n, max_bits), np.float32)
        nb = np.zeros(n, np.int32)
        for k, (bits, llr) in enumerate(sel):
            hb[k, :len(bits)] = bits
            nb[k] = len(bits)
            if llr is not None:
                sl[k, :len(llr)] = llr
        out = (DecodedOut * n)()
        rc = bitlayer_emul.bitlayer_emul_frame_decode(hb.ctypes.data_as(C.POINTER(C.c_uint8)), orc.fptr(sl),
                                                      nb.ctypes.data_as(C.POINTER(C.c_int)), n, int(use_llr), out)
        assert rc == 0
        types = {0: 0, 1: 0, 2: 0}
        for k, (bits, llr) in enumerate(sel):
            r, o = decode_with(L.orc_frame_decode, bits, llr)
            g = out[k]
            assert (g.type != 0) == bool(r), k
            types[g.type] += 1
            if not r:
                continue
            for f in ("type", "sat_id", "beam_id", "n_pages", "timeslot", "sv_blocking", "bc_type", "iri_time"):
                assert getattr(g, f) == getattr(o, f), (k, f)
            assert tuple(g.pos_xyz) == tuple(o.pos_xyz), k
            assert tuple(g.page_tmsi)[:g.n_pages] == tuple(o.page_tmsi)[:o.n_pages], k
            assert tuple(g.page_msc)[:g.n_pages] == tuple(o.page_msc)[:o.n_pages], k
        assert types[1] and types[2] and types[0], types
