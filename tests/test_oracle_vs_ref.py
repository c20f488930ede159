"""Pin the oracle's restatement against the REFERENCE's own sources compiled in
place (oracle/_ref: simd_generic.c, simd_avx2.c, fir_filter.c, window_func.c,
qpsk_demod.c, rotator.h).  Bit-for-bit on every scalar kernel, every tap design,
the rotator recurrence, and whole stage C on synthetic frames.

burst_detect.c / burst_downmix.c need FFTW3 (absent, unpinned) and cannot be
built here; their non-FFT pieces are exactly the functions tested below."""
import ctypes as C

import numpy as np
import pytest

import orc
import siggen

F = C.POINTER(C.c_float)


def fp(a):
    return a.ctypes.data_as(F)


def crand(rng, n, scale=1.0):
    return ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * scale).astype(np.complex64)


def test_fir_ccf_and_dec(oracle, reflib):
    rng = np.random.default_rng(1)
    taps = rng.standard_normal(801).astype(np.float32)
    x = crand(rng, 801 + 40 * 300)
    for dec, n_out in ((1, 500), (8, 300), (40, 300), (48, 250)):
        a = np.zeros(n_out, np.complex64)
        b = np.zeros(n_out, np.complex64)
        if dec == 1:
            oracle.orc_fir_ccf(fp(taps), 801, fp(x), fp(a), n_out)
            reflib.generic_fir_ccf(fp(taps), 801, fp(x), fp(b), n_out)
        else:
            oracle.orc_fir_ccf_dec(fp(taps), 801, fp(x), fp(a), n_out, dec)
            reflib.generic_fir_ccf_dec(fp(taps), 801, fp(x), fp(b), n_out, dec)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_small_kernels(oracle, reflib):
    rng = np.random.default_rng(2)
    n = 8192
    x = crand(rng, n, 0.01)
    w = rng.random(n).astype(np.float32)
    a = np.zeros(n, np.complex64); b = np.zeros(n, np.complex64)
    oracle.orc_window_cf(fp(x), fp(w), fp(a), n)
    reflib.generic_window_cf(fp(x), fp(w), fp(b), n)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))

    ma = np.zeros(n, np.float32); mb = np.zeros(n, np.float32)
    oracle.orc_fftshift_mag(fp(x), fp(ma), n)
    reflib.generic_fftshift_mag(fp(x), fp(mb), n)
    assert np.array_equal(ma.view(np.uint32), mb.view(np.uint32))

    s1 = rng.random(n).astype(np.float32) * 3; s2 = s1.copy()
    old = rng.random(n).astype(np.float32); new = rng.random(n).astype(np.float32)
    oracle.orc_baseline_update(fp(s1), fp(old), fp(new), n)
    reflib.generic_baseline_update(fp(s2), fp(old), fp(new), n)
    assert np.array_equal(s1.view(np.uint32), s2.view(np.uint32))

    base = rng.random(n).astype(np.float32); base[::17] = 0; base[5] = -1
    ra = np.zeros(n, np.float32); rb = np.zeros(n, np.float32)
    oracle.orc_relative_mag(fp(new), fp(base), fp(ra), n)
    reflib.generic_relative_mag(fp(new), fp(base), fp(rb), n)
    assert np.array_equal(ra.view(np.uint32), rb.view(np.uint32))

    i8 = rng.integers(-128, 128, 2 * n, dtype=np.int8)
    ca = np.zeros(n, np.complex64); cb = np.zeros(n, np.complex64)
    oracle.orc_convert_i8_cf(i8.ctypes.data_as(C.c_void_p), fp(ca), C.c_size_t(n))
    reflib.generic_convert_i8_cf(i8.ctypes.data_as(C.c_void_p), fp(cb), C.c_size_t(n))
    assert np.array_equal(ca.view(np.uint32), cb.view(np.uint32))

    oracle.orc_mag_squared(fp(x), fp(ma), n)
    reflib.generic_mag_squared(fp(x), fp(mb), n)
    assert np.array_equal(ma.view(np.uint32), mb.view(np.uint32))
    assert oracle.orc_max_float(fp(ma), n) == reflib.generic_max_float(fp(mb), n)

    w256 = rng.random(256).astype(np.float32)
    qa = np.zeros(256, np.complex64); qb = np.zeros(256, np.complex64)
    oracle.orc_csquare_window(fp(x), fp(w256), fp(qa), 256)
    reflib.generic_csquare_window(fp(x), fp(w256), fp(qb), 256)
    assert np.array_equal(qa.view(np.uint32), qb.view(np.uint32))

    t = rng.random(20).astype(np.float32); r = rng.random(1000).astype(np.float32)
    fa = np.zeros(900, np.float32); fb = np.zeros(900, np.float32)
    oracle.orc_fir_fff(fp(t), 20, fp(r), fp(fa), 900)
    reflib.generic_fir_fff(fp(t), 20, fp(r), fp(fb), 900)
    assert np.array_equal(fa.view(np.uint32), fb.view(np.uint32))


def test_windows_and_taps(oracle, reflib):
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    for n in (256, 2048, 8192, 16384):
        a = np.zeros(n, np.float32); b = np.zeros(n, np.float32)
        oracle.orc_blackman_window(fp(a), n)
        reflib.blackman_window(fp(b), n)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))

    def cmp(ours, theirs_ptr, nt):
        theirs = np.ctypeslib.as_array(theirs_ptr, (nt.value,)).copy()
        libc.free(C.cast(theirs_ptr, C.c_void_p))
        assert len(ours) == nt.value
        assert np.array_equal(ours.view(np.uint32), theirs.view(np.uint32))

    out = np.zeros(2048, np.float32)
    nt = C.c_int()
    n = oracle.orc_lpf_taps(fp(out), 2048, C.c_float(1.0), C.c_float(1e7), C.c_float(1e5), C.c_float(5e4))
    assert n == 801
    cmp(out[:n].copy(), reflib.lpf_taps(C.byref(nt), 1.0, 1e7, 1e5, 5e4), nt)
    n = oracle.orc_lpf_taps(fp(out), 2048, C.c_float(1.0), C.c_float(250000.0), C.c_float(20000.0), C.c_float(40000.0))
    assert n == 25
    cmp(out[:n].copy(), reflib.lpf_taps(C.byref(nt), 1.0, 250000.0, 20000.0, 40000.0), nt)
    n = oracle.orc_rrc_taps(fp(out), 2048, C.c_float(1.0), C.c_float(250000.0), C.c_float(25000.0), C.c_float(0.4), 51)
    cmp(out[:n].copy(), reflib.rrc_taps(C.byref(nt), 1.0, 250000.0, 25000.0, 0.4, 51), nt)
    n = oracle.orc_rc_taps(fp(out), 2048, C.c_float(250000.0), C.c_float(25000.0), C.c_float(0.4), 51)
    cmp(out[:n].copy(), reflib.rc_taps(C.byref(nt), 250000.0, 25000.0, 0.4, 51), nt)
    n = oracle.orc_box_taps(fp(out), 2048, 20)
    cmp(out[:n].copy(), reflib.box_taps(C.byref(nt), 20), nt)


@pytest.mark.parametrize("theta", [-2.3, -0.01, 0.4, 1.234])
def test_rotator_recurrence(oracle, reflib, theta):
    rng = np.random.default_rng(3)
    n = 300000
    x = crand(rng, n, 0.05)
    inc = np.array([np.cos(np.float32(theta)), np.sin(np.float32(theta))], np.float32)
    pa = np.array([1, 0], np.float32); pb = pa.copy()
    a = np.zeros(n, np.complex64); b = np.zeros(n, np.complex64)
    oracle.orc_rotator_rotate_n(fp(pa), fp(inc), fp(a), fp(x), n)
    reflib.ref_rotator_rotate_n(fp(pb), fp(inc), fp(b), fp(x), n)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32))


def _oracle_frames():
    """Downmixed frames from the oracle's own stages A+B on a seeded scene (stage C input)."""
    fs = 2_000_000
    iq, _ = siggen.standard_scene(fs, int(2.4 * fs), 8, seed=11, uplink_every=4)
    res = orc.run_stream(iq, fs)
    return [f for f in res.frames if f.drop_reason == 0]


def _ref_demod(reflib, f):
    n = f.num_samples
    s = np.ctypeslib.as_array(f.samples)[:2 * n].copy()
    d_out = C.c_int(); conf = C.c_int(); lvl = C.c_float(); ns = C.c_int(); npay = C.c_int(); nb = C.c_int()
    bits = (C.c_uint8 * 1024)(); llr = (C.c_float * 1024)(); cfo = C.c_double()
    r = reflib.ref_qpsk_demod(fp(s), n, f.samples_per_symbol, f.direction, f.center_frequency,
                              f.id, f.timestamp, f.magnitude, f.noise,
                              C.byref(d_out), C.byref(conf), C.byref(lvl), C.byref(ns), C.byref(npay),
                              C.byref(nb), bits, llr, C.byref(cfo))
    return r, d_out.value, conf.value, lvl.value, ns.value, npay.value, nb.value, bytes(bits[:nb.value]), \
        np.array(llr[:nb.value], np.float32), cfo.value


@pytest.mark.parametrize("gardner", [1, 0])
def test_stage_c_matches_reference(oracle, reflib, gardner):
    frames = _oracle_frames()
    assert len(frames) >= 6
    reflib.ref_set_use_gardner(gardner)
    n_ok = 0
    rng = np.random.default_rng(9)
    variants = []
    for f in frames:
        variants.append(f)
        # a noisy copy (exercises soft UW rescue / failures) and a truncated copy
        g = orc.Frame.from_buffer_copy(f)
        s = np.ctypeslib.as_array(g.samples)
        s[:2 * g.num_samples] += (rng.standard_normal(2 * g.num_samples) * 0.012).astype(np.float32)
        variants.append(g)
        h = orc.Frame.from_buffer_copy(f)
        h.num_samples = 1400
        variants.append(h)
    for f in variants:
        d = orc.Demod()
        r_o = oracle.orc_qpsk_demod(C.byref(f), gardner, C.byref(d))
        r = _ref_demod(reflib, f)
        assert r_o == r[0]
        if not r_o:
            continue
        n_ok += 1
        assert (d.direction, d.confidence, d.n_symbols, d.n_payload_symbols, d.n_bits) == (r[1], r[2], r[4], r[5], r[6])
        assert np.float32(d.level).view(np.uint32) == np.float32(r[3]).view(np.uint32)
        assert bytes(d.bits[:d.n_bits]) == r[7]
        assert np.array_equal(np.array(d.llr[:d.n_bits], np.float32).view(np.uint32), r[8].view(np.uint32))
        assert d.center_frequency == r[9]
    reflib.ref_set_use_gardner(1)
    assert n_ok >= 8


def test_fir_ccf_dec_avx2_order(oracle, reflib):
    """orc_fir_ccf_dec_avx2 == the reference's avx2_fir_ccf_dec (simd_avx2.c:62-108, compiled in place with the
    reference's own flags: -std=c99 -O3 -mavx2 -mfma), bit for bit: four fused accumulators, horizontal sum, scalar tail.
    This is the decimating FIR the reference runs by default on x86 (simd_init) and the order the product's decimator
    follows with option fir_order 1."""
    rng = np.random.default_rng(11)
    for ntaps in (801, 800, 803, 25):
        taps = np.zeros(ntaps + 8, np.float32)          # (the AVX2 kernel loads taps four at a time)
        taps[:ntaps] = rng.standard_normal(ntaps).astype(np.float32)
        for dec, n_out in ((8, 300), (40, 300), (48, 250), (1, 100)):
            x = crand(rng, ntaps + dec * n_out + 8, scale=float(rng.uniform(0.01, 10.0)))
            a = np.zeros(n_out, np.complex64)
            b = np.zeros(n_out, np.complex64)
            oracle.orc_fir_ccf_dec_avx2(fp(taps), ntaps, fp(x), fp(a), n_out, dec)
            reflib.avx2_fir_ccf_dec(fp(taps), ntaps, fp(x), fp(b), n_out, dec)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (ntaps, dec)
            c = np.zeros(n_out, np.complex64)
            oracle.orc_fir_ccf_dec(fp(taps), ntaps, fp(x), fp(c), n_out, dec)
            assert not np.array_equal(a.view(np.uint32), c.view(np.uint32)) or ntaps < 100      # the two orders differ in rounding


def test_avx2_forms_of_the_small_kernels(oracle, reflib):
    """The other dispatched kernels whose AVX2 form rounds differently (simd_avx2.c): fir_ccf (:28-55: outputs four at a
    time with fused multiply-adds, the last n % 4 generic), fir_fff (:115-138: eight at a time, the last n % 8 generic),
    fftshift_mag (:177-221) and mag_squared (:304-323): fma(re, re, im*im).  Each oracle restatement equals the
    reference's kernel bit for bit at lengths with and without a tail; the kernels whose two forms are the same
    operations agree with EACH OTHER bit for bit (window_cf, csquare_window, baseline_update, relative_mag, max_float)."""
    rng = np.random.default_rng(23)
    bits = lambda v: v.view(np.uint32)
    for ntaps in (25, 51, 20, 7):
        taps = rng.standard_normal(ntaps).astype(np.float32)
        for n in (0, 1, 3, 4, 5, 64, 1023, 4002):
            x = crand(rng, n + ntaps + 8, scale=float(rng.uniform(0.01, 10.0)))
            a = np.zeros(max(n, 1), np.complex64); b = np.zeros(max(n, 1), np.complex64); c = np.zeros(max(n, 1), np.complex64)
            oracle.orc_fir_ccf_avx2(fp(taps), ntaps, fp(x), fp(a), n)
            reflib.avx2_fir_ccf(fp(taps), ntaps, fp(x), fp(b), n)
            assert np.array_equal(bits(a), bits(b)), ("fir_ccf", ntaps, n)
            oracle.orc_fir_ccf(fp(taps), ntaps, fp(x), fp(c), n)
            if n % 4:                                   # the tail is the generic form
                assert np.array_equal(bits(a)[-2 * (n % 4):], bits(c)[-2 * (n % 4):])
            xr = rng.standard_normal(n + ntaps + 8).astype(np.float32)
            ar = np.zeros(max(n, 1), np.float32); br = np.zeros(max(n, 1), np.float32)
            oracle.orc_fir_fff_avx2(fp(taps), ntaps, fp(xr), fp(ar), n)
            reflib.avx2_fir_fff(fp(taps), ntaps, fp(xr), fp(br), n)
            assert np.array_equal(bits(ar), bits(br)), ("fir_fff", ntaps, n)
    differs = 0
    for n in (0, 1, 3, 4, 7, 8, 100, 4099):
        x = crand(rng, n + 8, scale=3.0)
        a = np.zeros(max(n, 1), np.float32); b = np.zeros(max(n, 1), np.float32); c = np.zeros(max(n, 1), np.float32)
        oracle.orc_mag_squared_avx2(fp(x), fp(a), n)
        reflib.avx2_mag_squared(fp(x), fp(b), n)
        oracle.orc_mag_squared(fp(x), fp(c), n)
        assert np.array_equal(bits(a), bits(b)), ("mag_squared", n)
        differs += int(not np.array_equal(bits(a), bits(c)))
    assert differs >= 2                                  # fused and separately rounded sums differ in the last bit
    for n in (4, 8, 12, 4096, 8192, 16384):
        x = crand(rng, n, scale=2.0)
        a = np.zeros(n, np.float32); b = np.zeros(n, np.float32)
        oracle.orc_fftshift_mag_avx2(fp(x), fp(a), n)
        reflib.avx2_fftshift_mag(fp(x), fp(b), n)
        assert np.array_equal(bits(a), bits(b)), ("fftshift_mag", n)
    # the same operations in both forms
    n = 1027
    x = crand(rng, n + 8, scale=2.0)
    w = rng.uniform(0.0, 1.0, n + 8).astype(np.float32)
    for name in ("window_cf", "csquare_window"):
        a = np.zeros(n, np.complex64); b = np.zeros(n, np.complex64)
        getattr(reflib, "generic_" + name)(fp(x), fp(w), fp(a), n)
        getattr(reflib, "avx2_" + name)(fp(x), fp(w), fp(b), n)
        assert np.array_equal(bits(a), bits(b)), name
    s0 = rng.uniform(0, 100, n).astype(np.float32); old = rng.uniform(0, 1, n).astype(np.float32); new = rng.uniform(0, 1, n).astype(np.float32)
    sa, sb = s0.copy(), s0.copy()
    reflib.generic_baseline_update(fp(sa), fp(old), fp(new), n)
    reflib.avx2_baseline_update(fp(sb), fp(old), fp(new), n)
    assert np.array_equal(bits(sa), bits(sb))
    base = s0.copy(); base[::7] = 0.0
    ra = np.zeros(n, np.float32); rb = np.zeros(n, np.float32)
    reflib.generic_relative_mag(fp(new), fp(base), fp(ra), n)
    reflib.avx2_relative_mag(fp(new), fp(base), fp(rb), n)
    assert np.array_equal(bits(ra), bits(rb))
    reflib.generic_max_float.restype = C.c_float
    reflib.avx2_max_float.restype = C.c_float
    assert reflib.generic_max_float(fp(new), n) == reflib.avx2_max_float(fp(new), n)


def test_whole_stream_in_the_avx2_forms_differs_only_in_rounding(oracle):
    """orc_run_stream with the AVX2 forms throughout (orc_set_fir_order 1: fftshift_mag, fir_ccf_dec, fir_ccf, mag_squared,
    fir_fff) finds the same bursts and frames as with the generic forms, with the frame samples a rounding apart."""
    import orc
    import siggen
    fs = 2_000_000
    n = int(0.6 * fs) // 32768 * 32768
    iq, _ = siggen.standard_scene(fs, n, 6, seed=2)
    try:
        orc.set_fir_order(0)
        r0 = orc.run_stream(iq, fs)
        orc.set_fir_order(1)
        r1 = orc.run_stream(iq, fs)
    finally:
        orc.set_fir_order(1)
    assert len(r0.bursts) == len(r1.bursts) >= 4 and len(r0.frames) == len(r1.frames) >= 3
    assert [(b.start, b.stop, b.center_bin) for b in r0.bursts] == [(b.start, b.stop, b.center_bin) for b in r1.bursts]
    same = 0
    for f0, f1 in zip(r0.frames, r1.frames):
        a, b = np.asarray(f0.samples), np.asarray(f1.samples)
        assert a.shape == b.shape and np.allclose(a, b, rtol=1e-3, atol=1e-4)
        same += int(np.array_equal(a, b))
    assert same < len(r0.frames)                          # ... but not the same bits


def test_avx2_variant_differs_only_in_float_rounding(reflib):
    """The reference's AVX2 path is NOT bit-identical to its scalar path: the oracle restates both forms of
    the decimating FIR (orc_fir_ccf_dec / orc_fir_ccf_dec_avx2) and pins each to its reference kernel."""
    rng = np.random.default_rng(4)
    taps = rng.standard_normal(801).astype(np.float32)
    pad = np.zeros(808, np.float32); pad[:801] = taps
    x = crand(rng, 801 + 40 * 64)
    a = np.zeros(64, np.complex64); b = np.zeros(64, np.complex64)
    reflib.generic_fir_ccf_dec(fp(pad), 801, fp(x), fp(a), 64, 40)
    reflib.avx2_fir_ccf_dec(fp(pad), 801, fp(x), fp(b), 64, 40)
    assert np.allclose(a, b, rtol=1e-4, atol=1e-4)


_RAW_CHILD = r'''
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import orc, irdm
R = orc.ref()
R.ref_frame_output_line.argtypes = [C.c_uint64, C.c_uint64, C.c_double, C.c_float, C.c_float, C.c_int, C.c_float,
                                    C.c_int, C.c_int, C.POINTER(C.c_uint8), C.c_char_p, C.c_int]
file_info = sys.argv[3]
fi_keep = C.create_string_buffer(file_info.encode())      # frame_output_init keeps the POINTER (frame_output.c:101-104)
R.ref_frame_output_init(fi_keep if file_info else None)
rng = np.random.default_rng(int(sys.argv[4]))
out = []
t = 1700000000 * 10**9 + int(rng.integers(0, 10**9))
demods = []
for k in range(40):
    d = irdm.Demod()
    d.id = int(rng.integers(0, 10**7)) * 10
    t += int(rng.integers(1, 10**9))
    d.timestamp = t
    d.center_frequency = float(rng.uniform(1.616e9, 1.6265e9))
    d.magnitude = float(np.float32(rng.uniform(-5, 60)))
    d.noise = float(np.float32(rng.uniform(-140, -80)))
    d.confidence = int(rng.integers(0, 101))
    d.level = float(np.float32(rng.uniform(0, 2)))
    d.n_symbols = int(rng.integers(20, 445))
    d.n_payload_symbols = d.n_symbols - 12
    d.n_bits = 2 * d.n_symbols
    d.ok = 1
    bits = rng.integers(0, 2, d.n_bits).astype(np.uint8)
    for i, b in enumerate(bits):
        d.bits[i] = int(b)
    demods.append(d)
    buf = C.create_string_buffer(4096)
    n = R.ref_frame_output_line(d.id, d.timestamp, d.center_frequency, d.magnitude, d.noise, d.confidence, d.level,
                                d.n_payload_symbols, d.n_bits, bits.ctypes.data_as(C.POINTER(C.c_uint8)), buf, 4096)
    assert n > 0
    out.append(buf.value.decode())
ours = irdm.format_raw(demods, file_info)                 # product (host C in libirdm_hip.so)
batch = irdm.format_raw_batch(demods, file_info)
oracle = []
t0 = C.c_uint64(0)
for d in demods:
    o = orc.Demod.from_buffer_copy(bytes(d))
    b = C.create_string_buffer(4096)
    orc.lib().orc_format_raw(C.byref(o), file_info.encode(), C.byref(t0), b, 4096)
    oracle.append(b.value.decode())
print(json.dumps(dict(ref=out, ours=ours, oracle=oracle, batch=batch)))
'''


@pytest.mark.parametrize("file_info,seed", [("golden", 1), ("", 2)])
def test_raw_line_printer_matches_frame_output_c(reflib, file_info, seed):
    """frame_output_print (frame_output.c:160-199, the reference's object code, stdout captured) vs the oracle's
    orc_format_raw vs the product's irdm_format_raw / _batch: byte-identical lines for random records, including the
    automatic "i-<t0>-t1" file info.  Runs in a child process: frame_output.c latches t0 and file_info in statics."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    pkg = os.path.join(os.path.dirname(here), "iridium-sniffer_amd")
    r = subprocess.run([sys.executable, "-c", _RAW_CHILD, here, pkg, file_info, str(seed)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert len(d["ref"]) == 40
    assert d["ref"] == d["oracle"] == d["ours"]
    assert d["batch"] == "".join(d["ref"])
    assert d["ref"][0].startswith("RAW: golden " if file_info else "RAW: i-17000000")
