#!/usr/bin/env python3
"""Generates tests/golden/*.npz.  Run in the authoring container only.

Two kinds of vectors:
  ref_*   produced by the REFERENCE's own sources compiled in place (oracle/_ref, needs
          /root/reference): stage C (qpsk_demod.c) inputs/outputs, tap designs, window,
          rotator recurrence.  These pin the oracle AND the HIP path to the real reference.
  e2e_*   produced by the CPU oracle on seeded synthetic scenes (the generator recipe is
          stored, not the IQ): burst records, frame probes, bits, RAW lines.  The FFT-bearing
          stages cannot be run from the reference here (FFTW3 absent), see DESIGN.md.
Data only: no reference source text is stored.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "iridium-sniffer_amd"))
import orc      # noqa: E402
import siggen   # noqa: E402

F = C.POINTER(C.c_float)


def fp(a):
    return a.ctypes.data_as(F)


def ref_stage_c(R):
    """Frames come from the oracle's stages A+B on a seeded scene; outputs from qpsk_demod.c."""
    fs = 2_000_000
    iq, _ = siggen.standard_scene(fs, int(2.4 * fs), 8, seed=11, uplink_every=4)
    res = orc.run_stream(iq, fs)
    frames = [f for f in res.frames if f.drop_reason == 0][:6]
    rng = np.random.default_rng(9)
    ins, dirs, outs = [], [], []
    for k, f in enumerate(frames):
        s = np.ctypeslib.as_array(f.samples)[:2 * f.num_samples].copy()
        variants = [s]
        noisy = s + (rng.standard_normal(len(s)) * 0.012).astype(np.float32)
        variants.append(noisy.astype(np.float32))
        variants.append(s[:2 * 1400].copy())
        for v in variants:
            n = len(v) // 2
            d_out = C.c_int(); conf = C.c_int(); lvl = C.c_float(); ns = C.c_int(); npay = C.c_int(); nb = C.c_int()
            bits = (C.c_uint8 * 1024)(); llr = (C.c_float * 1024)(); cfo = C.c_double()
            r = R.ref_qpsk_demod(fp(v), n, f.samples_per_symbol, f.direction, f.center_frequency, f.id,
                                 f.timestamp, f.magnitude, f.noise, C.byref(d_out), C.byref(conf), C.byref(lvl),
                                 C.byref(ns), C.byref(npay), C.byref(nb), bits, llr, C.byref(cfo))
            ins.append(v)
            dirs.append(f.direction)
            outs.append(dict(ok=r, direction=d_out.value, confidence=conf.value, level=float(np.float32(lvl.value)),
                             n_symbols=ns.value, n_bits=nb.value if r else 0,
                             bits=bytes(bits[:nb.value]).hex() if r else "",
                             llr=np.array(llr[:nb.value], np.float32).tobytes().hex() if r else "",
                             dfreq=(cfo.value - f.center_frequency) if r else 0.0))
    width = max(len(v) for v in ins)
    mat = np.zeros((len(ins), width), np.float32)
    lens = np.zeros(len(ins), np.int32)
    for i, v in enumerate(ins):
        mat[i, :len(v)] = v
        lens[i] = len(v) // 2
    np.savez_compressed(os.path.join(HERE, "ref_stage_c.npz"), samples=mat, num_samples=lens,
                        direction=np.array(dirs, np.int32), expected=json.dumps(outs))
    print("ref_stage_c:", len(ins), "frames,", sum(o["ok"] for o in outs), "accepted")


def ref_designs(R):
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    out = {}
    nt = C.c_int()

    def grab(ptr):
        a = np.ctypeslib.as_array(ptr, (nt.value,)).copy()
        libc.free(C.cast(ptr, C.c_void_p))
        return a
    out["lpf_in"] = grab(R.lpf_taps(C.byref(nt), 1.0, 1e7, 1e5, 5e4))
    out["lpf_noise"] = grab(R.lpf_taps(C.byref(nt), 1.0, 250000.0, 20000.0, 40000.0))
    out["rrc"] = grab(R.rrc_taps(C.byref(nt), 1.0, 250000.0, 25000.0, 0.4, 51))
    out["rc"] = grab(R.rc_taps(C.byref(nt), 250000.0, 25000.0, 0.4, 51))
    out["box"] = grab(R.box_taps(C.byref(nt), 20))
    for n in (256, 2048, 8192, 16384):
        w = np.zeros(n, np.float32)
        R.blackman_window(fp(w), n)
        out["blackman_%d" % n] = w
    # rotator recurrence (rotator.h:36-46): phases after k steps via rotating a vector of ones
    for name, theta in (("rot_a", -2.3), ("rot_b", 0.4)):
        n = 100000
        x = np.ones(n, np.complex64)
        y = np.zeros(n, np.complex64)
        ph = np.array([1, 0], np.float32)
        inc = np.array([np.cos(np.float32(theta)), np.sin(np.float32(theta))], np.float32)
        R.ref_rotator_rotate_n(fp(ph), fp(inc), fp(y), fp(x), n)
        out[name + "_incr"] = inc
        out[name + "_every1000"] = y[::1000].copy()
        out[name + "_final"] = ph
    np.savez_compressed(os.path.join(HERE, "ref_designs.npz"), **out)
    print("ref_designs:", sorted(out))


SCENES = {
    # 2 MHz ci16 plumbing case (3 bursts incl. the documented PRBS15 frame)
    "cfg1_2mhz_ci16": dict(fs=2_000_000, secs=1.2, fmt=1, seed=1, kind="known_answer"),
    "cfg1_2mhz_cf32": dict(fs=2_000_000, secs=1.6, fmt=2, seed=2, kind="standard", n_bursts=6, uplink_every=3),
    "cfg3_10mhz_cf32": dict(fs=10_000_000, secs=0.8, fmt=2, seed=3, kind="standard", n_bursts=5, uplink_every=0),
    "cfg4_12mhz_cf32": dict(fs=12_000_000, secs=1.0, fmt=2, seed=4, kind="standard", n_bursts=4, uplink_every=0),
}


def build_scene(name):
    sc = SCENES[name]
    fs = sc["fs"]
    n = int(sc["secs"] * fs) // 32768 * 32768
    if sc["kind"] == "known_answer":
        q = siggen.bits_to_quadrants(siggen.KNOWN_ANSWER_BITS)
        rng = np.random.default_rng(sc["seed"])
        bursts = [dict(start=520 * 2048 + 1000, freq_hz=siggen.channel_freq(5), quads=[0] * 16 + q, amp=0.05),
                  dict(start=520 * 2048 + 300000, freq_hz=siggen.channel_freq(-7),
                       payload=rng.integers(0, 4, 150).tolist()),
                  dict(start=520 * 2048 + 700000, freq_hz=siggen.channel_freq(11),
                       payload=rng.integers(0, 4, 179).tolist())]
        iq, _ = siggen.make_stream(fs, n, bursts, seed=sc["seed"])
    else:
        iq, _ = siggen.standard_scene(fs, n, sc["n_bursts"], seed=sc["seed"], uplink_every=sc["uplink_every"])
    if sc["fmt"] == 1:
        return fs, siggen.to_ci16(iq), 1
    return fs, iq, 2


def e2e(order=0):
    """order 0: the dispatched kernels in simd_generic.c's operation order (e2e_scenes.json, recorded before the oracle
    knew the other one and kept byte for byte); order 1: in simd_avx2.c's, the product's default (e2e_scenes_avx2.json)."""
    out = {}
    orc.set_fir_order(order)
    for name in SCENES:
        fs, iq, fmt = build_scene(name)
        res = orc.run_stream(iq, fs, fmt=fmt)
        rec = dict(
            n_tagged=int(res.n_tagged),
            iq_crc=int(np.frombuffer(np.ascontiguousarray(iq).tobytes(), np.uint32).astype(np.uint64).sum() & 0xFFFFFFFF),
            bursts=[dict(id=b.id, start=b.start, stop=b.stop, last_active=b.last_active, center_bin=b.center_bin,
                         num_samples=b.num_samples, magnitude=float(np.float32(b.magnitude)),
                         noise=float(np.float32(b.noise))) for b in res.bursts],
            frames=[dict(id=f.id, drop=f.drop_reason, dec_len=f.dec_len, start=f.start,
                         center_offset=float(np.float32(f.center_offset)), uw_start=f.uw_start_idx,
                         direction=f.direction, num_samples=f.num_samples) for f in res.frames],
            raw=res.raw_lines("golden"))
        out[name] = rec
        print(name, "bursts", len(rec["bursts"]), "raw lines", len(rec["raw"]))
    orc.set_fir_order(1)
    json.dump(out, open(os.path.join(HERE, "e2e_scenes_avx2.json" if order else "e2e_scenes.json"), "w"), indent=1)


def ref_bitlayer(R):
    """Post-demod bit layer: seeded encoded / corrupted / truncated frames (tests/test_oracle_bitlayer.py's generators)
    and what the REFERENCE's frame_decode.c / ida_decode.c return for them.  Stored: the input bits, the LLRs (or their
    absence), the direction, and every output field as integers (doubles as their bit patterns, texts as bytes)."""
    import test_oracle_bitlayer as T
    R.ref_frame_decode.restype = C.c_int
    R.ref_ida_decode.restype = C.c_int
    fd, ida = [], []
    for bits, llr in T.make_cases(4242, n=90):
        r, d = T.decode_with(R.ref_frame_decode, bits, llr)
        fd.append(dict(bits=bytes(bytearray(bits)).hex(), llr=None if llr is None else np.asarray(llr, np.float32).tobytes().hex(),
                       ret=int(r), out=[int(v) if not isinstance(v, tuple) else [int(x) for x in v] for v in T.as_tuple(d)]))
    for bits, llr, direction in T.make_ida_cases(4343, n=90):
        r, d = T.ida_decode_with(R.ref_ida_decode, bits, llr, direction)
        t = T.ida_tuple(d)
        ida.append(dict(bits=bytes(bytearray(bits)).hex(), llr=None if llr is None else np.asarray(llr, np.float32).tobytes().hex(),
                        direction=int(direction), ret=int(r),
                        out=[v.hex() if isinstance(v, (bytes, bytearray)) else
                             ([int(x) for x in v] if isinstance(v, (tuple, list)) else int(v)) for v in t]))
    import gzip
    with gzip.GzipFile(os.path.join(HERE, "ref_bitlayer.json.gz"), "wb", mtime=0) as f:
        f.write(json.dumps(dict(frame_decode=fd, ida_decode=ida)).encode())
    print("ref_bitlayer:", len(fd), "frame_decode cases,", len(ida), "ida_decode cases;",
          sum(c["ret"] for c in fd), "+", sum(c["ret"] for c in ida), "decoded")


if __name__ == "__main__":
    R = orc.ref()
    if R is None:
        raise SystemExit("needs oracle/_ref (the reference tree)")
    if len(sys.argv) > 1 and sys.argv[1] == "bitlayer":      # only the newer fixture; the others stay byte-identical
        ref_bitlayer(R)
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "e2e_avx2":      # likewise: the end-to-end scenes in the AVX2 kernels' order
        e2e(1)
        raise SystemExit(0)
    ref_stage_c(R)
    ref_designs(R)
    e2e(0)
    e2e(1)
    ref_bitlayer(R)
