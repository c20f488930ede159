#!/usr/bin/env python3
"""Authoring-container aid (needs oracle/_ref, i.e. the reference tree): extended differential run of the oracle's stage C
(orc_qpsk_demod) against the reference's qpsk_demod.c object code on many seeded scenes, noise levels and truncations --
the same comparison as tests/test_oracle_vs_ref.py::test_stage_c_matches_reference, just wider.  Prints a summary; exits
non-zero on the first mismatch.  Usage: python tools/fuzz_stage_c.py [n_seeds]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "iridium-sniffer_amd"))
import orc          # noqa: E402
import siggen       # noqa: E402
import test_oracle_vs_ref as T   # noqa: E402


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    oracle = orc.lib()
    reflib = orc.ref()
    if reflib is None:
        raise SystemExit("oracle/_ref is not built (needs /root/reference): make -C oracle ref")
    total = ok = 0
    for seed in range(100, 100 + n_seeds):
        rng = np.random.default_rng(seed)
        fs = 2_000_000
        amp = float(rng.choice([0.008, 0.015, 0.03, 0.06]))
        iq, _ = siggen.standard_scene(fs, int(2.0 * fs) // 32768 * 32768, int(rng.integers(4, 10)), seed=seed,
                                      uplink_every=int(rng.integers(2, 5)), amp=amp)
        res = orc.run_stream(iq, fs)
        frames = [f for f in res.frames if f.drop_reason == 0]
        for gardner in (1, 0):
            reflib.ref_set_use_gardner(gardner)
            for f in frames:
                variants = [f]
                base = np.ctypeslib.as_array(f.samples)[:2 * f.num_samples]
                rms = float(np.sqrt(np.mean(base.astype(np.float64) ** 2)))
                for rel in (0.05, 0.15, 0.3, 0.5):            # noise relative to the frame's rms: clean .. soft-rescue .. rejected
                    g = orc.Frame.from_buffer_copy(f)
                    s = np.ctypeslib.as_array(g.samples)
                    s[:2 * g.num_samples] += (rng.standard_normal(2 * g.num_samples) * rel * rms).astype(np.float32)
                    variants.append(g)
                h = orc.Frame.from_buffer_copy(f)
                h.num_samples = int(rng.integers(300, max(301, f.num_samples)))
                variants.append(h)
                for v in variants:
                    d = orc.Demod()
                    r_o = oracle.orc_qpsk_demod(C.byref(v), gardner, C.byref(d))
                    r = T._ref_demod(reflib, v)
                    total += 1
                    assert r_o == r[0], (seed, v.id, r_o, r[0])
                    if not r_o:
                        continue
                    ok += 1
                    assert (d.direction, d.confidence, d.n_symbols, d.n_payload_symbols, d.n_bits) == (r[1], r[2], r[4], r[5], r[6]), (seed, v.id)
                    assert np.float32(d.level).view(np.uint32) == np.float32(r[3]).view(np.uint32), (seed, v.id)
                    assert bytes(d.bits[:d.n_bits]) == r[7], (seed, v.id)
                    assert np.array_equal(np.array(d.llr[:d.n_bits], np.float32).view(np.uint32), r[8].view(np.uint32)), (seed, v.id)
                    assert d.center_frequency == r[9], (seed, v.id)
        reflib.ref_set_use_gardner(1)
        print("seed %d: amp %.3f, %d frames; cumulative %d comparisons, %d demodulated" % (seed, amp, len(frames), total, ok), flush=True)
    print("stage C: oracle == reference object code on %d inputs (%d demodulated, %d rejected by both)" % (total, ok, total - ok))


if __name__ == "__main__":
    main()
