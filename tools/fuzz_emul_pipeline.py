#!/usr/bin/env python3
"""Differential run of the WHOLE product on the CPU emulation (tests/emul_build.py -> tests/_build/libirdm_emul.so) against
the oracle over random scenes (tests/scenes.py: random_scene): whole stream at pipeline_depth 0, chunked at depth 1, chunked
at depth 2 fed in place with look-ahead.  Every record is compared by tests/parity.py, the code of the -m gpu parity tests.
No GPU.  Usage: python tools/fuzz_emul_pipeline.py [first_seed last_seed]   (seeds 0-11: 36 runs, all equal)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

CHILD = r'''
import sys, time
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(pkg)r)
import irdm, orc, parity, scenes
ok = bad = 0
for seed in range(%(lo)d, %(hi)d):
    fs, iq = scenes.random_scene(seed)
    ref = orc.run_stream(iq, fs)
    n = len(iq); blocks = n // 32768; parts = 3 + seed %% 3
    cuts = [blocks * (i + 1) // parts for i in range(parts)]
    ch, prev = [], 0
    for c in cuts:
        if c > prev:
            ch.append((c - prev) * 32768); prev = c
    if n %% 32768:
        ch[-1] += n %% 32768
    for kw in (dict(), dict(chunks=ch, depth=1), dict(chunks=ch, depth=2, feed="ingest_lookahead")):
        try:
            t = time.time()
            s = parity.compare(parity.run_gpu(iq, fs, **kw), ref)
            ok += 1
            print("seed", seed, fs, kw.get("depth", 0), kw.get("feed", "host"), s, round(time.time() - t, 1), flush=True)
        except Exception as ex:
            bad += 1
            print("DIFFERENT seed", seed, kw.get("depth", 0), kw.get("feed", "host"), repr(ex)[:300], flush=True)
print("%%d equal to the oracle, %%d different" %% (ok, bad))
sys.exit(1 if bad else 0)
'''


def main():
    import emul_build
    lib = emul_build.build()
    lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 12)
    code = CHILD % dict(tests=os.path.join(ROOT, "tests"), pkg=os.path.join(ROOT, "iridium-sniffer_amd"), lo=lo, hi=hi)
    return subprocess.call([sys.executable, "-c", code], env=dict(os.environ, IRDM_LIB=lib))


if __name__ == "__main__":
    sys.exit(main())
