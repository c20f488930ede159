#!/usr/bin/env python3
"""Authoring-container aid (needs oracle/_ref): the bit-layer comparisons of tests/test_oracle_bitlayer.py (oracle vs the
reference's frame_decode.c / ida_decode.c object code) over many more seeds.  Usage: python tools/fuzz_bitlayer.py [n_seeds]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "iridium-sniffer_amd"))
import orc                        # noqa: E402
import test_oracle_bitlayer as T  # noqa: E402


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    oracle, reflib = orc.lib(), orc.ref()
    if reflib is None:
        raise SystemExit("oracle/_ref is not built (needs /root/reference): make -C oracle ref")
    oracle.orc_frame_decode.restype = C.c_int
    reflib.ref_frame_decode.restype = C.c_int
    oracle.orc_ida_decode.restype = C.c_int
    reflib.ref_ida_decode.restype = C.c_int
    nf = ni = 0
    for seed in range(1000, 1000 + n_seeds):
        for bits, llr in T.make_cases(seed):
            ro, do = T.decode_with(oracle.orc_frame_decode, bits, llr)
            rr, dr = T.decode_with(reflib.ref_frame_decode, bits, llr)
            assert ro == rr and T.as_tuple(do) == T.as_tuple(dr), ("frame_decode", seed)
            nf += 1
        for bits, llr, direction in T.make_ida_cases(seed):
            ro, do = T.ida_decode_with(oracle.orc_ida_decode, bits, llr, direction)
            rr, dr = T.ida_decode_with(reflib.ref_ida_decode, bits, llr, direction)
            assert ro == rr and T.ida_tuple(do) == T.ida_tuple(dr), ("ida_decode", seed)
            ni += 1
    print("bit layer: oracle == reference object code on %d frame_decode and %d ida_decode inputs" % (nf, ni))


if __name__ == "__main__":
    main()
