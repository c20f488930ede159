"""TEST INFRASTRUCTURE: runs the product end to end on the CPU emulation (tests/_build/libirdm_emul.so, built by
tests/emul_build.py from the product's own sources) and compares every record with the oracle, exactly as the -m gpu
parity tests do on the real library (tests/parity.py).  Started by tests/test_pipeline_emul.py in a process of its own with
IRDM_LIB pointing at the emulated build.  Usage: python emul_pipeline_run.py <case>"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "iridium-sniffer_amd"))

import numpy as np      # noqa: E402

import irdm             # noqa: E402
import orc              # noqa: E402
import parity           # noqa: E402
import scenes           # noqa: E402
import siggen           # noqa: E402


def scene(fs, secs, nb, seed):
    n = int(secs * fs) // 32768 * 32768
    iq, _ = siggen.standard_scene(fs, n, nb, seed=seed)
    return iq


def chunks_of(n, parts):
    blocks = n // 32768
    cuts = [blocks * (i + 1) // parts for i in range(parts)]
    out, prev = [], 0
    for c in cuts:
        if c > prev:
            out.append((c - prev) * 32768)
            prev = c
    if n % 32768:
        out[-1] += n % 32768
    return out


def main():
    case = sys.argv[1]
    assert "libirdm_emul" in irdm.LIB_PATH, irdm.LIB_PATH
    res = {}
    if case == "2mhz":
        fs = 2_000_000
        iq = scene(fs, 1.3, 7, 3)
        ref = orc.run_stream(iq, fs)
        res["whole"] = parity.compare(parity.run_gpu(iq, fs), ref)
        res["chunked_depth1"] = parity.compare(parity.run_gpu(iq, fs, chunks=chunks_of(len(iq), 4), depth=1), ref)
        res["chunked_depth2_in_place_lookahead"] = parity.compare(
            parity.run_gpu(iq, fs, chunks=chunks_of(len(iq), 5), depth=2, feed="ingest_lookahead"), ref)
        # the compact records the bench polls, written to pinned memory by the demodulator's last kernel itself (demod_export)
        res["packed_depth3_in_place_lookahead"] = parity.compare_packed(
            parity.run_gpu(iq, fs, chunks=chunks_of(len(iq), 7), depth=3, feed="ingest_lookahead", packed=True), ref)
        res["packed_depth0"] = parity.compare_packed(parity.run_gpu(iq, fs, packed=True), ref)
        # every centre bin's checkpoint row prebuilt by one launch behind create (rot_prebuild; the default on the GPU for
        # pipeline_depth >= 1, asked for by option here): no build on any chain, same records
        pre = parity.run_gpu(iq, fs, chunks=chunks_of(len(iq), 5), depth=2, feed="ingest_lookahead", options={"rot_prebuild": 1})
        res["rows_prebuilt"] = parity.compare(pre, ref)
        assert pre["stats"]["rot_rows"] == 2048 and pre["stats"]["rot_ckpts"] == 0, pre["stats"]
        # five batch contexts (pipeline_depth 4): records five chunks late, same records, same order
        res["chunked_depth4_in_place_lookahead"] = parity.compare(
            parity.run_gpu(iq, fs, chunks=chunks_of(len(iq), 9), depth=4, feed="ingest_lookahead"), ref)
        x = siggen.to_ci8(iq)
        ref8 = orc.run_stream(x, fs, fmt=irdm.FMT_CI8)
        res["ci8"] = parity.compare(parity.run_gpu(x, fs, fmt=irdm.FMT_CI8), ref8)
        res["sequential_scan"] = parity.compare(parity.run_gpu(iq, fs, scan_mode=1), ref)
        # rotator checkpoint rows are built as far as needed and extended: a short burst, then a long one on the same carrier
        import test_gpu_footprint as tf
        fs2, iq2 = tf._short_then_long()
        ref2 = orc.run_stream(iq2, fs2)
        half = len(iq2) // 2 // 32768 * 32768
        got = parity.run_gpu(iq2, fs2, chunks=[half, len(iq2) - half], depth=1)
        res["rotator_row_extension"] = parity.compare(got, ref2)
        assert res["rotator_row_extension"]["bursts"] == 4
        assert got["stats"]["rot_runs"] > got["stats"]["rot_rows"] >= 2 and got["stats"]["rot_ckpts"] >= 3 * 2048, got["stats"]   # (a row was extended)
        # ... and the arena of checkpoint blocks grows when it is full (one whole row's worth of blocks to begin with), while
        # chains launched with the old arena are in flight
        got = parity.run_gpu(iq, fs, chunks=chunks_of(len(iq), 5), depth=2, feed="ingest_lookahead", options={"rot_pool_rows": 1})
        res["rotator_arena_growth"] = parity.compare(got, ref)
        assert got["stats"]["rot_grows"] >= 1 and got["stats"]["rot_blocks_cap"] >= got["stats"]["rot_blocks"], got["stats"]
        # the decimated / low-passed rows of a batch lie end to end by actual length: a scratch of 64 outputs to begin
        # with grows by doubling, per context, while the other contexts' chains are in flight
        got = parity.run_gpu(iq, fs, chunks=chunks_of(len(iq), 5), depth=2, feed="ingest_lookahead", options={"scratch_outputs": 64})
        res["scratch_growth"] = parity.compare(got, ref)
        res["scratch_growth"]["grows"] = got["stats"]["scratch_grows"]
        assert got["stats"]["scratch_grows"] >= 1 and got["stats"]["scratch_outputs"] >= 64
    elif case == "scene_zoo":
        for name in ("too_long", "squelch", "dc_and_edges"):
            fs, iq = scenes.ALL[name]()
            ref = orc.run_stream(iq, fs)
            res[name] = parity.compare(parity.run_gpu(iq, fs), ref)
    elif case == "10mhz":
        # 8192-point frames, decimation by 40: K1's radix-16 kernel (writing the band scan's candidate lists once the
        # detector is primed) and the register-resident decimator (fir_reg.hip: columns in registers, DPP shifts,
        # travelling accumulators); a priming chunk and three more at pipeline_depth 2, fed in place with look-ahead
        fs = 10_000_000
        iq = scene(fs, 0.95, 8, 77)
        ref = orc.run_stream(iq, fs)
        first = 512 * 8192
        c = ((len(iq) - first) // 5) // 32768 * 32768
        sizes = [first, c, c, c, c, len(iq) - first - 4 * c]
        got = parity.run_gpu(iq, fs, chunks=sizes, depth=2, feed="ingest_lookahead")
        res["default"] = parity.compare(got, ref)
        res["default"]["k1_lists"] = got["stats"]["k1_lists"]
        # fed with look-ahead, the chained scans open with round 1: their round 0 was a speculation pass on the second
        # workspace, enqueued with the previous chunk (band_spec)
        assert got["stats"]["spec_scans"] >= 2 and got["stats"]["spec_passes"] >= got["stats"]["spec_scans"], got["stats"]
        res["default"]["spec_scans"] = got["stats"]["spec_scans"]
        # the guess spoilt in one late frame (test hook band_selfcheck 32): the next round's sums pass restarts behind the prefix
        # of update steps both rounds share, from the state the earlier round stored there
        got = parity.run_gpu(iq, fs, chunks=sizes, depth=2, feed="ingest_lookahead", options={"band_selfcheck": 32})
        res["sums_pass_restart"] = parity.compare(got, ref)
        assert got["stats"]["sum_restarts"] >= 1, got["stats"]
        res["sums_pass_restart"]["restarts"] = got["stats"]["sum_restarts"]
        got = parity.run_gpu(iq, fs, chunks=sizes, depth=5, feed="ingest_lookahead")
        res["depth5"] = parity.compare(got, ref)
        assert got["stats"]["spec_scans"] >= 2, got["stats"]
        got = parity.run_gpu(iq, fs, chunks=sizes, depth=2, feed="ingest_lookahead", options={"band_spec": 0})
        res["without_speculation_pass"] = parity.compare(got, ref)
        assert got["stats"]["spec_scans"] == 0 and got["stats"]["scan_chained"] >= 2, got["stats"]
        # the same stream with the decimating FIR in the reference's scalar order (--no-simd: fir_decimate_kernel_r, one
        # accumulator per output travelling from lane to lane) against the oracle in that order
        orc.set_fir_order(0)
        ref0 = orc.run_stream(iq, fs)
        res["scalar_fir_order"] = parity.compare(parity.run_gpu(iq, fs, options={"fir_order": 0}), ref0)
        res["any_m_decimator_scalar_order"] = parity.compare(parity.run_gpu(iq, fs, options={"fir_order": 0, "fir_generic": 1}), ref0)
        orc.set_fir_order(1)
    elif case == "12mhz":
        # 16384-point frames (K1 <14>), decimation by 48 (the decimator's second instantiation), ci16 in two chunks
        fs = 12_000_000
        iq = scene(fs, 0.78, 3, 12)
        ref = orc.run_stream(iq, fs)
        res["default"] = parity.compare(parity.run_gpu(iq, fs), ref)
        x = siggen.to_ci16(iq)
        ref16 = orc.run_stream(x, fs, fmt=irdm.FMT_CI16)
        res["ci16_chunked_depth1"] = parity.compare(
            parity.run_gpu(x, fs, fmt=irdm.FMT_CI16, chunks=chunks_of(len(iq), 2), depth=1), ref16)
    elif case == "group":
        # ONE stream across the members of a group (irdm_group_*, csrc/group.cpp), every "device" the emulation and RCCL the
        # emulated send / receive pairs of tests/hip_emul/rccl/rccl.h: four chunks with a burst across every chunk boundary
        # (the stream of the two-rank time-shard tests).  2 members: two super-steps, the state goes round; 3 members: a
        # super-step of three chunks and one of one; 1 member with "group_loopback": overlap seed, export, send to itself,
        # import; staged ahead; fed from "device" memory of member 0 (the scatter); 1 member, no loopback: the plain path
        import test_gpu_timeshard as G
        fs = G.CASES["2mhz"][0]
        iq, chunk, ov = G._stream("2mhz")
        ref = orc.run_stream(iq, fs)
        for name, kw in (("two_members", dict(n_gpus=2)),
                         ("three_members_staged_ahead", dict(n_gpus=3, staged_ahead=True)),
                         ("two_members_scatter_from_member_0", dict(n_gpus=2, feed="device", staged_ahead=True, depth=2)),
                         ("one_member_loopback", dict(n_gpus=1, options={"group_loopback": 1})),
                         ("one_member", dict(n_gpus=1))):
            got = parity.run_group(iq, fs, chunk=chunk, **kw)
            res[name] = parity.compare(got, ref)
            res[name].update({k: got["stats"][k] for k in ("hops", "chunks", "overlap_bytes", "scatter_bytes")})
            assert got["stats"]["chunks"] == 4 and got["chunks_fed"] == 4, got["stats"]
            assert got["stats"]["overlap_samples"] == ov, (got["stats"], ov)
            proto = kw["n_gpus"] > 1 or "options" in kw
            assert got["stats"]["hops"] == (4 if proto else 0), got["stats"]
            assert got["stats"]["overlap_bytes"] == (3 * ov * 8 if proto else 0), got["stats"]
            assert got["stats"]["scatter_bytes"] == len(iq) * 8, got["stats"]
        # the stream ends inside a chunk (and not on a feed-block boundary): two members, the last super-step's second
        # chunk short; a further feed is refused
        cut = len(iq) - 3 * 32768 - 777
        ref_cut = orc.run_stream(iq[:cut], fs)
        got = parity.run_group(iq[:cut], fs, n_gpus=2, chunk=chunk, staged_ahead=True)
        res["two_members_ragged_end"] = parity.compare(got, ref_cut)
        assert got["stats"]["chunks"] == 4 and got["stats"]["hops"] == 4, got["stats"]
    else:
        raise SystemExit("unknown case")
    print("RESULT " + json.dumps(res))


if __name__ == "__main__":
    main()
