"""Measurement aid (not product): what the group protocol costs on ONE GPU.  A 12 MHz stream (BASELINE config 4's geometry)
in chunks of --samples, device-resident, through irdm_group_feed_device with the next super-step staged ahead:
  plain      a group of one member, no hand-off (the landing-buffer copy and the ordinary feed)
  loopback   the same member seeds the overlap from its landing buffer, exports its state, sends it to itself with a
             grouped ncclSend / ncclRecv pair and imports it in front of every chunk ("group_loopback"): every part of a
             hop except the link
Prints one JSON line per mode: Msamples/s over the timed super-steps (the feeding loop's steady state; the final flush and
the poll of the last chunks' records are timed apart), ms per chunk, host time per call, the bytes that moved per chunk.
Usage: python tools/group_bench.py [--samples N] [--steps K] [--warmup W] [--density D] [--depth P]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "iridium-sniffer_amd"))

import bench            # noqa: E402
import irdm             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=64 * 1024 * 1024)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--density", type=float, default=10.0)
    ap.add_argument("--sample-rate", type=int, default=12_000_000)
    ap.add_argument("--depth", type=int, default=2)
    args = ap.parse_args()
    fs, n = args.sample_rate, args.samples
    import torch
    dev = torch.device("cuda:0")
    x, nb = bench.build_scene(torch, dev, fs, n, args.density, seed=5)
    torch.cuda.synchronize()
    ptr = x.data_ptr()
    out = []
    for mode in ("plain", "loopback"):
        g = irdm.Group(fs, 1, max_chunk_samples=n, max_bursts_per_chunk=8192, pipeline_depth=args.depth)
        g.set_option("packed_records", 1)
        if mode == "loopback":
            g.set_option("group_loopback", 1)
        frames = 0
        # (raw polls into preallocated arrays: irdm.py's record objects cost 2-3 us each in Python, 3 ms a chunk)
        rec_buf = (irdm.DemodPacked * 8192)()
        burst_buf = (irdm.Burst * 8192)()

        def poll_all():
            k = 0
            while True:
                m = g.L.irdm_group_poll_demods_packed(g.g, rec_buf, 8192)
                if m <= 0:
                    break
                k += m
            while g.L.irdm_group_poll_bursts(g.g, burst_buf, 8192) > 0:
                pass
            return k
        t0 = None
        total = args.warmup + args.steps
        g.stage_device(ptr, n)
        tt = {"stage": 0.0, "feed": 0.0, "poll": 0.0}
        h0 = None
        names = ("k1_ring", "settle", "chain_enqueue", "scan_enqueue", "wait_older_chain", "final_sync", "settle_wait_scan",
                 "settle_counters", "settle_records", "build_records")
        for s in range(total):
            if s == args.warmup:
                t0 = time.perf_counter()
                tt = {k: 0.0 for k in tt}
                h0 = [g.stat("host_us_%d" % i) for i in range(10)]
            a = time.perf_counter()
            if s + 1 < total:
                g.stage_device(ptr, n)         # (the same buffer again: the stream repeats, the detector carries on)
            b = time.perf_counter()
            g.feed_device(ptr, n)
            c = time.perf_counter()
            frames += poll_all()
            d = time.perf_counter()
            tt["stage"] += b - a
            tt["feed"] += c - b
            tt["poll"] += d - c
        h1 = [g.stat("host_us_%d" % i) for i in range(10)]
        t_loop = time.perf_counter() - t0
        in_loop = frames
        g.flush()
        t_flush = time.perf_counter() - t0 - t_loop
        frames += poll_all()
        dt = time.perf_counter() - t0
        st = {k: g.stat(k) for k in ("hops", "hop_bytes", "overlap_bytes", "scatter_bytes", "late_history", "chunks")}
        out.append({"mode": mode, "Msamples_per_s": round(args.steps * n / t_loop / 1e6, 1), "ms_per_chunk": round(t_loop / args.steps * 1e3, 3),
                    "Msamples_per_s_with_the_tail": round(args.steps * n / dt / 1e6, 1),
                    "frames": frames, "per_chunk_MB": {k: round(st[k] / max(st["chunks"], 1) / 1e6, 1) for k in ("hop_bytes", "overlap_bytes", "scatter_bytes")},
                    "loop_ms_per_chunk": round(t_loop / args.steps * 1e3, 3), "final_flush_ms": round(t_flush * 1e3, 3),
                    "final_poll_ms": round((dt - t_loop - t_flush) * 1e3, 3), "frames_polled_in_loop": in_loop,
                    "call_ms_per_chunk": {k: round(v / args.steps * 1e3, 3) for k, v in tt.items()},
                    "member_host_us_per_chunk": {k: round((b_ - a_) / args.steps) for k, a_, b_ in zip(names, h0, h1)},
                    "hops": st["hops"], "late_history": st["late_history"], "sample_rate": fs, "chunk_samples": n, "pipeline_depth": args.depth})
        print(json.dumps(out[-1]), flush=True)
        g.close()


if __name__ == "__main__":
    main()
