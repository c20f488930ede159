#!/usr/bin/env python3
"""The detector-state hand-off of the time-sharded mode on ONE GPU: how long the export (settled scan -> blob in device
memory) and the two imports (head; history) take at 12 MHz / 16384-point frames (blob 33.7 MB, head 65 KB).  What travels
between two GPUs is the blob; these are the parts of a hop a single GPU can measure.  Usage: python tools/hop_timing.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "iridium-sniffer_amd"))
import irdm      # noqa: E402
import siggen    # noqa: E402

fs = 12_000_000
n = 16 * 1024 * 1024
iq, _ = siggen.standard_scene(fs, n, 40, seed=11)
p = irdm.Pipeline(fs, max_chunk_samples=n, max_bursts_per_chunk=4096, pipeline_depth=1)
p.feed_host(iq)
p.flush()
nb, head = p.state_bytes(), p.state_head_bytes()
buf = torch.empty(nb, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e6


t_exp = timed(lambda: p.export_state_device(buf.data_ptr(), nb))
t_head = timed(lambda: p.import_state_head_device(buf.data_ptr(), head))
t_hist = timed(lambda: p.import_state_history_device(buf.data_ptr() + head, nb - head))
t_all = timed(lambda: p.import_state_device(buf.data_ptr(), nb))
host = torch.empty(nb, dtype=torch.uint8).pin_memory()
t_d2h = timed(lambda: host.copy_(buf))
print("state blob %.1f MB, head %.1f KB" % (nb / 1e6, head / 1e3))
print("export (device blob)        %7.1f us" % t_exp)
print("import head                 %7.1f us" % t_head)
print("import history (sync path)  %7.1f us" % t_hist)
print("import whole blob           %7.1f us" % t_all)
print("blob device -> pinned host  %7.1f us (%.1f GB/s)" % (t_d2h, nb / t_d2h / 1e3))
p.close()
