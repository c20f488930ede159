#!/usr/bin/env python3
"""What the C99 binary's wall clock is made of on a 604 Msample cf32 recording in /dev/shm (noise only: the read and the H2D
copy set the pace): --read-threads 6 / 12 / 16, quick exit (default) against IRDM_CLEAN_EXIT=1.  GPU box; prints one line
per run: wall, start-up, stream, teardown."""
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, "iridium-sniffer_amd", "iridium-sniffer-hip")
n = 64 * 1024 * 1024
path = "/dev/shm/irdm_file_mode_%d.cf32" % os.getpid()
rng = np.random.default_rng(1)
chunk = (rng.standard_normal(2 * n, dtype=np.float32) * np.float32(0.01))
with open(path, "wb") as f:
    for _ in range(9):
        chunk.tofile(f)
del chunk
try:
    for rep in range(2):
        for threads in (6, 12, 16):
            for clean in (0, 1):
                env = dict(os.environ, IRDM_CLEAN_EXIT=str(clean))
                t0 = time.perf_counter()
                r = subprocess.run([exe, "-f", path, "-r", "10000000", "--format", "cf32", "--timing", "--read-threads", str(threads)],
                                   stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env, timeout=300)
                wall = time.perf_counter() - t0
                m = re.search(rb"startup ([0-9.]+) s .* stream ([0-9.]+) s", r.stderr)
                td = re.search(rb"teardown ([0-9.]+) s", r.stderr)
                print("threads %2d clean_exit %d: wall %.3f s  startup %s  stream %s  teardown %s  rc %d" % (
                    threads, clean, wall, m.group(1).decode() if m else "?", m.group(2).decode() if m else "?",
                    td.group(1).decode() if td else "-", r.returncode), flush=True)
finally:
    os.remove(path)
