#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_run6}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
D12="--density 40 --sample-rate 12000000"
show() { python - "$1" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], "clk", r.get("kernel_clock_ms"), "stage", r["stage_ms"])
except Exception as e: print(sys.argv[1], "ERR", e)
P
}
run() { name=$1; shift; timeout 120 python bench.py --steps 20 --warmup 5 $Q "$@" 2>/dev/null | tail -1 > "$OUT/$name.json"; show "$OUT/$name.json"; }
for rep in a b; do
run f_$rep
run g1536_$rep --opt fir_grid=1536
run g1280_$rep --opt fir_grid=1280
run g1792_$rep --opt fir_grid=1792
run r_$rep --opt fir_order=0
done
run c5_f --steps 10 --warmup 3 $D12
run c5_g1536 --steps 10 --warmup 3 $D12 --opt fir_grid=1536
run c5_g1280 --steps 10 --warmup 3 $D12 --opt fir_grid=1280
run c5_r --steps 10 --warmup 3 $D12 --opt fir_order=0
# the file reader: 604 Msample recording through the C99 binary, slices 6 / 12 / 1 / 0
python - <<'P'
import numpy as np, os, subprocess, time, sys, hashlib
sys.path.insert(0, "iridium-sniffer_amd")
import siggen
fs = 10_000_000
n = 9 * 64 * 1024 * 1024 + 12345
path = "/dev/shm/irdm_r4.cf32"
blk = 16 * 1024 * 1024
with open(path, "wb") as f:
    iq, _ = siggen.standard_scene(fs, blk * 4, 40, seed=3)
    left = n
    while left > 0:
        k = min(left, len(iq)); f.write(iq[:k].tobytes()); left -= k
exe = "iridium-sniffer_amd/iridium-sniffer-hip"
for rt in (6, 12, 1, 0, 6):
    t = time.perf_counter()
    r = subprocess.run([exe, "-f", path, "-r", str(fs), "--format", "cf32", "--file-info", "bench", "--timing", "--chunk", str(64 * 1024 * 1024), "--read-threads", str(rt)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    dt = time.perf_counter() - t
    lines = r.stdout.count(b"\nRAW:")
    print("read-threads", rt, "wall %.3f s" % dt, "lines", lines, hashlib.md5(r.stdout).hexdigest()[:8], [l for l in r.stderr.decode().splitlines() if "timing" in l])
os.remove(path)
P
