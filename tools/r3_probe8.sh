#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/p8
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests -x -q -m gpu > "$OUT/t1.log" 2>&1
Q="--cpu-samples 0 --host-steps 0 --detect-steps 0 --file-run 0"
timeout 120 python bench.py $Q 2>"$OUT/b.err" | tail -1 > "$OUT/b.json"
timeout 120 python bench.py $Q --opt scan_chain=0 2>"$OUT/b_nc.err" | tail -1 > "$OUT/b_nc.json"
timeout 120 python bench.py $Q --opt fir_strip=2 2>"$OUT/b_s2.err" | tail -1 > "$OUT/b_s2.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 --sample-rate 12000000 2>"$OUT/cfg5.err" | tail -1 > "$OUT/cfg5.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 2>"$OUT/d40.err" | tail -1 > "$OUT/d40.json"
tail -n 5 "$OUT/t1.log"
