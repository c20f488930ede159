#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/p16
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -x -q -m gpu > "$OUT/t1.log" 2>&1
Q="--cpu-samples 0 --host-steps 0 --detect-steps 0 --file-run 0 --alone-steps 3"
for rep in 1 2; do
timeout 120 python bench.py $Q 2>"$OUT/b_$rep.err" | tail -1 > "$OUT/b_$rep.json"
done
timeout 120 python bench.py $Q --opt fir_strip=2 2>/dev/null | tail -1 > "$OUT/b_s2.json"
timeout 120 python bench.py $Q --steps 100 --warmup 5 2>/dev/null | tail -1 > "$OUT/b_k100.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 --sample-rate 12000000 2>"$OUT/cfg5.err" | tail -1 > "$OUT/cfg5.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 2>/dev/null | tail -1 > "$OUT/d40.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 2 2>/dev/null | tail -1 > "$OUT/d2.json"
tail -n 5 "$OUT/t1.log"
