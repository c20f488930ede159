#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/p27
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --detect-steps 0 --file-run 0 --alone-steps 0"
for s in 3 2 1; do
timeout 120 python bench.py $Q --opt band_timeline=1 --opt fir_strip=$s 2>"$OUT/b.err" | tail -1 > "$OUT/b_$s.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 --sample-rate 12000000 --opt band_timeline=1 --opt fir_strip=$s 2>/dev/null | tail -1 > "$OUT/cfg5_$s.json"
done
timeout 120 python bench.py $Q --depth 0 --opt band_timeline=1 2>/dev/null | tail -1 > "$OUT/b0.json"
