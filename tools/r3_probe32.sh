#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/p32
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --detect-steps 0 --file-run 0 --alone-steps 0"
for v in 0 16; do
timeout 120 python bench.py $Q --opt band_timeline=1 --opt band_selfcheck=$v 2>"$OUT/b.err" | tail -1 > "$OUT/b_$v.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 --sample-rate 12000000 --opt band_timeline=1 --opt band_selfcheck=$v 2>/dev/null | tail -1 > "$OUT/cfg5_$v.json"
done
