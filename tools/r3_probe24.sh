#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/p24
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --detect-steps 0 --file-run 0 --alone-steps 0"
for i in 1 2; do
timeout 120 python bench.py $Q 2>/dev/null | tail -1 > "$OUT/b_ev1_$i.json"
timeout 120 python bench.py $Q --opt scan_events=0 2>"$OUT/b.err" | tail -1 > "$OUT/b_ev0_$i.json"
timeout 120 python bench.py $Q --opt scan_events=0 --opt band_timeline=1 2>"$OUT/b.err" | tail -1 > "$OUT/b_ev0tl_$i.json"
done
