#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/p12
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --detect-steps 0 --file-run 0 --alone-steps 0"
timeout 120 python bench.py $Q 2>"$OUT/b_base.err" | tail -1 > "$OUT/b_base.json"
for n in 16 32 64; do
for s in 1 8; do
IRDM_SCAN_CUS=$n IRDM_SCAN_CU_STRIDE=$s timeout 120 python bench.py $Q 2>"$OUT/b_n${n}_s$s.err" | tail -1 > "$OUT/b_n${n}_s$s.json"
done
done
IRDM_SCAN_CUS=32 IRDM_SCAN_CU_STRIDE=1 timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 --sample-rate 12000000 2>"$OUT/cfg5_n32.err" | tail -1 > "$OUT/cfg5_n32.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 --sample-rate 12000000 2>"$OUT/cfg5_base.err" | tail -1 > "$OUT/cfg5_base.json"
