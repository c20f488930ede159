#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/p29
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout 300 python -m pytest tests/test_gpu_fir_reg.py -x -q -m gpu > "$OUT/t0.log" 2>&1
tail -n 3 "$OUT/t0.log"
Q="--cpu-samples 0 --host-steps 0 --detect-steps 0 --file-run 0 --alone-steps 0"
for s in 0 1792 1536 1024; do
timeout 120 python bench.py $Q --opt band_timeline=1 --opt fir_grid=$s 2>"$OUT/b.err" | tail -1 > "$OUT/b_$s.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 --sample-rate 12000000 --opt band_timeline=1 --opt fir_grid=$s 2>/dev/null | tail -1 > "$OUT/cfg5_$s.json"
done
