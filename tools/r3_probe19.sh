#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/p19
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_gpu_scenes.py tests/test_gpu_cfg5.py tests/test_gpu_ingest.py tests/test_gpu_parity.py tests/test_gpu_timeshard.py -x -q -m gpu > "$OUT/t1.log" 2>&1
Q="--cpu-samples 0 --host-steps 0 --detect-steps 0 --file-run 0 --alone-steps 3"
for v in 64 32; do
timeout 120 python bench.py $Q --opt band_sum_bins=$v 2>"$OUT/b_$v.err" | tail -1 > "$OUT/b_$v.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 2 --opt band_sum_bins=$v 2>/dev/null | tail -1 > "$OUT/d2_$v.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 --sample-rate 12000000 --opt band_sum_bins=$v 2>/dev/null | tail -1 > "$OUT/cfg5_$v.json"
done
tail -n 5 "$OUT/t1.log"
