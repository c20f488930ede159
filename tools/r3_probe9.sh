#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/p9
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_compat.py tests/test_gpu_libm.py tests/test_gpu_ingest.py -x -q -m gpu > "$OUT/t1.log" 2>&1
Q="--cpu-samples 0 --host-steps 0 --detect-steps 0 --file-run 0 --alone-steps 0"
for c in 1 0; do
timeout 120 python bench.py $Q --opt scan_chain=$c 2>"$OUT/b_c$c.err" | tail -1 > "$OUT/b_c$c.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 --sample-rate 12000000 --opt scan_chain=$c 2>"$OUT/cfg5_c$c.err" | tail -1 > "$OUT/cfg5_c$c.json"
done
tail -n 5 "$OUT/t1.log"
