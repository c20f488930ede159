#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/p21
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_gpu_scenes.py tests/test_gpu_cfg5.py tests/test_gpu_ingest.py tests/test_gpu_parity.py tests/test_gpu_timeshard.py -x -q -m gpu > "$OUT/t1.log" 2>&1
Q="--cpu-samples 0 --host-steps 0 --detect-steps 0 --file-run 0 --alone-steps 3"
timeout 120 python bench.py $Q 2>"$OUT/b.err" | tail -1 > "$OUT/b.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 2 2>/dev/null | tail -1 > "$OUT/d2.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 --sample-rate 12000000 2>/dev/null | tail -1 > "$OUT/cfg5.json"
tail -n 5 "$OUT/t1.log"
