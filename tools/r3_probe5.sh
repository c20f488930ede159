#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/p6
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_scenes.py tests/test_gpu_cfg5.py tests/test_gpu_ingest.py -x -q -m gpu > "$OUT/t1.log" 2>&1
Q="--cpu-samples 0 --host-steps 0 --detect-steps 0 --file-run 0"
timeout 120 python bench.py $Q 2>"$OUT/b.err" | tail -1 > "$OUT/b.json"
timeout 120 python bench.py $Q --opt band_coop=0 2>"$OUT/b_nc.err" | tail -1 > "$OUT/b_nc.json"
timeout 120 python bench.py $Q --opt fir_strip=2 2>"$OUT/b_s2.err" | tail -1 > "$OUT/b_s2.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 --sample-rate 12000000 2>"$OUT/cfg5.err" | tail -1 > "$OUT/cfg5.json"
tail -n 5 "$OUT/t1.log"
