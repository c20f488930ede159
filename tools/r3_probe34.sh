#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/p34
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_scenes.py tests/test_gpu_cfg5.py tests/test_gpu_ingest.py -x -q -m gpu > "$OUT/t.log" 2>&1
tail -n 6 "$OUT/t.log"
Q="--cpu-samples 0 --host-steps 0 --detect-steps 0 --file-run 0 --alone-steps 0"
timeout 120 python bench.py $Q --opt band_timeline=1 2>"$OUT/b.err" | tail -1 > "$OUT/b.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 --sample-rate 12000000 --opt band_timeline=1 2>/dev/null | tail -1 > "$OUT/cfg5.json"
timeout 120 python bench.py $Q --depth 0 --opt band_timeline=1 2>/dev/null | tail -1 > "$OUT/b0.json"
timeout 120 python bench.py $Q 2>/dev/null | tail -1 > "$OUT/b_plain.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 --sample-rate 12000000 2>/dev/null | tail -1 > "$OUT/cfg5_plain.json"
