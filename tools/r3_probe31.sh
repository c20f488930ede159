#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/p31
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_scenes.py tests/test_gpu_cfg5.py tests/test_gpu_ingest.py tests/test_gpu_parity.py tests/test_gpu_timeshard.py -x -q -m gpu > "$OUT/t.log" 2>&1
tail -n 12 "$OUT/t.log"
Q="--cpu-samples 0 --host-steps 0 --detect-steps 0 --file-run 0 --alone-steps 0"
for v in 0 8; do
timeout 120 python bench.py $Q --depth 0 --opt band_timeline=1 --opt band_selfcheck=$v 2>/dev/null | tail -1 > "$OUT/b0_$v.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 --sample-rate 12000000 --depth 0 --opt band_timeline=1 --opt band_selfcheck=$v 2>/dev/null | tail -1 > "$OUT/cfg5_0_$v.json"
done
timeout 120 python bench.py $Q --opt band_timeline=1 2>"$OUT/b.err" | tail -1 > "$OUT/b.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 --sample-rate 12000000 --opt band_timeline=1 2>/dev/null | tail -1 > "$OUT/cfg5.json"
