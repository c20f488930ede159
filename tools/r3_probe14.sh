#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/p14
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout 300 python bench.py --host-steps 0 --detect-steps 0 --file-run 0 --alone-steps 2 --cpu-passes 1 2>"$OUT/b.err" | tail -1 > "$OUT/b.json"
timeout 900 python -m pytest tests/test_gpu_scenes.py tests/test_gpu_cfg5.py tests/test_gpu_ingest.py tests/test_gpu_timeshard.py tests/test_gpu_compat.py -x -q -m gpu > "$OUT/t1.log" 2>&1
tail -n 4 "$OUT/t1.log"
python -c "
import json; d=json.load(open('$OUT/b.json')); print(d['value'], d['ms_per_step'], d['parity_checked'], d['roofline']['stage_ms'])"
