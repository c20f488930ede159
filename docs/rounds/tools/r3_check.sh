#!/bin/bash
# scan-related GPU tests + the timeline bench lines (a quick check of a change to the scan on the GPU box)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-chk}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_scenes.py tests/test_gpu_cfg5.py tests/test_gpu_ingest.py tests/test_gpu_parity.py tests/test_gpu_timeshard.py -x -q -m gpu > "$OUT/t.log" 2>&1
tail -n 6 "$OUT/t.log"
tools/scan_timeline.sh "${1:-chk}"
