#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-cold}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_footprint.py tests/test_gpu_timeshard.py -x -q -m gpu > "$OUT/tests.log" 2>&1
tail -n 3 "$OUT/tests.log"
for w in 2 10; do
  timeout 120 python bench.py --shard time --steps 6 --warmup $w 2>/dev/null | tail -1 > "$OUT/ts_s6_w$w.json"
  IRDM_NO_ROT_PREBUILD=1 timeout 120 python bench.py --shard time --steps 6 --warmup $w 2>/dev/null | tail -1 > "$OUT/ts_s6_w${w}_ondemand.json"
done
timeout 120 python bench.py --shard time --steps 6 --warmup 2 --depth 2 2>/dev/null | tail -1 > "$OUT/ts_s6_w2_d2.json"
python - <<P
import json,glob,os
for f in sorted(glob.glob("$OUT/ts_*.json")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d["value"], d["ms_per_step"], d["config"].get("rot"), d["config"].get("host_us"))
    except Exception as e: print(f, e)
P
