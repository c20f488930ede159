#!/bin/bash
# whole -m gpu suite + smoke, then the cold-start lines (time-shard N=1: first eight chunks / steady) with and without rot_prebuild
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-suite}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout 1800 python -m pytest tests -x -q -m gpu > "$OUT/tests.log" 2>&1
tail -n 6 "$OUT/tests.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
tail -n 2 "$OUT/smoke.log"
timeout 120 python bench.py --shard time --steps 6 --warmup 2 2>/dev/null | tail -1 > "$OUT/cfg4_n1_first8chunks.json"
IRDM_NO_ROT_PREBUILD=1 timeout 120 python bench.py --shard time --steps 6 --warmup 2 2>/dev/null | tail -1 > "$OUT/cfg4_n1_first8chunks_ondemand.json"
timeout 120 python bench.py --shard time --steps 20 --warmup 5 2>/dev/null | tail -1 > "$OUT/cfg4_n1.json"
python - <<P
import json
for f in ("cfg4_n1_first8chunks","cfg4_n1_first8chunks_ondemand","cfg4_n1"):
    try:
        d=json.load(open("$OUT/%s.json"%f)); print(f, d["value"], d["ms_per_step"])
    except Exception as e: print(f, e)
P
