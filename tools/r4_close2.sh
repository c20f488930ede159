#!/bin/bash
# after the last host-side change of round 4: the whole -m gpu suite and the time-shard / headline bench lines
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_close2}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
timeout 600 python -m pytest tests -x -q -m gpu > "$OUT/tests.log" 2>&1
tail -n 3 "$OUT/tests.log"
timeout 100 python bench.py --shard time --steps 6 --warmup 2 2>/dev/null | tail -1 > "$OUT/cfg4_n1.json"
timeout 100 python bench.py --shard time --steps 20 --warmup 5 2>/dev/null | tail -1 > "$OUT/cfg4_n1_20.json"
timeout 100 python bench.py --steps 20 --warmup 5 $Q 2>/dev/null | tail -1 > "$OUT/b_quick.json"
python - "$OUT" <<'P'
import json, sys
for n in ("cfg4_n1", "cfg4_n1_20", "b_quick"):
    try:
        d = json.load(open("%s/%s.json" % (sys.argv[1], n))); print(n, d["value"], d["ms_per_step"])
    except Exception as e:
        print(n, "ERR", e)
P
