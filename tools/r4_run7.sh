#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_run7}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest.txt" 2>&1
tail -25 "$OUT/pytest.txt"
show() { python - "$1" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], "clk", r.get("kernel_clock_ms"), "stage", r.get("stage_ms") or r.get("stage_ms_rank0_last_step"))
except Exception as e: print(sys.argv[1], "ERR", e)
P
}
timeout 120 python bench.py --steps 20 --warmup 5 $Q 2>"$OUT/b.err" | tail -1 > "$OUT/b.json"; show "$OUT/b.json"
timeout 160 python bench.py --shard time --steps 8 --warmup 3 2>"$OUT/ts.err" | tail -1 > "$OUT/cfg4_n1.json"; show "$OUT/cfg4_n1.json"; tail -3 "$OUT/ts.err"
timeout 160 python bench.py --shard time --steps 8 --warmup 3 --depth 1 2>/dev/null | tail -1 > "$OUT/cfg4_n1_d1.json"; show "$OUT/cfg4_n1_d1.json"
