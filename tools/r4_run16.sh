#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_run16}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
timeout 900 python -m pytest tests -m gpu -q -x -s -k "footprint" > "$OUT/pytest_footprint.txt" 2>&1
grep -i "GB\|passed\|failed" "$OUT/pytest_footprint.txt" | tail -6
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest.txt" 2>&1
tail -4 "$OUT/pytest.txt"
show() { python - "$1" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], r["kernel"], r["frac"], "clk", r.get("kernel_clock_ms"), "stage", r["stage_ms"])
except Exception as e: print(sys.argv[1], "ERR", e)
P
}
run() { name=$1; shift; timeout 120 python bench.py --steps 20 --warmup 5 $Q "$@" 2>/dev/null | tail -1 > "$OUT/$name.json"; show "$OUT/$name.json"; }
run c3_a
run c3_b
run c5 --steps 10 --warmup 3 --density 40 --sample-rate 12000000
run d2 --steps 10 --warmup 3 --density 2
run d0 --depth 0
