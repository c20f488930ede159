#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_run14}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --detect-steps 0 --file-run 0"
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest.txt" 2>&1
tail -4 "$OUT/pytest.txt"
show() { python - "$1" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], "clk", r.get("kernel_clock_ms"), "stage", r["stage_ms"], "alone", r.get("stage_ms_alone"))
except Exception as e: print(sys.argv[1], "ERR", e)
P
}
run() { name=$1; shift; timeout 120 python bench.py --steps 20 --warmup 5 $Q "$@" 2>/dev/null | tail -1 > "$OUT/$name.json"; show "$OUT/$name.json"; }
run a --alone-steps 3
run b --alone-steps 0
run c --alone-steps 0
run c5 --steps 10 --warmup 3 --density 40 --sample-rate 12000000 --alone-steps 3
run c5b --steps 10 --warmup 3 --density 40 --sample-rate 12000000 --alone-steps 0
run d40 --steps 10 --warmup 3 --density 40 --alone-steps 0
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT" -o d0 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --depth 0 $Q --alone-steps 0 > "$OUT/kt0.log" 2>&1
python - "$OUT" <<'P'
import csv,sys,glob
for f in glob.glob(sys.argv[1]+'/**/d0_kernel_stats.csv', recursive=True):
    for row in list(csv.DictReader(open(f)))[:14]:
        print(row['Name'][:70], row['Calls'], row['AverageNs'])
P
