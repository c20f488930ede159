#!/bin/bash
# round-3 first probe: VALU issue ubench, baseline bench lines, rocprof kernel stats of the 12 MHz dense scene
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/p1
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout 120 tools/ubench/valu_issue > "$OUT/valu_issue.txt" 2>&1
timeout 120 tools/ubench/valu_chain > "$OUT/valu_chain.txt" 2>&1
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
timeout 120 python bench.py $Q 2>"$OUT/b.err" | tail -1 > "$OUT/b.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 --sample-rate 12000000 2>"$OUT/cfg5.err" | tail -1 > "$OUT/cfg5.json"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT" -o cfg5 --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 $Q --density 40 --sample-rate 12000000 > "$OUT/kt_cfg5.log" 2>&1
ls "$OUT"
