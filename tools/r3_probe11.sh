#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/p11
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
Q="--cpu-samples 0 --host-steps 0 --detect-steps 0 --file-run 0 --alone-steps 0"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT" -o c1 --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 $Q --opt scan_chain=1 > "$OUT/kt_c1.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT" -o c0 --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 $Q --opt scan_chain=0 > "$OUT/kt_c0.log" 2>&1
ls "$OUT"
