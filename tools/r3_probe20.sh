#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/p20
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
Q="--cpu-samples 0 --host-steps 0 --detect-steps 0 --file-run 0 --alone-steps 0"
timeout 120 python bench.py $Q 2>"$OUT/b.err" | tail -1 > "$OUT/b.json"
timeout 120 python bench.py $Q --depth 0 2>/dev/null | tail -1 > "$OUT/b0.json"
timeout 300 rocprofv3 --kernel-trace -d "$OUT/kt" -o r -- python bench.py $Q > "$OUT/kt.log" 2>&1
tail -1 "$OUT/kt.log" > "$OUT/b_traced.json"
find "$OUT/kt" -name "*kernel_trace.csv" -exec cp {} "$OUT/kernel_trace.csv" \;
rm -rf "$OUT/kt"
ls -la "$OUT"
