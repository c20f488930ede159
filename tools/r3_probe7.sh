#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/p7
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout 300 tools/check_sincosf_gpu 2.0 > "$OUT/sincosf_exhaustive.txt" 2>&1
timeout 1200 python -m pytest tests -x -q -m gpu > "$OUT/t1.log" 2>&1
Q="--cpu-samples 0 --host-steps 0 --detect-steps 0 --file-run 0"
timeout 120 python bench.py $Q 2>"$OUT/b.err" | tail -1 > "$OUT/b.json"
timeout 120 python bench.py $Q --opt host_cfo=1 2>"$OUT/b_hc.err" | tail -1 > "$OUT/b_hc.json"
timeout 120 python bench.py $Q --depth 0 2>"$OUT/b_d0.err" | tail -1 > "$OUT/b_d0.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 --sample-rate 12000000 2>"$OUT/cfg5.err" | tail -1 > "$OUT/cfg5.json"
cat "$OUT/sincosf_exhaustive.txt"; tail -n 5 "$OUT/t1.log"
