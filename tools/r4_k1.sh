#!/bin/bash
# K1 alone (tools/ubench/k1_bench), the GPU suite, and short in-run bench lines.  Usage: tools/r4_k1.sh <tag>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_k1}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
D12="--density 40 --sample-rate 12000000"
{
timeout 120 tools/ubench/k1_bench 13 8192 20 2 200
timeout 120 tools/ubench/k1_bench 14 4096 20 2 400
timeout 60 tools/ubench/k1_bench 13 2048 5 1 200
timeout 60 tools/ubench/k1_bench 14 1024 5 0 200
} > "$OUT/k1_bench.txt" 2>&1
timeout 600 python -m pytest tests -m gpu -x -q > "$OUT/pytest.txt" 2>&1
tail -3 "$OUT/pytest.txt"
timeout 120 python bench.py --steps 20 --warmup 5 $Q 2>"$OUT/b.err" | tail -1 > "$OUT/b.json"
timeout 120 python bench.py --steps 20 --warmup 5 $Q --opt k1_kernel=0 2>/dev/null | tail -1 > "$OUT/b_r16.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --alone-steps 3 $D12 2>/dev/null | tail -1 > "$OUT/c5.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --alone-steps 3 $D12 --opt k1_kernel=0 2>/dev/null | tail -1 > "$OUT/c5_r16.json"
cat "$OUT/k1_bench.txt"
for f in b b_r16 c5 c5_r16; do python - "$OUT/$f.json" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], r["stage_ms"], r.get("stage_ms_alone"))
except Exception as e: print(sys.argv[1], "ERR", e)
P
done
