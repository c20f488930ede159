#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_run4}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
D12="--density 40 --sample-rate 12000000"
show() { python - "$1" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], "frac", r["frac"], "clk", r.get("kernel_clock_ms"), "stage", r["stage_ms"], "alone", r.get("stage_ms_alone"), r.get("kernel_clock_ms_alone"), d.get("parity_checked",{}).get("first_mismatch") if isinstance(d.get("parity_checked"),dict) else None)
except Exception as e: print(sys.argv[1], "ERR", e)
P
}
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest.txt" 2>&1
tail -25 "$OUT/pytest.txt"
timeout 120 python bench.py --steps 20 --warmup 5 $Q --alone-steps 3 2>"$OUT/b.err" | tail -1 > "$OUT/b.json"; show "$OUT/b.json"
timeout 120 python bench.py --steps 20 --warmup 5 $Q --alone-steps 3 --opt fir_order=0 2>/dev/null | tail -1 > "$OUT/b_scalar.json"; show "$OUT/b_scalar.json"
timeout 120 python bench.py --steps 20 --warmup 5 $Q 2>/dev/null | tail -1 > "$OUT/b2.json"; show "$OUT/b2.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --alone-steps 3 $D12 2>/dev/null | tail -1 > "$OUT/c5.json"; show "$OUT/c5.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --alone-steps 3 $D12 --opt fir_order=0 2>/dev/null | tail -1 > "$OUT/c5_scalar.json"; show "$OUT/c5_scalar.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 2>/dev/null | tail -1 > "$OUT/d40.json"; show "$OUT/d40.json"
