#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_run15}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
timeout 900 python -m pytest tests -m gpu -q -x > "$OUT/pytest.txt" 2>&1
tail -4 "$OUT/pytest.txt"
show() { python - "$1" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    tl = d["config"].get("scan_timeline_us") or {}
    print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], "clk", r.get("kernel_clock_ms"), "stage", r["stage_ms"])
    if tl: print("   " + "  ".join("%s %.0f/%.0f" % (k, v[0], v[1]) for k, v in tl.items() if v[2] > 0.5))
except Exception as e: print(sys.argv[1], "ERR", e)
P
}
run() { name=$1; shift; timeout 120 python bench.py --steps 20 --warmup 5 $Q "$@" 2>/dev/null | tail -1 > "$OUT/$name.json"; show "$OUT/$name.json"; }
for rep in a b; do
run ring_$rep
run noring_$rep --opt band_sum_ring=0
done
run tl_ring --opt band_timeline=1
run tl_ring_d0 --opt band_timeline=1 --depth 0
run tl_noring_d0 --opt band_timeline=1 --depth 0 --opt band_sum_ring=0
run c5_ring --steps 10 --warmup 3 --density 40 --sample-rate 12000000
run c5_noring --steps 10 --warmup 3 --density 40 --sample-rate 12000000 --opt band_sum_ring=0
run d2_ring --steps 10 --warmup 3 --density 2 --opt band_timeline=1
run d2_noring --steps 10 --warmup 3 --density 2 --opt band_sum_ring=0
run d2_ring_b --steps 10 --warmup 3 --density 2
