#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_run3}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
D12="--density 40 --sample-rate 12000000"
show() { python - "$1" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], "clk", r.get("kernel_clock_ms"), "stage", r["stage_ms"], "alone", r.get("stage_ms_alone"), r.get("kernel_clock_ms_alone"))
except Exception as e: print(sys.argv[1], "ERR", e)
P
}
# A/B: the decimator with and without its clock stamps, alone on the chip
for v in a b; do
timeout 120 python bench.py --steps 6 --warmup 2 $Q --alone-steps 6 2>/dev/null | tail -1 > "$OUT/ab_kclk_$v.json"; show "$OUT/ab_kclk_$v.json"
IRDM_LIB=$GRAFT_REPO_ROOT/iridium-sniffer_amd/build/nokclk/libirdm_hip.so timeout 120 python bench.py --steps 6 --warmup 2 $Q --alone-steps 6 2>/dev/null | tail -1 > "$OUT/ab_nokclk_$v.json"; show "$OUT/ab_nokclk_$v.json"
done
for sl in 0 2048 4096 1024 2048; do
timeout 120 python bench.py --steps 20 --warmup 5 $Q --opt fir_slice=$sl 2>/dev/null | tail -1 > "$OUT/b_slice$sl.json"; show "$OUT/b_slice$sl.json"
done
timeout 120 python bench.py --steps 20 --warmup 5 $Q --opt fir_slice=2048 --opt band_plan_ahead=1 2>/dev/null | tail -1 > "$OUT/b_slice2048_ahead.json"; show "$OUT/b_slice2048_ahead.json"
for sl in 0 2048 4096; do
timeout 120 python bench.py --steps 10 --warmup 3 $Q $D12 --opt fir_slice=$sl 2>/dev/null | tail -1 > "$OUT/c5_slice$sl.json"; show "$OUT/c5_slice$sl.json"
done
timeout 120 python bench.py --steps 10 --warmup 3 $Q $D12 --opt fir_slice=2048 --opt band_plan_ahead=1 2>/dev/null | tail -1 > "$OUT/c5_slice2048_ahead.json"; show "$OUT/c5_slice2048_ahead.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 2 --opt fir_slice=2048 2>/dev/null | tail -1 > "$OUT/d2_slice2048.json"; show "$OUT/d2_slice2048.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 2 2>/dev/null | tail -1 > "$OUT/d2.json"; show "$OUT/d2.json"
