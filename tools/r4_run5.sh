#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_run5}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
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
run tl_f --opt band_timeline=1
run tl_r --opt band_timeline=1 --opt fir_order=0
GPU_MAX_HW_QUEUES=8 run hwq8
GPU_MAX_HW_QUEUES=8 run hwq8_r --opt fir_order=0
run depth1 --depth 1
run depth1_r --depth 1 --opt fir_order=0
run depth3 --depth 3
run f_slice8192 --opt fir_slice=8192
run f_grid2048 --opt fir_grid=2048
run f_grid1536 --opt fir_grid=1536
run f2
run r2 --opt fir_order=0
