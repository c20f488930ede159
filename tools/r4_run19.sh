#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_run19}
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
timeout 600 python -m pytest tests -m gpu -q -x -k "parity or golden or band or scan" > "$OUT/pytest.txt" 2>&1
tail -3 "$OUT/pytest.txt"
for pf in 0 4; do
run d2_pf${pf}_tl --steps 10 --warmup 3 --density 2 --opt band_timeline=1 --opt band_sum_prefetch=$pf
run d2_pf${pf} --steps 10 --warmup 3 --density 2 --opt band_sum_prefetch=$pf
run c3_pf${pf} --opt band_sum_prefetch=$pf
run c3_pf${pf}_b --opt band_sum_prefetch=$pf
run d0_pf${pf}_tl --depth 0 --opt band_timeline=1 --opt band_sum_prefetch=$pf
done
run c3_pf4_tl --opt band_timeline=1 --opt band_sum_prefetch=4
run c3_pf0_tl --opt band_timeline=1 --opt band_sum_prefetch=0
run c5_pf0 --steps 10 --warmup 3 --density 40 --sample-rate 12000000 --opt band_sum_prefetch=0
run c5_pf4 --steps 10 --warmup 3 --density 40 --sample-rate 12000000 --opt band_sum_prefetch=4
