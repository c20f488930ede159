#!/bin/bash
# A/B lines of profiles/r4_sched_experiments.txt: `bench.py` with an option set per line, 20 steps on the GPU box, one
# summary line each (Msamples/s, ms per step, kernel clocks of the decimator and K1, stage brackets; with band_timeline=1
# also the scan's passes as duration/wait in us).
# Usage (on the GPU box): tools/ab_bench.sh <out-dir under gpurun_out/> "<label>|<bench.py arguments>" ...
#   e.g. tools/ab_bench.sh ab1 "default|" "scalar order|--opt fir_order=0" "2/Ms timeline|--density 2 --opt band_timeline=1"
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
shift
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
i=0
for spec in "$@"; do
  i=$((i+1))
  label=${spec%%|*}
  args=${spec#*|}
  timeout 120 python bench.py --steps 20 --warmup 5 $Q $args 2>/dev/null | tail -1 > "$OUT/line$i.json"
  python - "$OUT/line$i.json" "$label" <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]; c = r.get("kernel_clock_ms") or {}; st = r["stage_ms"]
    print("%-61s %6.0f  %.3f  fir %.3f k1 %.3f | %.2f %.2f %.2f %.2f %.2f" % (sys.argv[2], d["value"], d["ms_per_step"], c.get("fir", 0), c.get("fft_mag", 0),
          st["fft_mag"], st["scan"], st["fir"], st["post"], st["demod"]))
    tl = d["config"].get("scan_timeline_us") or {}
    if tl:
        print("      passes us (duration/wait): " + "  ".join("%s %.0f/%.0f" % (k, v[0], v[1]) for k, v in tl.items() if v[2] > 0.5))
except Exception as e:
    print("%-61s (no result: %s)" % (sys.argv[2], e))
P
done
