#!/bin/bash
# (round 6, experiment) what the step does when the chain's lane-per-burst tails are cut short (results are WRONG with the
# IRDM_WHATIF_* variables set: timing only), at several pipeline depths; and a kernel trace of the default run.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-whatif}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
run() { # label, env, args
  local label=$1; shift
  local envs=$1; shift
  env $envs timeout 120 python bench.py --steps 20 --warmup 6 $Q "$@" 2>/dev/null | tail -1 > "$OUT/$label.json"
  python - "$OUT/$label.json" "$label" <<'P'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]; c = r.get("kernel_clock_ms") or {}; st = r["stage_ms"]
    h = d["config"].get("host_us_total") or {}
    n = d["steps"] + d["warmup"]
    print("%-28s %6.0f  %.3f  fir %.3f k1 %.3f | %.2f %.2f %.2f %.2f %.2f | settle %d older %d" % (sys.argv[2], d["value"], d["ms_per_step"], c.get("fir", 0), c.get("fft_mag", 0),
          st["fft_mag"], st["scan"], st["fir"], st["post"], st["demod"], h.get("settle", 0) / n, h.get("wait_older_chain", 0) / n))
except Exception as e:
    print("%-28s (no result: %s)" % (sys.argv[2], e))
P
}
run base_a "X=1"
run base_b "X=1"
run demod32 "IRDM_WHATIF_DEMOD_SYMS=32"
run demod32_rot512 "IRDM_WHATIF_DEMOD_SYMS=32 IRDM_WHATIF_ROT_STEPS=512"
for d in 2 4 5; do
  run base_d$d "X=1" --depth $d
  run demod32_rot512_d$d "IRDM_WHATIF_DEMOD_SYMS=32 IRDM_WHATIF_ROT_STEPS=512" --depth $d
done
run demod191 "IRDM_WHATIF_DEMOD_SYMS=191 IRDM_WHATIF_ROT_STEPS=2800"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT" -o kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 6 $Q > "$OUT/kt.log" 2>&1
cd "$GRAFT_REPO_ROOT"
f=$(ls $OUT/*kt_kernel_trace.csv $OUT/*/*kt_kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python tools/trace_gantt.py "$f" 4 > "$OUT/gantt.txt" && rm -f "$f"
ls "$OUT"
