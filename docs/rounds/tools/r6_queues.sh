#!/bin/bash
# (round 6, experiment) more batch contexts with more hardware queues: is the slowdown beyond four contexts head-of-line
# blocking of chain streams that share a hardware queue behind a 0.65 ms demod_seq launch?
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-queues}
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
run d3 "X=1"
run d3_q16 "GPU_MAX_HW_QUEUES=16"
run d5 "X=1" --depth 5
run d5_q8 "GPU_MAX_HW_QUEUES=8" --depth 5
run d5_q16 "GPU_MAX_HW_QUEUES=16" --depth 5
run d4_q16 "GPU_MAX_HW_QUEUES=16" --depth 4
run d5_q16_la2 "GPU_MAX_HW_QUEUES=16" --depth 5 --lookahead 2
run d3_la2 "X=1" --lookahead 2
run d5_q24 "GPU_MAX_HW_QUEUES=24" --depth 5
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace -d "$OUT" -o kt5 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 6 $Q --depth 5 > "$OUT/kt5.log" 2>&1
GPU_MAX_HW_QUEUES=16 timeout 200 rocprofv3 --kernel-trace -d "$OUT" -o kt5q --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 6 $Q --depth 5 > "$OUT/kt5q.log" 2>&1
cd "$GRAFT_REPO_ROOT"
for t in kt5 kt5q; do
  f=$(ls $OUT/*${t}_kernel_trace.csv $OUT/*/*${t}_kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python tools/trace_gantt.py "$f" 4 > "$OUT/gantt_$t.txt" && rm -f "$f"
done
ls "$OUT"
