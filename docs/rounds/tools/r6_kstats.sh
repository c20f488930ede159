#!/bin/bash
# round 6: per-kernel times alone (pipeline_depth 0) and in run, cfg3 and the 12 MHz dense scene
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-ks}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
D12="--density 40 --sample-rate 12000000"
if [ "${2:-}" = tests ]; then
  timeout 1200 python -m pytest tests/test_gpu_timed_config.py -x -q -m gpu > "$OUT/tests.log" 2>&1
  tail -n 5 "$OUT/tests.log"
fi
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py"
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT" -o r1d0 --output-format csv -- python $B --steps 10 --warmup 3 --depth 0 $Q > "$OUT/kt0.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT" -o r1 --output-format csv -- python $B --steps 20 --warmup 5 $Q > "$OUT/kt.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT" -o c5d0 --output-format csv -- python $B --steps 6 --warmup 2 --depth 0 $Q $D12 > "$OUT/kt_c5d0.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT" -o c5 --output-format csv -- python $B --steps 10 --warmup 3 $Q $D12 > "$OUT/kt_c5.log" 2>&1
cd "$GRAFT_REPO_ROOT"
for t in r1 c5; do
  f=$(ls $OUT/*${t}_kernel_trace.csv $OUT/*/*${t}_kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python tools/trace_gantt.py "$f" 4 > "$OUT/gantt_$t.txt"
done
find "$OUT" -name "*kernel_trace.csv" -delete
for t in r1d0 r1 c5d0 c5; do
  f=$(ls $OUT/*${t}_kernel_stats.csv $OUT/*/*${t}_kernel_stats.csv 2>/dev/null | head -1)
  echo "== $t"; [ -n "$f" ] && python - "$f" <<'P'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print("%-60s %4s %9.1f" % (r["Name"].replace("void ","").replace("irdm::","").replace("(anonymous namespace)::","").split("(")[0][:60], r["Calls"], float(r["AverageNs"])/1e3))
P
done
