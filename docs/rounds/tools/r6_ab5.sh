#!/bin/bash
# round 6: demod_seq / rot_phase_rows as two-wavefront workgroups -- GPU tests of the touched stages, same-box A/B against the
# library before the change (tests/_build/libirdm_hip_base.so), per-kernel times alone and in run
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-ab5}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_timed_config.py tests/test_gpu_scenes.py tests/test_golden.py -x -q -m gpu > "$OUT/tests.log" 2>&1
tail -n 4 "$OUT/tests.log"
for i in 1 2 3; do
  tools/ab_bench.sh ${1:-ab5}_new$i "two-wavefront chains|"
  IRDM_LIB=$GRAFT_REPO_ROOT/tests/_build/libirdm_hip_base.so tools/ab_bench.sh ${1:-ab5}_old$i "before|"
done
tools/ab_bench.sh ${1:-ab5}_c5new "12 MHz dense, new|--density 40 --sample-rate 12000000"
IRDM_LIB=$GRAFT_REPO_ROOT/tests/_build/libirdm_hip_base.so tools/ab_bench.sh ${1:-ab5}_c5old "12 MHz dense, before|--density 40 --sample-rate 12000000"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py"
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT" -o r1d0 --output-format csv -- python $B --steps 10 --warmup 3 --depth 0 $Q > "$OUT/kt0.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT" -o r1 --output-format csv -- python $B --steps 20 --warmup 5 $Q > "$OUT/kt.log" 2>&1
cd "$GRAFT_REPO_ROOT"
f=$(ls $OUT/*r1_kernel_trace.csv $OUT/*/*r1_kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python tools/trace_gantt.py "$f" 4 > "$OUT/gantt_r1.txt"
find "$OUT" -name "*kernel_trace.csv" -delete
for t in r1d0 r1; do
  f=$(ls $OUT/*${t}_kernel_stats.csv $OUT/*/*${t}_kernel_stats.csv 2>/dev/null | head -1)
  echo "== $t"; [ -n "$f" ] && python - "$f" <<'P'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print("%-50s %4s %9.1f" % (r["Name"].replace("void ","").replace("irdm::","").replace("(anonymous namespace)::","").split("(")[0][:50], r["Calls"], float(r["AverageNs"])/1e3))
P
done
