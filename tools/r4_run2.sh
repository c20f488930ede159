#!/bin/bash
# Usage: tools/r4_run2.sh <tag>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_run2}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
D12="--density 40 --sample-rate 12000000"
{
timeout 120 tools/ubench/k1_bench 13 8192 20 2 200
timeout 120 tools/ubench/k1_bench 14 4096 20 2 400
timeout 120 tools/ubench/k1_bench 13 8192 10 2 20
} > "$OUT/k1_bench.txt" 2>&1
cat "$OUT/k1_bench.txt"
IRDM_PLAN_AHEAD=1 timeout 600 python -m pytest tests -m gpu -x -q > "$OUT/pytest_ahead.txt" 2>&1
tail -3 "$OUT/pytest_ahead.txt"
timeout 120 python bench.py --steps 20 --warmup 5 $Q --alone-steps 3 2>"$OUT/b.err" | tail -1 > "$OUT/b.json"
timeout 120 python bench.py --steps 20 --warmup 5 $Q --opt band_plan_ahead=1 2>/dev/null | tail -1 > "$OUT/b_ahead.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --alone-steps 3 $D12 2>/dev/null | tail -1 > "$OUT/c5.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q $D12 --opt band_plan_ahead=1 2>/dev/null | tail -1 > "$OUT/c5_ahead.json"
for f in b b_ahead c5 c5_ahead; do python - "$OUT/$f.json" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], "frac", r["frac"], "ms", r["ms_per_launch"], r.get("kernel_clock_ms"), r["stage_ms"], r.get("stage_ms_alone"), r.get("kernel_clock_ms_alone"))
except Exception as e: print(sys.argv[1], "ERR", e)
P
done
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py"
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT" -o r1 --output-format csv -- python $B --steps 20 --warmup 5 $Q > "$OUT/kt.log" 2>&1
tail -1 "$OUT/kt.log" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('under rocprof:', d['value'], r['ms_per_launch'], r['kernel_clock_ms'], r['stage_ms'])"
python - "$OUT" <<'P'
import csv,sys,glob
for f in glob.glob(sys.argv[1]+'/**/r1_kernel_stats.csv', recursive=True):
    for row in list(csv.DictReader(open(f)))[:8]:
        print(row['Name'][:60], row['Calls'], row['AverageNs'])
P
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $pass --kernel-include-regex "fft_mag_p32" -d "$OUT/k1sq$i" -o pmc --output-format csv -- \
      $GRAFT_REPO_ROOT/tools/ubench/k1_bench 13 8192 3 2 200 > "$OUT/k1sq$i.log" 2>&1
done
python - "$OUT" <<'P'
import csv,sys,glob,collections
for d in ('k1sq1','k1sq2'):
    acc=collections.defaultdict(list)
    for f in glob.glob(sys.argv[1]+'/'+d+'/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            acc[(row['Kernel_Name'][:40],row['Counter_Name'])].append(float(row['Counter_Value']))
    for k,v in sorted(acc.items()): print(k, len(v), sum(v)/len(v))
P
