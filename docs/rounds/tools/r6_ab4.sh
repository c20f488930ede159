#!/bin/bash
# the full default bench line (new parity / pcie / cpu fields), then same-box A/B: current / four feed slots / before the prune
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-ab4}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
( time timeout 900 python bench.py ) > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
tail -c 600 "$OUT/bench_default.err"
python - "$OUT/bench_default.json" <<'P'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], "parity", d["parity_checked"], "\npcie", d["pcie_inclusive"], "\ncpu", d["cpu_baseline"], "\nfile", d["file_to_raw"])
P
for i in 1 2 3; do
  tools/ab_bench.sh ${1:-ab4}_new$i "current|"
  IRDM_LIB=$GRAFT_REPO_ROOT/tests/_build/libirdm_hip_slots4.so tools/ab_bench.sh ${1:-ab4}_s4$i "four feed slots|"
  IRDM_LIB=$GRAFT_REPO_ROOT/tests/_build/libirdm_hip_preprune.so tools/ab_bench.sh ${1:-ab4}_old$i "before the prune|"
done
