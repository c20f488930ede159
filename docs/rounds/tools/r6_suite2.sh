#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-suite}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests -x -q -m gpu > "$OUT/tests.log" 2>&1
tail -n 6 "$OUT/tests.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
tail -n 2 "$OUT/smoke.log"
tools/ab_bench.sh ${1:-suite} "default|" "default again|" "12 MHz dense|--density 40 --sample-rate 12000000" "sparse 2/Ms|--density 2" "scalar order|--opt fir_order=0"
