#!/bin/bash
# the round's closing run on the GPU box: the whole -m gpu suite, smoke(), then tools/measure_round.sh
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_final}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -x -q -m gpu > "$OUT/tests.log" 2>&1
tail -n 4 "$OUT/tests.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
tail -n 2 "$OUT/smoke.log"
tools/measure_round.sh "${1:-r4_final}" > "$OUT/measure.log" 2>&1
tail -n 3 "$OUT/measure.log"
