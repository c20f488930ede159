#!/bin/bash
# A round's closing run on the GPU box: the whole -m gpu suite, smoke(), then tools/measure_round.sh (usage:
# gpurun -- 'bash tools/round_end.sh r5_final'; then python profiles/summarize.py gpurun_out/r5_final r5).  "quick" as the
# second argument: the suite, smoke and one default bench line, without the profiler passes.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-final}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -x -q -m gpu > "$OUT/tests.log" 2>&1
tail -n 4 "$OUT/tests.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
tail -n 2 "$OUT/smoke.log"
if [ "${2:-}" = quick ]; then
  timeout 600 python bench.py 2>"$OUT/b.err" | tail -1 > "$OUT/b.json"
  head -c 400 "$OUT/b.json"; echo
else
  tools/measure_round.sh "${1:-final}" > "$OUT/measure.log" 2>&1
  tail -n 3 "$OUT/measure.log"
fi
