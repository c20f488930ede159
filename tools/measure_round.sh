#!/bin/bash
# Round-end measurement on the GPU box: bench lines, rocprofv3 kernel stats and the two PMC passes (their own runs, no
# other trace domains).  Usage: tools/measure_round.sh <out-dir under gpurun_out/>;  then, back in the authoring container,
# python profiles/summarize.py gpurun_out/<dir> <tag>.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-final}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout 200 python bench.py 2>"$OUT/b.err" | tail -1 > "$OUT/b.json"
timeout 60 python bench.py --depth 0 --cpu-samples 0 --host-steps 0 2>/dev/null | tail -1 > "$OUT/b0.json"
timeout 60 python bench.py --steps 6 --warmup 2 --cpu-samples 0 --host-steps 0 --density 2 2>/dev/null | tail -1 > "$OUT/d2.json"
timeout 60 python bench.py --steps 6 --warmup 2 --cpu-samples 0 --host-steps 0 --density 40 2>/dev/null | tail -1 > "$OUT/d40.json"
timeout 60 python bench.py --steps 6 --warmup 2 --cpu-samples 0 --host-steps 0 --opt scan_mode=2 2>/dev/null | tail -1 > "$OUT/single_cu.json"
timeout 60 python bench.py --steps 6 --warmup 2 --cpu-samples 0 --host-steps 0 --density 2 --opt scan_updaters=15 2>/dev/null | tail -1 > "$OUT/d2_u15.json"
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats -d "$OUT" -o r1 --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 2 --cpu-samples 0 --host-steps 0 > "$OUT/kt.log" 2>&1
timeout 120 rocprofv3 --pmc FETCH_SIZE -d "$OUT" -o pmc_fetch --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --depth 0 --cpu-samples 0 --host-steps 0 > "$OUT/pmc_fetch.log" 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE -d "$OUT" -o pmc_write --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --depth 0 --cpu-samples 0 --host-steps 0 > "$OUT/pmc_write.log" 2>&1
ls "$OUT"
