#!/bin/bash
# Round-end measurement on the GPU box: bench lines, rocprofv3 kernel stats and the two PMC passes (their own runs, no
# other trace domains).  Usage: tools/measure_round.sh <out-dir under gpurun_out/>;  then, back in the authoring container,
# python profiles/summarize.py gpurun_out/<dir> <tag>.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-final}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
timeout 400 python bench.py 2>"$OUT/b.err" | tail -1 > "$OUT/b.json"
timeout 90 python bench.py --depth 0 $Q 2>/dev/null | tail -1 > "$OUT/b0.json"
timeout 90 python bench.py --steps 10 --warmup 3 $Q --density 2 2>/dev/null | tail -1 > "$OUT/d2.json"
timeout 90 python bench.py --steps 10 --warmup 3 $Q --density 40 2>/dev/null | tail -1 > "$OUT/d40.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 --sample-rate 12000000 2>/dev/null | tail -1 > "$OUT/cfg5_12mhz_d40.json"
timeout 90 python bench.py --steps 6 --warmup 2 $Q --opt scan_mode=3 2>/dev/null | tail -1 > "$OUT/legacy_scan.json"
timeout 120 python bench.py --shard time --steps 6 --warmup 2 2>/dev/null | tail -1 > "$OUT/cfg4_n1.json"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT" -o r1 --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 5 $Q > "$OUT/kt.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT" -o r1d0 --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --depth 0 $Q > "$OUT/kt0.log" 2>&1
timeout 120 rocprofv3 --pmc FETCH_SIZE -d "$OUT" -o pmc_fetch --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --depth 0 $Q > "$OUT/pmc_fetch.log" 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE -d "$OUT" -o pmc_write --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --depth 0 $Q > "$OUT/pmc_write.log" 2>&1
ls "$OUT"
