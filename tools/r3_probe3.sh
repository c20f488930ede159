#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/p4
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "10_and_12 or variants or stale or depth_1" > "$OUT/t1.log" 2>&1
timeout 600 python -m pytest tests/test_gpu_cfg5.py tests/test_gpu_scenes.py tests/test_gpu_fir_reg.py -x -q -m gpu -k "cfg5 or 10mhz or fir_reg" > "$OUT/t2.log" 2>&1
Q="--cpu-samples 0 --host-steps 0 --detect-steps 0 --file-run 0"
for s in 2 3 4 6; do
timeout 120 python bench.py $Q --opt fir_strip=$s 2>"$OUT/b_s$s.err" | tail -1 > "$OUT/b_s$s.json"
done
timeout 120 python bench.py $Q --opt fir_layout=2 2>"$OUT/b_l2.err" | tail -1 > "$OUT/b_l2.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 --sample-rate 12000000 2>"$OUT/cfg5.err" | tail -1 > "$OUT/cfg5.json"
tail -3 "$OUT/t1.log" "$OUT/t2.log"
