#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/p10
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
python -c "
import torch
print(torch.cuda.get_device_properties(0))
import ctypes
" > "$OUT/info.txt" 2>&1
Q="--cpu-samples 0 --host-steps 0 --detect-steps 0 --file-run 0 --alone-steps 0"
for rep in 1 2; do
for c in 1 0; do
timeout 120 python bench.py $Q --opt scan_chain=$c 2>"$OUT/b_c${c}_$rep.err" | tail -1 > "$OUT/b_c${c}_$rep.json"
done
done
IRDM_K1_PRIO=-1 timeout 120 python bench.py $Q --opt scan_chain=1 2>"$OUT/b_k1hi.err" | tail -1 > "$OUT/b_k1hi.json"
GPU_MAX_HW_QUEUES=8 timeout 120 python bench.py $Q --opt scan_chain=1 2>"$OUT/b_q8.err" | tail -1 > "$OUT/b_q8.json"
GPU_MAX_HW_QUEUES=8 IRDM_K1_PRIO=-1 timeout 120 python bench.py $Q --opt scan_chain=1 2>"$OUT/b_q8hi.err" | tail -1 > "$OUT/b_q8hi.json"
for c in 1 0; do
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 --sample-rate 12000000 --opt scan_chain=$c 2>"$OUT/cfg5_c$c.err" | tail -1 > "$OUT/cfg5_c$c.json"
done
timeout 300 python -m pytest tests/test_gpu_compat.py -x -q -m gpu > "$OUT/t1.log" 2>&1
tail -n 3 "$OUT/t1.log"
