#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_run10}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
timeout 900 python -m pytest tests/test_gpu_timeshard.py tests/test_gpu_bench_multirank.py tests/test_gpu_cfg5.py -m gpu -q > "$OUT/pytest.txt" 2>&1
tail -6 "$OUT/pytest.txt"
timeout 200 python tools/hop_timing.py 2>/dev/null | tee "$OUT/hop_timing.txt"
export IRDM_BENCH_BACKEND=gloo IRDM_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --shard time --steps 3 --warmup 1 --sample-rate 12000000 $Q 2> "$OUT/ts2.err" | tail -1 > "$OUT/ts2.json"
echo "rc $?"; grep -i "irdm_hip" "$OUT/ts2.err" | head -5; python -c "
import json; d=json.load(open('$OUT/ts2.json')); print(d['value'], d['ms_per_step'], d['config']['records'])"
