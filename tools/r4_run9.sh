#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_run9}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
timeout 900 python -m pytest tests -m gpu -q > "$OUT/pytest.txt" 2>&1
tail -8 "$OUT/pytest.txt"
show() { python - "$1" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], "clk", r.get("kernel_clock_ms"), "stage", r.get("stage_ms") or r.get("stage_ms_rank0_last_step"), "alone", r.get("stage_ms_alone"))
except Exception as e: print(sys.argv[1], "ERR", e)
P
}
timeout 120 python bench.py --steps 20 --warmup 5 $Q --alone-steps 3 2>"$OUT/b.err" | tail -1 > "$OUT/b.json"; show "$OUT/b.json"
timeout 120 python bench.py --steps 20 --warmup 5 $Q --depth 0 2>/dev/null | tail -1 > "$OUT/b0.json"; show "$OUT/b0.json"
timeout 160 python bench.py --shard time --steps 8 --warmup 3 2>"$OUT/ts.err" | tail -1 > "$OUT/cfg4_n1.json"; show "$OUT/cfg4_n1.json"
export IRDM_BENCH_BACKEND=gloo IRDM_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --shard time --steps 3 --warmup 1 --sample-rate 12000000 $Q 2> "$OUT/ts2.err" | tail -1 > "$OUT/ts2.json"
echo "rc $?"; grep -i "irdm_hip" "$OUT/ts2.err" | head -5; show "$OUT/ts2.json"
