#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_run8}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
export IRDM_BENCH_BACKEND=gloo IRDM_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --shard time --steps 2 --warmup 1 --sample-rate 12000000 $Q > "$OUT/ts2.out" 2> "$OUT/ts2.err"
echo "rc $?"; grep -v "^\[W\|amdgpu.ids\|^W0\|\*\*\*" "$OUT/ts2.err" | grep -i "irdm\|error\|rank1\|failed" | head -30; tail -c 600 "$OUT/ts2.out"
unset IRDM_BENCH_BACKEND IRDM_BENCH_SHARE_GPU
{
timeout 120 tools/ubench/k1_bench 13 8192 20 2 200
timeout 120 tools/ubench/k1_bench 14 4096 20 2 400
} > "$OUT/k1_bench.txt" 2>&1
grep -E "p32|r16|differ" "$OUT/k1_bench.txt"
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_bench_multirank.py::test_time_shard_mode_two_ranks > "$OUT/pytest.txt" 2>&1
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
timeout 120 python bench.py --steps 20 --warmup 5 $Q 2>/dev/null | tail -1 > "$OUT/b2.json"; show "$OUT/b2.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --alone-steps 3 --density 40 --sample-rate 12000000 2>/dev/null | tail -1 > "$OUT/c5.json"; show "$OUT/c5.json"
