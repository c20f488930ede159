#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_run11}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
timeout 900 python -m pytest tests -m gpu -q -s > "$OUT/pytest.txt" 2>&1
grep -E "context, 16 Mi|passed|failed|FAILED" "$OUT/pytest.txt" | tail -8
IRDM_CREATE_DEBUG=1 python - <<'P' 2>&1 | grep -v amdgpu.ids | tail -30
import sys, time
sys.path.insert(0, "iridium-sniffer_amd")
import irdm, torch
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for fs, chunk in ((10_000_000, 16 * 1024 * 1024), (10_000_000, 64 * 1024 * 1024), (12_000_000, 64 * 1024 * 1024)):
    f0, t = torch.cuda.mem_get_info()
    t0 = time.perf_counter()
    p = irdm.Pipeline(fs, max_chunk_samples=chunk, max_bursts_per_chunk=4096, pipeline_depth=2)
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    f1, _ = torch.cuda.mem_get_info()
    print("CREATE fs %d chunk %d: %.1f ms, %.2f GB" % (fs, chunk, dt * 1e3, (f0 - f1) / 1e9))
    p.close()
P
show() { python - "$1" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], "clk", r.get("kernel_clock_ms"), "stage", r.get("stage_ms"))
except Exception as e: print(sys.argv[1], "ERR", e)
P
}
timeout 120 python bench.py --steps 20 --warmup 5 $Q 2>"$OUT/b.err" | tail -1 > "$OUT/b.json"; show "$OUT/b.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 --sample-rate 12000000 2>/dev/null | tail -1 > "$OUT/c5.json"; show "$OUT/c5.json"
