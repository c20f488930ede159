#!/bin/bash
# round 5, eighth GPU call: the resident decimator grid claiming its strips from a counter (fir_claim 1 / 0), k1_first variants
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5_h}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
D12="--density 40 --sample-rate 12000000"
timeout 600 python -m pytest tests/test_gpu_fir_reg.py -x -q -m gpu > "$OUT/tests.log" 2>&1
tail -n 3 "$OUT/tests.log"
run() { local name=$1; shift
  timeout 150 python bench.py --steps 20 --warmup 6 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"; }
run c1 
run c0 --opt fir_claim=0
run c1_b
run c0_b --opt fir_claim=0
run c1_k0 --opt k1_first=0
run c1_k0_b --opt k1_first=0
run c1_d4 --depth 4
run c1_d5 --depth 5
run c1_d0 --depth 0
run c1_g2048 --opt fir_grid=2048
run c1_d4_k0 --depth 4 --opt k1_first=0
timeout 150 python bench.py --steps 10 --warmup 6 $Q $D12 2>/dev/null | tail -1 > "$OUT/c5_c1.json"
timeout 150 python bench.py --steps 10 --warmup 6 $Q $D12 --opt fir_claim=0 2>/dev/null | tail -1 > "$OUT/c5_c0.json"
timeout 150 python bench.py --steps 10 --warmup 6 $Q $D12 --opt k1_first=0 2>/dev/null | tail -1 > "$OUT/c5_c1_k0.json"
python - "$OUT" <<'P'
import json, sys, glob, os
def find(d, key):
    if isinstance(d, dict):
        if key in d: return d[key]
        for v in d.values():
            r = find(v, key)
            if r is not None: return r
    return None
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f))
        st = find(d, "stage_ms") or {}
        h = find(d, "host_us_total") or {}
        n = d["steps"] + d["warmup"]
        print(os.path.basename(f), d["value"], d["ms_per_step"], "scan_ms", st.get("scan"), "k1", st.get("fft_mag"), "fir", st.get("fir"), "post", st.get("post"),
              "host/step: settle", round(h.get("settle", 0) / n), "older_chain", round(h.get("wait_older_chain", 0) / n),
              "kclk", find(d, "kernel_clock_ms"), "frac", find(d, "frac"))
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
P
