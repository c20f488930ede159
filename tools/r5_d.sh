#!/bin/bash
# round 5, fourth GPU call: the decimator on the matrix cores (fir_layout 4) -- its tests, then bench lines against the
# VALU kernel at pipeline_depth 2 / 3, alone (depth 0) and in the dense 12 MHz scene
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5_d}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
D12="--density 40 --sample-rate 12000000"
timeout 900 python -m pytest tests/test_gpu_fir_reg.py -x -q -m gpu -k "matrix_core" > "$OUT/tests.log" 2>&1
tail -n 15 "$OUT/tests.log"
run() { # name, args...
  local name=$1; shift
  timeout 150 python bench.py --steps 20 --warmup 6 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"
}
run d0_l4 --depth 0 --opt fir_layout=4
run d0_l3 --depth 0
run d3_l4 --depth 3 --opt fir_layout=4
run d3_l3 --depth 3
run d2_l4 --depth 2 --opt fir_layout=4
run d3_l4_b --depth 3 --opt fir_layout=4
run d3_l3_b --depth 3
timeout 150 python bench.py --steps 10 --warmup 6 $Q $D12 --depth 3 --opt fir_layout=4 2>/dev/null | tail -1 > "$OUT/c5_d3_l4.json"
timeout 150 python bench.py --steps 10 --warmup 6 $Q $D12 --depth 3 2>/dev/null | tail -1 > "$OUT/c5_d3_l3.json"
timeout 150 python bench.py --steps 6 --warmup 3 $Q $D12 --depth 0 --opt fir_layout=4 2>/dev/null | tail -1 > "$OUT/c5_d0_l4.json"
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
              "host/step: settle", round(h.get("settle", 0) / n), "older_chain", round(h.get("wait_older_chain", 0) / n), "parity", (find(d, "parity_checked") or {}).get("ok"),
              "kclk", find(d, "kernel_clock_ms"), "frac", find(d, "frac"))
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
P
