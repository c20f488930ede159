#!/bin/bash
# round 5, seventh GPU call: CUs kept free of the per-burst chains' streams (IRDM_CHAIN_CU_RESERVE) so that the scan's
# 1024-thread plan passes never wait for the decimator's resident grid to drain; the speculation pass's prep as 256 threads
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5_g}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
D12="--density 40 --sample-rate 12000000"
run() { # name, env, args...
  local name=$1; shift
  local r=$1; shift
  IRDM_CHAIN_CU_RESERVE=$r timeout 150 python bench.py --steps 20 --warmup 6 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"
}
run r0_d3 0 --depth 3
run r8_d3 8 --depth 3
run r16_d3 16 --depth 3
run r4_d3 4 --depth 3
run r8_d4 8 --depth 4
run r8_d5 8 --depth 5
run r16_d5 16 --depth 5
run r8_d3_s0 8 --depth 3 --opt band_spec=0
run r8_d3_b 8 --depth 3
run r0_d3_b 0 --depth 3
run r32_d4 32 --depth 4
run tl_r8_d3 8 --depth 3 --opt band_timeline=1
IRDM_CHAIN_CU_RESERVE=8 timeout 150 python bench.py --steps 10 --warmup 6 $Q $D12 --depth 3 2>/dev/null | tail -1 > "$OUT/c5_r8_d3.json"
IRDM_CHAIN_CU_RESERVE=8 timeout 150 python bench.py --steps 10 --warmup 6 $Q $D12 --depth 5 2>/dev/null | tail -1 > "$OUT/c5_r8_d5.json"
IRDM_CHAIN_CU_RESERVE=8 timeout 150 python bench.py --steps 10 --warmup 6 $Q --density 2 --depth 3 2>/dev/null | tail -1 > "$OUT/dens2_r8_d3.json"
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
              "kclk", find(d, "kernel_clock_ms"))
        tl = find(d, "scan_timeline_us")
        if tl: print("   ", {k: v[:2] for k, v in tl.items()})
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
P
