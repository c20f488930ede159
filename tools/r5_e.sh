#!/bin/bash
# round 5, fifth GPU call: the next chunk's scan chained BEFORE the feeding thread waits for the oldest chain
# (scan_chain_early) x pipeline_depth x speculation pass x history copy on the side stream; a kernel trace of the best
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5_e}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
D12="--density 40 --sample-rate 12000000"
run() { # name, args...
  local name=$1; shift
  timeout 150 python bench.py --steps 20 --warmup 6 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"
}
run d3_e1_s1 --depth 3
run d3_e0_s1 --depth 3 --opt scan_chain_early=0
run d3_e1_s0 --depth 3 --opt band_spec=0
run d2_e1_s1 --depth 2
run d4_e1_s1 --depth 4
run d5_e1_s1 --depth 5
run d3_e1_s1_h1 --depth 3 --opt band_hist_side=1
run d4_e1_s1_h1 --depth 4 --opt band_hist_side=1
run d4_e1_s0 --depth 4 --opt band_spec=0
run d3_e1_s1_b --depth 3
run tl_d3_e1_s1 --depth 3 --opt band_timeline=1
for d in 3 4; do
  timeout 150 python bench.py --steps 10 --warmup 6 $Q $D12 --depth $d 2>/dev/null | tail -1 > "$OUT/c5_d$d.json"
  timeout 150 python bench.py --steps 10 --warmup 6 $Q --density 2 --depth $d 2>/dev/null | tail -1 > "$OUT/dens2_d$d.json"
done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT" -o kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 6 --depth 3 $Q > "$OUT/kt.log" 2>&1
cd "$GRAFT_REPO_ROOT"
ls "$OUT" | head -50
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
        sc = find(d, "scan") or {}
        h = find(d, "host_us_total") or {}
        n = d["steps"] + d["warmup"]
        print(os.path.basename(f), d["value"], d["ms_per_step"], "scan_ms", st.get("scan"), "k1", st.get("fft_mag"), "fir", st.get("fir"), "post", st.get("post"),
              "rounds/chunks", sc.get("band_rounds"), sc.get("band_chunks"), "undone", sc.get("scan_chain_undone"), "spec", sc.get("spec_scans"),
              "host/step: settle", round(h.get("settle", 0) / n), "older_chain", round(h.get("wait_older_chain", 0) / n), "parity", (find(d, "parity_checked") or {}).get("ok"),
              "kclk", find(d, "kernel_clock_ms"))
        tl = find(d, "scan_timeline_us")
        if tl: print("   ", {k: v[:2] for k, v in tl.items()})
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
P
