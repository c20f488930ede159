#!/bin/bash
# round 5, third GPU call: more batch contexts (pipeline_depth 2..5) x the scan's forms.  The feeding thread waited 0.6 ms of
# every 1.05 ms step for the oldest per-burst chain (profiles/r5_spec_ab.json): the period was chain latency / 3 contexts.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5_c}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
D12="--density 40 --sample-rate 12000000"
run() { # name, args...
  local name=$1; shift
  timeout 150 python bench.py --steps 20 --warmup 6 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"
}
run d2_s1 --depth 2
run d3_s1 --depth 3
run d4_s1 --depth 4
run d5_s1 --depth 5
run d3_s0 --depth 3 --opt band_spec=0
run d4_s0 --depth 4 --opt band_spec=0
run d5_s0 --depth 5 --opt band_spec=0
run d4_s1_h1 --depth 4 --opt band_hist_side=1
run d5_s1_h1 --depth 5 --opt band_hist_side=1
run d4_s1_fir0 --depth 4 --opt fir_order=0
run tl_d4_s1 --depth 4 --opt band_timeline=1
run tl_d4_s0 --depth 4 --opt band_spec=0 --opt band_timeline=1
run d4_s1_b --depth 4
for d in 2 4 5; do
  timeout 150 python bench.py --steps 10 --warmup 6 $Q $D12 --depth $d 2>/dev/null | tail -1 > "$OUT/c5_d$d.json"
  timeout 150 python bench.py --steps 10 --warmup 6 $Q --density 2 --depth $d 2>/dev/null | tail -1 > "$OUT/dens2_d$d.json"
done
timeout 150 python bench.py --steps 10 --warmup 6 $Q --density 40 --depth 4 2>/dev/null | tail -1 > "$OUT/dens40_d4.json"
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
        print(os.path.basename(f), d["value"], d["ms_per_step"], "scan_ms", st.get("scan"), "k1", st.get("fft_mag"), "fir", st.get("fir"), "demod", st.get("demod"),
              "rounds/chunks", sc.get("band_rounds"), sc.get("band_chunks"), "aborts", sc.get("band_aborts"), "undone", sc.get("scan_chain_undone"), "spec", sc.get("spec_scans"),
              "host/step: settle", round(h.get("settle", 0) / n), "older_chain", round(h.get("wait_older_chain", 0) / n), "parity", (find(d, "parity_checked") or {}).get("ok"),
              "kclk", find(d, "kernel_clock_ms"))
        tl = find(d, "scan_timeline_us")
        if tl: print("   ", {k: v[:2] for k, v in tl.items()})
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
P
