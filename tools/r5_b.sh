#!/bin/bash
# round 5, second GPU call: round 0 as a speculation pass beside the previous chunk's scan (band_spec, default on) against
# the classical scan; the whole -m gpu suite first
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5_b}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
D12="--density 40 --sample-rate 12000000"
timeout 900 python -m pytest tests -x -q -m gpu > "$OUT/tests.log" 2>&1
tail -n 5 "$OUT/tests.log"
run() { # name, args...
  local name=$1; shift
  timeout 120 python bench.py --steps 20 --warmup 5 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"
}
run s1 --opt band_spec=1
run s0 --opt band_spec=0
run s1b --opt band_spec=1
run s0b --opt band_spec=0
run s1_h1 --opt band_spec=1 --opt band_hist_side=1
run s0_h1 --opt band_spec=0 --opt band_hist_side=1
run s1_h1b --opt band_spec=1 --opt band_hist_side=1
run tl_s1_h1 --opt band_spec=1 --opt band_hist_side=1 --opt band_timeline=1
run tl_s1 --opt band_spec=1 --opt band_timeline=1
run s1_fir0 --opt band_spec=1 --opt fir_order=0
run s1_d3 --opt band_spec=1 --depth 3
timeout 120 python bench.py --steps 10 --warmup 3 $Q $D12 --opt band_spec=1 2>/dev/null | tail -1 > "$OUT/c5_s1.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q $D12 --opt band_spec=0 2>/dev/null | tail -1 > "$OUT/c5_s0.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q $D12 --opt band_spec=1 --opt band_hist_side=1 2>/dev/null | tail -1 > "$OUT/c5_s1_h1.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q $D12 --opt band_spec=1 --opt band_timeline=1 2>/dev/null | tail -1 > "$OUT/c5_tl_s1.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q $D12 --opt band_spec=1 --opt fir_grid=1536 2>/dev/null | tail -1 > "$OUT/c5_s1_grid1536.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q $D12 --opt band_spec=1 --opt fir_grid=1280 2>/dev/null | tail -1 > "$OUT/c5_s1_grid1280.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 2 --opt band_spec=1 2>/dev/null | tail -1 > "$OUT/d2_s1.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 2 --opt band_spec=0 2>/dev/null | tail -1 > "$OUT/d2_s0.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 --opt band_spec=1 2>/dev/null | tail -1 > "$OUT/d40_s1.json"
timeout 200 python bench.py --shard time --steps 20 --warmup 5 2>/dev/null | tail -1 > "$OUT/cfg4_n1.json"
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
        print(os.path.basename(f), d["value"], d["ms_per_step"], "scan_ms", st.get("scan"), "k1", st.get("fft_mag"), "fir", st.get("fir"),
              "rounds/chunks", sc.get("band_rounds"), sc.get("band_chunks"), "aborts", sc.get("band_aborts"), "chained", sc.get("scan_chained"),
              "undone", sc.get("scan_chain_undone"), "spec", find(d, "spec_scans"), "host", find(d, "host_ms"), "parity", (find(d, "parity_checked") or {}).get("ok"))
        tl = find(d, "scan_timeline_us")
        if tl: print("   ", {k: v[:2] for k, v in tl.items()})
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
P
