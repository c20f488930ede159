#!/bin/bash
# round 5, sixth GPU call: the whole -m gpu suite at the round's defaults (speculation pass, early chaining; bench at
# pipeline_depth 3), decimator grid sizes at depth 3, and the full default bench line
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5_f}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
timeout 900 python -m pytest tests -x -q -m gpu > "$OUT/tests.log" 2>&1
tail -n 5 "$OUT/tests.log"
run() { # name, args...
  local name=$1; shift
  timeout 150 python bench.py --steps 20 --warmup 6 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"
}
run g_default
run g_1536 --opt fir_grid=1536
run g_1280 --opt fir_grid=1280
run g_2048 --opt fir_grid=2048
run g_0 --opt fir_grid=0
run k1first2 --opt k1_first=2
run k1first0 --opt k1_first=0
run g_default_b
timeout 600 python bench.py 2>"$OUT/full.err" | tail -1 > "$OUT/full.json"
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
