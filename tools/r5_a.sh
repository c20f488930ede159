#!/bin/bash
# round 5, first GPU call: the whole -m gpu suite with the tail form of the scan (default), the MFMA decimator ubench, and
# A/B bench lines of the scan's forms (band_tail 0 / 1, workgroup width of the walk that carries the tail), with timelines
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5_a}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
D12="--density 40 --sample-rate 12000000"
timeout 900 python -m pytest tests -x -q -m gpu > "$OUT/tests.log" 2>&1
tail -n 5 "$OUT/tests.log"
timeout 300 tools/ubench/mfma_fir 40 2048 512 5 > "$OUT/mfma_fir.txt" 2>&1
timeout 200 tools/ubench/mfma_fir 48 2048 512 5 >> "$OUT/mfma_fir.txt" 2>&1
cat "$OUT/mfma_fir.txt"
run() { # name, args...
  local name=$1; shift
  timeout 120 python bench.py --steps 20 --warmup 5 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"
}
run t1_1024 --opt band_tail=1
run t0 --opt band_tail=0
run t1_512 --opt band_tail=1 --opt band_tail_threads=512
run t1_256 --opt band_tail=1 --opt band_tail_threads=256
run t1_1024b --opt band_tail=1
run t0b --opt band_tail=0
run tl_t1 --opt band_tail=1 --opt band_timeline=1
run tl_t0 --opt band_tail=0 --opt band_timeline=1
run tl_t1_d0 --depth 0 --opt band_tail=1 --opt band_timeline=1
run tl_t1_256 --opt band_tail=1 --opt band_tail_threads=256 --opt band_timeline=1
timeout 120 python bench.py --steps 10 --warmup 3 $Q $D12 --opt band_tail=1 2>/dev/null | tail -1 > "$OUT/c5_t1.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q $D12 --opt band_tail=0 2>/dev/null | tail -1 > "$OUT/c5_t0.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q $D12 --opt band_tail=1 --opt band_timeline=1 2>/dev/null | tail -1 > "$OUT/c5_tl_t1.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 2 --opt band_tail=1 2>/dev/null | tail -1 > "$OUT/d2_t1.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 2 --opt band_tail=0 2>/dev/null | tail -1 > "$OUT/d2_t0.json"
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
              "aborts", sc.get("band_aborts") if isinstance(sc, dict) else sc, "chained", sc.get("scan_chained") if isinstance(sc, dict) else None,
              "host", find(d, "host_ms"))
        tl = find(d, "scan_timeline_us")
        if tl: print("   ", {k: v[:2] for k, v in tl.items()})
        pp = find(d, "plan_phase_us")
        if pp: print("    plan phases", pp)
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
P
