#!/bin/bash
# The detector scan's device timeline on the GPU box, for A/B runs of a scan-related option:
#   gpurun -- tools/scan_timeline.sh <out-dir under gpurun_out/> [bench.py options, e.g. --opt fir_strip=2]
# writes the bench lines of the 10 MHz scene and the 12 MHz dense scene, in run and alone (config.scan_timeline_us:
# per pass [us from the first wavefront's start to the last one's end, idle us in front of the pass, launches per chunk]).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-tl}
shift || true
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --detect-steps 0 --file-run 0 --alone-steps 0 --opt band_timeline=1"
D12="--steps 10 --warmup 3 --density 40 --sample-rate 12000000"
timeout 120 python bench.py $Q "$@" 2>"$OUT/err.log" | tail -1 > "$OUT/cfg3.json"
timeout 120 python bench.py $Q --depth 0 "$@" 2>/dev/null | tail -1 > "$OUT/cfg3_alone.json"
timeout 120 python bench.py $Q $D12 "$@" 2>/dev/null | tail -1 > "$OUT/cfg5.json"
timeout 120 python bench.py $Q $D12 --depth 0 "$@" 2>/dev/null | tail -1 > "$OUT/cfg5_alone.json"
python - "$OUT" <<'PY'
import json, sys
for name in ("cfg3", "cfg3_alone", "cfg5", "cfg5_alone"):
    try:
        e = json.load(open("%s/%s.json" % (sys.argv[1], name)))
    except Exception as ex:
        print(name, "no line:", ex)
        continue
    tl = e["config"].get("scan_timeline_us") or {}
    print(name, round(e["value"]), "Msamples/s, scan stage", e["roofline"]["stage_ms"]["scan"], "ms")
    print("   " + "  ".join("%s %.0f/%.0f" % (k, v[0], v[1]) for k, v in tl.items() if v[2] > 0.5))
PY
