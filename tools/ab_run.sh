#!/bin/bash
# GPU box: parity (scene zoo + randomised scenes, scan-relevant) and the bench line for the in-tree library and for every
# library under iridium-sniffer_amd/build/ab/*/ (tools/ab_build.sh).  One line per library:
#   <name>  tests: <pytest summary>   Msamples/s  ms/step  scan ms   (depth 1)   |   scan ms alone (depth 0)
# IRDM_LIB selects the library (irdm.py); the CLI tests are not part of this (the binary is looked up next to the library).
set -u
cd "$GRAFT_REPO_ROOT"
DENS=${DENS:-10}
line() {
    python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%9.1f Msps %7.3f ms/step  scan %6.3f ms" % (d["value"], d["ms_per_step"], d["roofline"]["stage_ms"]["scan"]), end="")
except Exception as e:
    print("bench failed (%s)" % e, end="")
PY
}
for lib in iridium-sniffer_amd/libirdm_hip.so iridium-sniffer_amd/build/ab/*/libirdm_hip.so; do
    [ -f "$lib" ] || continue
    name=$(basename "$(dirname "$lib")")
    export IRDM_LIB=$GRAFT_REPO_ROOT/$lib
    t=$(timeout 200 python -m pytest tests/test_gpu_scenes.py -m gpu -x -q 2>&1 | tail -1)
    timeout 60 python bench.py --steps 10 --warmup 2 --cpu-samples 0 --host-steps 0 --density "$DENS" 2>/dev/null | tail -1 > /tmp/ab1.json
    timeout 60 python bench.py --steps 6 --warmup 2 --cpu-samples 0 --host-steps 0 --density "$DENS" --depth 0 2>/dev/null | tail -1 > /tmp/ab0.json
    printf "%-28s tests: %-28s " "$name" "$t"; line /tmp/ab1.json; printf "  |  depth 0: "; line /tmp/ab0.json; echo
done
