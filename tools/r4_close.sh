#!/bin/bash
# the round's last GPU call: the whole -m gpu suite, smoke(), and the bench lines (no rocprofv3 passes: those are
# tools/measure_round.sh's, taken by tools/r4_final.sh)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_close}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
D12="--density 40 --sample-rate 12000000"
timeout 900 python -m pytest tests -x -q -m gpu > "$OUT/tests.log" 2>&1
tail -n 3 "$OUT/tests.log"
timeout 120 python bench.py --shard time --steps 6 --warmup 2 2>/dev/null | tail -1 > "$OUT/cfg4_n1.json"
timeout 120 python bench.py --shard time --steps 20 --warmup 5 2>/dev/null | tail -1 > "$OUT/cfg4_n1_20.json"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
tail -n 1 "$OUT/smoke.log" | cut -c1-100
timeout 600 python bench.py --steps 20 --warmup 5 2>"$OUT/b.err" | tail -1 > "$OUT/b.json"
timeout 90 python bench.py --steps 20 --warmup 5 --depth 0 $Q 2>/dev/null | tail -1 > "$OUT/b0.json"
timeout 90 python bench.py --steps 10 --warmup 3 $Q --density 2 2>/dev/null | tail -1 > "$OUT/d2.json"
timeout 90 python bench.py --steps 10 --warmup 3 $Q --density 40 2>/dev/null | tail -1 > "$OUT/d40.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q --alone-steps 3 $D12 2>/dev/null | tail -1 > "$OUT/cfg5_12mhz_d40.json"
timeout 90 python bench.py --steps 20 --warmup 5 $Q --alone-steps 3 --opt fir_order=0 2>/dev/null | tail -1 > "$OUT/scalar_fir.json"
python - "$OUT" <<'P'
import json, sys
for n in ("b", "b0", "cfg4_n1", "cfg4_n1_20", "d2", "d40", "cfg5_12mhz_d40", "scalar_fir"):
    try:
        d = json.load(open("%s/%s.json" % (sys.argv[1], n))); print(n, d["value"], d["ms_per_step"])
    except Exception as e:
        print(n, "ERR", e)
P
