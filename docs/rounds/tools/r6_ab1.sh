#!/bin/bash
# round 6: parity of the rebuilt post stage on the GPU, then A/B lines
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-ab1}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_packed.py tests/test_gpu_scenes.py tests/test_gpu_fir_reg.py -x -q -m gpu > "$OUT/tests.log" 2>&1
tail -n 5 "$OUT/tests.log"
tools/ab_bench.sh ${1:-ab1} "default|--alone-steps 3" "post_split 0|--opt post_split=0 --alone-steps 3" "default again|" "post_split 0 again|--opt post_split=0" "12 MHz dense|--density 40 --sample-rate 12000000 --alone-steps 3" "12 MHz dense post_split 0|--density 40 --sample-rate 12000000 --opt post_split=0 --alone-steps 3" "depth 4|--depth 4" "depth 2|--depth 2"
python - <<'P'
import json,glob,os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/%s/line*.json" % "${1:-ab1}")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d["roofline"].get("stage_ms_alone"))
    except Exception as e: print(f, e)
P
