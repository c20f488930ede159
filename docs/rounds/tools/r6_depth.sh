#!/bin/bash
# round 6, behind the two-wavefront chains: pipeline_depth sweep (cfg3 and the 12 MHz dense scene), then the whole GPU suite
set -u
T=${1:-depth1}
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
  tools/ab_bench.sh ${T}_d2_$i "depth 2|--depth 2"
  tools/ab_bench.sh ${T}_d3_$i "depth 3|--depth 3"
  tools/ab_bench.sh ${T}_d4_$i "depth 4|--depth 4"
  tools/ab_bench.sh ${T}_d5_$i "depth 5|--depth 5"
done
tools/ab_bench.sh ${T}_c5d2 "12 MHz dense depth 2|--depth 2 --density 40 --sample-rate 12000000"
tools/ab_bench.sh ${T}_c5d3 "12 MHz dense depth 3|--depth 3 --density 40 --sample-rate 12000000"
tools/ab_bench.sh ${T}_c5d4 "12 MHz dense depth 4|--depth 4 --density 40 --sample-rate 12000000"
mkdir -p gpurun_out/$T
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/$T/tests.log 2>&1
tail -n 4 gpurun_out/$T/tests.log
