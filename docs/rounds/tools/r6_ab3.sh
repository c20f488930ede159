#!/bin/bash
# same-box A/B: the library before the prune (tests/_build/libirdm_hip_preprune.so, built from commit c37d38e) against the current one
set -u
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_timed_config.py tests/test_gpu_ingest.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do
  tools/ab_bench.sh ${1:-ab3}_new$i "current|" 
  IRDM_LIB=$GRAFT_REPO_ROOT/tests/_build/libirdm_hip_preprune.so tools/ab_bench.sh ${1:-ab3}_old$i "before the prune|"
done
tools/ab_bench.sh ${1:-ab3}_c5new "current 12 MHz dense|--density 40 --sample-rate 12000000"
IRDM_LIB=$GRAFT_REPO_ROOT/tests/_build/libirdm_hip_preprune.so tools/ab_bench.sh ${1:-ab3}_c5old "before the prune 12 MHz dense|--density 40 --sample-rate 12000000"
