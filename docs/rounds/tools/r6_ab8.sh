#!/bin/bash
# round 6: can the scan's passes be placed beside the decimator's resident grid (one 256-register slot free per CU)?
# A: band_walk_wave_kernel<4> capped at 64 registers (four wavefronts fit one slot); B: A + the plan pass as 256 threads; C: plan 256 only
set -u
T=${1:-ab8}
cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
  tools/ab_bench.sh ${T}_cur$i "current|"
  IRDM_LIB=$GRAFT_REPO_ROOT/tests/_build/libirdm_hip_A.so tools/ab_bench.sh ${T}_A$i "walk <= 64 registers|"
  IRDM_LIB=$GRAFT_REPO_ROOT/tests/_build/libirdm_hip_B.so tools/ab_bench.sh ${T}_B$i "walk <= 64 registers, plan 256 threads|"
  IRDM_LIB=$GRAFT_REPO_ROOT/tests/_build/libirdm_hip_C.so tools/ab_bench.sh ${T}_C$i "plan 256 threads|"
done
