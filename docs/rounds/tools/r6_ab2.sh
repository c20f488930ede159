#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
tools/ab_bench.sh ${1:-ab2} "default|" "band_hist_side 1|--opt band_hist_side=1" "fir_grid 6/CU|--opt fir_grid=1536" "fir_grid 5/CU|--opt fir_grid=1280" \
  "hist_side + fir_grid 6|--opt band_hist_side=1 --opt fir_grid=1536" "lookahead 2|--lookahead 2" "default again|" "band_hist_side 1 again|--opt band_hist_side=1" \
  "12 MHz dense|--density 40 --sample-rate 12000000" "12 MHz dense hist_side|--density 40 --sample-rate 12000000 --opt band_hist_side=1" \
  "12 MHz dense fir_grid 6|--density 40 --sample-rate 12000000 --opt fir_grid=1536" "sparse 2/Ms|--density 2" "sparse 2/Ms hist_side|--density 2 --opt band_hist_side=1"
GPU_MAX_HW_QUEUES=8 tools/ab_bench.sh ${1:-ab2}q "q8 depth 3|" "q8 depth 4|--depth 4" "q8 depth 5|--depth 5" "q8 depth 4 hist_side|--depth 4 --opt band_hist_side=1"
