#!/bin/bash
# SQ counters of the decimator alone (pipeline_depth 0), two passes of eight counters; condensed by hand into
# profiles/r2_fir_pmc.json.  Usage on the GPU box: tools/fir_pmc.sh <out-dir under gpurun_out/>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-firpmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
Q="--steps 2 --warmup 1 --file-run 0 --cpu-samples 0 --detect-steps 0 --host-steps 0 --alone-steps 0 --depth 0"
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $pass --kernel-include-regex "fir_decimate" -d "$OUT/p$i" -o pmc --output-format csv -- \
      python "$GRAFT_REPO_ROOT/bench.py" $Q > "$OUT/p$i.log" 2>&1
done
ls -R "$OUT" | head
