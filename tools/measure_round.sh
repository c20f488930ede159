#!/bin/bash
# Round-end measurement on the GPU box: bench lines, rocprofv3 kernel stats and the PMC passes (their own runs, no other
# trace domains).  Usage: tools/measure_round.sh <out-dir under gpurun_out/>;  then, back in the authoring container,
# python profiles/summarize.py gpurun_out/<dir> <tag>.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-final}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
D12="--density 40 --sample-rate 12000000"
timeout 600 python bench.py --steps 20 --warmup 5 2>"$OUT/b.err" | tail -1 > "$OUT/b.json"
timeout 90 python bench.py --steps 20 --warmup 5 --depth 0 $Q 2>/dev/null | tail -1 > "$OUT/b0.json"
# (20 steps behind 6 of warm-up, as the driver's line: over 10 steps behind 3 the dense and the sparse scene read 10-15 % lower)
timeout 90 python bench.py --steps 20 --warmup 6 $Q --density 2 2>/dev/null | tail -1 > "$OUT/d2.json"
timeout 120 python bench.py --steps 20 --warmup 6 $Q --density 40 2>/dev/null | tail -1 > "$OUT/d40.json"
timeout 150 python bench.py --steps 20 --warmup 6 $Q --alone-steps 3 $D12 2>/dev/null | tail -1 > "$OUT/cfg5_12mhz_d40.json"
timeout 90 python bench.py --steps 20 --warmup 5 $Q --alone-steps 3 --opt fir_order=0 2>/dev/null | tail -1 > "$OUT/scalar_fir.json"
timeout 120 python bench.py --shard time --steps 20 --warmup 5 2>/dev/null | tail -1 > "$OUT/cfg4_n1.json"
# (the first eight chunks of that stream: every chunk brings centre bins never seen before -- rotator checkpoint builds)
timeout 120 python bench.py --shard time --steps 6 --warmup 2 2>/dev/null | tail -1 > "$OUT/cfg4_n1_first8chunks.json"
# what the group protocol costs on one GPU (one member plain / handing its state to itself over RCCL)
timeout 300 python tools/group_bench.py > "$OUT/group_bench.txt" 2>/dev/null
# the detector scan's own device timeline (option band_timeline: first workgroup's start / last one's end per pass), in
# run and alone, both scenes
timeout 90 python bench.py $Q --opt band_timeline=1 2>/dev/null | tail -1 > "$OUT/tl.json"
timeout 90 python bench.py --depth 0 $Q --opt band_timeline=1 2>/dev/null | tail -1 > "$OUT/tl_depth0.json"
timeout 120 python bench.py --steps 10 --warmup 3 $Q $D12 --opt band_timeline=1 2>/dev/null | tail -1 > "$OUT/tl_cfg5.json"
timeout 120 python bench.py --steps 6 --warmup 2 --depth 0 $Q $D12 --opt band_timeline=1 2>/dev/null | tail -1 > "$OUT/tl_cfg5_depth0.json"
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py"
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT" -o r1 --output-format csv -- python $B --steps 20 --warmup 5 $Q > "$OUT/kt.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT" -o r1d0 --output-format csv -- python $B --steps 10 --warmup 3 --depth 0 $Q > "$OUT/kt0.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT" -o c5 --output-format csv -- python $B --steps 10 --warmup 3 $Q $D12 > "$OUT/kt_c5.log" 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT" -o c5d0 --output-format csv -- python $B --steps 6 --warmup 2 --depth 0 $Q $D12 > "$OUT/kt_c5d0.log" 2>&1
timeout 120 rocprofv3 --pmc FETCH_SIZE -d "$OUT" -o pmc_fetch --output-format csv -- python $B --steps 2 --warmup 1 --depth 0 $Q > "$OUT/pmc_fetch.log" 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE -d "$OUT" -o pmc_write --output-format csv -- python $B --steps 2 --warmup 1 --depth 0 $Q > "$OUT/pmc_write.log" 2>&1
timeout 120 rocprofv3 --pmc FETCH_SIZE -d "$OUT" -o c5_pmc_fetch --output-format csv -- python $B --steps 2 --warmup 1 --depth 0 $Q $D12 > "$OUT/c5_pmc_fetch.log" 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE -d "$OUT" -o c5_pmc_write --output-format csv -- python $B --steps 2 --warmup 1 --depth 0 $Q $D12 > "$OUT/c5_pmc_write.log" 2>&1
# SQ counters of the decimator alone (pipeline_depth 0), two passes of eight counters, 10 MHz and 12 MHz dense
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $pass --kernel-include-regex "fir_decimate" -d "$OUT/sq$i" -o pmc --output-format csv -- \
      python $B --steps 2 --warmup 1 --depth 0 $Q > "$OUT/sq$i.log" 2>&1
  timeout 200 rocprofv3 --pmc $pass --kernel-include-regex "fir_decimate" -d "$OUT/c5_sq$i" -o pmc --output-format csv -- \
      python $B --steps 2 --warmup 1 --depth 0 $Q $D12 > "$OUT/c5_sq$i.log" 2>&1
done
# (K1's own SQ-counter passes ran on docs/rounds/tools/k1_bench.hip until round 5: profiles/r5_k1_pmc_final.json; the kernel has not changed since)
cd "$GRAFT_REPO_ROOT"
f=$(ls $OUT/*r1_kernel_trace.csv $OUT/*/*r1_kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python tools/trace_gantt.py "$f" 4 > "$OUT/gantt_r1.txt"
find "$OUT" -name "*kernel_trace.csv" -delete
timeout 200 python tools/hop_timing.py > "$OUT/hop_timing.txt" 2>/dev/null
ls "$OUT"
