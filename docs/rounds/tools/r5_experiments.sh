#!/bin/bash
# Round 5's experiment runs on the GPU box, one function per gpurun call (tools/r5_experiments.sh <a..h> [out-dir]); the
# condensed results (profiles/condense_runs.py) are the profiles/r5_*.json files named in DESIGN.md.  (tools/measure_round.sh is the round-end measurement.)
set -u
EXP=${1:?which experiment: a .. w}
OUT=$GRAFT_REPO_ROOT/gpurun_out/${2:-r5_$EXP}
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"

# round 5, first GPU call: the whole -m gpu suite with the tail form of the scan (default), the MFMA decimator ubench, and
# A/B bench lines of the scan's forms (band_tail 0 / 1, workgroup width of the walk that carries the tail), with timelines
exp_a() {
  Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
  D12="--density 40 --sample-rate 12000000"
  timeout 900 python -m pytest tests -x -q -m gpu > "$OUT/tests.log" 2>&1
  tail -n 5 "$OUT/tests.log"
  timeout 300 tools/ubench/mfma_fir 40 2048 512 5 > "$OUT/mfma_fir.txt" 2>&1
  timeout 200 tools/ubench/mfma_fir 48 2048 512 5 >> "$OUT/mfma_fir.txt" 2>&1
  cat "$OUT/mfma_fir.txt"
  run() { # name, args...
    local name=$1; shift
    timeout 120 python bench.py --steps 20 --warmup 5 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"
  }
  run t1_1024 --opt band_tail=1
  run t0 --opt band_tail=0
  run t1_512 --opt band_tail=1 --opt band_tail_threads=512
  run t1_256 --opt band_tail=1 --opt band_tail_threads=256
  run t1_1024b --opt band_tail=1
  run t0b --opt band_tail=0
  run tl_t1 --opt band_tail=1 --opt band_timeline=1
  run tl_t0 --opt band_tail=0 --opt band_timeline=1
  run tl_t1_d0 --depth 0 --opt band_tail=1 --opt band_timeline=1
  run tl_t1_256 --opt band_tail=1 --opt band_tail_threads=256 --opt band_timeline=1
  timeout 120 python bench.py --steps 10 --warmup 3 $Q $D12 --opt band_tail=1 2>/dev/null | tail -1 > "$OUT/c5_t1.json"
  timeout 120 python bench.py --steps 10 --warmup 3 $Q $D12 --opt band_tail=0 2>/dev/null | tail -1 > "$OUT/c5_t0.json"
  timeout 120 python bench.py --steps 10 --warmup 3 $Q $D12 --opt band_tail=1 --opt band_timeline=1 2>/dev/null | tail -1 > "$OUT/c5_tl_t1.json"
  timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 2 --opt band_tail=1 2>/dev/null | tail -1 > "$OUT/d2_t1.json"
  timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 2 --opt band_tail=0 2>/dev/null | tail -1 > "$OUT/d2_t0.json"
}

# round 5, second GPU call: round 0 as a speculation pass beside the previous chunk's scan (band_spec, default on) against
# the classical scan; the whole -m gpu suite first
exp_b() {
  Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
  D12="--density 40 --sample-rate 12000000"
  timeout 900 python -m pytest tests -x -q -m gpu > "$OUT/tests.log" 2>&1
  tail -n 5 "$OUT/tests.log"
  run() { # name, args...
    local name=$1; shift
    timeout 120 python bench.py --steps 20 --warmup 5 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"
  }
  run s1 --opt band_spec=1
  run s0 --opt band_spec=0
  run s1b --opt band_spec=1
  run s0b --opt band_spec=0
  run s1_h1 --opt band_spec=1 --opt band_hist_side=1
  run s0_h1 --opt band_spec=0 --opt band_hist_side=1
  run s1_h1b --opt band_spec=1 --opt band_hist_side=1
  run tl_s1_h1 --opt band_spec=1 --opt band_hist_side=1 --opt band_timeline=1
  run tl_s1 --opt band_spec=1 --opt band_timeline=1
  run s1_fir0 --opt band_spec=1 --opt fir_order=0
  run s1_d3 --opt band_spec=1 --depth 3
  timeout 120 python bench.py --steps 10 --warmup 3 $Q $D12 --opt band_spec=1 2>/dev/null | tail -1 > "$OUT/c5_s1.json"
  timeout 120 python bench.py --steps 10 --warmup 3 $Q $D12 --opt band_spec=0 2>/dev/null | tail -1 > "$OUT/c5_s0.json"
  timeout 120 python bench.py --steps 10 --warmup 3 $Q $D12 --opt band_spec=1 --opt band_hist_side=1 2>/dev/null | tail -1 > "$OUT/c5_s1_h1.json"
  timeout 120 python bench.py --steps 10 --warmup 3 $Q $D12 --opt band_spec=1 --opt band_timeline=1 2>/dev/null | tail -1 > "$OUT/c5_tl_s1.json"
  timeout 120 python bench.py --steps 10 --warmup 3 $Q $D12 --opt band_spec=1 --opt fir_grid=1536 2>/dev/null | tail -1 > "$OUT/c5_s1_grid1536.json"
  timeout 120 python bench.py --steps 10 --warmup 3 $Q $D12 --opt band_spec=1 --opt fir_grid=1280 2>/dev/null | tail -1 > "$OUT/c5_s1_grid1280.json"
  timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 2 --opt band_spec=1 2>/dev/null | tail -1 > "$OUT/d2_s1.json"
  timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 2 --opt band_spec=0 2>/dev/null | tail -1 > "$OUT/d2_s0.json"
  timeout 120 python bench.py --steps 10 --warmup 3 $Q --density 40 --opt band_spec=1 2>/dev/null | tail -1 > "$OUT/d40_s1.json"
  timeout 200 python bench.py --shard time --steps 20 --warmup 5 2>/dev/null | tail -1 > "$OUT/cfg4_n1.json"
}

# round 5, third GPU call: more batch contexts (pipeline_depth 2..5) x the scan's forms.  The feeding thread waited 0.6 ms of
# every 1.05 ms step for the oldest per-burst chain (profiles/r5_spec_ab.json): the period was chain latency / 3 contexts.
exp_c() {
  Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
  D12="--density 40 --sample-rate 12000000"
  run() { # name, args...
    local name=$1; shift
    timeout 150 python bench.py --steps 20 --warmup 6 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"
  }
  run d2_s1 --depth 2
  run d3_s1 --depth 3
  run d4_s1 --depth 4
  run d5_s1 --depth 5
  run d3_s0 --depth 3 --opt band_spec=0
  run d4_s0 --depth 4 --opt band_spec=0
  run d5_s0 --depth 5 --opt band_spec=0
  run d4_s1_h1 --depth 4 --opt band_hist_side=1
  run d5_s1_h1 --depth 5 --opt band_hist_side=1
  run d4_s1_fir0 --depth 4 --opt fir_order=0
  run tl_d4_s1 --depth 4 --opt band_timeline=1
  run tl_d4_s0 --depth 4 --opt band_spec=0 --opt band_timeline=1
  run d4_s1_b --depth 4
  for d in 2 4 5; do
    timeout 150 python bench.py --steps 10 --warmup 6 $Q $D12 --depth $d 2>/dev/null | tail -1 > "$OUT/c5_d$d.json"
    timeout 150 python bench.py --steps 10 --warmup 6 $Q --density 2 --depth $d 2>/dev/null | tail -1 > "$OUT/dens2_d$d.json"
  done
  timeout 150 python bench.py --steps 10 --warmup 6 $Q --density 40 --depth 4 2>/dev/null | tail -1 > "$OUT/dens40_d4.json"
}

# round 5, fourth GPU call: the decimator on the matrix cores (fir_layout 4) -- its tests, then bench lines against the
# VALU kernel at pipeline_depth 2 / 3, alone (depth 0) and in the dense 12 MHz scene
exp_d() {
  Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
  D12="--density 40 --sample-rate 12000000"
  timeout 900 python -m pytest tests/test_gpu_fir_reg.py -x -q -m gpu -k "matrix_core" > "$OUT/tests.log" 2>&1
  tail -n 15 "$OUT/tests.log"
  run() { # name, args...
    local name=$1; shift
    timeout 150 python bench.py --steps 20 --warmup 6 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"
  }
  run d0_l4 --depth 0 --opt fir_layout=4
  run d0_l3 --depth 0
  run d3_l4 --depth 3 --opt fir_layout=4
  run d3_l3 --depth 3
  run d2_l4 --depth 2 --opt fir_layout=4
  run d3_l4_b --depth 3 --opt fir_layout=4
  run d3_l3_b --depth 3
  timeout 150 python bench.py --steps 10 --warmup 6 $Q $D12 --depth 3 --opt fir_layout=4 2>/dev/null | tail -1 > "$OUT/c5_d3_l4.json"
  timeout 150 python bench.py --steps 10 --warmup 6 $Q $D12 --depth 3 2>/dev/null | tail -1 > "$OUT/c5_d3_l3.json"
  timeout 150 python bench.py --steps 6 --warmup 3 $Q $D12 --depth 0 --opt fir_layout=4 2>/dev/null | tail -1 > "$OUT/c5_d0_l4.json"
}

# round 5, fifth GPU call: the next chunk's scan chained BEFORE the feeding thread waits for the oldest chain
# (scan_chain_early) x pipeline_depth x speculation pass x history copy on the side stream; a kernel trace of the best
exp_e() {
  Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
  D12="--density 40 --sample-rate 12000000"
  run() { # name, args...
    local name=$1; shift
    timeout 150 python bench.py --steps 20 --warmup 6 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"
  }
  run d3_e1_s1 --depth 3
  run d3_e0_s1 --depth 3 --opt scan_chain_early=0
  run d3_e1_s0 --depth 3 --opt band_spec=0
  run d2_e1_s1 --depth 2
  run d4_e1_s1 --depth 4
  run d5_e1_s1 --depth 5
  run d3_e1_s1_h1 --depth 3 --opt band_hist_side=1
  run d4_e1_s1_h1 --depth 4 --opt band_hist_side=1
  run d4_e1_s0 --depth 4 --opt band_spec=0
  run d3_e1_s1_b --depth 3
  run tl_d3_e1_s1 --depth 3 --opt band_timeline=1
  for d in 3 4; do
    timeout 150 python bench.py --steps 10 --warmup 6 $Q $D12 --depth $d 2>/dev/null | tail -1 > "$OUT/c5_d$d.json"
    timeout 150 python bench.py --steps 10 --warmup 6 $Q --density 2 --depth $d 2>/dev/null | tail -1 > "$OUT/dens2_d$d.json"
  done
  cd /tmp && export TMPDIR=/tmp
  timeout 200 rocprofv3 --kernel-trace --stats -d "$OUT" -o kt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 6 --depth 3 $Q > "$OUT/kt.log" 2>&1
  cd "$GRAFT_REPO_ROOT"
  ls "$OUT" | head -50
}

# round 5, sixth GPU call: the whole -m gpu suite at the round's defaults (speculation pass, early chaining; bench at
# pipeline_depth 3), decimator grid sizes at depth 3, and the full default bench line
exp_f() {
  Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
  timeout 900 python -m pytest tests -x -q -m gpu > "$OUT/tests.log" 2>&1
  tail -n 5 "$OUT/tests.log"
  run() { # name, args...
    local name=$1; shift
    timeout 150 python bench.py --steps 20 --warmup 6 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"
  }
  run g_default
  run g_1536 --opt fir_grid=1536
  run g_1280 --opt fir_grid=1280
  run g_2048 --opt fir_grid=2048
  run g_0 --opt fir_grid=0
  run k1first2 --opt k1_first=2
  run k1first0 --opt k1_first=0
  run g_default_b
  timeout 600 python bench.py 2>"$OUT/full.err" | tail -1 > "$OUT/full.json"
}

# round 5, seventh GPU call: CUs kept free of the per-burst chains' streams (IRDM_CHAIN_CU_RESERVE) so that the scan's
# 1024-thread plan passes never wait for the decimator's resident grid to drain; the speculation pass's prep as 256 threads
exp_g() {
  Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
  D12="--density 40 --sample-rate 12000000"
  run() { # name, env, args...
    local name=$1; shift
    local r=$1; shift
    IRDM_CHAIN_CU_RESERVE=$r timeout 150 python bench.py --steps 20 --warmup 6 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"
  }
  run r0_d3 0 --depth 3
  run r8_d3 8 --depth 3
  run r16_d3 16 --depth 3
  run r4_d3 4 --depth 3
  run r8_d4 8 --depth 4
  run r8_d5 8 --depth 5
  run r16_d5 16 --depth 5
  run r8_d3_s0 8 --depth 3 --opt band_spec=0
  run r8_d3_b 8 --depth 3
  run r0_d3_b 0 --depth 3
  run r32_d4 32 --depth 4
  run tl_r8_d3 8 --depth 3 --opt band_timeline=1
  IRDM_CHAIN_CU_RESERVE=8 timeout 150 python bench.py --steps 10 --warmup 6 $Q $D12 --depth 3 2>/dev/null | tail -1 > "$OUT/c5_r8_d3.json"
  IRDM_CHAIN_CU_RESERVE=8 timeout 150 python bench.py --steps 10 --warmup 6 $Q $D12 --depth 5 2>/dev/null | tail -1 > "$OUT/c5_r8_d5.json"
  IRDM_CHAIN_CU_RESERVE=8 timeout 150 python bench.py --steps 10 --warmup 6 $Q --density 2 --depth 3 2>/dev/null | tail -1 > "$OUT/dens2_r8_d3.json"
}

# round 5, eighth GPU call: the resident decimator grid claiming its strips from a counter (fir_claim 1 / 0), k1_first variants
exp_h() {
  Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
  D12="--density 40 --sample-rate 12000000"
  timeout 600 python -m pytest tests/test_gpu_fir_reg.py -x -q -m gpu > "$OUT/tests.log" 2>&1
  tail -n 3 "$OUT/tests.log"
  run() { local name=$1; shift
    timeout 150 python bench.py --steps 20 --warmup 6 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"; }
  run c1 
  run c0 --opt fir_claim=0
  run c1_b
  run c0_b --opt fir_claim=0
  run c1_k0 --opt k1_first=0
  run c1_k0_b --opt k1_first=0
  run c1_d4 --depth 4
  run c1_d5 --depth 5
  run c1_d0 --depth 0
  run c1_g2048 --opt fir_grid=2048
  run c1_d4_k0 --depth 4 --opt k1_first=0
  timeout 150 python bench.py --steps 10 --warmup 6 $Q $D12 2>/dev/null | tail -1 > "$OUT/c5_c1.json"
  timeout 150 python bench.py --steps 10 --warmup 6 $Q $D12 --opt fir_claim=0 2>/dev/null | tail -1 > "$OUT/c5_c0.json"
  timeout 150 python bench.py --steps 10 --warmup 6 $Q $D12 --opt k1_first=0 2>/dev/null | tail -1 > "$OUT/c5_c1_k0.json"
}

# round 5, ninth GPU call: two chunks begun ahead (K1 of chunk k + 2 on the GPU a period early) against one
exp_i() {
  Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
  D12="--density 40 --sample-rate 12000000"
  timeout 900 python -m pytest tests/test_gpu_ingest.py -x -q -m gpu > "$OUT/tests.log" 2>&1
  tail -n 3 "$OUT/tests.log"
  run() { local name=$1; shift
    timeout 150 python bench.py --steps 20 --warmup 6 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"; }
  run l1
  run l2 --lookahead 2
  run l1_b
  run l2_b --lookahead 2
  run l2_d4 --lookahead 2 --depth 4
  run l2_d5 --lookahead 2 --depth 5
  run l2_k0 --lookahead 2 --opt k1_first=0
  run l2_d4_k0 --lookahead 2 --depth 4 --opt k1_first=0
  run l2_s0 --lookahead 2 --opt band_spec=0
  run l2_d2 --lookahead 2 --depth 2
  timeout 150 python bench.py --steps 10 --warmup 6 $Q $D12 --lookahead 2 2>/dev/null | tail -1 > "$OUT/c5_l2.json"
  timeout 150 python bench.py --steps 10 --warmup 6 $Q $D12 --lookahead 2 --opt k1_first=0 2>/dev/null | tail -1 > "$OUT/c5_l2_k0.json"
  timeout 150 python bench.py --steps 10 --warmup 6 $Q --density 2 --lookahead 2 2>/dev/null | tail -1 > "$OUT/dens2_l2.json"
}

# j: the final commit: whole suite, smoke, the default line (round_end quick), then the sums-pass restart on and off on the
# sparse scene, the default scene and the 12 MHz dense one
exp_j() {
  Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
  D12="--density 40 --sample-rate 12000000"
  bash tools/round_end.sh "$(basename "$OUT")" quick
  run() { local name=$1; shift
    timeout 150 python bench.py --steps 20 --warmup 6 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"; }
  run d2_r1 --density 2
  run d2_r0 --density 2 --opt band_sum_restart=0
  run c3_r1
  run c3_r0 --opt band_sum_restart=0
  run d2_r1_b --density 2
  run d2_r0_b --density 2 --opt band_sum_restart=0
  run d2_r1_tl --density 2 --opt band_timeline=1
  run d2_r0_tl --density 2 --opt band_sum_restart=0 --opt band_timeline=1
  run c5_r1 $D12
  run c5_r0 $D12 --opt band_sum_restart=0
}

# k: demod_seq with the loop skewed by one symbol: the stage-C / libm / ingest tests, then the default line twice, every
# stage serial (pipeline_depth 0: the kernel alone), the sparse and the dense scene and one stream in time-chunks
exp_k() {
  Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
  D12="--density 40 --sample-rate 12000000"
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ingest.py tests/test_gpu_scenes.py -x -q -m gpu > "$OUT/tests.log" 2>&1
  tail -n 3 "$OUT/tests.log"
  run() { local name=$1; shift
    timeout 150 python bench.py --steps 20 --warmup 6 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"; }
  run c3
  run c3_b
  run c3_d0 --depth 0
  run c3_d2 --depth 2
  run d2 --density 2
  run c5 $D12
  timeout 200 python bench.py --shard time --steps 20 --warmup 5 2>/dev/null | tail -1 > "$OUT/cfg4_n1.json"
  timeout 120 python bench.py --shard time --steps 6 --warmup 2 2>/dev/null | tail -1 > "$OUT/cfg4_n1_first8chunks.json"
  timeout 200 python bench.py --steps 20 --warmup 5 --cpu-samples 0 --file-run 0 2>/dev/null | tail -1 > "$OUT/b_alone.json"
  python - "$OUT" <<'P'
import json, sys
d = json.load(open(sys.argv[1] + "/b_alone.json"))
def find(o, k):
    if isinstance(o, dict):
        if k in o: return o[k]
        for v in o.values():
            r = find(v, k)
            if r is not None: return r
    return None
print("alone", find(d, "stage_ms_alone"), "kernel_ms_alone", find(d, "kernel_ms_alone"))
print("cfg4_n1", json.load(open(sys.argv[1] + "/cfg4_n1.json"))["value"], "first eight chunks", json.load(open(sys.argv[1] + "/cfg4_n1_first8chunks.json"))["value"])
P
}

# l: the group API on the real library and the real RCCL (one member handing its state to itself), the C binary's --gpus
exp_l() {
  timeout 1200 python -m pytest tests/test_gpu_group.py -x -q -m gpu > "$OUT/tests.log" 2>&1
  tail -n 25 "$OUT/tests.log"
}

# m: rot_phase's stores through an LDS transpose: parity tests, the default line twice, every stage serial; what the group
# protocol costs on one GPU (tools/group_bench.py); does RCCL write to the C binary's stdout?
exp_m() {
  Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_group.py -x -q -m gpu > "$OUT/tests.log" 2>&1
  tail -n 3 "$OUT/tests.log"
  run() { local name=$1; shift
    timeout 150 python bench.py --steps 20 --warmup 6 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"; }
  run c3
  run c3_b
  run c3_d0 --depth 0
  run c5 --density 40 --sample-rate 12000000
  timeout 200 python bench.py --steps 20 --warmup 5 --cpu-samples 0 --file-run 0 2>/dev/null | tail -1 > "$OUT/b_alone.json"
  python - "$OUT" <<'P'
import json, sys
d = json.load(open(sys.argv[1] + "/b_alone.json"))
def find(o, k):
    if isinstance(o, dict):
        if k in o: return o[k]
        for v in o.values():
            r = find(v, k)
            if r is not None: return r
    return None
print("alone", find(d, "stage_ms_alone"))
P
  timeout 400 python tools/group_bench.py > "$OUT/group_bench.json" 2>"$OUT/group_bench.err"
  cat "$OUT/group_bench.json"; tail -n 3 "$OUT/group_bench.err"
  timeout 300 python tools/group_bench.py --sample-rate 10000000 --depth 3 > "$OUT/group_bench_10mhz.json" 2>>"$OUT/group_bench.err"
  cat "$OUT/group_bench_10mhz.json"
  # the C binary with a loopback group: anything but RAW lines on stdout?
  python - <<'P' > "$OUT/cli_stdout.txt" 2>&1
import os, subprocess, sys
sys.path.insert(0, "tests"); sys.path.insert(0, "iridium-sniffer_amd")
import numpy as np
import test_gpu_timeshard as G
iq, chunk, ov = G._stream("2mhz")
np.ascontiguousarray(iq).tofile("/tmp/g.cf32")
out = subprocess.run(["iridium-sniffer_amd/iridium-sniffer-hip", "-f", "/tmp/g.cf32", "-r", "2000000", "--gpus", "1", "--group-loopback",
                      "--chunk", str(chunk), "--timing"], capture_output=True, text=True)
lines = out.stdout.splitlines()
print("rc", out.returncode, "stdout lines", len(lines), "not RAW:", [l for l in lines if not l.startswith("RAW:")][:10])
print("stderr:", out.stderr[-600:])
P
  cat "$OUT/cli_stdout.txt"
}

# n: exp_m again after rot_phase went back to its direct stores (the LDS transpose measured slower: post stage alone 0.34 ->
# 0.62 ms), the group's same-device copies by kernel, librccl's banner kept off stdout
exp_n() { exp_m; }

# o: rot_phase with its phases leaving as rows through LDS, skewed by a tile and without barriers (rot_store 1) against the
# direct stores (rot_store 0), alternating on one box; the group bench with the wide copy kernel
exp_o() {
  Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_group.py -x -q -m gpu > "$OUT/tests.log" 2>&1
  tail -n 3 "$OUT/tests.log"
  run() { local name=$1; shift
    timeout 150 python bench.py --steps 20 --warmup 6 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"; }
  run rs1
  run rs0 --opt rot_store=0
  run rs1_b
  run rs0_b --opt rot_store=0
  run rs1_d0 --depth 0
  run rs0_d0 --depth 0 --opt rot_store=0
  run rs1_c
  run rs0_c --opt rot_store=0
  run rs1_c5 --density 40 --sample-rate 12000000
  run rs0_c5 --density 40 --sample-rate 12000000 --opt rot_store=0
  timeout 400 python tools/group_bench.py > "$OUT/group_bench.txt" 2>"$OUT/group_bench.err"
  cat "$OUT/group_bench.txt"
}

# p: where the frames differ with rot_store 1 on the hardware (tools/rot_store_debug.py); the PLL's cosf / sinf as glibc's
# polynomial: the parity tests and the default line, every stage serial, with the demod stage alone
exp_p() {
  Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
  timeout 300 python tools/rot_store_debug.py > "$OUT/rot_store_debug.txt" 2>&1
  cat "$OUT/rot_store_debug.txt" | cut -c1-260
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_libm.py -x -q -m gpu > "$OUT/tests.log" 2>&1
  tail -n 3 "$OUT/tests.log"
  run() { local name=$1; shift
    timeout 150 python bench.py --steps 20 --warmup 6 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"; }
  run c3
  run c3_b
  run c3_d0 --depth 0
  timeout 200 python bench.py --steps 20 --warmup 5 --cpu-samples 0 --file-run 0 2>/dev/null | tail -1 > "$OUT/b_alone.json"
  python - "$OUT" <<'P'
import json, sys
d = json.load(open(sys.argv[1] + "/b_alone.json"))
def find(o, k):
    if isinstance(o, dict):
        if k in o: return o[k]
        for v in o.values():
            r = find(v, k)
            if r is not None: return r
    return None
print("alone", find(d, "stage_ms_alone"))
P
}

# q: the rows kernel's variants on the hardware (rot_store 1: LDS pitch 17 + buffer stores, 2: pitch 18, 3: pitch 17 + plain
# stores under a branch) against the per-lane stores
exp_q() {
  timeout 300 python tools/rot_store_debug.py > "$OUT/rot_store_debug.txt" 2>&1
  grep -c same "$OUT/rot_store_debug.txt"; grep "rot_store\|DIFF" "$OUT/rot_store_debug.txt" | cut -c1-200
}

# r: the rows kernel with the tile's offset in the vector offset (no register in the scalar-offset field: the hazard
# recognizer keeps the chain's next multiply off the store's data registers): debug script, parity tests, A/B against rot_store 0
exp_r() {
  Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
  timeout 300 python tools/rot_store_debug.py > "$OUT/rot_store_debug.txt" 2>&1
  grep -c same "$OUT/rot_store_debug.txt"; grep "rot_store\|DIFF" "$OUT/rot_store_debug.txt" | cut -c1-200
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_ingest.py -x -q -m gpu > "$OUT/tests.log" 2>&1
  tail -n 3 "$OUT/tests.log"
  run() { local name=$1; shift
    timeout 150 python bench.py --steps 20 --warmup 6 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"; }
  run rs1
  run rs0 --opt rot_store=0
  run rs1_b
  run rs0_b --opt rot_store=0
  run rs1_c
  run rs0_c --opt rot_store=0
  run rs1_d0 --depth 0
  run rs0_d0 --depth 0 --opt rot_store=0
  run rs1_c5 --density 40 --sample-rate 12000000
  run rs0_c5 --density 40 --sample-rate 12000000 --opt rot_store=0
}

# s: chunks NOT fed in place (--ingest 0: every chunk is copied into the history ring): the ring copy by kernel (copy_wide 1,
# default) against hipMemcpyAsync (0); the group bench again
exp_s() {
  Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
  run() { local name=$1; shift
    timeout 150 python bench.py --steps 20 --warmup 6 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"; }
  run i0_w1 --ingest 0
  run i0_w0 --ingest 0 --opt copy_wide=0
  run i0_w1_b --ingest 0
  run i0_w0_b --ingest 0 --opt copy_wide=0
  run i1
  timeout 400 python tools/group_bench.py > "$OUT/group_bench.txt" 2>"$OUT/group_bench.err"
  cat "$OUT/group_bench.txt"
  timeout 300 python -m pytest tests/test_gpu_group.py tests/test_gpu_ingest.py -x -q -m gpu > "$OUT/tests.log" 2>&1
  tail -n 3 "$OUT/tests.log"
}

# t: where a chunk's 4.4 ms go in the group's plain mode (timers in tools/group_bench.py)
exp_t() {
  timeout 400 python tools/group_bench.py > "$OUT/group_bench.txt" 2>"$OUT/group_bench.err"
  cat "$OUT/group_bench.txt"; tail -n 2 "$OUT/group_bench.err"
}

# u: the batch contexts again at the round's last commit (the chains got shorter: does the best depth move?)
exp_u() {
  Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
  run() { local name=$1; shift
    timeout 150 python bench.py --steps 20 --warmup 6 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"; }
  run d3
  run d4 --depth 4
  run d2 --depth 2
  run d3_b
  run d4_b --depth 4
  run d5 --depth 5
  run d3_c5 --density 40 --sample-rate 12000000
  run d4_c5 --density 40 --sample-rate 12000000 --depth 4
}

# v: IRDM_CHAIN_CU_RESERVE again, with the decimator's resident grid sized for the CUs its stream may use (the first
# measurement, exp_g, launched 7 x 256 workgroups onto 256 - R CUs: a second round, 0.61 ms): r<R>_d<pipeline_depth>
exp_v() {
  Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
  run() { local name=$1 r=$2; shift 2
    IRDM_CHAIN_CU_RESERVE=$r timeout 150 python bench.py --steps 20 --warmup 6 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"; }
  run r0_d3 0
  run r8_d3 8
  run r16_d3 16
  run r32_d3 32
  run r8_d4 8 --depth 4
  run r16_d4 16 --depth 4
  run r16_d5 16 --depth 5
  run r0_d3_b 0
  run r8_d3_b 8
  run r16_d3_b 16
  run r16_c5 16 --density 40 --sample-rate 12000000
  run r0_c5 0 --density 40 --sample-rate 12000000
}

# w: the chain's little copy / threshold kernels as single-wavefront workgroups (small_wg 64, default) against 256 threads
exp_w() {
  Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0"
  D12="--density 40 --sample-rate 12000000"
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ingest.py -x -q -m gpu > "$OUT/tests.log" 2>&1
  tail -n 2 "$OUT/tests.log"
  run() { local name=$1; shift
    timeout 150 python bench.py --steps 20 --warmup 6 $Q "$@" 2>"$OUT/$name.err" | tail -1 > "$OUT/$name.json"; }
  run w64
  run w256 --opt small_wg=256
  run w64_b
  run w256_b --opt small_wg=256
  run w64_c5 $D12
  run w256_c5 $D12 --opt small_wg=256
  run w64_c5_b $D12
  run w256_c5_b $D12 --opt small_wg=256
  run w64_d0 --depth 0
  run w256_d0 --depth 0 --opt small_wg=256
}

exp_$EXP

# one line per bench result of the call
python - "$OUT" <<'P'
import json, sys, glob, os
def find(d, key):
    if isinstance(d, dict):
        if key in d: return d[key]
        for v in d.values():
            r = find(v, key)
            if r is not None: return r
    return None
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f))
        st = find(d, "stage_ms") or {}
        sc = find(d, "scan") or {}
        h = find(d, "host_us_total") or {}
        n = d["steps"] + d["warmup"]
        print(os.path.basename(f), d["value"], d["ms_per_step"], "scan_ms", st.get("scan"), "k1", st.get("fft_mag"), "fir", st.get("fir"), "post", st.get("post"),
              "rounds/chunks", sc.get("band_rounds"), sc.get("band_chunks"), "undone", sc.get("scan_chain_undone"), "spec", sc.get("spec_scans"), "restarts", sc.get("sum_restarts"),
              "host/step: settle", round(h.get("settle", 0) / n), "older_chain", round(h.get("wait_older_chain", 0) / n), "parity", (find(d, "parity_checked") or {}).get("ok"),
              "kclk", find(d, "kernel_clock_ms"), "frac", find(d, "frac"))
        tl = find(d, "scan_timeline_us")
        if tl: print("   ", {k: v[:2] for k, v in tl.items()})
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
P
