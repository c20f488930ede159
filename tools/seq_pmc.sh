#!/bin/bash
# SQ counters of the chain's two lane-per-burst kernels alone (pipeline_depth 0), two passes of eight counters
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-seqpmc}
mkdir -p "$OUT"
Q="--cpu-samples 0 --host-steps 0 --alone-steps 0 --detect-steps 0 --file-run 0 --big-chunk-steps 0"
B="$GRAFT_REPO_ROOT/bench.py"
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $pass --kernel-include-regex "demod_seq|rot_phase_rows" -d "$OUT/sq$i" -o pmc --output-format csv -- \
      python $B --steps 2 --warmup 1 --depth 0 $Q > "$OUT/sq$i.log" 2>&1
done
cd "$GRAFT_REPO_ROOT"
python - "$OUT" <<'P'
import csv, glob, json, sys, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/sq*/**/*counter_collection.csv", recursive=True) + glob.glob(sys.argv[1] + "/sq*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("irdm::", "").split("(")[0]
        out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: round(sum(v) / len(v), 1) for c, v in d.items()} for k, d in out.items()}
for k, d in res.items():
    d["launches"] = max(len(v) for v in out[k].values())
json.dump(res, open(sys.argv[1] + "/seq_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
P
