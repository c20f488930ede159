mkdir -p gpurun_out/fir
run() { # name, opts...
  n=$1; shift
  timeout 300 python bench.py --steps 10 --warmup 4 --file-run 0 --cpu-samples 0 --detect-steps 0 --host-steps 0 --alone-steps 0 "$@" > gpurun_out/fir/$n.json 2> gpurun_out/fir/$n.err
}
run i0l0_d1 --depth 1 --ingest 0 --lookahead 0
run i1l0_d1 --depth 1 --ingest 1 --lookahead 0
run i0l1_d1 --depth 1 --ingest 0 --lookahead 1
run i1l1_d1 --depth 1 --ingest 1 --lookahead 1
run i1l1_d2 --depth 2 --ingest 1 --lookahead 1
run i1l1_d2k0 --depth 2 --ingest 1 --lookahead 1 --opt k1_first=0
python - <<'PY'
import json,glob
for f in ("i0l0_d1","i1l0_d1","i0l1_d1","i1l1_d1","i1l1_d2","i1l1_d2k0"):
    try:
        j=json.loads(open("gpurun_out/fir/%s.json"%f).read().strip().splitlines()[-1])
        K=j["steps"]+j["warmup"]
        print(f, j["value"], j["ms_per_step"], j["roofline"]["stage_ms"], {k: round(v/1e3/K,3) for k,v in j["config"]["host_us_total"].items()})
    except Exception as e: print(f, "ERR", e, open("gpurun_out/fir/%s.err"%f).read()[-600:])
PY
