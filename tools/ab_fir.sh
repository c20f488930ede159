mkdir -p gpurun_out/fir
timeout 900 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_scenes.py tests/test_gpu_cfg5.py tests/test_gpu_timeshard.py -m gpu -x -q > gpurun_out/fir/t.log 2>&1; tail -12 gpurun_out/fir/t.log
run() { # name, opts...
  n=$1; shift
  timeout 300 python bench.py --file-run 0 --cpu-samples 0 --detect-steps 0 --host-steps 0 "$@" > gpurun_out/fir/$n.json 2> gpurun_out/fir/$n.err
}
run d1 --depth 1 --steps 20 --warmup 5 --alone-steps 0
run d2 --depth 2 --steps 20 --warmup 5 --alone-steps 0
python - <<'PY'
import json,glob
for f in ("d1","d2"):
    try:
        j=json.loads(open("gpurun_out/fir/%s.json"%f).read().strip().splitlines()[-1])
        K=j["steps"]+j["warmup"]
        print(f, j["value"], j["ms_per_step"], j["roofline"]["host_ms"], j["roofline"]["stage_ms"], {k: round(v/1e3/K,3) for k,v in j["config"]["host_us_total"].items()})
    except Exception as e: print(f, "ERR", e, open("gpurun_out/fir/%s.err"%f).read()[-600:])
PY
