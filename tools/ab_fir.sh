mkdir -p gpurun_out/fir
timeout 900 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_scenes.py tests/test_gpu_cfg5.py -m gpu -x -q > gpurun_out/fir/t.log 2>&1; tail -12 gpurun_out/fir/t.log
run() { # name, opts...
  n=$1; shift
  timeout 300 python bench.py --file-run 0 --cpu-samples 0 --detect-steps 0 --host-steps 0 --alone-steps 0 --steps 20 --warmup 5 "$@" > gpurun_out/fir/$n.json 2> gpurun_out/fir/$n.err
}
run a --depth 2
run b --depth 2 --detect-steps 10
run c --depth 2 --density 2
run d --depth 1
python - <<'PY'
import json,glob
for f in "abcd":
    try:
        j=json.loads(open("gpurun_out/fir/%s.json"%f).read().strip().splitlines()[-1])
        K=j["steps"]+j["warmup"]
        h={k: round(v/1e3/K,3) for k,v in j["config"]["host_us_total"].items()}
        print(f, j["value"], j["ms_per_step"], j["roofline"]["host_ms"], j["roofline"]["stage_ms"], h["settle"], h["wait_older_chain"], j["config"]["scan"])
    except Exception as e: print(f, "ERR", e, open("gpurun_out/fir/%s.err"%f).read()[-600:])
PY
