mkdir -p gpurun_out/fir
run() { # name, opts...
  n=$1; shift
  timeout 300 python bench.py --file-run 0 --cpu-samples 0 --detect-steps 0 --host-steps 0 --alone-steps 0 --steps 20 --warmup 5 "$@" > gpurun_out/fir/$n.json 2> gpurun_out/fir/$n.err
}
GPU_MAX_HW_QUEUES=8 run a --depth 2
GPU_MAX_HW_QUEUES=8 run b --depth 1
GPU_MAX_HW_QUEUES=2 run c --depth 2
GPU_MAX_HW_QUEUES=16 run d --depth 2
run e --depth 2
python - <<'PY'
import json,glob
for f in "abcde":
    try:
        j=json.loads(open("gpurun_out/fir/%s.json"%f).read().strip().splitlines()[-1])
        K=j["steps"]+j["warmup"]
        h={k: round(v/1e3/K,3) for k,v in j["config"]["host_us_total"].items()}
        print(f, j["value"], j["ms_per_step"], j["roofline"]["host_ms"], j["roofline"]["stage_ms"], h["settle"], h["wait_older_chain"])
    except Exception as e: print(f, "ERR", e, open("gpurun_out/fir/%s.err"%f).read()[-600:])
PY
