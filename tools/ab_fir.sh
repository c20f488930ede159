mkdir -p gpurun_out/fir
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cfg5.py tests/test_gpu_scenes.py -m gpu -x -q > gpurun_out/fir/t.log 2>&1; tail -2 gpurun_out/fir/t.log
run() { # name, opts...
  n=$1; shift
  timeout 300 python bench.py --steps 10 --warmup 4 --file-run 0 --cpu-samples 0 --detect-steps 0 --host-steps 0 --alone-steps 2 "$@" > gpurun_out/fir/$n.json 2> gpurun_out/fir/$n.err
}
run c_d1 --depth 1 --opt fir_layout=1
run w_b0_d1 --depth 1 --opt fir_budget=0
run w_b4_d1 --depth 1 --opt fir_budget=4
run w_b8_d1 --depth 1 --opt fir_budget=8
run w_b16_d1 --depth 1 --opt fir_budget=16
run w_b8_r32_d1 --depth 1 --opt fir_budget=8 --opt fir_reserve_cus=32
python - <<'PY'
import json,glob
for f in ("c_d1","w_b0_d1","w_b4_d1","w_b8_d1","w_b16_d1","w_b8_r32_d1"):
    try:
        j=json.loads(open("gpurun_out/fir/%s.json"%f).read().strip().splitlines()[-1])
        K=j["steps"]+j["warmup"]
        print(f, j["value"], j["ms_per_step"], j["roofline"]["stage_ms"], j["roofline"]["stage_ms_alone"]["fir"], {k: round(v/1e3/K,3) for k,v in j["config"]["host_us_total"].items() if k in ("settle","wait_older_chain")})
    except Exception as e: print(f, "ERR", e)
PY
