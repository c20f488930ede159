mkdir -p gpurun_out/trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/trace/d1 -o tr --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --file-run 0 --cpu-samples 0 --detect-steps 0 --host-steps 0 --alone-steps 0 --depth 1 --opt fir_layout=1 > $R/gpurun_out/trace/d1.log 2>&1
ls -la $R/gpurun_out/trace/d1
