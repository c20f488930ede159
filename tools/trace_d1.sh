mkdir -p gpurun_out/trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/trace/d0 -o tr --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --file-run 0 --cpu-samples 0 --detect-steps 0 --host-steps 0 --alone-steps 0 --depth 0 > $R/gpurun_out/trace/d0.log 2>&1
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/trace/d1 -o tr --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --file-run 0 --cpu-samples 0 --detect-steps 0 --host-steps 0 --alone-steps 0 --depth 1 > $R/gpurun_out/trace/d1.log 2>&1
ls $R/gpurun_out/trace/d0 $R/gpurun_out/trace/d1
