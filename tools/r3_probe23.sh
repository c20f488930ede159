#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/p23
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
python - > "$OUT/gen.log" 2>&1 <<'PY'
import numpy as np
rng = np.random.default_rng(1)
x = (rng.standard_normal(2 * (32 << 20)).astype(np.float32) * 0.01)
x.tofile('/dev/shm/noise.cf32')
PY
for i in 1 2; do
IRDM_CREATE_DEBUG=1 timeout 120 ./iridium-sniffer_amd/iridium-sniffer-hip -f /dev/shm/noise.cf32 -r 10000000 -c 1622000000 --format cf32 --timing > /dev/null 2> "$OUT/host10_$i.err"
done
IRDM_CREATE_DEBUG=1 timeout 120 ./iridium-sniffer_amd/iridium-sniffer-hip -f /dev/shm/noise.cf32 -r 12000000 -c 1622000000 --format cf32 --timing > /dev/null 2> "$OUT/host12.err"
IRDM_CREATE_DEBUG=1 timeout 120 python - > "$OUT/py.log" 2>&1 <<'PY'
import sys, time
sys.path.insert(0, 'iridium-sniffer_amd'); sys.path.insert(0, 'tests')
import irdm
t = time.time()
p = irdm.Pipeline(10_000_000, max_chunk_samples=64 << 20, max_bursts_per_chunk=4096, pipeline_depth=2)
print('create', time.time() - t)
p.close()
PY
cat "$OUT"/host10_2.err "$OUT/host12.err" "$OUT/py.log"
