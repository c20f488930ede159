#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/p30
mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -x -q -m gpu > "$OUT/t.log" 2>&1
tail -n 8 "$OUT/t.log"
IRDM_CREATE_DEBUG=1 timeout 120 ./iridium-sniffer_amd/iridium-sniffer-hip -f /dev/null -r 10000000 -c 1622000000 --format cf32 --timing > /dev/null 2> "$OUT/host10.err"
tail -n 14 "$OUT/host10.err"
