"""-m gpu: bench.py's multi-rank paths executed for real -- two ranks launched the way the driver launches them
(python -m torch.distributed.run --nproc-per-node 2 ...), sharing GPU 0 with gloo as the backend (IRDM_BENCH_BACKEND /
IRDM_BENCH_SHARE_GPU: RCCL needs one GPU per rank, a one-GPU box runs everything else of the path): the per-rank
streams with the asynchronous record gather to rank 0 (BASELINE config 5 / the scaling of the headline metric), and
one stream cut into time-chunks with the detector state handed from rank to rank (config 4: burst_detect.c:438-454,
:594-631 is the sequential dependency).  Every record a rank produces must arrive on rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUICK = ["--cpu-samples", "0", "--host-steps", "0", "--alone-steps", "0", "--detect-steps", "0", "--file-run", "0"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(extra, timeout=600):
    env = dict(os.environ, IRDM_BENCH_BACKEND="gloo", IRDM_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + extra + QUICK
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]            # rank 0 prints ONE line
    return json.loads(lines[0])


def test_streams_mode_two_ranks_gathers_every_record():
    out = _run(["--steps", "3", "--warmup", "1", "--samples", str(8 * 1024 * 1024), "--density", "20"])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    rec = out["config"]["records"]
    assert rec["produced"] == rec["sent"] == rec["gathered_on_rank0"] > 0, rec
    assert out["config"]["scan"]["scan_fallbacks"] == 0
    # what a SCALE record is checked with: the group's real size, the backend that carried it (gloo here: rccl_world 0)
    assert out["collective_world"] == 2 and out["collective_backend"] == "gloo" and out["rccl_world"] == 0
    assert 0 < out["rank_Msamples_per_s"]["min"] <= out["rank_Msamples_per_s"]["max"]


def test_time_shard_mode_two_ranks():
    out = _run(["--shard", "time", "--steps", "2", "--warmup", "1", "--sample-rate", "12000000"])
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert out["config"]["parallelism"].startswith("time")
