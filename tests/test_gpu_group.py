"""ONE stream across the GPUs of one process (irdm_group_*, csrc/group.cpp; SURVEY 8e, main.c:667-694's layout for N > 1)
on the real library with the real RCCL.  A box with one GPU runs the whole protocol on a group of ONE member with
"group_loopback": the overlap seeded from the member's own landing buffer, the detector state exported, sent to itself
with a grouped ncclSend / ncclRecv pair, and imported in front of every chunk -- every record must equal the oracle's for
the whole stream, in stream order.  With two GPUs present the same stream goes over two members.  (The CPU suite runs 2 and
3 members on the emulated devices: tests/test_pipeline_emul.py::test_group_of_members_equals_one_context.)"""
import os
import subprocess

import numpy as np
import pytest

import irdm
import orc
import parity
import test_gpu_timeshard as G

pytestmark = pytest.mark.gpu


def _n_devices():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 1


@pytest.fixture(scope="module")
def stream():
    fs = G.CASES["2mhz"][0]
    iq, chunk, ov = G._stream("2mhz")
    return fs, iq, chunk, ov, orc.run_stream(iq, fs)


@pytest.mark.parametrize("kw", [dict(), dict(staged_ahead=True, depth=2), dict(feed="device", staged_ahead=True)],
                         ids=["host", "host_staged_ahead_depth2", "scatter_from_member_0"])
def test_group_of_one_member_hands_its_state_to_itself(stream, kw):
    fs, iq, chunk, ov, ref = stream
    got = parity.run_group(iq, fs, n_gpus=1, chunk=chunk, options={"group_loopback": 1}, **kw)
    parity.compare(got, ref)
    st = got["stats"]
    assert st["chunks"] == 4 and st["hops"] == 4 and st["overlap_samples"] == ov, st
    assert st["hop_bytes"] == 4 * (st["hop_bytes"] // 4) > 0 and st["overlap_bytes"] == 3 * ov * 8, st
    assert st["scan_fallbacks"] == 0 and st["band_aborts"] == 0, st


def test_group_of_one_member_is_the_plain_context(stream):
    fs, iq, chunk, ov, ref = stream
    got = parity.run_group(iq, fs, n_gpus=1, chunk=chunk)
    parity.compare(got, ref)
    assert got["stats"]["hops"] == 0 and got["stats"]["chunks"] == 4, got["stats"]


@pytest.mark.skipif(_n_devices() < 2, reason="needs two GPUs in this process")
@pytest.mark.parametrize("kw", [dict(), dict(feed="device", staged_ahead=True)], ids=["host", "scatter_from_member_0"])
def test_group_of_two_members(stream, kw):
    fs, iq, chunk, ov, ref = stream
    got = parity.run_group(iq, fs, n_gpus=2, chunk=chunk, **kw)
    parity.compare(got, ref)
    assert got["stats"]["hops"] == 4 and got["stats"]["chunks"] == 4, got["stats"]


def test_group_12mhz_loopback_with_a_ragged_last_chunk():
    """BASELINE config 4's geometry (16384-point frames, a 33.7 MB state, a 25 M-sample overlap) through the loopback
    group, the stream cut so that its last chunk is shorter than the others"""
    fs = G.CASES["12mhz"][0]
    iq, chunk, ov = G._stream("12mhz")
    n = (len(iq) - 5 * 32768 - 1234)
    iq = iq[:n]
    ref = orc.run_stream(iq, fs)
    got = parity.run_group(iq, fs, n_gpus=1, chunk=chunk, options={"group_loopback": 1})
    parity.compare(got, ref)
    assert got["stats"]["chunks"] == 2 and got["stats"]["hops"] == 2, got["stats"]
    # (the scan of chunk 1 starts on the head and takes the 512-frame history behind its round 0)
    assert got["stats"]["late_history"] == 1, got["stats"]


def test_cli_gpus_flag(tmp_path, stream):
    """iridium-sniffer-hip --gpus 1: the file goes through the group API (super-steps of one chunk) and prints the lines the
    oracle prints; with --group-loopback the member hands the detector state to itself over RCCL between the chunks"""
    fs, iq, chunk, ov, ref = stream
    exe = os.path.join(os.path.dirname(irdm.LIB_PATH), "iridium-sniffer-hip")
    if not os.path.exists(exe):
        irdm.build(force=True)
    path = tmp_path / "scene.cf32"
    np.ascontiguousarray(iq).tofile(path)
    want = [l.strip() for l in ref.raw_lines("golden")]
    runs = [["--gpus", "1", "--chunk", str(chunk)], ["--gpus", "1", "--group-loopback", "--chunk", str(chunk)]]
    if _n_devices() >= 2:
        runs.append(["--gpus", "2", "--chunk", str(chunk)])
    for extra in runs:
        out = subprocess.run([exe, "-f", str(path), "-r", str(fs), "--file-info", "golden"] + extra,
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        assert "tagged %d bursts total" % ref.n_tagged in out.stderr, out.stderr
        lines = [l for l in out.stdout.splitlines() if l.startswith("RAW:")]
        assert len(lines) == len(want) >= 30
        for a, b in zip(lines, want):
            fa, fb = parity.raw_fields(a), parity.raw_fields(b)
            assert fa[0] == fb[0] and fa[3:6] == fb[3:6] and fa[7:] == fb[7:]
            assert abs(fa[2] - fb[2]) <= 1 and abs(fa[6] - fb[6]) <= 1e-4
