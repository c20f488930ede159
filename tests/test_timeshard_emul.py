"""Time-chunk sharding of ONE stream across two ranks (BASELINE config 4) without a GPU: the workers of
tests/test_gpu_timeshard.py -- sharding.TimeShard over gloo, the detector state travelling rank to rank between the scans,
every rank seeding the samples in front of its chunk, K1 of a chunk before its state arrives -- on the emulated build of
the product (tests/emul_build.py; "device" tensors are CPU tensors).  The merged records of the two ranks must equal the
oracle's for the whole stream, bursts across every chunk boundary included (burst_detect.c:438-454, :594-631)."""
import os

import pytest

import emul_build
import test_gpu_timeshard as G


@pytest.mark.parametrize("depth", [0, 1])
def test_two_rank_time_shard_on_the_emulation(depth, monkeypatch):
    monkeypatch.setenv("IRDM_LIB", emul_build.build())
    monkeypatch.setenv("IRDM_TEST_DEVICE", "cpu")
    G.test_two_rank_time_shard_equals_the_oracle("2mhz", depth)
