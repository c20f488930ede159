"""-m gpu: BASELINE config 5 and the 12 MHz leg of config 4 at test scale.

* 12 MHz cf32, 16384-point frames, 16 Mi samples, 40 bursts per Msample (about 25 bursts active at any time, below
  max_bursts = 240): the band scan (default), the dense sequential scan, and chunked at pipeline_depth 1 -- every
  record must equal the oracle's (burst indices and hard bits exact, soft outputs within 1e-4);
* the 12 MHz detector-state hand-off (32 MiB history + sums + active bursts through irdm_export_state /
  irdm_import_state, plus the seeded sample history) with the cut inside a burst: what rank k+1 does in a
  time-sharded run (burst_detect.c:180-226 for the 12 MHz constants, :438-454 / :594-631 for the state).
"""
import numpy as np
import pytest

import irdm
import orc
import parity
import sharding
import siggen

pytestmark = pytest.mark.gpu

FS = 12_000_000
NFFT = 16384


def _dense_scene(n_samples, density_per_msample, seed):
    rng = np.random.default_rng(seed)
    first = 520 * NFFT
    nb = int(round(density_per_msample * (n_samples - first) / 1e6))
    starts = np.sort(rng.integers(first, n_samples - int(0.03 * FS), nb))
    half_ch = int((FS / 2 - 60e3) // (1e6 / 24.0))
    bursts = [dict(start=int(s), freq_hz=siggen.channel_freq(int(rng.integers(-half_ch, half_ch + 1)) or 1),
                   payload=rng.integers(0, 4, int(rng.integers(119, 180))).tolist()) for s in starts]
    iq, _ = siggen.make_stream(FS, n_samples, bursts, seed=seed)
    return iq


@pytest.fixture(scope="module")
def cfg5_scene():
    n = 16 * 1024 * 1024
    iq = _dense_scene(n, 40.0, seed=50)
    ref = orc.run_stream(iq, FS, cap_bursts=8192)
    assert len(ref.bursts) >= 250 and len(ref.demods) >= 200, (len(ref.bursts), len(ref.demods))
    return iq, ref


def test_cfg5_12mhz_dense_band_scan(cfg5_scene):
    iq, ref = cfg5_scene
    got = parity.run_gpu(iq, FS)
    s = parity.compare(got, ref)
    assert got["stats"]["band_chunks"] == 1 and got["stats"]["scan_fallbacks"] == 0, got["stats"]
    assert s["demods"] >= 200


def test_cfg5_12mhz_dense_sequential_scan(cfg5_scene):
    iq, ref = cfg5_scene
    got = parity.run_gpu(iq, FS, scan_mode=1)
    parity.compare(got, ref)
    assert got["stats"]["band_chunks"] == 0


def test_cfg5_12mhz_dense_chunked_depth_1(cfg5_scene):
    iq, ref = cfg5_scene
    n = len(iq)
    blocks = n // 32768
    cuts = [blocks * 2 // 7, blocks * 3 // 7, blocks * 5 // 7, blocks]
    chunks, prev = [], 0
    for c in cuts:
        chunks.append((c - prev) * 32768)
        prev = c
    got = parity.run_gpu(iq, FS, chunks=chunks, depth=1)
    parity.compare(got, ref)
    assert got["stats"]["scan_fallbacks"] == 0, got["stats"]


def test_12mhz_state_handoff_inside_a_burst():
    n = (520 * NFFT + 5 * 1024 * 1024) // 32768 * 32768
    iq = _dense_scene(n, 6.0, seed=51)
    ref = orc.run_stream(iq, FS)
    assert len(ref.bursts) >= 20
    cut = None
    for rb in ref.bursts:
        c = (rb.start + rb.num_samples // 2) // 32768 * 32768
        if rb.start < c < rb.start + rb.num_samples and c > 600 * NFFT:
            cut = int(c)
            break
    assert cut is not None
    a = irdm.Pipeline(FS, max_chunk_samples=n, max_bursts_per_chunk=1024)
    b = irdm.Pipeline(FS, max_chunk_samples=n, max_bursts_per_chunk=1024)
    try:
        for p in (a, b):
            p.set_option("keep_frame_samples", 1)
        a.feed_host(iq[:cut])
        blob = a.export_state()
        assert len(blob) == a.L.irdm_state_bytes(a.h) and len(blob) > 32 * 1024 * 1024
        ov = min(cut, sharding.required_overlap(FS, NFFT))
        b.seed_history(iq[cut - ov:cut], cut)
        b.import_state(blob)
        b.feed_host(iq[cut:])
        got = dict(bursts=a.poll_bursts() + b.poll_bursts(), demods=a.poll_demods() + b.poll_demods(),
                   tagged=b.tagged, n_samples=b.sample_count)
        ia, sa = a.poll_frames()
        ib, sb_ = b.poll_frames()
        got["infos"], got["samples"] = ia + ib, sa + sb_
        parity.compare(got, ref)
        assert any(bb.start < cut < bb.start + bb.num_samples for bb in got["bursts"])
    finally:
        a.close()
        b.close()
