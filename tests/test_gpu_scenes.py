"""-m gpu: the HIP path against the oracle on the edge-case scenes of scenes.py -- squelch and history reset, forced
burst ends, the DC notch and guard bands, simultaneous strong bursts (multi-delete / long-list paths of the sparse
scan), and more concurrent bursts than the sparse scan holds (dense fallback).  Each scene runs through the sparse
scan, the dense scan, and chunked in throughput mode (pipeline_depth 1); every record must equal the oracle's."""
import numpy as np
import pytest

import irdm
import orc
import parity
import scenes

pytestmark = pytest.mark.gpu


def _chunks(n, parts):
    blocks = n // 32768
    cuts = [blocks * (i + 1) // parts for i in range(parts)]
    out, prev = [], 0
    for c in cuts:
        if c > prev:
            out.append((c - prev) * 32768)
            prev = c
    if n % 32768:
        out[-1] += n % 32768
    return out


CASES = {name: ({}, {}) for name in scenes.ALL}
# capture centred above 1626 MHz: the simplex frame-length rules (burst_downmix.c:764-767)
CASES["frame_lengths_simplex"] = (dict(simplex=True), dict(center_frequency=1626.4e6))


@pytest.mark.parametrize("case", sorted(CASES))
def test_scene_parity_sparse_dense_and_chunked(case):
    name = case.replace("_simplex", "")
    skw, ckw = CASES[case]
    fs, iq = scenes.ALL[name](**skw)
    ref = orc.run_stream(iq, fs, **ckw)
    got = parity.run_gpu(iq, fs, **ckw)                            # band scan (sequential kernels if it declines)
    s = parity.compare(got, ref)
    dense = parity.run_gpu(iq, fs, scan_mode=1, **ckw)
    parity.compare(dense, ref)
    assert dense["stats"]["scan_fast_chunks"] == 0
    chunked = parity.run_gpu(iq, fs, chunks=_chunks(len(iq), 5), depth=1, **ckw)
    parity.compare(chunked, ref)
    if name == "squelch":
        # more simultaneous bursts than max_bursts: the band scan must decline and the sequential kernels take over
        assert got["stats"]["scan_fallbacks"] >= 1 and got["stats"]["band_aborts"] >= 1, got["stats"]
    else:
        assert got["stats"]["scan_fallbacks"] == 0 and got["stats"]["band_chunks"] >= 1, got["stats"]
    assert s["bursts"] == len(ref.bursts)


@pytest.mark.parametrize("case", sorted(CASES))
def test_scene_parity_single_cu_scan(case):
    """scan_mode 2: the sparse scan confined to one workgroup (the default, scan_mode 0, spreads the baseline updates over
    updater workgroups on MI355X -- scan_fast.hip, MC form) -- same scenes, one chunk and chunked in throughput mode;
    every record must equal the oracle's"""
    name = case.replace("_simplex", "")
    skw, ckw = CASES[case]
    fs, iq = scenes.ALL[name](**skw)
    ref = orc.run_stream(iq, fs, **ckw)
    got = parity.run_gpu(iq, fs, scan_mode=2, **ckw)
    parity.compare(got, ref)
    chunked = parity.run_gpu(iq, fs, chunks=_chunks(len(iq), 5), depth=1, scan_mode=2, **ckw)
    parity.compare(chunked, ref)
    if name in ("many_active_10m", "squelch"):
        assert got["stats"]["scan_fallbacks"] >= 1, got["stats"]
    else:
        assert got["stats"]["scan_fallbacks"] == 0 and got["stats"]["scan_fast_chunks"] >= 1, got["stats"]


@pytest.mark.parametrize("seed", (3, 7, 100))
def test_scan_forms_random_scenes_and_final_baseline(seed):
    """randomised emitters through the single-CU (2) and multi-CU (3) sparse scans against the oracle, and the carried
    noise-floor sums after the stream bit-identical to the dense scan's (in the multi-CU form the updaters own them)"""
    fs, iq = scenes.random_scene(seed, fs=10_000_000, secs=0.8) if seed >= 100 else scenes.random_scene(seed)
    ref = orc.run_stream(iq, fs)
    parity.compare(parity.run_gpu(iq, fs, scan_mode=3), ref)
    parity.compare(parity.run_gpu(iq, fs, chunks=_chunks(len(iq), 7), depth=1, scan_mode=2), ref)
    sums = []
    for mode in (1, 2, 3):
        cs = _chunks(len(iq), 3)
        p = irdm.Pipeline(fs, max_chunk_samples=max(cs), pipeline_depth=1)
        p.set_option("scan_mode", mode)
        try:
            off = 0
            for c in cs:
                p.feed_host(iq[off:off + c])
                off += c
            p.flush()
            sums.append(p.baseline_sum().copy())
            if mode != 1:
                assert p.stat("scan_fallbacks") == 0, "the sparse scan fell back to the dense scan"
        finally:
            p.close()
    assert np.array_equal(sums[0].view(np.uint32), sums[1].view(np.uint32))
    assert np.array_equal(sums[0].view(np.uint32), sums[2].view(np.uint32))


def test_squelch_scene_ci8():
    """the same squelch / reset sequence through the ci8 ingest path"""
    import siggen
    fs, iq = scenes.squelch()
    i8 = siggen.to_ci8(iq * 8)
    ref = orc.run_stream(i8, fs, fmt=0)
    got = parity.run_gpu(i8, fs, fmt=irdm.FMT_CI8, chunks=_chunks(len(iq), 3), depth=1)
    parity.compare(got, ref)
    assert len(ref.bursts) >= 30


@pytest.mark.parametrize("seed", range(12))
def test_random_scenes_differential(seed):
    """randomised emitters (scenes.random_scene): sparse scan in one chunk, and 7 ragged chunks in throughput mode,
    against the oracle; every third seed also through the dense scan"""
    fs, iq = scenes.random_scene(seed)
    ref = orc.run_stream(iq, fs)
    assert len(ref.bursts) >= 10
    parity.compare(parity.run_gpu(iq, fs), ref)
    parity.compare(parity.run_gpu(iq, fs, chunks=_chunks(len(iq), 7), depth=1), ref)
    if seed % 3 == 0:
        parity.compare(parity.run_gpu(iq, fs, scan_mode=1), ref)


@pytest.mark.parametrize("seed", (100, 101))
def test_random_scenes_10mhz(seed):
    fs, iq = scenes.random_scene(seed, fs=10_000_000, secs=0.8)
    ref = orc.run_stream(iq, fs)
    parity.compare(parity.run_gpu(iq, fs), ref)
    parity.compare(parity.run_gpu(iq, fs, chunks=_chunks(len(iq), 4), depth=1), ref)


@pytest.mark.parametrize("thr_db,gardner", [(12.0, 1), (22.0, 1), (16.0, 0)])
def test_threshold_and_no_gardner_options(thr_db, gardner):
    """-d <dB> (burst_detect.c:213-226 threshold_lin) and --no-gardner (qpsk_demod.c:409-415 decimate_simple): other
    operating points of the same state machine / demodulator, against the oracle"""
    import siggen
    fs = 2_000_000
    iq, _ = siggen.standard_scene(fs, int(2.0 * fs) // 32768 * 32768, 8, seed=51, uplink_every=3, amp=0.03)
    ref = orc.run_stream(iq, fs, threshold_db=thr_db, use_gardner=gardner)
    got = parity.run_gpu(iq, fs, threshold_db=thr_db, use_gardner=gardner)
    s = parity.compare(got, ref)
    assert s["bursts"] >= 4
    chunked = parity.run_gpu(iq, fs, chunks=_chunks(len(iq), 4), depth=1, threshold_db=thr_db, use_gardner=gardner)
    parity.compare(chunked, ref)


def test_burst_record_buffer_grows_when_a_chunk_has_more_bursts_than_configured():
    """max_bursts_per_chunk is a sizing hint, not a limit: 263 bursts (12 dB threshold: noise triggers) in one chunk with
    room for 64 -> the chunk is redone with a larger record buffer, records equal the oracle's"""
    import siggen
    fs = 2_000_000
    iq, _ = siggen.standard_scene(fs, int(2.0 * fs) // 32768 * 32768, 8, seed=51, uplink_every=3, amp=0.03)
    ref = orc.run_stream(iq, fs, threshold_db=12.0)
    assert len(ref.bursts) > 200
    for depth in (0, 1):
        p = irdm.Pipeline(fs, threshold_db=12.0, max_chunk_samples=len(iq), max_bursts_per_chunk=64, pipeline_depth=depth)
        p.set_option("keep_frame_samples", 1)
        try:
            p.feed_host(iq)
            p.flush()
            infos, samples = p.poll_frames()
            got = dict(bursts=p.poll_bursts(), infos=infos, samples=samples, demods=p.poll_demods(), tagged=p.tagged)
            assert p.stat("scan_fallbacks") >= 1
        finally:
            p.close()
        parity.compare(got, ref)


def _zone_scene():
    """10 MHz, bursts on random channels: with 8192 bins in 64 bands of 128 and burst_width/2 = 14 bins each side of a
    boundary, about a fifth of the bursts lie in a boundary zone"""
    import siggen
    fs = 10_000_000
    n = int(1.2 * fs) // 32768 * 32768
    iq, _ = siggen.standard_scene(fs, n, 40, seed=2024)
    return fs, iq


def test_boundary_test_forms_agree_and_both_object():
    """the plan pass's boundary test (scan_band.hip: all boundaries at once from zone lists in LDS) against the plain
    wavefront search it replaced: test hook band_selfcheck 1 runs both on every verdict and declines the chunk with
    BAND_F_CHECK (4096) if their answers differ; 3 additionally spoils one band's copy of every record in a boundary
    zone, so both must object (BAND_F_AGREE, 32) and the sequential kernels take the chunk.  Parity either way
    (burst_detect.c:426-632)."""
    fs, iq = _zone_scene()
    ref = orc.run_stream(iq, fs)
    for chunks in (None, _chunks(len(iq), 3)):
        got = parity.run_gpu(iq, fs, chunks=chunks, depth=1 if chunks else 0, options={"band_selfcheck": 1})
        parity.compare(got, ref)
        assert got["stats"]["band_aborts"] == 0 and got["stats"]["band_chunks"] >= 1, got["stats"]
    spoiled = parity.run_gpu(iq, fs, options={"band_selfcheck": 3})
    parity.compare(spoiled, ref)
    st = spoiled["stats"]
    assert st["band_aborts"] >= 1 and st["scan_fallbacks"] >= 1, st
    assert st["band_last_flags"] & 32 and not st["band_last_flags"] & 4096, st


@pytest.mark.parametrize("case", ["too_long", "strong_simultaneous", "many_active_10m"])
def test_wave_walk_without_look_ahead_agrees(case):
    """test hook band_selfcheck 8: the wavefront walk event by event (band_wave.hpp without skim()) -- the records must not
    depend on the 64-frame look-ahead"""
    fs, iq = scenes.ALL[case]()
    ref = orc.run_stream(iq, fs)
    parity.compare(parity.run_gpu(iq, fs, options={"band_selfcheck": 8}), ref)


@pytest.mark.parametrize("case", ["too_long", "strong_simultaneous"])
def test_plan_without_lds_and_commit_as_a_launch_of_its_own_agree(case):
    """test hook band_selfcheck 16: the plan pass through the workspace arrays instead of its LDS (what a chunk of more than
    8192 frames takes), the commit as its own launch behind the rounds instead of inside the accepting plan pass -- same
    records"""
    fs, iq = scenes.ALL[case]()
    ref = orc.run_stream(iq, fs)
    parity.compare(parity.run_gpu(iq, fs, options={"band_selfcheck": 16}), ref)
    parity.compare(parity.run_gpu(iq, fs, chunks=_chunks(len(iq), 4), depth=1, options={"band_selfcheck": 16}), ref)

