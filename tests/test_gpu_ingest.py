"""-m gpu: the throughput-mode feeding variants (include/irdm_hip.h, irdm_ingest_ptr / look-ahead) give the oracle's
records:
  ingest      every chunk written in place into its slot of the history ring and fed from there (no ring copy)
  lookahead   irdm_feed_begin(k+1) before irdm_feed_end(k) on device buffers (K1 of the next chunk ahead of the host's
              wait for the previous scan)
  both together, at pipeline_depth 1 and 2, over a stream long enough for the ring to wrap (2 MHz: the ring is about
  5.6 M samples, the stream 10 M), and on the scenes that end the stream with a ragged chunk / exceed max_bursts."""
import numpy as np
import pytest

import irdm
import orc
import parity
import scenes
import siggen

pytestmark = pytest.mark.gpu


def _equal_chunks(n, blocks_per_chunk):
    c = blocks_per_chunk * 32768
    out = [c] * (n // c)
    if n % c:
        out.append(n % c)
    return out


@pytest.fixture(scope="module")
def long_scene():
    fs = 2_000_000
    iq = siggen.standard_scene(fs, 5 * fs, 14, seed=23, uplink_every=4)[0]
    return fs, iq, orc.run_stream(iq, fs)


@pytest.mark.parametrize("depth", [1, 2])
@pytest.mark.parametrize("feed", ["ingest", "lookahead", "ingest_lookahead"])
def test_feed_variants_long_stream(long_scene, feed, depth):
    fs, iq, ref = long_scene
    got = parity.run_gpu(iq, fs, chunks=_equal_chunks(len(iq), 8), depth=depth, feed=feed)
    s = parity.compare(got, ref)
    assert s["bursts"] >= 14 and got["n_samples"] == len(iq)


@pytest.mark.parametrize("name", ["squelch", "many_active_10m", "too_long"])
def test_feed_variants_scenes(name):
    if name not in scenes.ALL:
        pytest.skip("no such scene")
    fs, iq = scenes.ALL[name]()
    ref = orc.run_stream(iq, fs)
    blocks = max(1, (len(iq) // 32768) // 5)
    for feed in ("ingest_lookahead", "lookahead"):
        got = parity.run_gpu(iq, fs, chunks=_equal_chunks(len(iq), blocks), depth=1, feed=feed)
        parity.compare(got, ref)


def test_ingest_ptr_contract():
    fs = 2_000_000
    chunk = 8 * 32768
    p = irdm.Pipeline(fs, max_chunk_samples=chunk, pipeline_depth=1)
    p0 = irdm.Pipeline(fs, max_chunk_samples=chunk, pipeline_depth=0)
    try:
        assert not p0.ingest_ptr(chunk)                 # no ring copy to save at pipeline_depth 0
        base, n = p.ring()
        assert base and n % chunk == 0
        assert p.ingest_ptr(chunk) == base              # the stream starts at ring index 0
        assert not p.ingest_ptr(chunk + 32768)          # larger than max_chunk_samples
        z = np.zeros(chunk, np.complex64)
        p.feed_host(z)                                  # an ordinary feed moves the position as well
        assert p.ingest_ptr(chunk) == base + chunk * 8
        # two begins may be pending (one chunk of look-ahead), not three; flush refuses while one is
        a, b = irdm.device_buffer(z), irdm.device_buffer(z)
        p.feed_begin(a, chunk)
        p.feed_begin(b, chunk)
        with pytest.raises(RuntimeError):
            p.feed_begin(a, chunk)
        assert p.L.irdm_flush(p.h) == -1
        p.feed_end()
        p.feed_end()
        with pytest.raises(RuntimeError):
            p.feed_end()
        p.flush()
        irdm.device_free(a)
        irdm.device_free(b)
    finally:
        p.close()
        p0.close()


def test_k1_builds_the_candidate_lists():
    """10 MHz (8192-point frames, the radix-16 K1): after the first chunk has primed the detector, K1 writes the band
    scan's candidate lists itself (fft_mag_r16_kernel<.., LISTS>); same records as with the prefilter pass, at every
    pipeline depth, also when the chunk is fed in place with look-ahead."""
    fs, iq = scenes.ALL["many_active_10m"]()
    ref = orc.run_stream(iq, fs)
    # the detector is primed after 512 frames (4.2 M samples of the 7.4 M): chunks of ~0.6 M samples leave four or five
    # chunks behind that point, minus the one or two in flight when the host learns of it
    blocks = max(1, (len(iq) // 32768) // 12)
    chunks = _equal_chunks(len(iq), blocks)
    for depth, feed in ((0, "host"), (1, "host"), (2, "ingest_lookahead")):
        got = parity.run_gpu(iq, fs, chunks=chunks, depth=depth, feed=feed)
        parity.compare(got, ref)
        assert got["stats"]["k1_lists"] >= 2, got["stats"]
        assert got["stats"]["scan_fallbacks"] == 0, got["stats"]
    off = parity.run_gpu(iq, fs, chunks=chunks, depth=1, options={"k1_lists": 0})
    parity.compare(off, ref)
    assert off["stats"]["k1_lists"] == 0


@pytest.mark.parametrize("name", ["strong_simultaneous", "too_long", "many_active_10m"])
def test_band_scan_continues_its_rounds(name):
    """Only the first rounds of the band scan are enqueued up front; when their verdict is still open the host enqueues
    the rest on the same workspace.  band_first = 1 forces that path on scenes that need two or three rounds."""
    fs, iq = scenes.ALL[name]()
    ref = orc.run_stream(iq, fs)
    blocks = max(1, (len(iq) // 32768) // 4)
    for depth in (0, 1):
        got = parity.run_gpu(iq, fs, chunks=_equal_chunks(len(iq), blocks), depth=depth, options={"band_first": 1})
        parity.compare(got, ref)
        assert got["stats"]["band_extra"] >= 1 and got["stats"]["scan_fallbacks"] == 0, got["stats"]


def test_scan_opens_with_round_1_behind_a_speculation_pass():
    """band_spec (default): fed with look-ahead, the chained scans have no round 0 of their own -- a speculation pass on a
    second workspace and stream, enqueued with the previous chunk, made the guess of the update vector -- and open with
    round 1; same records as without (test hook band_spec 0), and when the guess is spoilt chunk after chunk by a different
    chunking (bursts carried across every boundary)."""
    fs, iq = scenes.ALL["many_active_10m"]()
    ref = orc.run_stream(iq, fs)
    for parts, depth in ((12, 2), (7, 2), (9, 3)):
        blocks = max(1, (len(iq) // 32768) // parts)
        chunks = _equal_chunks(len(iq), blocks)
        got = parity.run_gpu(iq, fs, chunks=chunks, depth=depth, feed="ingest_lookahead")
        parity.compare(got, ref)
        assert got["stats"]["spec_scans"] >= 2, got["stats"]
        assert got["stats"]["spec_passes"] >= got["stats"]["spec_scans"], got["stats"]
        assert got["stats"]["scan_fallbacks"] == 0 and got["stats"]["band_aborts"] == 0, got["stats"]
        # (rounds per chunk stay what they were: a good guess is accepted by the first verdict)
        assert got["stats"]["band_rounds"] <= 3 * got["stats"]["band_chunks"], got["stats"]
    off = parity.run_gpu(iq, fs, chunks=chunks, depth=2, feed="ingest_lookahead", options={"band_spec": 0})
    parity.compare(off, ref)
    assert off["stats"]["spec_scans"] == 0 and off["stats"]["scan_chained"] >= 2, off["stats"]


def test_sums_pass_restart_behind_an_unchanged_prefix():
    """With the speculation pass's guess spoilt in one late frame (test hook band_selfcheck 32) the scans' next round has update
    steps that agree with the previous round's up to that frame -- its sums pass starts from the state stored there (one in
    64 steps; a sparse scene, so that a chunk of ~130 frames has more than 64 update steps in front of the spoilt frame); same
    records as the oracle."""
    import siggen
    fs = 10_000_000
    n = int(0.95 * fs) // 32768 * 32768
    iq, _ = siggen.standard_scene(fs, n, 8, seed=77)
    ref = orc.run_stream(iq, fs)
    first = 512 * 8192
    c = ((len(iq) - first) // 5) // 32768 * 32768
    chunks = [first, c, c, c, c, len(iq) - first - 4 * c]
    got = parity.run_gpu(iq, fs, chunks=chunks, depth=2, feed="ingest_lookahead", options={"band_selfcheck": 32})
    parity.compare(got, ref)
    assert got["stats"]["spec_scans"] >= 1 and got["stats"]["sum_restarts"] >= 1, got["stats"]
    assert got["stats"]["scan_fallbacks"] == 0, got["stats"]
    plain = parity.run_gpu(iq, fs, chunks=chunks, depth=2, feed="ingest_lookahead")
    parity.compare(plain, ref)
