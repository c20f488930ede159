"""-m gpu: the configuration bench.py times -- 10 MHz cf32, pipeline_depth 3 (four batch contexts), every chunk fed in place
(irdm_ingest_ptr) one chunk of look-ahead ahead -- over a WHOLE stream of twelve chunks instead of its first chunk: the
history ring (four chunks of 5 Mi samples here) wraps three times, four chains are in flight from the fourth chunk on, the
speculation passes are fed by a previous chunk, bursts straddle every chunk boundary, simplex-class channels carry frames of
up to 444 symbols next to the normal class's 191, and the rotator-checkpoint arena starts so small (rot_pool_rows) that it
grows while the contexts are in flight.  Once with the full records (frame samples, LLRs: parity.compare on every record)
and once with the compact records the bench polls (parity.compare_packed)."""
import numpy as np
import pytest

import orc
import parity
import siggen

pytestmark = pytest.mark.gpu

FS = 10_000_000
CHUNK = 160 * 32768                   # 5 Mi samples
N_CHUNKS = 12


@pytest.fixture(scope="module")
def stream():
    n = N_CHUNKS * CHUNK
    rng = np.random.default_rng(606)
    first = 520 * 8192
    half_ch = int((FS / 2 - 60e3) // (1e6 / 24.0))
    bursts = []
    # 170 bursts spread over the stream, every fifth an uplink-style one (two detections, no frame)
    for i, s in enumerate(np.sort(rng.integers(first, n - int(0.06 * FS), size=170))):
        ch = int(rng.integers(-half_ch, half_ch + 1)) or 1
        bursts.append(dict(start=int(s), freq_hz=siggen.channel_freq(ch), uplink=(i % 5 == 4),
                           payload=rng.integers(0, 4, size=int(rng.integers(119, 180))).tolist()))
    # one burst across every chunk boundary (its window starts 2 * 8192 samples before it and ends 0.016 s behind it)
    for k in range(1, N_CHUNKS):
        ch = int(rng.integers(-90, 91)) or 2
        bursts.append(dict(start=k * CHUNK - int(rng.integers(5_000, 70_000)), freq_hz=siggen.channel_freq(ch),
                           payload=rng.integers(0, 4, size=170).tolist()))
    # simplex-class channels (> 1626 MHz: more than 4 MHz above the centre): long frames
    for k in range(10):
        bursts.append(dict(start=int(rng.integers(first, n - int(0.08 * FS))), freq_hz=siggen.channel_freq(int(rng.integers(100, 117))),
                           payload=rng.integers(0, 4, size=int(rng.integers(300, 420))).tolist()))
    iq, _ = siggen.make_stream(FS, n, bursts, seed=61)
    return iq, orc.run_stream(iq, FS)


def test_whole_stream_full_records(stream):
    iq, ref = stream
    got = parity.run_gpu(iq, FS, chunks=[CHUNK] * N_CHUNKS, depth=3, feed="ingest_lookahead", options={"rot_pool_rows": 8})
    s = parity.compare(got, ref)
    assert s["bursts"] >= 190 and s["demods"] >= 150, s
    st = got["stats"]
    assert st["rot_grows"] >= 1, st                       # the arena grew with chains in flight
    assert st["spec_scans"] >= N_CHUNKS - 4 and st["scan_chained"] >= N_CHUNKS - 3, st
    assert st["scan_fallbacks"] == 0 and st["band_aborts"] == 0, st
    assert max(d.n_symbols for d in got["demods"]) > 300          # simplex-class frames came through
    assert got["n_samples"] == len(iq)


def test_whole_stream_packed_records(stream):
    iq, ref = stream
    got = parity.run_gpu(iq, FS, chunks=[CHUNK] * N_CHUNKS, depth=3, feed="ingest_lookahead", options={"rot_pool_rows": 8},
                         packed=True)
    s = parity.compare_packed(got, ref)
    assert s["demods"] >= 150, s
    assert got["stats"]["rot_grows"] >= 1 and got["stats"]["scan_fallbacks"] == 0, got["stats"]
