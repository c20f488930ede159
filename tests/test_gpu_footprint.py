"""-m gpu: device memory of a context.  The rotator checkpoints (rotator.h:36-46 restated as checkpoints of the phase
recurrence) are kept per centre bin that bursts have appeared on, as far as those bursts have needed them: blocks of 2048
checkpoints out of an arena.  The decimated
and low-passed bursts of a batch lie end to end by their actual length in a scratch that grows on demand."""
import numpy as np
import pytest
import torch

import irdm
import orc
import parity
import siggen

pytestmark = pytest.mark.gpu


def _used():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return total - free


def test_rotator_checkpoint_rows_on_demand_and_pool_growth():
    """Blocks of the rotator checkpoint arena are handed out when a burst first appears on a centre bin (or needs more of
    its row) and built on the chain that needs them; an arena that runs full doubles (here from one whole row's worth of
    blocks) while chains are in flight; the records are the oracle's either way."""
    fs = 2_000_000
    n = int(0.9 * fs) // 32768 * 32768
    iq, _ = siggen.standard_scene(fs, n, 9, seed=5)
    ref = orc.run_stream(iq, fs)
    bins = {b.center_bin for b in ref.bursts}
    assert len(bins) >= 3

    def run(p, chunks=None):
        p.set_option("keep_frame_samples", 1)
        if chunks:
            off = 0
            for c in chunks:
                p.feed_host(iq[off:off + c])
                off += c
            p.flush()
        else:
            p.feed_host(iq)
        bursts = p.poll_bursts()
        infos, samples = p.poll_frames()
        return dict(bursts=bursts, infos=infos, samples=samples, demods=p.poll_demods(), tagged=p.tagged)

    a = irdm.Pipeline(fs, max_chunk_samples=n, max_bursts_per_chunk=256)
    assert a.stat("rot_rows") == 0 and a.stat("rot_rows_cap") == 1024
    parity.compare(run(a), ref)
    assert a.stat("rot_rows") == len(bins) and a.stat("rot_builds") >= 1 and a.stat("rot_grows") == 0
    blocks, cap0 = a.stat("rot_blocks"), a.stat("rot_blocks_cap")
    assert len(bins) <= blocks < cap0 // 8           # (a few blocks per row, not whole rows)
    a.close()
    # one whole row's worth of blocks to begin with, the stream in chunks at pipeline_depth 2: the arena doubles while
    # earlier chains still run
    c = (n // 5) // 32768 * 32768
    b = irdm.Pipeline(fs, max_chunk_samples=n - 4 * c, max_bursts_per_chunk=256, pipeline_depth=2)
    b.set_option("rot_pool_rows", 1)
    small = b.stat("rot_blocks_cap")
    assert small < blocks
    parity.compare(run(b, chunks=[c, c, c, c, n - 4 * c]), ref)
    assert b.stat("rot_rows") == len(bins) and b.stat("rot_grows") >= 1 and b.stat("rot_blocks_cap") >= b.stat("rot_blocks") == blocks
    b.close()


def _short_then_long(fs=2_000_000):
    """a short burst and, later, a long one on the same carrier, and a long one on another carrier"""
    rng = np.random.default_rng(3)
    n = int(1.3 * fs) // 32768 * 32768
    f0, f1 = siggen.channel_freq(2), siggen.channel_freq(-5)
    bursts = [dict(start=520 * 2048 + 300, freq_hz=f0, payload=list(rng.integers(0, 4, 100))),
              dict(start=int(0.62 * fs), freq_hz=f0, payload=list(rng.integers(0, 4, 1350))),
              dict(start=int(0.80 * fs), freq_hz=f1, payload=list(rng.integers(0, 4, 1200))),
              dict(start=int(0.98 * fs), freq_hz=f0, payload=list(rng.integers(0, 4, 300)))]
    iq, _ = siggen.make_stream(fs, n, bursts, seed=8)
    return fs, iq


def test_rotator_rows_are_built_as_far_as_needed_and_extended():
    """A row of rotator checkpoints is built only as far as the bursts on its bin have needed so far (runs of 2048
    checkpoints) and continued from its last checkpoint when a longer burst arrives on the bin -- in the same chunk or a
    later one; the records are the oracle's."""
    fs, iq = _short_then_long()
    ref = orc.run_stream(iq, fs)
    assert len(ref.bursts) == 4 and 2 <= len({b.center_bin for b in ref.bursts}) <= 3
    n = len(iq)
    for chunks in ([n // 2 // 32768 * 32768, n - n // 2 // 32768 * 32768],):
        p = irdm.Pipeline(fs, max_chunk_samples=n, max_bursts_per_chunk=64, pipeline_depth=1 if chunks else 0)
        p.set_option("rot_prebuild", 0)               # rows on demand (a pipeline_depth >= 1 context prebuilds them by default)
        p.set_option("keep_frame_samples", 1)
        off = 0
        for c in (chunks or [n]):
            p.feed_host(iq[off:off + c])
            off += c
        p.flush()
        infos, samples = p.poll_frames()
        got = dict(bursts=p.poll_bursts(), infos=infos, samples=samples, demods=p.poll_demods(), tagged=p.tagged)
        st = {k: p.stat(k) for k in ("rot_rows", "rot_runs", "rot_ckpts", "rot_builds")}
        p.close()
        parity.compare(got, ref)
        # a row per centre bin; the first carrier's row in two builds (runs of 2048 checkpoints: the short burst's in the
        # first chunk, the long one's extension in the second)
        assert st["rot_runs"] > st["rot_rows"] >= 2, st
        assert st["rot_ckpts"] % 2048 == 0 and st["rot_ckpts"] >= 3 * 2048, st


def test_burst_scratch_rows_by_length_and_growth():
    """The per-burst scratch behind the decimator holds a batch's rows end to end (BurstWork::dec_off); one that starts
    with room for 64 outputs doubles, context by context, while other chains are in flight; records are the oracle's."""
    fs = 2_000_000
    n = int(0.9 * fs) // 32768 * 32768
    iq, _ = siggen.standard_scene(fs, n, 9, seed=5)
    ref = orc.run_stream(iq, fs)
    c = (n // 5) // 32768 * 32768
    got = parity.run_gpu(iq, fs, chunks=[c, c, c, c, n - 4 * c], depth=2, feed="ingest_lookahead", options={"scratch_outputs": 64})
    parity.compare(got, ref)
    assert got["stats"]["scratch_grows"] >= 1 and got["stats"]["scratch_peak"] > 64
    # the default scratch takes this stream without growing
    got = parity.run_gpu(iq, fs, chunks=[c, c, c, c, n - 4 * c], depth=2, feed="ingest_lookahead")
    parity.compare(got, ref)
    assert got["stats"]["scratch_grows"] == 0 and got["stats"]["scratch_peak"] <= got["stats"]["scratch_outputs"]


def test_context_footprint_10mhz():
    """a 10 MHz context for 16 Mi-sample chunks: no table of a row per FFT bin (4.5 GB in rounds 1-3), an arena of
    checkpoint blocks (0.57 GB to begin with) instead; no decimated / low-passed scratch for 4096 bursts of the longest length per chain (3 x 1.8 GB
    up to round 4), 1/16 of that to begin with; since round 6 plus every bin's row prebuilt as far as an ordinary burst needs it
    (rot_prebuild: 8192 bins x 10 runs x 16 KB = 1.3 GB instead of the 0.57 GB the empty arena had)"""
    u0 = _used()
    p = irdm.Pipeline(10_000_000, max_chunk_samples=16 * 1024 * 1024, max_bursts_per_chunk=4096, pipeline_depth=2)
    u1 = _used()
    p.close()
    assert u1 - u0 < 5.4e9, u1 - u0        # (8.8 GB with full-length scratch rows; rounds 1-3: 6.1 GB + the 4.5 GB table)
    u0 = _used()
    p = irdm.Pipeline(10_000_000, max_chunk_samples=16 * 1024 * 1024, max_bursts_per_chunk=4096, pipeline_depth=2)
    p.set_option("rot_prebuild", 0)
    u2 = _used()
    p.close()
    assert u2 - u0 < 4.4e9, u2 - u0        # rows on demand: as in round 5
    print("10 MHz context, rows on demand: %.2f GB" % ((u2 - u0) / 1e9))
    print("10 MHz context, 16 Mi-sample chunks, depth 2: %.2f GB" % ((u1 - u0) / 1e9))


def test_rotator_rows_prebuilt_in_the_background():
    """rot_prebuild (the default of a context with pipeline_depth >= 1): one background launch behind create builds every
    centre bin's row as far as a burst of ordinary length needs it -- a stream's first chunks then bring no checkpoint build in
    front of their chains (bursts of ordinary length: no build at all; a 54 ms burst extends its bin's row on demand) -- at
    n bins x runs x 16 KB of device memory; same records as the oracle, and as with the rows on demand."""
    fs = 2_000_000
    n = int(0.9 * fs) // 32768 * 32768
    iq, _ = siggen.standard_scene(fs, n, 9, seed=5)
    ref = orc.run_stream(iq, fs)
    c = (n // 5) // 32768 * 32768
    chunks = [c, c, c, c, n - 4 * c]
    before = _used()
    p = irdm.Pipeline(fs, max_chunk_samples=max(chunks), max_bursts_per_chunk=256, pipeline_depth=2)
    runs = p.stat("rot_prebuilt_runs")
    assert runs >= 2 and p.stat("rot_rows") == 2048 and p.stat("rot_blocks") == 2048 * runs
    p.close()
    got = parity.run_gpu(iq, fs, chunks=chunks, depth=2, feed="ingest_lookahead")
    parity.compare(got, ref)
    assert got["stats"]["rot_rows"] == 2048 and got["stats"]["rot_ckpts"] == 0 and got["stats"]["rot_grows"] == 0, got["stats"]
    off = parity.run_gpu(iq, fs, chunks=chunks, depth=2, feed="ingest_lookahead", options={"rot_prebuild": 0})
    parity.compare(off, ref)
    assert off["stats"]["rot_rows"] == len({b.center_bin for b in ref.bursts}) and off["stats"]["rot_ckpts"] > 0, off["stats"]
    # long bursts: their rows are continued from the prebuilt part
    fs2, iq2 = _short_then_long()
    ref2 = orc.run_stream(iq2, fs2)
    half = len(iq2) // 2 // 32768 * 32768
    got = parity.run_gpu(iq2, fs2, chunks=[half, len(iq2) - half], depth=1)
    parity.compare(got, ref2)
    assert got["stats"]["rot_rows"] == 2048 and got["stats"]["rot_ckpts"] >= 2048, got["stats"]
    # footprint at 10 MHz: 8192 bins x 10 runs x 16 KB on top of the on-demand margin
    base = _used()
    q = irdm.Pipeline(10_000_000, max_chunk_samples=16 * 1024 * 1024, max_bursts_per_chunk=4096, pipeline_depth=3)
    r10 = q.stat("rot_prebuilt_runs")
    with_pre = _used() - base
    q.close()
    assert 8 <= r10 <= 12
    assert with_pre < 9.0e9, with_pre            # (four batch contexts + ring + the prebuilt rows)
    del before
