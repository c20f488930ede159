"""-m gpu: option packed_records -- the compact frame record irdm_demod_packed_t (what frame_output_print reads of a
demod_frame_t, frame_output.c:168-197; hard bits 8 per byte, no LLRs; 136 bytes per burst over PCIe instead of 4.5 KB)
against the full records of the same stream and against the oracle, at pipeline_depth 0 and 2."""
import numpy as np
import pytest

import irdm
import orc
import siggen

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth", [0, 2])
def test_packed_records_equal_the_full_ones(depth):
    fs = 2_000_000
    n = int(2.0 * fs) // 32768 * 32768
    iq, _ = siggen.standard_scene(fs, n, 9, seed=23, uplink_every=4)
    ref = orc.run_stream(iq, fs)
    chunks = [n // 4 // 32768 * 32768] * 3
    chunks.append(n - sum(chunks))

    def run(packed):
        p = irdm.Pipeline(fs, max_chunk_samples=max(chunks), max_bursts_per_chunk=256, pipeline_depth=depth)
        p.set_option("packed_records", packed)
        off = 0
        for c in chunks:
            p.feed_host(iq[off:off + c])
            off += c
        if depth:
            p.flush()
        out = (p.poll_bursts(), p.poll_demods_packed() if packed else p.poll_demods(), p.poll_demods() if packed else [],
               p.poll_frames()[0] if packed else [])
        p.close()
        return out

    fb, fd, _, _ = run(0)
    pb, pd, leftover, frames = run(1)
    assert leftover == [] and frames == []                      # the packed mode queues nothing else
    assert len(fd) == len(pd) == len(ref.demods) >= 5
    assert [b.id for b in pb] == [b.id for b in fb] == [r.id for r in ref.bursts]
    for f, q, r in zip(fd, pd, ref.demods):
        for fld in ("id", "timestamp", "direction", "confidence", "n_symbols", "n_payload_symbols", "n_bits", "ok"):
            assert getattr(f, fld) == getattr(q, fld), fld
        for fld in ("center_frequency", "magnitude", "noise", "level", "total_phase"):
            assert np.float64(getattr(f, fld)).view(np.uint64) == np.float64(getattr(q, fld)).view(np.uint64), fld
        bits = np.unpackbits(np.frombuffer(bytes(q.bits), np.uint8))[:q.n_bits]
        assert bytes(bits) == bytes(f.bits[:f.n_bits]) == bytes(r.bits[:r.n_bits])
        assert not np.unpackbits(np.frombuffer(bytes(q.bits), np.uint8))[q.n_bits:].any()
