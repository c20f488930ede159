"""-m gpu: the register-resident decimator (csrc/fir_reg.hip, `fir_layout` 3: columns of M samples in VGPRs, travelling
accumulators) against the oracle and against the LDS kernel it replaces (burst_downmix.c:663-672, :417-437;
rotator.h:36-46; simd_generic.c:86-96).

Everything downstream of the decimator is compared bit for bit (downmixed frame samples, start index, CFO, correlation
peaks, hard bits), so one wrong rounding in any of the 801 multiply-adds of any output shows.  Cases chosen for the
kernel's own edges: strip lengths from one double block to many (carries between blocks and strips' NR-column overlap),
all three sample formats (the column fetch converts in the load stage), a stream longer than the history ring fed in
several chunks (columns that wrap the ring, columns in the chunk and in the ring, stale tails read one reference ring
length back) and a ragged stream end (avail_end not a multiple of the 8-sample fetch piece)."""
import numpy as np
import pytest

import irdm
import orc
import parity
import siggen

pytestmark = pytest.mark.gpu


def _scene(fs, secs, nb, seed):
    n = int(secs * fs) // 32768 * 32768
    iq, _ = siggen.standard_scene(fs, n, nb, seed=seed)
    return iq


@pytest.fixture(scope="module")
def scene10():
    iq = _scene(10_000_000, 0.9, 7, seed=31)
    return iq, orc.run_stream(iq, 10_000_000)


@pytest.mark.parametrize("strip", [1, 2, 3, 7])
def test_strip_lengths_10mhz(scene10, strip):
    iq, ref = scene10
    try:
        got = parity.run_gpu(iq, 10_000_000, options={"fir_layout": 3, "fir_strip": strip})
    finally:
        _restore()
    s = parity.compare(got, ref)
    assert s["demods"] >= 4, s


def _restore():
    p = irdm.Pipeline(2_000_000, max_chunk_samples=65536, max_bursts_per_chunk=64)
    p.set_option("fir_layout", 3)
    p.set_option("fir_strip", 3)
    p.close()


def test_lds_kernel_still_agrees_10mhz(scene10):
    iq, ref = scene10
    try:
        got = parity.run_gpu(iq, 10_000_000, options={"fir_layout": 2})
    finally:
        _restore()
    parity.compare(got, ref)


@pytest.mark.parametrize("fs,fmt", [(10_000_000, irdm.FMT_CI16), (10_000_000, irdm.FMT_CI8),
                                    (12_000_000, irdm.FMT_CI16), (12_000_000, irdm.FMT_CI8), (12_000_000, irdm.FMT_CF32)])
def test_formats(fs, fmt):
    iq = _scene(fs, 1.0, 6, seed=fs // 1_000_000 + fmt)
    if fmt == irdm.FMT_CI16:
        x = siggen.to_ci16(iq)
    elif fmt == irdm.FMT_CI8:
        x = siggen.to_ci8(iq)
    else:
        x = iq
    ref = orc.run_stream(x, fs, fmt=fmt)
    got = parity.run_gpu(x, fs, fmt=fmt)
    s = parity.compare(got, ref)
    assert s["demods"] >= 3, s


def test_ring_wrap_chunks_and_ragged_end_10mhz_ci8():
    """3.1 s of 10 MHz ci8 in chunks of 4 Mi samples: the history ring (2 s reference ring + the longest burst window +
    the chunk) wraps, burst windows straddle chunk boundaries, and the stream ends 1234 samples past a feed block."""
    fs = 10_000_000
    n = int(3.1 * fs) // 32768 * 32768 + 1234
    iq, _ = siggen.standard_scene(fs, n, 24, seed=77)
    x = siggen.to_ci8(iq)
    ref = orc.run_stream(x, fs, fmt=irdm.FMT_CI8)
    chunk = 4 * 1024 * 1024
    sizes = [chunk] * (n // chunk) + ([n % chunk] if n % chunk else [])
    got = parity.run_gpu(x, fs, fmt=irdm.FMT_CI8, chunks=sizes)
    s = parity.compare(got, ref)
    assert s["demods"] >= 12, s
    stale = [b for b in ref.bursts if b.avail_end < b.start + b.num_samples]
    assert stale, "no burst of the scene had a stale tail"
