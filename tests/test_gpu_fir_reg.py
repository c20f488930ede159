"""-m gpu: the register-resident decimators (csrc/fir_reg.hip: columns of M samples in VGPRs, travelling
accumulators) -- fir_decimate_kernel_f in the order of the reference's AVX2 kernel (simd_avx2.c:62-108, the default) and
fir_decimate_kernel_r in its scalar order (simd_generic.c:86-96, option fir_order 0) -- against the oracle in the same
order, and the any-M kernel (what 2 / 4 MHz take; test hook fir_generic) in both (burst_downmix.c:663-672, :417-437;
rotator.h:36-46).

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
    """the scene, the oracle's records with the decimating FIR in the order of the reference's AVX2 kernel (the default:
    fir_decimate_kernel_f) and in its scalar order (option fir_order 0: fir_decimate_kernel_r and the LDS kernels)"""
    iq = _scene(10_000_000, 0.9, 7, seed=31)
    ref = orc.run_stream(iq, 10_000_000)
    try:
        orc.set_fir_order(0)
        ref0 = orc.run_stream(iq, 10_000_000)
    finally:
        orc.set_fir_order(1)
    return iq, ref, ref0


def test_fused_four_accumulator_kernel_10mhz(scene10):
    """the default decimator (fir_decimate_kernel_f: four fused accumulators per output, simd_avx2.c:62-108) whole and in
    chunks (strips in the chunk, in the ring, across the boundary)"""
    iq, ref, ref0 = scene10
    assert any(a.center_offset != b.center_offset for a, b in zip(ref.frames, ref0.frames)), "the two orders should differ in rounding"
    s = parity.compare(parity.run_gpu(iq, 10_000_000), ref)
    assert s["demods"] >= 4, s
    n = len(iq)
    c = (n // 3) // 32768 * 32768
    parity.compare(parity.run_gpu(iq, 10_000_000, chunks=[c, c, n - 2 * c], depth=1), ref)


def test_scalar_order_kernel_10mhz(scene10):
    """fir_order 0 (--no-simd): fir_decimate_kernel_r, one accumulator per output, every product and sum rounded
    (simd_generic.c:86-96), whole and in chunks"""
    iq, _, ref0 = scene10
    s = parity.compare(parity.run_gpu(iq, 10_000_000, options={"fir_order": 0}), ref0)
    assert s["demods"] >= 4, s
    n = len(iq)
    c = (n // 3) // 32768 * 32768
    parity.compare(parity.run_gpu(iq, 10_000_000, chunks=[c, c, n - 2 * c], depth=1, options={"fir_order": 0}), ref0)


def test_runtime_m_kernel_in_both_orders_10mhz(scene10):
    """the any-M kernel (what 2 / 4 MHz and unaligned sources take; test hook fir_generic), two contexts of one process in
    different orders side by side (the switches are per pipeline)"""
    iq, ref, ref0 = scene10
    parity.compare(parity.run_gpu(iq, 10_000_000, options={"fir_generic": 1}), ref)
    parity.compare(parity.run_gpu(iq, 10_000_000, options={"fir_generic": 1, "fir_order": 0}), ref0)
    parity.compare(parity.run_gpu(iq, 10_000_000), ref)              # (nothing sticks to the process)


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
