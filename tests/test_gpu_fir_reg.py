"""-m gpu: the register-resident decimators (csrc/fir_reg.hip, `fir_layout` 3: columns of M samples in VGPRs, travelling
accumulators) -- fir_decimate_kernel_f in the order of the reference's AVX2 kernel (simd_avx2.c:62-108, the default) and
fir_decimate_kernel_r in its scalar order (simd_generic.c:86-96, option fir_order 0) -- against the oracle in the same
order and against the LDS kernel (burst_downmix.c:663-672, :417-437; rotator.h:36-46).

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
    """the default decimator (fir_decimate_kernel_f: four fused accumulators per output, simd_avx2.c:62-108) whole, in
    chunks (strips in the chunk, in the ring, across the boundary) and with a capped grid (workgroups walking the strips)"""
    iq, ref, ref0 = scene10
    assert any(a.center_offset != b.center_offset for a, b in zip(ref.frames, ref0.frames)), "the two orders should differ in rounding"
    s = parity.compare(parity.run_gpu(iq, 10_000_000), ref)
    assert s["demods"] >= 4, s
    n = len(iq)
    c = (n // 3) // 32768 * 32768
    parity.compare(parity.run_gpu(iq, 10_000_000, chunks=[c, c, n - 2 * c], depth=1), ref)
    try:
        parity.compare(parity.run_gpu(iq, 10_000_000, options={"fir_grid": 300}), ref)      # fixed shares
        parity.compare(parity.run_gpu(iq, 10_000_000, options={"fir_grid": 300, "fir_claim": 1}), ref)   # strips claimed from a counter
        parity.compare(parity.run_gpu(iq, 10_000_000, options={"fir_grid": 0}), ref)      # one workgroup per strip
    finally:
        _restore()


@pytest.mark.parametrize("strip", [1, 2, 3, 7])
def test_strip_lengths_10mhz(scene10, strip):
    iq, _, ref0 = scene10
    try:
        got = parity.run_gpu(iq, 10_000_000, options={"fir_order": 0, "fir_layout": 3, "fir_strip": strip})
    finally:
        _restore()
    s = parity.compare(got, ref0)
    assert s["demods"] >= 4, s


def _restore():
    p = irdm.Pipeline(2_000_000, max_chunk_samples=65536, max_bursts_per_chunk=64)
    p.set_option("fir_layout", 3)
    p.set_option("fir_strip", 3)
    p.set_option("fir_grid", -1)
    p.set_option("fir_claim", 0)
    p.set_option("fir_order", 1)
    p.close()


def test_lds_kernel_still_agrees_10mhz(scene10):
    iq, _, ref0 = scene10
    try:
        got = parity.run_gpu(iq, 10_000_000, options={"fir_order": 0, "fir_layout": 2})
    finally:
        _restore()
    parity.compare(got, ref0)


def test_runtime_m_kernel_in_both_orders_10mhz(scene10):
    """the runtime-M kernel (what 2 / 4 MHz and unaligned sources take in the AVX2 order; option fir_generic)"""
    iq, ref, ref0 = scene10
    try:
        parity.compare(parity.run_gpu(iq, 10_000_000, options={"fir_generic": 1}), ref)
        parity.compare(parity.run_gpu(iq, 10_000_000, options={"fir_generic": 1, "fir_order": 0}), ref0)
    finally:
        p = irdm.Pipeline(2_000_000, max_chunk_samples=65536, max_bursts_per_chunk=64)
        p.set_option("fir_generic", 0)
        p.close()
        _restore()


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


def test_matrix_core_decimator_10mhz(scene10):
    """fir_layout 4 (fir_decimate_kernel_x): the AVX2 order -- four fused accumulators per output -- as a Toeplitz-taps x
    samples product on v_mfma_f32_16x16x4_f32; downmixed samples bit for bit the oracle's: whole, in chunks (strips in the
    chunk, in the ring, across the boundary), with a capped grid (a few workgroups walking all octets of strips)"""
    iq, ref, _ = scene10
    try:
        s = parity.compare(parity.run_gpu(iq, 10_000_000, options={"fir_layout": 4}), ref)
        assert s["demods"] >= 4, s
        n = len(iq)
        c = (n // 3) // 32768 * 32768
        parity.compare(parity.run_gpu(iq, 10_000_000, chunks=[c, c, n - 2 * c], depth=1, options={"fir_layout": 4}), ref)
        parity.compare(parity.run_gpu(iq, 10_000_000, options={"fir_layout": 4, "fir_grid": 3}), ref)
    finally:
        _restore()


@pytest.mark.parametrize("fs,fmt", [(10_000_000, irdm.FMT_CI16), (12_000_000, irdm.FMT_CI8), (12_000_000, irdm.FMT_CF32)])
def test_matrix_core_decimator_formats(fs, fmt):
    iq = _scene(fs, 1.0, 6, seed=fs // 1_000_000 + fmt)
    x = siggen.to_ci16(iq) if fmt == irdm.FMT_CI16 else siggen.to_ci8(iq) if fmt == irdm.FMT_CI8 else iq
    ref = orc.run_stream(x, fs, fmt=fmt)
    try:
        got = parity.run_gpu(x, fs, fmt=fmt, options={"fir_layout": 4})
    finally:
        _restore()
    s = parity.compare(got, ref)
    assert s["demods"] >= 3, s


def test_matrix_core_decimator_ring_wrap_and_stale_tails_10mhz_ci8():
    """the ring wraps, windows straddle chunk boundaries, stale tails, a ragged end (the scene of the test above)"""
    fs = 10_000_000
    n = int(3.1 * fs) // 32768 * 32768 + 1234
    iq, _ = siggen.standard_scene(fs, n, 24, seed=77)
    x = siggen.to_ci8(iq)
    ref = orc.run_stream(x, fs, fmt=irdm.FMT_CI8)
    chunk = 4 * 1024 * 1024
    sizes = [chunk] * (n // chunk) + ([n % chunk] if n % chunk else [])
    try:
        got = parity.run_gpu(x, fs, fmt=irdm.FMT_CI8, chunks=sizes, options={"fir_layout": 4})
    finally:
        _restore()
    s = parity.compare(got, ref)
    assert s["demods"] >= 12, s
