"""-m gpu: the HIP hot path, called through the C-ABI, against the CPU oracle
on the same seeded inputs.  Integer fields, burst indices, downmixed frame
samples and hard bits: bit-exact.  Soft demod outputs: 1e-4 (north_star)."""
import ctypes as C
import os

import numpy as np
import pytest

import irdm
import orc
import parity
import siggen

pytestmark = pytest.mark.gpu


def _window(n):
    w = np.zeros(n, np.float32)
    orc.lib().orc_blackman_window(orc.fptr(w), n)
    return (w / np.float32(0.42)).astype(np.float32)


def _oracle_mags(fs, frames):
    L = orc.lib()
    det = L.orc_detector_create(1.622e9, fs, 0.0, 0)
    n = L.orc_detector_fft_size(det)
    out = np.zeros((len(frames), n), np.float32)
    for i, f in enumerate(frames):
        L.orc_detector_magnitude_frame(det, orc.fptr(np.ascontiguousarray(f).view(np.float32)), orc.fptr(out[i]))
    L.orc_detector_destroy(det)
    return out


@pytest.mark.parametrize("fs,n", [(2_000_000, 2048), (4_000_000, 4096), (10_000_000, 8192), (12_000_000, 16384)])
def test_gpu_burst_fft_bit_exact(fs, n):
    """a3/a12: gpu_burst_fft_process == window + pinned FFT + fftshift |.|^2, bit for bit."""
    rng = np.random.default_rng(n)
    frames = ((rng.standard_normal((16, n)) + 1j * rng.standard_normal((16, n))) * 0.01).astype(np.complex64)
    frames[3, :] = 0
    frames[4, :] = 0.25                      # DC: the reference's Vulkan self-test input
    frames[5] += (0.05 * np.exp(2j * np.pi * 0.123 * np.arange(n))).astype(np.complex64)
    g = irdm.GpuBurstFFT(n, 16, _window(n))
    got = g.process(frames)
    want = _oracle_mags(fs, frames)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.all(got[3] == 0)
    # partial batch and error conventions (opencl/burst_fft.c:325-326, burst_detect.c:659-665)
    assert np.array_equal(g.process(frames[:5]).view(np.uint32), want[:5].view(np.uint32))
    assert g.process_rc(frames, 0) == -1
    assert g.process_rc(frames, 17) == -1
    g.close()


def test_gpu_burst_fft_create_rejects_bad_arguments():
    L = irdm.lib()
    w = _window(2048)
    assert not L.gpu_burst_fft_create(2000, 16, irdm._fp(w))     # not a power of two -> NULL => CPU fallback in caller
    assert not L.gpu_burst_fft_create(2048, 0, irdm._fp(w))
    L.gpu_burst_fft_destroy(None)                                # NULL-safe
    # the DC self-test at create (vulkan/burst_fft.c:324-394): a context that does not compute is not handed out
    bad = w.copy()
    bad[7] = np.nan
    assert not L.gpu_burst_fft_create(2048, 16, irdm._fp(bad))
    g = L.gpu_burst_fft_create(2048, 16, irdm._fp(w))
    assert g
    L.gpu_burst_fft_destroy(g)


def _scene_2m(seed=11, n_bursts=8, secs=2.4, **kw):
    fs = 2_000_000
    return fs, siggen.standard_scene(fs, int(secs * fs), n_bursts, seed=seed, uplink_every=4, **kw)[0]


def test_pipeline_2mhz_cf32_full_parity():
    fs, iq = _scene_2m()
    ref = orc.run_stream(iq, fs)
    got = parity.run_gpu(iq, fs)
    s = parity.compare(got, ref)
    assert s["bursts"] >= 8 and s["demods"] >= 5, s
    # RAW lines: identical text except (possibly) the last printed digit of soft fields
    for a, b in zip(irdm.format_raw(got["demods"]), ref.raw_lines()):
        fa, fb = parity.raw_fields(a), parity.raw_fields(b)
        assert fa[0] == fb[0] and fa[1] == fb[1] and fa[3:6] == fb[3:6] and fa[7:] == fb[7:]
        assert abs(fa[2] - fb[2]) <= 1 and abs(fa[6] - fb[6]) <= 1e-4


def test_pipeline_chunked_equals_single_chunk():
    """Detector state, active bursts and the IQ history ring carry across feed calls."""
    fs, iq = _scene_2m(seed=12)
    ref = orc.run_stream(iq, fs)
    n = len(iq)
    blk = 32768
    chunks = [blk * 7, blk * 3, blk * 40, blk * 1, blk * 20]
    rest = n - sum(chunks)
    chunks += [rest // blk * blk, rest % blk] if rest % blk else [rest]
    chunks = [c for c in chunks if c > 0]
    got = parity.run_gpu(iq, fs, chunks=chunks)
    parity.compare(got, ref)


def test_pipeline_ci16_and_ci8():
    fs, iq = _scene_2m(seed=13, n_bursts=5, secs=2.0)
    i16 = siggen.to_ci16(iq)
    ref = orc.run_stream(i16, fs, fmt=1)
    got = parity.run_gpu(i16, fs, fmt=irdm.FMT_CI16)
    s = parity.compare(got, ref)
    assert s["demods"] >= 3, s
    i8 = siggen.to_ci8(iq * 8)
    ref = orc.run_stream(i8, fs, fmt=0)
    got = parity.run_gpu(i8, fs, fmt=irdm.FMT_CI8)
    parity.compare(got, ref)


def test_ci16_device_resident_and_pinned_host_feed():
    """Raw int16 pairs are narrowed in the kernels' load stage (main.c:245-246): a device-resident ci16 chunk
    (irdm_feed_device) and a pinned host buffer (irdm_host_alloc + irdm_feed_host), both chunked at pipeline_depth 1,
    give the records the oracle gives for the whole stream."""
    fs, iq = _scene_2m(seed=17, n_bursts=5, secs=2.0)
    i16 = siggen.to_ci16(iq)
    n = len(i16) // 2
    ref = orc.run_stream(i16, fs, fmt=1)
    chunks = [32768 * 30, 32768 * 17, n - 32768 * 47]
    # device-resident: the buffer is allocated and filled through the library itself (irdm_device_alloc / _upload), so
    # this half never depends on another HIP runtime in the process
    dptr = irdm.device_buffer(i16)
    try:
        p = irdm.Pipeline(fs, fmt=irdm.FMT_CI16, max_chunk_samples=max(chunks), max_bursts_per_chunk=1024, pipeline_depth=1)
        p.set_option("keep_frame_samples", 1)
        off = 0
        for c in chunks:
            p.feed_device(dptr + off * 4, c, None)
            off += c
        p.flush()
        infos, samples = p.poll_frames()
        got = dict(bursts=p.poll_bursts(), infos=infos, samples=samples, demods=p.poll_demods(), tagged=p.tagged)
        p.close()
    finally:
        irdm.device_free(dptr)
    s1 = parity.compare(got, ref)
    assert s1["demods"] >= 3, s1
    # pinned host buffer
    ptr, view = irdm.host_alloc(max(chunks) * 4)
    p = irdm.Pipeline(fs, fmt=irdm.FMT_CI16, max_chunk_samples=max(chunks), max_bursts_per_chunk=1024, pipeline_depth=1)
    p.set_option("keep_frame_samples", 1)
    raw = i16.view(np.uint8)
    off = 0
    for c in chunks:
        view[:c * 4] = raw[off * 4:(off + c) * 4]
        p.feed_host_ptr(ptr, c)
        off += c
    p.flush()
    infos, samples = p.poll_frames()
    got = dict(bursts=p.poll_bursts(), infos=infos, samples=samples, demods=p.poll_demods(), tagged=p.tagged)
    p.close()
    irdm.host_free(ptr)
    parity.compare(got, ref)


def test_detect_only_mode_emits_the_same_bursts():
    """option "detect_only" (stage A alone, BASELINE config 2): the burst records of the full pipeline, nothing else"""
    fs, iq = _scene_2m(seed=23, n_bursts=6, secs=2.0)
    ref = orc.run_stream(iq, fs)
    p = irdm.Pipeline(fs, max_chunk_samples=32768 * 40, max_bursts_per_chunk=1024, pipeline_depth=1)
    p.set_option("detect_only", 1)
    try:
        off = 0
        while off < len(iq):
            c = min(32768 * 40, len(iq) - off)
            p.feed_host(iq[off:off + c])
            off += c
        p.flush()
        bursts = p.poll_bursts()
        assert p.poll_demods() == [] and p.poll_frames()[0] == []
        assert p.tagged == ref.n_tagged and len(bursts) == len(ref.bursts) >= 6
        for g, r in zip(bursts, ref.bursts):
            for f in ("id", "start", "stop", "last_active", "center_bin", "num_samples", "avail_end"):
                assert getattr(g, f) == getattr(r, f), f
            assert parity.bits_of(g.magnitude) == parity.bits_of(r.magnitude)
            assert parity.bits_of(g.noise) == parity.bits_of(r.noise)
    finally:
        p.close()


def test_kernel_variants_agree():
    """The paths a default run at the standard rates never takes: the runtime-tap-count instances of the post filters (test
    hook post_generic), the any-M decimator in both orders (fir_generic), the reference's generic forms throughout
    (fir_order 0) -- same records as the oracle in the same order.  The switches are fields of the pipeline: nothing has to be
    restored afterwards."""
    fs, iq = _scene_2m(seed=19, n_bursts=6, secs=2.0)
    ref = orc.run_stream(iq, fs)
    try:
        orc.set_fir_order(0)                   # the dispatched kernels in the reference's generic forms (--no-simd)
        ref0 = orc.run_stream(iq, fs)
    finally:
        orc.set_fir_order(1)
    for opts in ({"fir_generic": 1}, {"post_generic": 1}, {"fir_order": 0}, {"fir_order": 0, "fir_generic": 1},
                 {"fir_order": 0, "post_generic": 1}):
        parity.compare(parity.run_gpu(iq, fs, options=opts), ref if opts.get("fir_order", 1) else ref0)
    parity.compare(parity.run_gpu(iq, fs), ref)

def test_known_answer_bits_from_reference_docs():
    """ARCHITECTURE.md:264/:270 -- the documented PRBS15 RAW line: 179 payload symbols, same bits."""
    fs = 2_000_000
    q = siggen.bits_to_quadrants(siggen.KNOWN_ANSWER_BITS)
    iq, _ = siggen.make_stream(fs, int(1.0 * fs) // 32768 * 32768,
                               [dict(start=520 * 2048 + 777, freq_hz=siggen.channel_freq(3), quads=[0] * 16 + q)],
                               seed=5)
    got = parity.run_gpu(iq, fs)
    assert len(got["demods"]) == 1
    d = got["demods"][0]
    assert d.n_payload_symbols == 179
    assert "".join(str(b) for b in d.bits[:d.n_bits]) == siggen.KNOWN_ANSWER_BITS
    parity.compare(got, orc.run_stream(iq, fs))


def test_burst_window_samples_and_stale_tail():
    """a11: ringbuf_extract semantics incl. the not-yet-written tail of the ring."""
    fs, iq = _scene_2m(seed=14, n_bursts=4, secs=1.6)
    L = orc.lib()
    recs, sams = [], []

    @orc.BURST_CB
    def cb(rec, samples, user):
        r = rec.contents
        recs.append(orc.BurstRec.from_buffer_copy(r))
        sams.append(np.ctypeslib.as_array(samples, (2 * r.num_samples,)).copy().view(np.complex64))

    det = L.orc_detector_create(1.622e9, fs, 0.0, 0)
    for off in range(0, len(iq), 32768):
        blk = np.ascontiguousarray(iq[off:off + 32768])
        L.orc_detector_feed_cf32(det, orc.fptr(blk.view(np.float32)), len(blk), cb, None)
    L.orc_detector_destroy(det)
    assert len(recs) >= 4
    p = irdm.Pipeline(fs, max_chunk_samples=len(iq), max_bursts_per_chunk=256)
    p.feed_host(iq)
    bursts = p.poll_bursts()
    assert [b.id for b in bursts] == [r.id for r in recs]
    stale = 0
    for i, (b, r, s) in enumerate(zip(bursts, recs, sams)):
        g = p.burst_samples(i, int(b.num_samples))
        assert np.array_equal(g.view(np.uint32), s.view(np.uint32)), b.id
        stale += int(b.start + b.num_samples > b.avail_end)
    p.close()


def test_detector_magnitudes_and_baseline_10mhz():
    """cfg2 shape (10 MHz, N=8192): magnitudes of a chunk and the baseline sum after priming."""
    fs = 10_000_000
    n = 8192 * 600
    iq, _ = siggen.make_stream(fs, n, [dict(start=8192 * 560, freq_hz=siggen.channel_freq(20), payload=[0, 1, 2, 3] * 40)], seed=21)
    L = orc.lib()
    det = L.orc_detector_create(1.622e9, fs, 0.0, 0)
    sink = np.zeros((600, 8192), np.float32)
    L.orc_detector_set_mag_sink(det, orc.fptr(sink), 600)
    for off in range(0, n, 32768):
        blk = np.ascontiguousarray(iq[off:off + 32768])
        L.orc_detector_feed_cf32(det, orc.fptr(blk.view(np.float32)), len(blk), orc.BURST_CB(0), None)
    base = np.ctypeslib.as_array(L.orc_detector_baseline_sum(det), (8192,)).copy()
    L.orc_detector_destroy(det)
    p = irdm.Pipeline(fs, max_chunk_samples=n, max_bursts_per_chunk=256)
    p.feed_host(iq)
    mags = p.last_magnitudes(600)
    assert np.array_equal(mags.view(np.uint32), sink.view(np.uint32))
    assert np.array_equal(p.baseline_sum().view(np.uint32), base.view(np.uint32))
    p.close()


@pytest.mark.parametrize("fs,secs,nb", [(10_000_000, 0.9, 6), (12_000_000, 1.1, 5)])
def test_pipeline_10_and_12_mhz(fs, secs, nb):
    """cfg3 / cfg4 shapes at oracle-sized lengths."""
    n = int(secs * fs) // 32768 * 32768
    iq, _ = siggen.standard_scene(fs, n, nb, seed=fs // 1_000_000)
    ref = orc.run_stream(iq, fs)
    got = parity.run_gpu(iq, fs)
    s = parity.compare(got, ref)
    assert s["demods"] >= 3, s


def test_eof_burst_not_emitted_and_ragged_last_chunk():
    """bursts still active at EOF are never emitted; trailing < N samples unprocessed."""
    fs = 2_000_000
    n = 520 * 2048 + 60000 + 1234
    iq, _ = siggen.make_stream(fs, n, [dict(start=520 * 2048 + 20000, freq_hz=siggen.channel_freq(4), payload=[1] * 150)], seed=3)
    ref = orc.run_stream(iq, fs)
    got = parity.run_gpu(iq, fs, chunks=[32768 * 20, n - 32768 * 20])
    assert ref.n_tagged == 0 and got["tagged"] == 0
    assert got["n_samples"] == n


def test_sparse_scan_is_used_and_equals_dense_scan():
    """The sparse detector scan (scan_fast.hip) is what normally runs, and it yields exactly
    the state and bursts of the dense scan (and of the oracle)."""
    fs, iq = _scene_2m(seed=15, n_bursts=12, secs=3.0)
    ref = orc.run_stream(iq, fs)
    blk = 32768
    n = len(iq)
    chunks = [blk * 40, blk * 50, n - blk * 90]
    fast = parity.run_gpu(iq, fs, chunks=chunks, scan_mode=0)
    dense = parity.run_gpu(iq, fs, chunks=chunks, scan_mode=1)
    parity.compare(fast, ref)
    parity.compare(dense, ref)
    assert fast["stats"]["scan_fast_chunks"] == 3 and fast["stats"]["scan_fallbacks"] == 0, fast["stats"]
    assert fast["stats"]["scan_dense_frames"] == 512          # only the priming frames of the stream
    assert dense["stats"]["scan_fast_chunks"] == 0


def test_sparse_scan_falls_back_when_noise_floor_drops():
    """A 12 dB drop of the noise floor inside a chunk makes the prefilter lists stale; the scan
    must notice (safety net / validation), fall back to the dense scan, and stay exact."""
    fs = 2_000_000
    n = 32768 * 110
    iq, _ = siggen.standard_scene(fs, n, 6, seed=16, first_start=600 * 2048)
    iq = iq.copy()
    cut = 32768 * 45
    # weaken everything (noise and bursts) after `cut`; bursts stay well above the new floor
    iq[cut:] *= np.float32(0.25)
    iq[cut:] += siggen.standard_scene(fs, n - cut, 4, seed=17, first_start=700 * 2048, amp=0.03)[0] * np.float32(0.0)
    ref = orc.run_stream(iq, fs)
    got = parity.run_gpu(iq, fs, chunks=[32768 * 40, n - 32768 * 40])
    parity.compare(got, ref)
    assert got["stats"]["scan_fast_chunks"] + got["stats"]["scan_fallbacks"] == 2


def test_time_chunk_handoff_equals_single_context():
    """chunk k+1 processed by a DIFFERENT context that received the detector state
    (irdm_export_state/irdm_import_state) and the preceding samples (irdm_seed_history)
    gives exactly the single-context result (what rank k+1 does in a time-sharded run)."""
    import sharding
    fs, iq = _scene_2m(seed=18, n_bursts=10, secs=2.6)
    ref = orc.run_stream(iq, fs)
    n = len(iq)
    # cut the stream inside a burst window so that it straddles the two contexts
    cut = None
    for rb in ref.bursts:
        c = (rb.start + rb.num_samples // 2) // 32768 * 32768
        if rb.start < c < rb.start + rb.num_samples and c > 600 * 2048:
            cut = int(c)
            break
    assert cut is not None
    a = irdm.Pipeline(fs, max_chunk_samples=n, max_bursts_per_chunk=512)
    b = irdm.Pipeline(fs, max_chunk_samples=n, max_bursts_per_chunk=512)
    for p in (a, b):
        p.set_option("keep_frame_samples", 1)
    a.feed_host(iq[:cut])
    blob = a.export_state()
    assert len(blob) == a.L.irdm_state_bytes(a.h)
    ov = min(cut, sharding.required_overlap(fs, 2048))
    b.seed_history(iq[cut - ov:cut], cut)
    b.import_state(blob)
    b.feed_host(iq[cut:])
    got = dict(bursts=a.poll_bursts() + b.poll_bursts(), demods=a.poll_demods() + b.poll_demods(),
               tagged=b.tagged, n_samples=b.sample_count)
    ia, sa = a.poll_frames()
    ib, sb_ = b.poll_frames()
    got["infos"], got["samples"] = ia + ib, sa + sb_
    s = parity.compare(got, ref)
    assert s["bursts"] >= 10
    # at least one burst window straddles the cut (reads the seeded history)
    assert any(bb.start < cut < bb.start + bb.num_samples for bb in got["bursts"])
    a.close()
    b.close()


@pytest.mark.parametrize("fmt", ["cf32", "ci8"])
def test_pipeline_depth_1_overlapped_equals_synchronous(fmt):
    """pipeline_depth 1: the per-burst stages of chunk k run during feed(k+1) from the history ring,
    overlapped with the detector; after irdm_flush the records equal the synchronous ones."""
    fs, iq = _scene_2m(seed=19, n_bursts=12, secs=3.0)
    blk = 32768
    n = len(iq)
    chunks = [blk * 30, blk * 41, blk * 50, n - blk * 121]
    if fmt == "cf32":
        ref = orc.run_stream(iq, fs)
        got = parity.run_gpu(iq, fs, chunks=chunks, depth=1)
    else:
        i8 = siggen.to_ci8(iq * 8)
        ref = orc.run_stream(i8, fs, fmt=0)
        got = parity.run_gpu(i8, fs, fmt=irdm.FMT_CI8, chunks=chunks, depth=1)
    s = parity.compare(got, ref)
    assert s["bursts"] >= 12


def test_stage_b_alone_matches_oracle_downmix():
    """burst_downmix_process for individual bursts (stage B entry point) == oracle, bit for bit."""
    fs, iq = _scene_2m(seed=20, n_bursts=6, secs=2.0)
    L = orc.lib()
    recs, sams = [], []

    @orc.BURST_CB
    def cb(rec, samples, user):
        r = rec.contents
        recs.append(orc.BurstRec.from_buffer_copy(r))
        sams.append(np.ctypeslib.as_array(samples, (2 * r.num_samples,)).copy().view(np.complex64))

    det = L.orc_detector_create(1.622e9, fs, 0.0, 0)
    for off in range(0, len(iq), 32768):
        blk = np.ascontiguousarray(iq[off:off + 32768])
        L.orc_detector_feed_cf32(det, orc.fptr(blk.view(np.float32)), len(blk), cb, None)
    L.orc_detector_destroy(det)
    dm = L.orc_downmix_create()
    p = irdm.Pipeline(fs, max_chunk_samples=32768 * 8, max_bursts_per_chunk=64)
    n_frames = 0
    for r, s_ in zip(recs, sams):
        want = orc.Frame()
        ok = L.orc_downmix_process(dm, C.byref(r), orc.fptr(s_.view(np.float32)), 1.622e9, fs, 2048,
                                   1700000000 * 10**9, C.byref(want))
        info = irdm.Burst.from_buffer_copy(bytes(r))
        fi, frame = p.downmix_burst(info, s_)
        assert fi.drop_reason == want.drop_reason and (frame is not None) == bool(ok)
        if ok:
            n_frames += 1
            ws = np.ctypeslib.as_array(want.samples)[:2 * want.num_samples].view(np.complex64)
            assert fi.num_samples == want.num_samples and fi.uw_start_idx == want.uw_start_idx
            assert fi.timestamp == want.timestamp and fi.center_frequency == want.center_frequency
            assert np.array_equal(frame.view(np.uint32), ws.view(np.uint32))
    assert n_frames >= 4
    L.orc_downmix_destroy(dm)
    p.close()


def test_cli_binary_raw_lines(tmp_path):
    """The plain-C file-mode binary over the C-ABI prints the RAW lines the oracle prints
    (test-configurations.sh methodology: same file, compare lines; soft digits within tolerance)."""
    import subprocess
    exe = os.path.join(os.path.dirname(irdm.LIB_PATH), "iridium-sniffer-hip")
    if not os.path.exists(exe):
        irdm.build(force=True)
    fs, iq = _scene_2m(seed=21, n_bursts=6, secs=1.8)
    for ext, data, fmt in (("cf32", iq, 2), ("ci16", siggen.to_ci16(iq), 1)):
        path = tmp_path / ("scene." + ext)
        np.ascontiguousarray(data).tofile(path)
        ref = orc.run_stream(data, fs, fmt=fmt)
        out = subprocess.run([exe, "-f", str(path), "-r", str(fs), "--file-info", "golden", "--chunk", str(32768 * 16)],
                             capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert "tagged %d bursts total" % ref.n_tagged in out.stderr
        lines = [l for l in out.stdout.splitlines() if l.startswith("RAW:")]
        want = [l.strip() for l in ref.raw_lines("golden")]
        assert len(lines) == len(want) >= 3
        for a, b in zip(lines, want):
            fa, fb = parity.raw_fields(a), parity.raw_fields(b)
            # field 2 (timestamp) differs by the wall-clock base the binary reads; all else equal
            assert fa[0] == fb[0] and fa[3:6] == fb[3:6] and fa[7:] == fb[7:]
            assert abs(fa[2] - fb[2]) <= 1 and abs(fa[6] - fb[6]) <= 1e-4


def _cli_lines_equal(lines, want):
    assert len(lines) == len(want)
    for a, b in zip(lines, want):
        fa, fb = parity.raw_fields(a), parity.raw_fields(b)
        assert fa[0] == fb[0] and fa[3:6] == fb[3:6] and fa[7:] == fb[7:]
        assert abs(fa[2] - fb[2]) <= 1 and abs(fa[6] - fb[6]) <= 1e-4


def test_cli_no_simd_selects_the_generic_order(tmp_path):
    """--no-simd (options.c:240, :351; main.c:567 simd_init(no_simd)): the binary follows simd_generic.c's operation order
    (option fir_order 0) and prints what the oracle prints in that order; without the flag, simd_avx2.c's.  At 10 MHz, where
    the two decimators are different kernels (fir_decimate_kernel_r / _f)."""
    import subprocess
    exe = os.path.join(os.path.dirname(irdm.LIB_PATH), "iridium-sniffer-hip")
    if not os.path.exists(exe):
        irdm.build(force=True)
    fs = 10_000_000
    iq = siggen.standard_scene(fs, int(0.62 * fs) // 32768 * 32768, 5, seed=77)[0]
    path = tmp_path / "scene.cf32"
    np.ascontiguousarray(iq).tofile(path)
    refs = {}
    try:
        for order in (0, 1):
            orc.set_fir_order(order)
            refs[order] = orc.run_stream(iq, fs)
    finally:
        orc.set_fir_order(1)
    outs = {}
    for order, flag in ((1, []), (0, ["--no-simd"])):
        out = subprocess.run([exe, "-f", str(path), "-r", str(fs), "--file-info", "golden", "--chunk", str(32768 * 64)] + flag,
                             capture_output=True, text=True, timeout=180)
        assert out.returncode == 0, out.stderr
        ref = refs[order]
        assert "tagged %d bursts total" % ref.n_tagged in out.stderr
        lines = [l for l in out.stdout.splitlines() if l.startswith("RAW:")]
        assert len(lines) >= 3
        _cli_lines_equal(lines, [l.strip() for l in ref.raw_lines("golden")])
        outs[order] = lines
    # (--no-gpu stays refused: this binary has no CPU path)
    out = subprocess.run([exe, "-f", str(path), "-r", str(fs), "--no-gpu"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 2


def test_cli_file_tail_that_does_not_divide_into_the_read_slices(tmp_path):
    """A regular file is read by --read-threads pread() slices per chunk.  A tail of T * 4096 * k + r bytes (r < T) used to
    be cut into T + 1 slices -- one more than there are threads: the reader waited for it forever."""
    import subprocess
    exe = os.path.join(os.path.dirname(irdm.LIB_PATH), "iridium-sniffer-hip")
    if not os.path.exists(exe):
        irdm.build(force=True)
    fs = 2_000_000
    chunk = 32768 * 16
    iq = siggen.standard_scene(fs, int(0.81 * fs), 3, seed=5)[0]
    data = siggen.to_ci16(iq)
    n = 3 * chunk + (6 * 4096 * 3 + 4) // 4                 # ci16: 4 bytes a sample; the tail is 6 * 4096 * 3 + 4 bytes
    data = np.ascontiguousarray(data.reshape(-1)[:2 * n])
    path = tmp_path / "tail.ci16"
    data.tofile(path)
    assert os.path.getsize(path) == 4 * n
    ref = orc.run_stream(data, fs, fmt=1)
    for threads in (6, 5, 3):
        out = subprocess.run([exe, "-f", str(path), "-r", str(fs), "--file-info", "golden", "--chunk", str(chunk),
                              "--read-threads", str(threads)], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert "tagged %d bursts total" % ref.n_tagged in out.stderr
        lines = [l for l in out.stdout.splitlines() if l.startswith("RAW:")]
        _cli_lines_equal(lines, [l.strip() for l in ref.raw_lines("golden")])


def test_cli_save_bursts_dumps_every_downmixed_frame(tmp_path):
    """--save-bursts (qpsk_demod.c:339-389): one .cf32/.meta pair per frame handed to the demodulator, named with the
    direction the demodulator settled on (UN when the unique word was rejected); payload == the frame samples."""
    import glob
    import subprocess
    import scenes
    exe = os.path.join(os.path.dirname(irdm.LIB_PATH), "iridium-sniffer-hip")
    if not os.path.exists(exe):
        irdm.build(force=True)
    fs, iq = scenes.junk()                       # accepted, rescued and rejected unique words
    ref = orc.run_stream(iq, fs)
    path = tmp_path / "junk.cf32"
    np.ascontiguousarray(iq).tofile(path)
    d = tmp_path / "dump"
    out = subprocess.run([exe, "-f", str(path), "-r", str(fs), "--save-bursts", str(d), "--chunk", str(32768 * 24)],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    metas = sorted(glob.glob(str(d / "*.meta")))
    frames = [f for f in ref.frames if f.drop_reason == 0]
    assert len(metas) == len(frames) == len(glob.glob(str(d / "*.cf32"))) >= 5
    ok_ids = {dm.id: dm.direction for dm in ref.demods}
    by_id = {}
    for m in metas:
        txt = dict(l.split(": ") for l in open(m).read().splitlines())
        by_id[int(txt["burst_id"])] = (m, txt)
    for f in frames:
        m, txt = by_id[f.id]
        want_dir = {1: "DL", 2: "UL"}[ok_ids[f.id]] if f.id in ok_ids else "UN"
        assert txt["direction"] == want_dir and m.endswith("_%d_%s.meta" % (f.id, want_dir))
        assert int(txt["num_samples"]) == f.num_samples
        got = np.fromfile(m[:-5] + ".cf32", np.float32)
        assert np.array_equal(got.view(np.uint32), np.ctypeslib.as_array(f.samples)[:2 * f.num_samples].view(np.uint32))
    assert any(f.id not in ok_ids for f in frames)            # the scene really has rejected frames


def test_pipeline_1mhz_fft_1024():
    """1 MHz -> 1024-point frames, /4 decimation: below the band scan's and the sparse scan's geometry (both need
    >= 2048 bins), so the dense scan and the runtime-M decimator serve it -- every feed must work and match the oracle
    (round-1 review: the sparse path returned -1 after priming at this size)."""
    fs = 1_000_000
    n = int(3.0 * fs) // 32768 * 32768
    rng = np.random.default_rng(31)
    first = 520 * 1024
    bursts = [dict(start=first + 3000 + 260_000 * i, freq_hz=siggen.channel_freq(int(rng.integers(-10, 11)) or 1),
                   payload=rng.integers(0, 4, 150).tolist()) for i in range(8)]
    iq, _ = siggen.make_stream(fs, n, bursts, seed=31)
    ref = orc.run_stream(iq, fs)
    assert len(ref.bursts) >= 8 and len(ref.demods) >= 6
    got = parity.run_gpu(iq, fs, chunks=[32768 * 30, n - 32768 * 30])
    parity.compare(got, ref)
    assert got["stats"]["band_chunks"] == 0 and got["stats"]["scan_fast_chunks"] == 0
