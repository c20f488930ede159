"""Shared comparison helpers: HIP pipeline (through the C-ABI) vs the CPU oracle."""
import numpy as np

import irdm
import orc

SOFT_TOL = 1e-4          # north_star tolerance on soft outputs


def bits_of(x):
    return np.float32(x).view(np.uint32)


def run_gpu(iq, fs, fmt=irdm.FMT_CF32, chunks=None, scan_mode=0, depth=0, feed="host", options=None, packed=False, **kw):
    """Feed `iq` through the HIP pipeline in the given chunk sizes (samples).

    feed: "host" irdm_feed_host per chunk; "ingest" every chunk written in place (irdm_ingest_ptr) and fed from there;
    "lookahead" device buffers, irdm_feed_begin(k+1) before irdm_feed_end(k); "ingest_lookahead" both.
    packed: option packed_records -- the compact frame records only (gpu["packed"]; no frame samples, no LLRs)."""
    n = len(iq) if fmt == irdm.FMT_CF32 else len(iq) // 2
    max_chunk = max(chunks) if chunks else n
    p = irdm.Pipeline(fs, fmt=fmt, max_chunk_samples=max_chunk, max_bursts_per_chunk=1024,
                      pipeline_depth=depth, **kw)
    if packed:
        p.set_option("packed_records", 1)
    else:
        p.set_option("keep_frame_samples", 1)
    p.set_option("scan_mode", scan_mode)
    for k, v in (options or {}).items():
        p.set_option(k, v)
    per = 1 if fmt == irdm.FMT_CF32 else 2
    off = 0
    sizes = list(chunks or [n])
    L = irdm.lib()
    if feed == "host":
        for c in sizes:
            p.feed_host(iq[off * per:(off + c) * per])
            off += c
    else:
        import ctypes as C
        assert depth >= 1
        ingest = feed.startswith("ingest")
        look = feed.endswith("lookahead")
        held = []                # device buffers of chunks begun (not "ingest"): released after their feed_end

        def begin(c, off):
            part = np.ascontiguousarray(iq[off * per:(off + c) * per])
            if ingest:
                ptr = p.ingest_ptr(c)
                assert ptr, "irdm_ingest_ptr refused a chunk of %d samples" % c
                assert L.irdm_device_upload(C.c_void_p(ptr), part.ctypes.data_as(C.c_void_p), part.nbytes) == 0
                held.append(None)
            else:
                ptr = irdm.device_buffer(part)
                held.append(ptr)
            p.feed_begin(ptr, c)

        def end():
            p.feed_end()
            ptr = held.pop(0)
            if ptr:
                irdm.device_free(ptr)

        pending = 0
        ahead = 1 if look else 0      # chunks begun ahead of the one that is ended
        for c in sizes:
            begin(c, off)
            off += c
            pending += 1
            if pending > ahead:
                end()
                pending -= 1
        while pending:
            end()
            pending -= 1
    assert off == n
    if depth:
        p.flush()
    bursts = p.poll_bursts()
    infos, samples = p.poll_frames()
    demods = p.poll_demods()
    packed_recs = p.poll_demods_packed() if packed else []
    res = dict(bursts=bursts, infos=infos, samples=samples, demods=demods, packed=packed_recs, tagged=p.tagged,
               n_samples=p.sample_count, timings=p.timings(),
               stats={k: p.stat(k) for k in ("scan_fast_chunks", "scan_fallbacks", "scan_dense_frames", "band_chunks",
                                             "band_rounds", "band_retries", "band_aborts", "band_last_flags", "k1_lists", "band_extra",
                                             "scratch_outputs", "scratch_grows", "scratch_peak", "rot_rows", "rot_runs", "rot_ckpts", "rot_blocks", "rot_blocks_cap", "rot_grows",
                                             "spec_passes", "spec_scans", "scan_chained", "scan_chain_undone", "sum_restarts")})
    p.close()
    return res


def run_group(iq, fs, n_gpus, chunk, fmt=irdm.FMT_CF32, depth=1, feed="host", staged_ahead=False, options=None, devices=None):
    """The same stream through a group of n_gpus members (irdm_group_*, csrc/group.cpp) in super-steps of n_gpus chunks of
    `chunk` samples: chunk k on member k mod n_gpus, the records merged by the library.  feed "host": irdm_group_feed_host;
    "device": the super-step uploaded to member 0's device and scattered from there.  staged_ahead: super-step s + 1 is
    staged before super-step s is fed."""
    import ctypes as C
    n = len(iq) if fmt == irdm.FMT_CF32 else len(iq) // 2
    per = 1 if fmt == irdm.FMT_CF32 else 2
    g = irdm.Group(fs, n_gpus, devices=devices, fmt=fmt, max_chunk_samples=chunk, max_bursts_per_chunk=1024, pipeline_depth=depth)
    g.set_option("keep_frame_samples", 1)
    for k, v in (options or {}).items():
        g.set_option(k, v)
    step = n_gpus * chunk
    parts = [np.ascontiguousarray(iq[o * per:min(o + step, n) * per]) for o in range(0, n, step)]
    held = []

    def handle(part):
        if feed == "host":
            return part
        ptr = irdm.device_buffer(part, device=(devices[0] if devices else 0))
        held.append(ptr)
        return (ptr, len(part) // per)

    def stage(h):
        if feed == "host":
            g.stage_host(h)
        else:
            g.stage_device(*h)

    def run(h):
        return g.feed_host(h) if feed == "host" else g.feed_device(*h)

    handles = [handle(p) for p in parts]
    fed = 0
    if staged_ahead and handles:
        stage(handles[0])
    for i, h in enumerate(handles):
        if staged_ahead and i + 1 < len(handles):
            stage(handles[i + 1])
        fed += run(h)
    g.flush()
    bursts = g.poll_bursts()
    infos, samples = g.poll_frames()
    demods = g.poll_demods()
    res = dict(bursts=bursts, infos=infos, samples=samples, demods=demods, tagged=g.tagged, chunks_fed=fed,
               stats={k: g.stat(k) for k in ("hops", "hop_bytes", "scatter_bytes", "overlap_bytes", "late_history", "chunks",
                                             "overlap_samples", "scan_fallbacks", "band_aborts", "band_chunks")})
    g.close()
    for ptr in held:
        irdm.device_free(ptr)
    return res


def compare(gpu, ref, exact_frames=True):
    """ref: orc.StreamResult.  Returns a small summary dict; asserts on any mismatch."""
    assert gpu["tagged"] == ref.n_tagged, (gpu["tagged"], ref.n_tagged)
    assert len(gpu["bursts"]) == len(ref.bursts)
    for g, r in zip(gpu["bursts"], ref.bursts):
        for f in ("id", "start", "stop", "last_active", "center_bin", "num_samples", "avail_end"):
            assert getattr(g, f) == getattr(r, f), (f, g.id, getattr(g, f), getattr(r, f))
        for f in ("magnitude", "noise", "peak_rel", "base_sum"):        # bit-identical floats
            assert bits_of(getattr(g, f)) == bits_of(getattr(r, f)), (f, g.id)
    assert len(gpu["infos"]) == len(ref.frames)
    n_frames = 0
    for g, s, r in zip(gpu["infos"], gpu["samples"], ref.frames):
        assert g.id == r.id and g.drop_reason == r.drop_reason, (g.id, g.drop_reason, r.drop_reason)
        assert g.dec_len == r.dec_len
        if r.drop_reason in (0, 3, 4, 5):
            assert g.start == r.start, (g.id, g.start, r.start)
        if r.drop_reason in (0, 4, 5):
            assert bits_of(g.center_offset) == bits_of(r.center_offset), (g.id, g.center_offset, r.center_offset)
            assert g.uw_start_idx == r.uw_start_idx and g.direction == r.direction
            assert bits_of(g.corr_re) == bits_of(r.corr_re) and bits_of(g.corr_im) == bits_of(r.corr_im)
        if r.drop_reason == 0:
            n_frames += 1
            assert g.timestamp == r.timestamp and g.center_frequency == r.center_frequency
            assert g.num_samples == r.num_samples
            assert bits_of(g.uw_start) == bits_of(r.uw_start)
            rs = np.ctypeslib.as_array(r.samples)[:2 * r.num_samples].view(np.complex64)
            if exact_frames:
                assert np.array_equal(s.view(np.uint32), rs.view(np.uint32)), \
                    (g.id, float(np.abs(s - rs).max()))
            else:
                assert np.allclose(s, rs, atol=1e-6)
    assert len(gpu["demods"]) == len(ref.demods), (len(gpu["demods"]), len(ref.demods))
    max_soft = 0.0
    for g, r in zip(gpu["demods"], ref.demods):
        assert g.id == r.id and g.timestamp == r.timestamp
        assert (g.direction, g.n_symbols, g.n_payload_symbols, g.n_bits) == \
               (r.direction, r.n_symbols, r.n_payload_symbols, r.n_bits), g.id
        assert bytes(g.bits[:g.n_bits]) == bytes(r.bits[:r.n_bits]), g.id     # hard bits identical
        assert g.confidence == r.confidence, (g.id, g.confidence, r.confidence)
        assert abs(g.level - r.level) <= SOFT_TOL
        gl = np.array(g.llr[:g.n_bits], np.float32)
        rl = np.array(r.llr[:r.n_bits], np.float32)
        assert np.max(np.abs(gl - rl)) <= SOFT_TOL
        max_soft = max(max_soft, float(np.max(np.abs(gl - rl))), abs(g.level - r.level))
        assert abs(g.center_frequency - r.center_frequency) <= 0.05      # Hz; printed rounded to 1 Hz
        assert bits_of(g.magnitude) == bits_of(r.magnitude) and bits_of(g.noise) == bits_of(r.noise)
    return dict(bursts=len(ref.bursts), frames=n_frames, demods=len(ref.demods), max_soft=max_soft)


def raw_fields(line):
    """RAW line -> (file_info, ts, freq, N-field, id, conf, level, payload, bits)"""
    p = line.split()
    return p[1], float(p[2]), int(p[3]), p[4], p[5], p[6], float(p[7]), int(p[8]), p[9]


def compare_records(bursts, demods, ref):
    """Burst and demodulated-frame records only (no per-frame samples): what a sharded run gathers.  ids, indices, centre
    bins, dB fields, hard bits and confidence exact; level / LLR within SOFT_TOL."""
    assert len(bursts) == len(ref.bursts), (len(bursts), len(ref.bursts))
    for g, r in zip(bursts, ref.bursts):
        for f in ("id", "start", "stop", "last_active", "center_bin", "num_samples", "avail_end"):
            assert getattr(g, f) == getattr(r, f), (f, g.id, getattr(g, f), getattr(r, f))
        for f in ("magnitude", "noise", "peak_rel", "base_sum"):
            assert bits_of(getattr(g, f)) == bits_of(getattr(r, f)), (f, g.id)
    assert len(demods) == len(ref.demods), (len(demods), len(ref.demods))
    max_soft = 0.0
    for g, r in zip(demods, ref.demods):
        assert g.id == r.id and g.timestamp == r.timestamp
        assert (g.direction, g.n_symbols, g.n_payload_symbols, g.n_bits) == \
               (r.direction, r.n_symbols, r.n_payload_symbols, r.n_bits), g.id
        assert bytes(g.bits[:g.n_bits]) == bytes(r.bits[:r.n_bits]), g.id
        assert g.confidence == r.confidence
        gl = np.array(g.llr[:g.n_bits], np.float32)
        rl = np.array(r.llr[:r.n_bits], np.float32)
        max_soft = max(max_soft, abs(g.level - r.level), float(np.max(np.abs(gl - rl))) if g.n_bits else 0.0)
        assert bits_of(g.magnitude) == bits_of(r.magnitude) and bits_of(g.noise) == bits_of(r.noise)
    assert max_soft <= SOFT_TOL
    return dict(bursts=len(bursts), demods=len(demods), max_soft=max_soft)


def compare_packed(gpu, ref):
    """A packed_records run (run_gpu(..., packed=True)) against the oracle: burst records as in compare(); the compact frame
    records -- everything frame_output_print reads -- ids, timestamps, direction, symbol counts, hard bits (8 per byte, MSB
    first, zero behind n_bits), confidence and the dB fields exact; level within SOFT_TOL, the refined frequency within
    0.05 Hz."""
    assert gpu["tagged"] == ref.n_tagged, (gpu["tagged"], ref.n_tagged)
    assert len(gpu["bursts"]) == len(ref.bursts)
    for g, r in zip(gpu["bursts"], ref.bursts):
        for f in ("id", "start", "stop", "last_active", "center_bin", "num_samples", "avail_end"):
            assert getattr(g, f) == getattr(r, f), (f, g.id, getattr(g, f), getattr(r, f))
        for f in ("magnitude", "noise", "peak_rel", "base_sum"):
            assert bits_of(getattr(g, f)) == bits_of(getattr(r, f)), (f, g.id)
    assert gpu["demods"] == [] and len(gpu["infos"]) == 0                 # the packed mode queues nothing else
    assert len(gpu["packed"]) == len(ref.demods), (len(gpu["packed"]), len(ref.demods))
    max_soft = 0.0
    for q, r in zip(gpu["packed"], ref.demods):
        assert q.id == r.id and q.timestamp == r.timestamp, (q.id, r.id)
        assert (q.direction, q.n_symbols, q.n_payload_symbols, q.n_bits, q.ok) == \
               (r.direction, r.n_symbols, r.n_payload_symbols, r.n_bits, 1), q.id
        allbits = np.unpackbits(np.frombuffer(bytes(q.bits), np.uint8))
        assert bytes(allbits[:q.n_bits]) == bytes(r.bits[:r.n_bits]), q.id
        assert not allbits[q.n_bits:].any(), q.id
        assert q.confidence == r.confidence, (q.id, q.confidence, r.confidence)
        assert abs(q.level - r.level) <= SOFT_TOL
        max_soft = max(max_soft, abs(q.level - r.level))
        assert abs(q.center_frequency - r.center_frequency) <= 0.05
        assert bits_of(q.magnitude) == bits_of(r.magnitude) and bits_of(q.noise) == bits_of(r.noise)
    return dict(bursts=len(ref.bursts), demods=len(ref.demods), max_soft=max_soft)
