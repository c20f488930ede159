"""The C-ABI library loads (no GPU needed) and exports every symbol that
include/irdm_hip.h declares; struct layouts in the ctypes mirror match the header."""
import ctypes as C
import os
import re

import irdm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "irdm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b((?:gpu_burst_fft|irdm)_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    irdm.build()
    L = C.CDLL(irdm.LIB_PATH)
    names = declared_functions()
    assert "gpu_burst_fft_create" in names and "irdm_feed_device" in names and len(names) >= 18
    for n in names:
        assert hasattr(L, n), n


def test_reference_stage_level_names_present():
    # include/irdm_compat.h: the reference's stage API (burst_detect.h:67-94, burst_downmix.h:64-73, qpsk_demod.h:42)
    src = open(os.path.join(ROOT, "include", "irdm_compat.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b((?:burst_detector|burst_downmix|qpsk_demod|irdm_compat)[a-z0-9_]*)\s*\(", src)))
    assert "burst_detector_feed_cf32" in names and "burst_downmix_process" in names and "qpsk_demod" in names
    L = C.CDLL(irdm.LIB_PATH)
    for n in names:
        assert hasattr(L, n), n


def test_reference_plug_point_names_present():
    # opencl/burst_fft.h:35-47 -- the three symbols burst_detect.c links against
    L = irdm.lib()
    for n in ("gpu_burst_fft_create", "gpu_burst_fft_process", "gpu_burst_fft_destroy"):
        assert hasattr(L, n)
    assert irdm.lib().irdm_version().startswith(b"irdm_hip")


def test_struct_sizes_match_header():
    # sizeof() of the C structs in include/irdm_hip.h, x86-64 SysV (checked with gcc)
    assert C.sizeof(irdm.Burst) == 72
    assert C.sizeof(irdm.Demod) == 4544
    assert C.sizeof(irdm.FrameInfo) == 88
    assert C.sizeof(irdm.Config) == 64


def test_format_raw_matches_oracle_format():
    """RAW line printer is host C (frame_output.c:160-199): byte-identical to the oracle's."""
    import orc
    d = irdm.Demod()
    d.id = 120
    d.timestamp = 1700000000 * 10**9 + 533072000
    d.center_frequency = 1622209567.6
    d.magnitude, d.noise, d.confidence, d.level = 23.804, -114.127, 97, 0.016812
    d.n_symbols, d.n_payload_symbols, d.n_bits, d.ok = 191, 179, 382, 1
    for i in range(382):
        d.bits[i] = (i * 7 // 3) & 1
    o = orc.Demod.from_buffer_copy(bytes(d))
    for fi in ("golden", ""):
        ours = irdm.format_raw([d], fi)[0]
        t0 = C.c_uint64(0)
        buf = C.create_string_buffer(4096)
        orc.lib().orc_format_raw(C.byref(o), fi.encode(), C.byref(t0), buf, 4096)
        assert ours == buf.value.decode()
    assert ours.startswith("RAW: i-1700000000-t1 0000533.0720 1622209568 N:23.80-114.13 I:00000000120  97% 0.01681 179 ")


def test_format_raw_batch_is_the_concatenation_of_lines():
    """irdm_format_raw_batch (one write per poll batch) produces exactly the per-line bytes, t0 carried across."""
    ds = []
    for k in range(5):
        d = irdm.Demod()
        d.id = 10 * k
        d.timestamp = 1700000000 * 10**9 + 533072000 + 90_000_000 * k
        d.center_frequency = 1622209567.6 + 41666.7 * k
        d.magnitude, d.noise, d.confidence, d.level = 20.0 + k, -114.1, 90 + k, 0.016 + 0.001 * k
        d.n_symbols, d.n_payload_symbols, d.n_bits, d.ok = 191, 179, 382 - 2 * k, 1
        for i in range(d.n_bits):
            d.bits[i] = (i * (k + 3) // 5) & 1
        ds.append(d)
    for fi in ("golden", ""):
        assert irdm.format_raw_batch(ds, fi) == "".join(irdm.format_raw(ds, fi))
    assert irdm.format_raw_batch([], "x") == ""


def test_packed_records_print_the_same_raw_lines():
    """irdm_format_raw_packed[_batch]: the RAW line of a compact record (hard bits 8 per byte, MSB first) is byte for byte the
    line of the full record (frame_output.c:160-199)."""
    import ctypes as C
    import numpy as np
    L = irdm.lib()
    L.irdm_format_raw_packed_batch.restype = C.c_longlong
    L.irdm_format_raw_packed_batch.argtypes = [C.POINTER(irdm.DemodPacked), C.c_int, C.c_char_p, C.POINTER(C.c_uint64),
                                               C.c_char_p, C.c_size_t]
    rng = np.random.default_rng(4)
    full, packed = [], (irdm.DemodPacked * 6)()
    for k in range(6):
        d = irdm.Demod()
        d.id = 10 * k
        d.timestamp = 1700000000 * 10**9 + 533072000 + 90_000_000 * k
        d.center_frequency = 1622209567.6 + 41666.7 * k
        d.magnitude, d.noise, d.confidence, d.level = 20.0 + k, -114.1, 90 + k, 0.016 + 0.001 * k
        d.n_symbols, d.n_payload_symbols, d.n_bits, d.ok = 191 - k, 179 - k, 2 * (191 - k), 1
        bits = rng.integers(0, 2, d.n_bits).astype(np.uint8)
        for i, bv in enumerate(bits):
            d.bits[i] = int(bv)
        full.append(d)
        q = packed[k]
        for fld in ("id", "timestamp", "center_frequency", "direction", "magnitude", "noise", "confidence", "level",
                    "n_symbols", "n_payload_symbols", "n_bits", "ok", "total_phase"):
            setattr(q, fld, getattr(d, fld))
        pb = np.packbits(np.concatenate([bits, np.zeros(irdm.MAX_BITS - len(bits), np.uint8)]))
        for i, bv in enumerate(pb):
            q.bits[i] = int(bv)
    for fi in ("golden", ""):
        t0 = C.c_uint64(0)
        buf = C.create_string_buffer(6 * 2048)
        n = L.irdm_format_raw_packed_batch(packed, 6, fi.encode(), C.byref(t0), buf, len(buf))
        assert n > 0 and buf.raw[:n].decode() == irdm.format_raw_batch(full, fi)


def test_save_burst_writes_the_reference_file_pair(tmp_path):
    """irdm_save_burst == save_burst_iq (qpsk_demod.c:339-389): file names, .meta text, raw cf32 payload."""
    import numpy as np
    f = irdm.FrameInfo()
    f.id, f.timestamp, f.center_frequency = 130, 1700000000533072000, 1626270833.0
    f.sample_rate, f.samples_per_symbol = 250000.0, 10.0
    f.magnitude, f.noise, f.uw_start, f.num_samples = 23.456, -114.127, 3.25, 1910
    f.drop_reason, f.demod_ok, f.demod_direction = 0, 1, 2
    x = (np.arange(2 * 1910, dtype=np.float32) * 0.001).astype(np.float32)
    d = str(tmp_path / "bursts")
    assert irdm.save_burst(f, x, d) == 0
    base = "%s/%020d_%011.0f_%d_UL" % (d, f.timestamp, f.center_frequency, f.id)
    assert np.array_equal(np.fromfile(base + ".cf32", np.float32), x)
    assert open(base + ".meta").read() == (
        "burst_id: 130\ntimestamp_ns: 1700000000533072000\ncenter_freq_hz: 1626270833\nsample_rate_hz: 250000\n"
        "samples_per_symbol: 10.00\ndirection: UL\nmagnitude_db: 23.46\nnoise_dbfs_hz: -114.13\nnum_samples: 1910\n"
        "uw_start_offset: 3.25\n")
    f.demod_ok, f.demod_direction = 0, 0
    assert irdm.save_burst(f, x, d) == 0 and os.path.exists(base[:-2] + "UN.meta")
    f.drop_reason = 3
    assert irdm.save_burst(f, x, d) == -1


def test_cli_refuses_more_gpus_than_the_host_has(tmp_path):
    """iridium-sniffer-hip --gpus N on a host with fewer devices: exit code 2 and a message that says how many there are (here:
    none without a GPU, one on the GPU box) -- before any context or communicator is built."""
    import subprocess
    irdm.build()
    exe = os.path.join(os.path.dirname(irdm.LIB_PATH), "iridium-sniffer-hip")
    f = tmp_path / "x.cf32"
    f.write_bytes(b"\0" * 8 * 32768)
    n_dev = irdm.lib().irdm_device_count()
    out = subprocess.run([exe, "-f", str(f), "-r", "2000000", "--gpus", str(n_dev + 1)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 2, (out.returncode, out.stderr)
    assert "--gpus %d: this host has %d GPU" % (n_dev + 1, n_dev) in out.stderr, out.stderr
    assert out.stdout == ""
