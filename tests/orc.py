"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, when present,
the reference build (oracle/_ref/libiridium_ref.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

MAX_FRAME_SAMPLES = 4440
MAX_BITS = 896


class BurstRec(C.Structure):
    _fields_ = [("id", C.c_uint64), ("start", C.c_uint64), ("stop", C.c_uint64),
                ("last_active", C.c_uint64), ("center_bin", C.c_int32),
                ("magnitude", C.c_float), ("noise", C.c_float), ("peak_rel", C.c_float),
                ("base_sum", C.c_float), ("num_samples", C.c_uint64),
                ("avail_end", C.c_uint64)]


class Frame(C.Structure):
    _fields_ = [("id", C.c_uint64), ("timestamp", C.c_uint64),
                ("center_frequency", C.c_double), ("sample_rate", C.c_float),
                ("samples_per_symbol", C.c_float), ("direction", C.c_int32),
                ("magnitude", C.c_float), ("noise", C.c_float), ("uw_start", C.c_float),
                ("num_samples", C.c_int32), ("dec_len", C.c_int32), ("start", C.c_int32),
                ("center_offset", C.c_float), ("uw_start_idx", C.c_int32),
                ("corr_re", C.c_float), ("corr_im", C.c_float), ("drop_reason", C.c_int32),
                ("samples", C.c_float * (2 * MAX_FRAME_SAMPLES))]


class Demod(C.Structure):
    _fields_ = [("id", C.c_uint64), ("timestamp", C.c_uint64),
                ("center_frequency", C.c_double), ("direction", C.c_int32),
                ("magnitude", C.c_float), ("noise", C.c_float), ("confidence", C.c_int32),
                ("level", C.c_float), ("n_symbols", C.c_int32),
                ("n_payload_symbols", C.c_int32), ("n_bits", C.c_int32), ("ok", C.c_int32),
                ("total_phase", C.c_float), ("bits", C.c_uint8 * MAX_BITS),
                ("llr", C.c_float * MAX_BITS)]


class StreamCfg(C.Structure):
    _fields_ = [("center_frequency", C.c_double), ("sample_rate", C.c_int),
                ("threshold_db", C.c_float), ("format", C.c_int), ("block", C.c_int),
                ("use_gardner", C.c_int), ("start_time_ns", C.c_uint64)]


class StreamOut(C.Structure):
    _fields_ = [("bursts", C.POINTER(BurstRec)), ("n_bursts", C.c_size_t), ("cap_bursts", C.c_size_t),
                ("frames", C.POINTER(Frame)), ("n_frames", C.c_size_t), ("cap_frames", C.c_size_t),
                ("demods", C.POINTER(Demod)), ("n_demods", C.c_size_t), ("cap_demods", C.c_size_t),
                ("n_tagged", C.c_uint64), ("n_samples", C.c_uint64)]


BURST_CB = C.CFUNCTYPE(None, C.POINTER(BurstRec), C.POINTER(C.c_float), C.c_void_p)

_lib = None
_ref = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "ref"])


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def set_fir_order(order):
    """0: stage B's decimating FIR as simd_generic.c (--no-simd), 1: as simd_avx2.c (default)"""
    lib().orc_set_fir_order(int(order))


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_fft.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int]
        L.orc_fir_ccf_dec_avx2.argtypes = [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int]
        L.orc_set_fir_order.argtypes = [C.c_int]
        # the decimating FIR of stage B in the order of the reference's AVX2 kernel (simd_avx2.c:62-108: the reference's
        # default on x86, and the product's default "fir_order" 1); tests of the scalar order call set_fir_order(0)
        L.orc_set_fir_order(1)
        L.orc_max_float.restype = C.c_float
        L.orc_detector_create.restype = C.c_void_p
        L.orc_detector_create.argtypes = [C.c_double, C.c_int, C.c_float, C.c_int]
        L.orc_detector_destroy.argtypes = [C.c_void_p]
        L.orc_detector_feed_cf32.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_size_t, BURST_CB, C.c_void_p]
        L.orc_detector_feed_i8.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, BURST_CB, C.c_void_p]
        L.orc_detector_fft_size.argtypes = [C.c_void_p]
        L.orc_detector_tagged.argtypes = [C.c_void_p]
        L.orc_detector_tagged.restype = C.c_uint64
        L.orc_detector_baseline_sum.argtypes = [C.c_void_p]
        L.orc_detector_baseline_sum.restype = C.POINTER(C.c_float)
        L.orc_detector_magnitude_frame.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_detector_set_mag_sink.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_size_t]
        L.orc_detector_frames_done.argtypes = [C.c_void_p]
        L.orc_detector_frames_done.restype = C.c_size_t
        L.orc_downmix_create.restype = C.c_void_p
        L.orc_downmix_destroy.argtypes = [C.c_void_p]
        L.orc_downmix_process.argtypes = [C.c_void_p, C.POINTER(BurstRec), C.POINTER(C.c_float),
                                          C.c_double, C.c_int, C.c_int, C.c_uint64, C.POINTER(Frame)]
        L.orc_downmix_taps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.orc_downmix_taps.restype = C.POINTER(C.c_float)
        L.orc_downmix_sync_fft.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.orc_downmix_sync_fft.restype = C.POINTER(C.c_float)
        L.orc_downmix_cfo_window.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.orc_downmix_cfo_window.restype = C.POINTER(C.c_float)
        L.orc_qpsk_demod.argtypes = [C.POINTER(Frame), C.c_int, C.POINTER(Demod)]
        L.orc_format_raw.argtypes = [C.POINTER(Demod), C.c_char_p, C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]
        L.orc_run_stream.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(StreamCfg), C.POINTER(StreamOut)]
        L.orc_rotator_rotate_n.argtypes = [C.POINTER(C.c_float)] * 2 + [C.POINTER(C.c_float)] * 2 + [C.c_int]
        _lib = L
    return _lib


def ref():
    """The reference's own sources compiled in place (None on the GPU box if absent)."""
    global _ref
    if _ref is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libiridium_ref.so")
        if not os.path.exists(path):
            if os.path.isdir("/root/reference"):
                build()
            else:
                return None
        R = C.CDLL(path)
        R.generic_max_float.restype = C.c_float
        R.avx2_max_float.restype = C.c_float
        for name in ("lpf_taps", "rrc_taps", "rc_taps", "box_taps"):
            getattr(R, name).restype = C.POINTER(C.c_float)
        R.lpf_taps.argtypes = [C.POINTER(C.c_int), C.c_float, C.c_float, C.c_float, C.c_float]
        R.rrc_taps.argtypes = [C.POINTER(C.c_int), C.c_float, C.c_float, C.c_float, C.c_float, C.c_int]
        R.rc_taps.argtypes = [C.POINTER(C.c_int), C.c_float, C.c_float, C.c_float, C.c_int]
        R.box_taps.argtypes = [C.POINTER(C.c_int), C.c_int]
        R.ref_qpsk_demod.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_float, C.c_int, C.c_double,
                                     C.c_uint64, C.c_uint64, C.c_float, C.c_float,
                                     C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float),
                                     C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                     C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_double)]
        _ref = R
    return _ref


class StreamResult:
    def __init__(self, bursts, frames, demods, n_tagged, n_samples):
        self.bursts, self.frames, self.demods = bursts, frames, demods
        self.n_tagged, self.n_samples = n_tagged, n_samples

    def raw_lines(self, file_info="golden"):
        L = lib()
        t0 = C.c_uint64(0)
        buf = C.create_string_buffer(4096)
        out = []
        for d in self.demods:
            n = L.orc_format_raw(C.byref(d), file_info.encode(), C.byref(t0), buf, 4096)
            assert n > 0
            out.append(buf.value.decode())
        return out


def run_stream(iq, sample_rate, fmt=2, center_frequency=1622000000.0, threshold_db=0.0,
               block=32768, use_gardner=1, start_time_ns=1700000000 * 10**9,
               cap_bursts=4096):
    """iq: complex64 array (fmt 2), int16 interleaved (fmt 1) or int8 interleaved (fmt 0)."""
    L = lib()
    iq = np.ascontiguousarray(iq)
    n = len(iq) if fmt == 2 else len(iq) // 2
    cfg = StreamCfg(center_frequency, int(sample_rate), threshold_db, fmt, block, use_gardner,
                    start_time_ns)
    bursts = (BurstRec * cap_bursts)()
    frames = (Frame * cap_bursts)()
    demods = (Demod * cap_bursts)()
    out = StreamOut(bursts, 0, cap_bursts, frames, 0, cap_bursts, demods, 0, cap_bursts, 0, 0)
    rc = L.orc_run_stream(iq.ctypes.data_as(C.c_void_p), n, C.byref(cfg), C.byref(out))
    assert rc == 0, "oracle stream capacity exceeded"
    return StreamResult([bursts[i] for i in range(out.n_bursts)],
                        [frames[i] for i in range(out.n_frames)],
                        [demods[i] for i in range(out.n_demods)],
                        out.n_tagged, out.n_samples)
