"""ctypes view of the C-ABI in include/irdm_hip.h (libirdm_hip.so, gfx950).

Host-side mirror used by tests, bench.py and __graft_entry__: the same entry
points a C host binds.  There is no CPU fallback here: if the
HIP library is missing or no GPU is present, creation fails loudly.
"""
import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# IRDM_LIB: another build of the same library (A/B timing of kernel variants on one GPU box)
LIB_PATH = os.environ.get("IRDM_LIB") or os.path.join(PKG_DIR, "libirdm_hip.so")

FMT_CI8, FMT_CI16, FMT_CF32 = 0, 1, 2
MAX_FRAME_SAMPLES = 4440
MAX_BITS = 896


class Config(C.Structure):
    _fields_ = [("center_frequency", C.c_double), ("sample_rate", C.c_int),
                ("threshold_db", C.c_float), ("format", C.c_int), ("feed_block", C.c_int),
                ("use_gardner", C.c_int), ("start_time_ns", C.c_uint64), ("device", C.c_int),
                ("max_chunk_samples", C.c_size_t), ("max_bursts_per_chunk", C.c_int),
                ("pipeline_depth", C.c_int)]


class Burst(C.Structure):
    _fields_ = [("id", C.c_uint64), ("start", C.c_uint64), ("stop", C.c_uint64),
                ("last_active", C.c_uint64), ("center_bin", C.c_int32),
                ("magnitude", C.c_float), ("noise", C.c_float), ("peak_rel", C.c_float),
                ("base_sum", C.c_float), ("num_samples", C.c_uint64),
                ("avail_end", C.c_uint64)]


class FrameInfo(C.Structure):
    _fields_ = [("id", C.c_uint64), ("timestamp", C.c_uint64),
                ("center_frequency", C.c_double), ("sample_rate", C.c_float),
                ("samples_per_symbol", C.c_float), ("direction", C.c_int32),
                ("magnitude", C.c_float), ("noise", C.c_float), ("uw_start", C.c_float),
                ("num_samples", C.c_int32), ("dec_len", C.c_int32), ("start", C.c_int32),
                ("center_offset", C.c_float), ("uw_start_idx", C.c_int32),
                ("corr_re", C.c_float), ("corr_im", C.c_float), ("drop_reason", C.c_int32),
                ("demod_ok", C.c_int32), ("demod_direction", C.c_int32)]


class Demod(C.Structure):
    _fields_ = [("id", C.c_uint64), ("timestamp", C.c_uint64),
                ("center_frequency", C.c_double), ("direction", C.c_int32),
                ("magnitude", C.c_float), ("noise", C.c_float), ("confidence", C.c_int32),
                ("level", C.c_float), ("n_symbols", C.c_int32),
                ("n_payload_symbols", C.c_int32), ("n_bits", C.c_int32), ("ok", C.c_int32),
                ("total_phase", C.c_float), ("bits", C.c_uint8 * MAX_BITS),
                ("llr", C.c_float * MAX_BITS)]


class DemodPacked(C.Structure):
    """irdm_demod_packed_t: the frame without LLRs, hard bits 8 per byte (MSB first)"""
    _fields_ = [("id", C.c_uint64), ("timestamp", C.c_uint64),
                ("center_frequency", C.c_double), ("direction", C.c_int32),
                ("magnitude", C.c_float), ("noise", C.c_float), ("confidence", C.c_int32),
                ("level", C.c_float), ("n_symbols", C.c_int32),
                ("n_payload_symbols", C.c_int32), ("n_bits", C.c_int32), ("ok", C.c_int32),
                ("total_phase", C.c_float), ("bits", C.c_uint8 * (MAX_BITS // 8))]


class Decoded(C.Structure):
    _fields_ = [("type", C.c_int32), ("sat_id", C.c_int32), ("beam_id", C.c_int32), ("pos_xyz", C.c_int32 * 3),
                ("alt", C.c_int32), ("n_pages", C.c_int32), ("lat", C.c_double), ("lon", C.c_double),
                ("page_tmsi", C.c_uint32 * 12), ("page_msc", C.c_int32 * 12), ("timeslot", C.c_int32),
                ("sv_blocking", C.c_int32), ("bc_type", C.c_int32), ("iri_time", C.c_uint32),
                ("bch_len", C.c_int32), ("pad", C.c_int32), ("id", C.c_uint64), ("timestamp", C.c_uint64),
                ("frequency", C.c_double)]


class Ida(C.Structure):
    _fields_ = [("ok", C.c_int32), ("ft", C.c_int32), ("lcw_ft", C.c_int32), ("lcw_code", C.c_int32),
                ("ec_lcw", C.c_int32), ("lcw3_val", C.c_uint32), ("da_ctr", C.c_int32), ("da_len", C.c_int32),
                ("cont", C.c_int32), ("crc_ok", C.c_int32), ("stored_crc", C.c_uint32), ("computed_crc", C.c_uint32),
                ("fixederrs", C.c_int32), ("payload_len", C.c_int32), ("bch_len", C.c_int32), ("direction", C.c_int32),
                ("payload", C.c_uint8 * 32), ("bch_stream", C.c_uint8 * 256), ("lcw_header", C.c_char * 128),
                ("id", C.c_uint64), ("timestamp", C.c_uint64), ("frequency", C.c_double), ("magnitude", C.c_float),
                ("noise", C.c_float), ("level", C.c_float), ("confidence", C.c_int32), ("n_symbols", C.c_int32),
                ("pad", C.c_int32)]


_lib = None


def build(force=False):
    """hipcc --offload-arch=gfx950 ... -> iridium-sniffer_amd/libirdm_hip.so (in-tree)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-s", "-C", PKG_DIR, "-j8"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libirdm_hip.so is not built (run __graft_entry__.build()); "
                               "there is no CPU fallback for the product path")
        L = C.CDLL(LIB_PATH)
        L.gpu_burst_fft_create.restype = C.c_void_p
        L.gpu_burst_fft_create.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_float)]
        L.gpu_burst_fft_destroy.argtypes = [C.c_void_p]
        L.gpu_burst_fft_process.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int]
        L.gpu_burst_fft_process_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.irdm_create.restype = C.c_void_p
        L.irdm_create.argtypes = [C.POINTER(Config)]
        L.irdm_destroy.argtypes = [C.c_void_p]
        L.irdm_feed_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.irdm_feed_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.irdm_flush.argtypes = [C.c_void_p]
        if hasattr(L, "irdm_advance"):
            L.irdm_advance.argtypes = [C.c_void_p]
        L.irdm_host_alloc.argtypes = [C.c_size_t]
        L.irdm_host_alloc.restype = C.c_void_p
        L.irdm_host_free.argtypes = [C.c_void_p]
        L.irdm_host_free.restype = None
        L.irdm_feed_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.irdm_feed_end.argtypes = [C.c_void_p]
        L.irdm_ingest_ptr.argtypes = [C.c_void_p, C.c_size_t]
        L.irdm_ingest_ptr.restype = C.c_void_p
        L.irdm_ring_ptr.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.irdm_ring_ptr.restype = C.c_void_p
        L.irdm_export_state_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.irdm_export_state_device.restype = C.c_longlong
        L.irdm_import_state_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.irdm_state_head_bytes.argtypes = [C.c_void_p]
        L.irdm_state_head_bytes.restype = C.c_size_t
        L.irdm_import_state_head_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.irdm_expect_history.argtypes = [C.c_void_p, C.c_void_p]
        L.irdm_import_state_history_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.irdm_seed_history_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64]
        L.irdm_device_alloc.argtypes = [C.c_int, C.c_size_t]
        L.irdm_device_alloc.restype = C.c_void_p
        L.irdm_device_free.argtypes = [C.c_void_p]
        L.irdm_device_free.restype = None
        L.irdm_device_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.irdm_device_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.irdm_poll_bursts.argtypes = [C.c_void_p, C.POINTER(Burst), C.c_int]
        L.irdm_poll_frames.argtypes = [C.c_void_p, C.POINTER(FrameInfo), C.POINTER(C.c_float), C.c_int]
        L.irdm_poll_demods.argtypes = [C.c_void_p, C.POINTER(Demod), C.c_int]
        L.irdm_poll_demods_packed.argtypes = [C.c_void_p, C.POINTER(DemodPacked), C.c_int]
        L.irdm_poll_decoded.argtypes = [C.c_void_p, C.POINTER(Decoded), C.c_int]
        L.irdm_poll_ida.argtypes = [C.c_void_p, C.POINTER(Ida), C.c_int]
        L.irdm_ida_decode_batch.argtypes = [C.c_void_p, C.POINTER(Demod), C.c_int, C.c_int, C.POINTER(Ida)]
        L.irdm_frame_decode_batch.argtypes = [C.c_void_p, C.POINTER(Demod), C.c_int, C.c_int, C.POINTER(Decoded)]
        L.irdm_tagged_bursts.argtypes = [C.c_void_p]
        L.irdm_tagged_bursts.restype = C.c_uint64
        L.irdm_sample_count.argtypes = [C.c_void_p]
        L.irdm_sample_count.restype = C.c_uint64
        L.irdm_fft_size.argtypes = [C.c_void_p]
        L.irdm_last_magnitudes.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_size_t]
        L.irdm_baseline_sum.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.irdm_burst_samples.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_size_t]
        L.irdm_downmix_burst.argtypes = [C.c_void_p, C.POINTER(Burst), C.POINTER(C.c_float), C.c_size_t,
                                         C.POINTER(FrameInfo), C.POINTER(C.c_float)]
        L.irdm_qpsk_demod_batch.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int),
                                            C.POINTER(C.c_int), C.c_int, C.POINTER(Demod)]
        L.irdm_state_bytes.argtypes = [C.c_void_p]
        L.irdm_state_bytes.restype = C.c_size_t
        L.irdm_export_state.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.irdm_export_state.restype = C.c_longlong
        L.irdm_import_state.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.irdm_seed_history.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64]
        L.irdm_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.irdm_get_stat.argtypes = [C.c_void_p, C.c_char_p]
        L.irdm_get_stat.restype = C.c_int64
        L.irdm_last_timings.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
        L.irdm_kernel_clock.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.c_int]
        L.irdm_format_raw.argtypes = [C.POINTER(Demod), C.c_char_p, C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]
        L.irdm_version.restype = C.c_char_p
        # a group: one stream across several GPUs of this process (csrc/group.cpp)
        L.irdm_group_create.restype = C.c_void_p
        L.irdm_group_create.argtypes = [C.POINTER(Config), C.c_int, C.POINTER(C.c_int)]
        L.irdm_group_destroy.argtypes = [C.c_void_p]
        L.irdm_group_destroy.restype = None
        L.irdm_group_size.argtypes = [C.c_void_p]
        L.irdm_group_member.argtypes = [C.c_void_p, C.c_int]
        L.irdm_group_member.restype = C.c_void_p
        L.irdm_group_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.irdm_group_get_stat.argtypes = [C.c_void_p, C.c_char_p]
        L.irdm_group_get_stat.restype = C.c_int64
        for name in ("irdm_group_stage_host", "irdm_group_stage_device", "irdm_group_feed_host", "irdm_group_feed_device"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.irdm_group_flush.argtypes = [C.c_void_p]
        L.irdm_group_poll_bursts.argtypes = [C.c_void_p, C.POINTER(Burst), C.c_int]
        L.irdm_group_poll_frames.argtypes = [C.c_void_p, C.POINTER(FrameInfo), C.POINTER(C.c_float), C.c_int]
        L.irdm_group_poll_demods.argtypes = [C.c_void_p, C.POINTER(Demod), C.c_int]
        L.irdm_group_poll_demods_packed.argtypes = [C.c_void_p, C.POINTER(DemodPacked), C.c_int]
        L.irdm_group_poll_decoded.argtypes = [C.c_void_p, C.POINTER(Decoded), C.c_int]
        L.irdm_group_poll_ida.argtypes = [C.c_void_p, C.POINTER(Ida), C.c_int]
        L.irdm_chunks_complete.argtypes = [C.c_void_p]
        L.irdm_chunks_complete.restype = C.c_uint64
        L.irdm_required_overlap.argtypes = [C.c_void_p]
        L.irdm_required_overlap.restype = C.c_size_t
        L.irdm_max_chunk_samples.argtypes = [C.c_void_p]
        L.irdm_max_chunk_samples.restype = C.c_size_t
        L.irdm_bytes_per_sample.argtypes = [C.c_void_p]
        L.irdm_bytes_per_sample.restype = C.c_size_t
        L.irdm_wait_ingest.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def host_alloc(nbytes):
    """Pinned host buffer (irdm_host_alloc) as (pointer, numpy uint8 view); release with host_free(pointer)."""
    L = lib()
    ptr = L.irdm_host_alloc(nbytes)
    if not ptr:
        raise MemoryError("irdm_host_alloc(%d) failed" % nbytes)
    view = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr))
    return ptr, view


def device_buffer(array, device=0):
    """Copy a numpy array into a device buffer allocated by the library (irdm_device_alloc + irdm_device_upload);
    returns the device pointer, release with device_free(pointer)."""
    L = lib()
    a = np.ascontiguousarray(array)
    ptr = L.irdm_device_alloc(device, a.nbytes)
    if not ptr:
        raise MemoryError("irdm_device_alloc(%d) failed" % a.nbytes)
    if L.irdm_device_upload(ptr, a.ctypes.data_as(C.c_void_p), a.nbytes) != 0:
        L.irdm_device_free(ptr)
        raise RuntimeError("irdm_device_upload failed")
    return ptr


def device_free(ptr):
    lib().irdm_device_free(ptr)


def host_free(ptr):
    lib().irdm_host_free(C.c_void_p(ptr))


class GpuBurstFFT:
    """gpu_burst_fft_* (opencl/burst_fft.h:35-47)."""

    def __init__(self, fft_size, batch_size, window):
        self.L = lib()
        self.n, self.batch = fft_size, batch_size
        w = np.ascontiguousarray(window, np.float32)
        self.h = self.L.gpu_burst_fft_create(fft_size, batch_size, _fp(w))
        if not self.h:
            raise RuntimeError("gpu_burst_fft_create failed (no GPU?)")

    def process(self, frames):
        """frames: complex64 [count, n] -> float32 [count, n] (mag^2, DC-shifted); raises on -1."""
        x = np.ascontiguousarray(frames, np.complex64)
        count = x.shape[0]
        out = np.empty((count, self.n), np.float32)
        rc = self.L.gpu_burst_fft_process(self.h, _fp(x.view(np.float32)), _fp(out), count)
        if rc != 0:
            raise RuntimeError("gpu_burst_fft_process returned %d" % rc)
        return out

    def process_rc(self, frames, count):
        x = np.ascontiguousarray(frames, np.complex64)
        out = np.empty((max(count, 1), self.n), np.float32)
        return self.L.gpu_burst_fft_process(self.h, _fp(x.view(np.float32)), _fp(out), count)

    def close(self):
        if self.h:
            self.L.gpu_burst_fft_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class Pipeline:
    """irdm_* batched detect -> downmix -> demod context."""

    def __init__(self, sample_rate, fmt=FMT_CF32, center_frequency=1622000000.0, threshold_db=0.0,
                 feed_block=0, use_gardner=1, start_time_ns=1700000000 * 10**9, device=0,
                 max_chunk_samples=0, max_bursts_per_chunk=0, pipeline_depth=0):
        self.L = lib()
        self.cfg = Config(center_frequency, int(sample_rate), threshold_db, fmt, feed_block,
                          use_gardner, start_time_ns, device, max_chunk_samples, max_bursts_per_chunk,
                          pipeline_depth)
        self.h = self.L.irdm_create(C.byref(self.cfg))
        if not self.h:
            raise RuntimeError("irdm_create failed (no GPU, or bad config)")
        self.fmt = fmt
        self.fft_size = self.L.irdm_fft_size(self.h)

    def set_option(self, key, value):
        if self.L.irdm_set_option(self.h, key.encode(), int(value)) != 0:
            raise ValueError("unknown option %r" % key)

    def stat(self, key):
        return int(self.L.irdm_get_stat(self.h, key.encode()))

    def feed_host(self, iq):
        iq = np.ascontiguousarray(iq)
        n = len(iq) if self.fmt == FMT_CF32 else len(iq) // 2
        rc = self.L.irdm_feed_host(self.h, iq.ctypes.data_as(C.c_void_p), n)
        if rc < 0:
            raise RuntimeError("irdm_feed_host failed")
        return rc

    def poll_decoded(self):
        """irdm_poll_decoded: one Decoded per polled Demod (option "decode_frames" = 1)."""
        return self._poll(self.L.irdm_poll_decoded, Decoded)

    def poll_ida(self):
        """irdm_poll_ida: one Ida per polled Demod (option "decode_ida" = 1)."""
        return self._poll(self.L.irdm_poll_ida, Ida)

    def ida_decode_batch(self, demods, use_llr=True):
        """irdm_ida_decode_batch: ida_decode() for a list of Demod records (direction field = demodulator's)."""
        n = len(demods)
        arr = (Demod * max(n, 1))(*demods)
        out = (Ida * max(n, 1))()
        if self.L.irdm_ida_decode_batch(self.h, arr, n, 1 if use_llr else 0, out) != 0:
            raise RuntimeError("irdm_ida_decode_batch failed")
        return [out[i] for i in range(n)]

    def frame_decode_batch(self, demods, use_llr=True):
        """irdm_frame_decode_batch: frame_decode() for a list of Demod records."""
        n = len(demods)
        arr = (Demod * max(n, 1))(*demods)
        out = (Decoded * max(n, 1))()
        if self.L.irdm_frame_decode_batch(self.h, arr, n, 1 if use_llr else 0, out) != 0:
            raise RuntimeError("irdm_frame_decode_batch failed")
        return [out[i] for i in range(n)]

    def feed_host_ptr(self, ptr, n_samples):
        """irdm_feed_host on a raw host pointer (e.g. a pinned buffer from host_alloc)."""
        rc = self.L.irdm_feed_host(self.h, C.c_void_p(ptr), n_samples)
        if rc < 0:
            raise RuntimeError("irdm_feed_host failed")
        return rc

    def feed_device(self, ptr, n_samples, stream=None):
        rc = self.L.irdm_feed_device(self.h, C.c_void_p(ptr), n_samples, C.c_void_p(stream or 0))
        if rc < 0:
            raise RuntimeError("irdm_feed_device failed")
        return rc

    def feed_begin(self, ptr, n_samples, stream=None):
        """K1 + history-ring copy of a device-resident chunk (what does not depend on the detector state)."""
        if self.L.irdm_feed_begin(self.h, C.c_void_p(ptr), n_samples, C.c_void_p(stream or 0)) != 0:
            raise RuntimeError("irdm_feed_begin failed")

    def feed_end(self):
        rc = self.L.irdm_feed_end(self.h)
        if rc < 0:
            raise RuntimeError("irdm_feed_end failed")
        return rc

    def ingest_ptr(self, n_samples):
        """Device address the next chunk may be written to in place (its slot of the history ring), or None."""
        return self.L.irdm_ingest_ptr(self.h, n_samples)

    def ring(self):
        """(device address, length in samples) of the history ring."""
        n = C.c_uint64(0)
        ptr = self.L.irdm_ring_ptr(self.h, C.byref(n))
        return ptr, int(n.value)

    def state_bytes(self):
        return int(self.L.irdm_state_bytes(self.h))

    def export_state_device(self, dptr, cap):
        if self.L.irdm_export_state_device(self.h, C.c_void_p(dptr), cap) < 0:
            raise RuntimeError("irdm_export_state_device failed")

    def state_head_bytes(self):
        return int(self.L.irdm_state_head_bytes(self.h))

    def import_state_head_device(self, dptr, n):
        if self.L.irdm_import_state_head_device(self.h, C.c_void_p(dptr), n) != 0:
            raise RuntimeError("irdm_import_state_head_device failed")

    def expect_history(self, dptr):
        """True: the scan the next feed_end enqueues waits on the device until import_state_history_device(dptr, n) says
        the history has arrived at dptr"""
        return bool(self.L.irdm_expect_history(self.h, C.c_void_p(dptr)))

    def import_state_history_device(self, dptr, n):
        if self.L.irdm_import_state_history_device(self.h, C.c_void_p(dptr), n) != 0:
            raise RuntimeError("irdm_import_state_history_device failed")

    def import_state_device(self, dptr, n):
        if self.L.irdm_import_state_device(self.h, C.c_void_p(dptr), n) != 0:
            raise RuntimeError("irdm_import_state_device failed")

    def seed_history_device(self, dptr, n_samples, abs_start):
        if self.L.irdm_seed_history_device(self.h, C.c_void_p(dptr), n_samples, abs_start) != 0:
            raise RuntimeError("irdm_seed_history_device failed")

    def flush(self):
        rc = self.L.irdm_flush(self.h)
        if rc < 0:
            raise RuntimeError("irdm_flush failed")
        return rc

    def advance(self):
        """irdm_flush without the waiting: the scan in flight settled, its bursts' chain enqueued, finished records out"""
        rc = self.L.irdm_advance(self.h)
        if rc < 0:
            raise RuntimeError("irdm_advance failed")
        return rc

    def _poll(self, fn, typ, chunk=256):
        out = []
        buf = (typ * chunk)()
        while True:
            n = fn(self.h, buf, chunk)
            if n <= 0:
                break
            out += [typ.from_buffer_copy(buf[i]) for i in range(n)]
        return out

    def _poll_raw(self, fn, typ, chunk):
        """Drain a result queue into one numpy byte matrix [n, sizeof(typ)] (no per-record objects)."""
        size = C.sizeof(typ)
        parts = []
        while True:
            buf = np.empty((chunk, size), np.uint8)
            n = fn(self.h, buf.ctypes.data_as(C.POINTER(typ)), chunk)
            if n <= 0:
                break
            parts.append(buf[:n])
            if n < chunk:
                break
        return np.concatenate(parts) if parts else np.empty((0, size), np.uint8)

    def poll_bursts_raw(self, chunk=4096):
        return self._poll_raw(self.L.irdm_poll_bursts, Burst, chunk)

    def poll_demods_raw(self, chunk=2048):
        return self._poll_raw(self.L.irdm_poll_demods, Demod, chunk)

    def poll_demods_packed_raw(self, chunk=8192):
        """option packed_records 1: [n, 176] bytes, irdm_demod_packed_t records"""
        return self._poll_raw(self.L.irdm_poll_demods_packed, DemodPacked, chunk)

    def poll_demods_packed(self):
        return self._poll(self.L.irdm_poll_demods_packed, DemodPacked)

    def drop_frames(self, chunk=4096):
        """Discard queued frame records (metadata only path)."""
        buf = (FrameInfo * chunk)()
        total = 0
        while True:
            n = self.L.irdm_poll_frames(self.h, buf, None, chunk)
            if n <= 0:
                break
            total += n
            if n < chunk:
                break
        return total

    def poll_bursts(self):
        return self._poll(self.L.irdm_poll_bursts, Burst)

    def poll_demods(self):
        return self._poll(self.L.irdm_poll_demods, Demod)

    def poll_frames(self, chunk=64):
        infos, samples = [], []
        buf = (FrameInfo * chunk)()
        sb = np.zeros((chunk, 2 * MAX_FRAME_SAMPLES), np.float32)
        while True:
            n = self.L.irdm_poll_frames(self.h, buf, _fp(sb), chunk)
            if n <= 0:
                break
            for i in range(n):
                fi = FrameInfo.from_buffer_copy(buf[i])
                infos.append(fi)
                samples.append(sb[i, :2 * fi.num_samples].copy().view(np.complex64))
        return infos, samples

    def downmix_burst(self, info, samples):
        """burst_downmix_process for one burst (host samples, complex64) -> (FrameInfo, complex64 frame | None)."""
        x = np.ascontiguousarray(samples, np.complex64)
        fi = FrameInfo()
        out = np.zeros(2 * MAX_FRAME_SAMPLES, np.float32)
        rc = self.L.irdm_downmix_burst(self.h, C.byref(info), _fp(x.view(np.float32)), len(x), C.byref(fi), _fp(out))
        if rc < 0:
            raise RuntimeError("irdm_downmix_burst failed")
        return fi, (out[:2 * fi.num_samples].view(np.complex64).copy() if rc == 1 else None)

    def qpsk_demod_batch(self, frames, directions):
        """frames: list of complex64 arrays (<= 4440 samples); returns list of Demod."""
        n = len(frames)
        buf = np.zeros((n, 2 * MAX_FRAME_SAMPLES), np.float32)
        ns = np.zeros(n, np.int32)
        for i, f in enumerate(frames):
            f = np.ascontiguousarray(f, np.complex64)
            ns[i] = len(f)
            buf[i, :2 * len(f)] = f.view(np.float32)
        dirs = np.ascontiguousarray(directions, np.int32)
        out = (Demod * n)()
        rc = self.L.irdm_qpsk_demod_batch(self.h, _fp(buf), ns.ctypes.data_as(C.POINTER(C.c_int)),
                                          dirs.ctypes.data_as(C.POINTER(C.c_int)), n, out)
        if rc != 0:
            raise RuntimeError("irdm_qpsk_demod_batch failed")
        return [Demod.from_buffer_copy(out[i]) for i in range(n)]

    def export_state(self):
        n = self.L.irdm_state_bytes(self.h)
        buf = np.empty(n, np.uint8)
        if self.L.irdm_export_state(self.h, buf.ctypes.data_as(C.c_void_p), n) != n:
            raise RuntimeError("irdm_export_state failed")
        return buf

    def import_state(self, buf):
        buf = np.ascontiguousarray(buf, np.uint8)
        if self.L.irdm_import_state(self.h, buf.ctypes.data_as(C.c_void_p), len(buf)) != 0:
            raise RuntimeError("irdm_import_state failed")

    def seed_history(self, iq_tail, abs_start):
        iq_tail = np.ascontiguousarray(iq_tail)
        n = len(iq_tail) if self.fmt == FMT_CF32 else len(iq_tail) // 2
        if self.L.irdm_seed_history(self.h, iq_tail.ctypes.data_as(C.c_void_p), n, abs_start) != 0:
            raise RuntimeError("irdm_seed_history failed")

    def last_magnitudes(self, max_frames):
        out = np.zeros((max_frames, self.fft_size), np.float32)
        n = self.L.irdm_last_magnitudes(self.h, _fp(out), max_frames)
        return out[:n]

    def baseline_sum(self):
        out = np.zeros(self.fft_size, np.float32)
        self.L.irdm_baseline_sum(self.h, _fp(out))
        return out

    def burst_samples(self, i, n):
        out = np.zeros(2 * n, np.float32)
        got = self.L.irdm_burst_samples(self.h, i, _fp(out), n)
        if got < 0:
            raise RuntimeError("irdm_burst_samples failed")
        return out[:2 * got].view(np.complex64)

    def kernel_clock(self, which, reset=False):
        """(sum of the launches' device spans in ms, launches, last span in ms) of the decimator (0) / K1 (1); option
        kernel_clock must be on"""
        sm, n, last = C.c_double(0), C.c_uint64(0), C.c_double(0)
        if self.L.irdm_kernel_clock(self.h, which, C.byref(sm), C.byref(n), C.byref(last), 1 if reset else 0) != 0:
            raise RuntimeError("irdm_kernel_clock failed")
        return sm.value, int(n.value), last.value

    def timings(self):
        t = (C.c_float * 6)()
        self.L.irdm_last_timings(self.h, t, 6)
        return dict(zip(["fft_mag", "scan", "fir", "post", "demod", "total"], [float(v) for v in t]))

    @property
    def tagged(self):
        return self.L.irdm_tagged_bursts(self.h)

    @property
    def sample_count(self):
        return self.L.irdm_sample_count(self.h)

    def close(self):
        if self.h:
            self.L.irdm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Group:
    """irdm_group_*: ONE stream across n_gpus GPUs of this process, chunk k on member k mod n_gpus, the detector's state
    handed from member to member with RCCL (csrc/group.cpp).  The polls return the members' records merged in stream
    order -- what one Pipeline fed with the same samples returns."""
    _poll = Pipeline._poll            # (fn(self.h, buffer, max): the group's polls have the pipeline's shape)

    def __init__(self, sample_rate, n_gpus, devices=None, fmt=FMT_CF32, center_frequency=1622000000.0, threshold_db=0.0,
                 feed_block=0, use_gardner=1, start_time_ns=1700000000 * 10**9, max_chunk_samples=0,
                 max_bursts_per_chunk=0, pipeline_depth=1):
        self.L = lib()
        self.cfg = Config(center_frequency, int(sample_rate), threshold_db, fmt, feed_block, use_gardner, start_time_ns, 0,
                          max_chunk_samples, max_bursts_per_chunk, pipeline_depth)
        devs = (C.c_int * n_gpus)(*devices) if devices is not None else None
        self.g = self.L.irdm_group_create(C.byref(self.cfg), n_gpus, devs)
        if not self.g:
            raise RuntimeError("irdm_group_create failed (devices, RCCL, or bad config)")
        self.h = self.g                   # (Pipeline's poll helpers call fn(self.h, ...))
        self.fmt = fmt
        self.n_gpus = n_gpus
        self.fft_size = self.L.irdm_fft_size(self.L.irdm_group_member(self.g, 0))
        self.chunk = int(self.L.irdm_max_chunk_samples(self.L.irdm_group_member(self.g, 0)))

    def member(self, i):
        return self.L.irdm_group_member(self.g, i)

    def set_option(self, key, value):
        if self.L.irdm_group_set_option(self.g, key.encode(), int(value)) != 0:
            raise ValueError("option %r refused" % key)

    def stat(self, key):
        return int(self.L.irdm_group_get_stat(self.g, key.encode()))

    def _samples(self, iq):
        return len(iq) if self.fmt == FMT_CF32 else len(iq) // 2

    def stage_host(self, iq):
        if self.L.irdm_group_stage_host(self.g, iq.ctypes.data_as(C.c_void_p), self._samples(iq)) != 0:
            raise RuntimeError("irdm_group_stage_host failed")

    def feed_host(self, iq):
        """iq: a C-contiguous array of at most n_gpus chunks; it must stay alive until the call returns (and, if it was
        staged first, from the stage call on)"""
        rc = self.L.irdm_group_feed_host(self.g, iq.ctypes.data_as(C.c_void_p), self._samples(iq))
        if rc < 0:
            raise RuntimeError("irdm_group_feed_host failed")
        return rc

    def stage_device(self, ptr, n_samples):
        if self.L.irdm_group_stage_device(self.g, C.c_void_p(ptr), n_samples) != 0:
            raise RuntimeError("irdm_group_stage_device failed")

    def feed_device(self, ptr, n_samples, stream=None):
        rc = self.L.irdm_group_feed_device(self.g, C.c_void_p(ptr), n_samples)
        if rc < 0:
            raise RuntimeError("irdm_group_feed_device failed")
        return rc

    def flush(self):
        rc = self.L.irdm_group_flush(self.g)
        if rc < 0:
            raise RuntimeError("irdm_group_flush failed")
        return rc

    def poll_bursts(self):
        return self._poll(self.L.irdm_group_poll_bursts, Burst)

    def poll_demods(self):
        return self._poll(self.L.irdm_group_poll_demods, Demod)

    def poll_demods_packed(self):
        return self._poll(self.L.irdm_group_poll_demods_packed, DemodPacked)

    def poll_decoded(self):
        return self._poll(self.L.irdm_group_poll_decoded, Decoded)

    def poll_ida(self):
        return self._poll(self.L.irdm_group_poll_ida, Ida)

    def poll_frames(self, chunk=64):
        infos, samples = [], []
        buf = (FrameInfo * chunk)()
        sb = np.zeros((chunk, 2 * MAX_FRAME_SAMPLES), np.float32)
        while True:
            n = self.L.irdm_group_poll_frames(self.g, buf, _fp(sb), chunk)
            if n <= 0:
                break
            for i in range(n):
                fi = FrameInfo.from_buffer_copy(buf[i])
                infos.append(fi)
                samples.append(sb[i, :2 * fi.num_samples].copy().view(np.complex64))
        return infos, samples

    @property
    def tagged(self):
        return self.stat("tagged")

    def close(self):
        if self.g:
            self.L.irdm_group_destroy(self.g)
            self.g = self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def format_raw(demods, file_info="golden"):
    """frame_output_print for a list of Demod records (t0 from the first, frame_output.c:144-158)."""
    L = lib()
    t0 = C.c_uint64(0)
    buf = C.create_string_buffer(4096)
    out = []
    for d in demods:
        n = L.irdm_format_raw(C.byref(d), file_info.encode() if file_info else None, C.byref(t0), buf, 4096)
        if n < 0:
            raise RuntimeError("irdm_format_raw failed")
        out.append(buf.value.decode())
    return out


def format_raw_batch(demods, file_info="golden"):
    """irdm_format_raw_batch: all lines in one buffer (one write per batch); returns the text."""
    L = lib()
    L.irdm_format_raw_batch.restype = C.c_longlong
    n = len(demods)
    arr = (Demod * n)(*demods)
    cap = max(n, 1) * 1280
    buf = C.create_string_buffer(cap)
    t0 = C.c_uint64(0)
    rc = L.irdm_format_raw_batch(arr, n, file_info.encode() if file_info else None, C.byref(t0), buf, cap)
    if rc < 0:
        raise RuntimeError("irdm_format_raw_batch failed")
    return buf.raw[:rc].decode()


def save_burst(info, samples, dirname):
    """irdm_save_burst: the reference's --save-bursts file pair for one frame (qpsk_demod.c:339-389)."""
    L = lib()
    L.irdm_save_burst.argtypes = [C.POINTER(FrameInfo), C.POINTER(C.c_float), C.c_char_p]
    s = np.ascontiguousarray(samples, np.float32)
    return L.irdm_save_burst(C.byref(info), s.ctypes.data_as(C.POINTER(C.c_float)), dirname.encode())

