"""-m gpu: the device's cexpf(i x) (csrc/libm_port.hpp: glibc's sincosf restated, the fine-CFO step of the per-burst
chain, burst_downmix.c:716-717) against the HOST's libm, bit for bit.

The step's argument is -2 pi offset with |offset| <= 1/4.  Here: 4 Mi random floats over [-1.7, 1.7] (uniform in
value and uniform in bit pattern), every float of the windows around the branch points of the routine (0, 2^-126,
2^-12, pi/4, pi/2) and the quadrant boundaries.  tools/check_sincosf_gpu.c runs EVERY float of [-2, 2]
(profiles/r3_sincosf_exhaustive.txt); tools/check_sincosf.cpp does the same for the host-compiled header."""
import ctypes as C

import numpy as np
import pytest

import irdm

pytestmark = pytest.mark.gpu


def _host_sincos(x):
    libm = C.CDLL("libm.so.6")
    libm.sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    s, c = C.c_float(), C.c_float()
    re = np.empty(len(x), np.float32)
    im = np.empty(len(x), np.float32)
    for i, v in enumerate(x):
        libm.sincosf(float(v), C.byref(s), C.byref(c))
        re[i], im[i] = c.value, s.value
    tiny = np.abs(x) <= np.float32(2.0 ** -126)        # cexpf skips sincosf there (s_cexp_template.c)
    re[tiny] = 1.0
    im[tiny] = x[tiny]
    return re, im


def _device(x):
    L = irdm.lib()
    L.irdm_sincosf_probe.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    x = np.ascontiguousarray(x, np.float32)
    re = np.empty_like(x)
    im = np.empty_like(x)
    assert L.irdm_sincosf_probe(0, x.ctypes.data, len(x), re.ctypes.data, im.ctypes.data) == 0
    return re, im


def _window(center, half):
    c = np.float32(center).view(np.uint32).item()
    u = np.arange(c - half, c + half + 1, dtype=np.uint32)
    return np.concatenate([u.view(np.float32), (u | np.uint32(0x80000000)).view(np.float32)])


def test_device_cexpf_equals_host_libm():
    rng = np.random.default_rng(5)
    xs = [rng.uniform(-1.7, 1.7, 150_000).astype(np.float32),
          rng.integers(0, np.float32(1.7).view(np.uint32).item(), 150_000, dtype=np.uint32).view(np.float32),
          np.array([0.0, -0.0, 1e-45, -1e-45, 2.0 ** -126, -(2.0 ** -126)], np.float32)]
    for c in (2.0 ** -126, 2.0 ** -12, np.pi / 4, np.pi / 2, 3 * np.pi / 4, 1.0, 1.5707964):
        xs.append(_window(c, 2000))
    x = np.concatenate(xs)
    hr, hi = _host_sincos(x)
    dr, di = _device(x)
    bad = (hr.view(np.uint32) != dr.view(np.uint32)) | (hi.view(np.uint32) != di.view(np.uint32))
    assert not bad.any(), (int(bad.sum()), x[bad][:5], hr[bad][:5], dr[bad][:5])


def test_host_cfo_path_still_agrees():
    """the helper-thread form of the step (option host_cfo: what a host whose libm the port does not reproduce falls
    back to) gives the same records as the device form"""
    import orc
    import parity
    import siggen
    fs = 2_000_000
    n = int(1.2 * fs) // 32768 * 32768
    iq, _ = siggen.standard_scene(fs, n, 6, seed=41)
    ref = orc.run_stream(iq, fs)
    for depth in (0, 1):
        got = parity.run_gpu(iq, fs, depth=depth, options={"host_cfo": 1},
                             chunks=[n // 2 // 32768 * 32768, n - n // 2 // 32768 * 32768])
        parity.compare(got, ref)
