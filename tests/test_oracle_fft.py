"""The pinned radix-2 DIT float32 FFT (oracle) is a correct FFT: checked against
numpy.fft in double precision (<= 1e-5 relative)."""
import ctypes as C

import numpy as np
import pytest

import orc


@pytest.mark.parametrize("n", [8, 64, 2048, 4096, 8192, 16384])
@pytest.mark.parametrize("direction", [-1, 1])
def test_fft_matches_numpy(oracle, n, direction):
    rng = np.random.default_rng(n + direction)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    buf = x.copy()
    oracle.orc_fft(buf.ctypes.data_as(C.POINTER(C.c_float)), n, direction)
    ref = np.fft.fft(x.astype(np.complex128)) if direction < 0 else np.fft.ifft(x.astype(np.complex128)) * n
    err = np.linalg.norm(buf - ref) / np.linalg.norm(ref)
    assert err < 1e-5, err


def test_fft_impulse_and_dc(oracle):
    n = 8192
    x = np.zeros(n, np.complex64)
    x[0] = 1
    oracle.orc_fft(x.ctypes.data_as(C.POINTER(C.c_float)), n, -1)
    assert np.all(x == 1)
    x = np.ones(n, np.complex64)
    oracle.orc_fft(x.ctypes.data_as(C.POINTER(C.c_float)), n, -1)
    assert x[0] == n and np.all(x[1:] == 0)     # the self-test the reference's Vulkan backend runs (vulkan/burst_fft.c:324-394)


def test_fft_linearity_round_trip(oracle):
    n = 2048
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    y = x.copy()
    oracle.orc_fft(y.ctypes.data_as(C.POINTER(C.c_float)), n, -1)
    oracle.orc_fft(y.ctypes.data_as(C.POINTER(C.c_float)), n, 1)
    assert np.allclose(y / n, x, atol=2e-5)
