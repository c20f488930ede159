"""Cheap pins against the reference's own headers and object code (CPU only; oracle/_ref = the reference's sources compiled in
place, authoring container only):
  * the struct layouts include/irdm_compat.h RESTATES (burst_downmix.h:31-53, qpsk_demod.h:24-38, the direction enum) equal
    sizeof / offsetof of the reference's headers -- read out of oracle/_ref (ref_layout, oracle/ref_glue.c), recorded in
    tests/golden/ref_layouts.json so that the check of irdm_compat.h also runs where the reference is absent.
    (burst_detect.h's three structs are not covered: that header includes <fftw3.h>, which this image lacks.)
  * --save-bursts: the file pair irdm_save_burst writes -- names, the .meta text byte for byte, the .cf32 payload -- equals
    what the reference's save_burst_iq (qpsk_demod.c:339-389) writes when its qpsk_demod is run with save_bursts_dir set, for
    an accepted downlink frame, an uplink frame and a frame whose unique word fails (DIR_UNDEF, "UN")."""
import ctypes as C
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

import irdm
import siggen

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden", "ref_layouts.json")

COMPAT_PROBE = r"""
#include <stdio.h>
#include <stddef.h>
#include "irdm_compat.h"
#define LAY(T, F) printf("%s.%s %ld\n", #T, #F, (long)offsetof(T, F))
#define SZ(T) printf("sizeof %s %ld\n", #T, (long)sizeof(T))
int main(void)
{
    SZ(downmix_frame_t); LAY(downmix_frame_t, id); LAY(downmix_frame_t, timestamp); LAY(downmix_frame_t, center_frequency);
    LAY(downmix_frame_t, sample_rate); LAY(downmix_frame_t, samples_per_symbol); LAY(downmix_frame_t, direction);
    LAY(downmix_frame_t, magnitude); LAY(downmix_frame_t, noise); LAY(downmix_frame_t, uw_start);
    LAY(downmix_frame_t, num_samples); LAY(downmix_frame_t, samples);
    SZ(downmix_config_t); LAY(downmix_config_t, output_sample_rate); LAY(downmix_config_t, search_depth);
    LAY(downmix_config_t, handle_multiple_frames);
    SZ(demod_frame_t); LAY(demod_frame_t, id); LAY(demod_frame_t, timestamp); LAY(demod_frame_t, center_frequency);
    LAY(demod_frame_t, direction); LAY(demod_frame_t, magnitude); LAY(demod_frame_t, noise); LAY(demod_frame_t, confidence);
    LAY(demod_frame_t, level); LAY(demod_frame_t, n_symbols); LAY(demod_frame_t, n_payload_symbols); LAY(demod_frame_t, bits);
    LAY(demod_frame_t, llr); LAY(demod_frame_t, n_bits);
    SZ(ir_direction_t);
    printf("DIR_UNDEF %d\nDIR_DOWNLINK %d\nDIR_UPLINK %d\n", DIR_UNDEF, DIR_DOWNLINK, DIR_UPLINK);
    return 0;
}
"""


def _ref_layouts(reflib):
    reflib.ref_layout.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_long)]
    reflib.ref_layout.restype = C.c_int
    out, i = {}, 0
    name, val = C.c_char_p(), C.c_long()
    while reflib.ref_layout(i, C.byref(name), C.byref(val)):
        out[name.value.decode().replace("sizeof ", "sizeof ")] = int(val.value)
        i += 1
    return out


def _compat_layouts():
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "probe.c")
        open(src, "w").write(COMPAT_PROBE)
        exe = os.path.join(td, "probe")
        subprocess.check_call(["gcc", "-std=gnu99", "-I", os.path.join(ROOT, "include"), "-o", exe, src])
        out = {}
        for line in subprocess.check_output([exe]).decode().splitlines():
            k, v = line.rsplit(" ", 1)
            out[k] = int(v)
        return out


def test_golden_layouts_are_the_reference_headers(reflib):
    """(authoring container) the committed table is what the reference's headers give here"""
    ref = _ref_layouts(reflib)
    assert len(ref) >= 34
    if os.environ.get("IRDM_WRITE_GOLDEN") == "1":
        json.dump(ref, open(GOLDEN, "w"), indent=1, sort_keys=True)
    assert json.load(open(GOLDEN)) == ref


def test_compat_header_layouts_equal_the_reference():
    want = json.load(open(GOLDEN))
    got = _compat_layouts()
    assert got == want, {k: (got.get(k), want.get(k)) for k in set(got) | set(want) if got.get(k) != want.get(k)}


def _frame(uplink, spoil, seed):
    """a frame at 250 kHz as stage B delivers it: unique word at sample 0, 10 samples per symbol"""
    rng = np.random.default_rng(seed)
    quads = list(siggen.UW_UL if uplink else siggen.UW_DL) + rng.integers(0, 4, size=150).tolist()
    if spoil:
        quads[:12] = [1, 3, 1, 3, 1, 3, 1, 3, 1, 3, 1, 3]      # no unique word of either direction
    sig = siggen.make_burst(250000, quads, 0.0, 0.3, amp=0.05)
    lead = 5 * 10                                               # the pulse's span in front of the first symbol
    x = sig[lead:lead + len(quads) * 10].astype(np.complex64)
    x += (rng.standard_normal(len(x)) + 1j * rng.standard_normal(len(x))).astype(np.complex64) * np.float32(0.0005)
    return np.ascontiguousarray(x)


@pytest.mark.parametrize("case", ["downlink", "uplink", "uw_fails"])
def test_save_bursts_file_pair_is_the_reference(reflib, case, tmp_path):
    x = _frame(case == "uplink", case == "uw_fails", seed={"downlink": 1, "uplink": 2, "uw_fails": 3}[case])
    given_dir = 2 if case == "uplink" else 1
    args = dict(id=730, ts=1700000000123456789, cf=1626270833.0 + 0.4, mag=23.456, noise=-112.345, uw=3.21)
    ref_dir, our_dir = str(tmp_path / "ref"), str(tmp_path / "ours")
    reflib.ref_set_save_bursts_dir.argtypes = [C.c_char_p]
    reflib.ref_qpsk_demod_save.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_float, C.c_float, C.c_int, C.c_double, C.c_uint64,
                                           C.c_uint64, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_int)]
    reflib.ref_qpsk_demod_save.restype = C.c_int
    left = C.c_int(-1)
    reflib.ref_set_save_bursts_dir(ref_dir.encode())
    try:
        ok = reflib.ref_qpsk_demod_save(x.view(np.float32).ctypes.data_as(C.POINTER(C.c_float)), len(x), 250000.0, 10.0, given_dir,
                                        args["cf"], args["id"], args["ts"], args["mag"], args["noise"], args["uw"], C.byref(left))
    finally:
        reflib.ref_set_save_bursts_dir(None)
    assert ok == (0 if case == "uw_fails" else 1)
    assert left.value == {"downlink": 1, "uplink": 2, "uw_fails": 0}[case]
    info = irdm.FrameInfo()
    info.id, info.timestamp, info.center_frequency = args["id"], args["ts"], args["cf"]
    info.sample_rate, info.samples_per_symbol = 250000.0, 10.0
    info.direction, info.magnitude, info.noise, info.uw_start = given_dir, args["mag"], args["noise"], args["uw"]
    info.num_samples, info.drop_reason = len(x), 0
    info.demod_ok = 0 if case == "uw_fails" else 1
    info.demod_direction = left.value              # (what the pipeline reports in irdm_frame_info_t.demod_direction, qpsk_demod.c:444)
    assert irdm.save_burst(info, x.view(np.float32), our_dir) == 0
    ref_files, our_files = sorted(os.listdir(ref_dir)), sorted(os.listdir(our_dir))
    assert ref_files == our_files and len(ref_files) == 2, (ref_files, our_files)
    for f in ref_files:
        assert open(os.path.join(ref_dir, f), "rb").read() == open(os.path.join(our_dir, f), "rb").read(), f
    meta = open(os.path.join(our_dir, [f for f in our_files if f.endswith(".meta")][0])).read()
    assert "direction: %s\n" % {"downlink": "DL", "uplink": "UL", "uw_fails": "UN"}[case] in meta
