"""The product end to end without a GPU: every source of iridium-sniffer_amd/csrc (all kernel files as the gfx950 build
compiles them, the host sources, compat.cpp, host_design.cpp: the C-ABI of include/irdm_hip.h) is compiled with g++ against the
HIP emulation of tests/hip_emul (tests/emul_build.py says which lines are substituted: the dynamic-LDS declarations, six
inline-assembly statements, and the generated assembly of fir_mac.inc, restated in C++) and driven through irdm.py exactly
like the real library; every burst, downmixed frame (samples bit for bit), hard bit and LLR is compared with the oracle by
the same code as the -m gpu parity tests (tests/parity.py).  This is test infrastructure: the product never loads it."""
import json
import os
import subprocess
import sys

import pytest

import emul_build

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emul_lib():
    return emul_build.build()


def run_case(lib, case, timeout=900, order=None):
    env = dict(os.environ, IRDM_LIB=lib)
    if order:
        env["HIP_EMUL_ORDER"] = order
    p = subprocess.run([sys.executable, os.path.join(HERE, "emul_pipeline_run.py"), case], env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_whole_path_2mhz(emul_lib):
    """2 MHz: whole stream, chunked at pipeline_depth 1, chunked at depth 2 fed in place with look-ahead (chained scans),
    ci8 input, the sequential scan instead of the band scan, and a per-burst scratch that has to grow"""
    res = run_case(emul_lib, "2mhz")
    assert set(res) == {"whole", "chunked_depth1", "chunked_depth2_in_place_lookahead", "ci8", "sequential_scan", "scratch_growth",
                        "rotator_row_extension", "rotator_arena_growth", "chunked_depth4_in_place_lookahead",
                        "packed_depth3_in_place_lookahead", "packed_depth0", "rows_prebuilt"}
    assert res["scratch_growth"]["grows"] >= 1
    for name, s in res.items():
        assert s["bursts"] >= 4 and s["demods"] >= 3, (name, s)


@pytest.mark.parametrize("order", ["reverse", "shuffle"])
def test_workgroup_order_does_not_matter(emul_lib, order):
    """the hardware promises no order among the workgroups of a launch: the emulation runs them last to first / in a
    pseudo-random order that changes from launch to launch (HIP_EMUL_ORDER) -- same records"""
    res = run_case(emul_lib, "2mhz", order=order)
    for name, s in res.items():
        assert s["bursts"] >= 4 and s["demods"] >= 3, (name, s)


def test_whole_path_scene_zoo(emul_lib):
    res = run_case(emul_lib, "scene_zoo")
    assert res["too_long"]["bursts"] >= 2 and res["squelch"]["bursts"] >= 1 and res["dc_and_edges"]["bursts"] >= 1, res


def test_whole_path_10mhz_register_resident_decimator(emul_lib):
    """10 MHz in four chunks at pipeline_depth 2, fed in place with look-ahead: 8192-point frames through K1's
    32-points-per-lane kernel (with the candidate lists once the detector is primed), decimation by 40 through fir_reg.hip
    -- columns of rotated samples in registers, accumulators travelling from lane to lane by DPP shifts: the fused
    four-accumulator form of the reference's AVX2 kernel (default), and with the dispatched kernels in the reference's generic forms
    (IRDM_EMUL_FULL=1 adds the LDS decimator, another minute)"""
    res = run_case(emul_lib, "10mhz", timeout=1500)
    for name in res:
        assert res[name]["bursts"] >= 6 and res[name]["frames"] >= 4, res
    assert res["default"]["k1_lists"] >= 1, res          # a chunk whose candidate lists K1 wrote
    assert res["default"]["spec_scans"] >= 2, res        # scans that opened with round 1 behind a speculation pass
    assert {"default", "scalar_fir_order", "without_speculation_pass", "sums_pass_restart", "depth5", "any_m_decimator_scalar_order"} <= set(res)


@pytest.mark.skipif(not os.environ.get("IRDM_EMUL_FULL"), reason="four minutes of emulation: set IRDM_EMUL_FULL=1")
def test_whole_path_12mhz(emul_lib):
    """12 MHz: 16384-point frames, decimation by 48, cf32 whole and ci16 in two chunks at pipeline_depth 1"""
    res = run_case(emul_lib, "12mhz", timeout=2400)
    for name in ("default", "ci16_chunked_depth1"):
        assert res[name]["bursts"] >= 3 and res[name]["frames"] >= 2, res


def test_group_of_members_equals_one_context(emul_lib):
    """irdm_group_*: one stream over 2 and 3 members, a member handing the detector state to itself, super-steps staged
    ahead, the scatter from member 0 -- every record equal to the oracle's for the whole stream, in stream order"""
    res = run_case(emul_lib, "group", timeout=1800)
    assert set(res) == {"two_members", "three_members_staged_ahead", "two_members_scatter_from_member_0", "one_member_loopback",
                        "one_member", "two_members_ragged_end"}
    for name, s in res.items():
        assert s["bursts"] >= 36 and s["demods"] >= 30, (name, s)
