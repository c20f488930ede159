"""A guard for a gfx950 hazard the compiler does not know about (found with rot_phase_rows_kernel): a
buffer store of more than 64 bits whose scalar-offset field holds an SGPR, followed at once by a VALU instruction that
writes the store's data registers, stored the NEW values for part of the wavefront's lanes (LLVM's hazard recognizer
exempts exactly this form -- a register in the soffset field -- from its "wide store, then VALU write of vdata" rule).
The product's kernels are compiled to gfx950 assembly here and scanned: no such pair within two instructions of each
other.  No GPU needed (hipcc cross-compiles); skipped where hipcc is absent."""
import concurrent.futures
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "iridium-sniffer_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FILES = ["detect.hip", "scan_band.hip", "scan_fast.hip", "fir_reg.hip", "downmix.hip", "demod.hip", "bitlayer.hip"]


def _asm(name, out_dir):
    out = os.path.join(out_dir, name + ".s")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-x", "hip",
                    "--cuda-device-only", "-S", os.path.join(CSRC, name), "-o", out], check=True, capture_output=True)
    return open(out).read()


def hazards(asm):
    lines = asm.split("\n")
    func = "?"
    ins = []
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            func = m.group(1)
        elif re.match(r"^\s+[a-z][a-z_0-9]+", l) and not l.strip().startswith("."):
            ins.append((func, l.strip()))
    found = []
    for n, (fn, l) in enumerate(ins):
        m = re.match(r"buffer_store_(?:dwordx[34]|format_xyzw?|b(?:96|128)) v\[(\d+):(\d+)\], (?:v\d+|v\[\d+:\d+\]|off), s\[\d+:\d+\], (\S+)", l)
        if not m or not m.group(3).startswith(("s", "m0", "vcc", "ttmp")):
            continue                                  # (a constant in the scalar-offset field: the compiler guards that form)
        lo, hi = int(m.group(1)), int(m.group(2))
        for d in (1, 2):
            if n + d >= len(ins) or ins[n + d][0] != fn:
                break
            nl = ins[n + d][1]
            mm = re.match(r"(v_\w+) (?:v\[(\d+):(\d+)\]|v(\d+))", nl)
            if not mm or mm.group(1).startswith(("v_cmp", "v_cmpx")):
                continue
            a, b = (int(mm.group(2)), int(mm.group(3))) if mm.group(2) else (int(mm.group(4)),) * 2
            if a <= hi and b >= lo:
                found.append((fn[:70], d, l, nl))
    return found


def test_the_scanner_sees_the_pattern():
    bad = """_Zk:
	buffer_store_dwordx4 v[32:35], v68, s[0:3], s9 offen
	v_pk_mul_f32 v[32:33], v[8:9], v[64:65] op_sel_hi:[1,0]
	buffer_store_dwordx4 v[36:39], v13, s[0:3], 0 offen
	v_pk_add_f32 v[36:37], v[32:33], v[34:35]
	buffer_store_dwordx4 v[40:43], v13, s[0:3], s9 offen
	s_nop 0
	v_mov_b32_e32 v2, v3
	v_pk_add_f32 v[40:41], v[32:33], v[34:35]
"""
    got = hazards(bad)
    assert len(got) == 1 and got[0][1] == 1 and "v[32:35]" in got[0][2], got


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not present")
def test_no_wide_buffer_store_is_followed_by_a_write_of_its_data_registers(tmp_path):
    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        asms = list(ex.map(lambda f: _asm(f, str(tmp_path)), FILES))
    found = []
    stores = 0
    for name, asm in zip(FILES, asms):
        stores += len(re.findall(r"buffer_store_dwordx[34]", asm))
        found += [(name,) + h for h in hazards(asm)]
    assert stores >= 16                                # (K1's and the rotator's stores are of this kind: the scan saw code)
    assert not found, found
