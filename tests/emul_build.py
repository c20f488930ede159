"""TEST INFRASTRUCTURE: builds tests/_build/libirdm_emul.so -- the product's sources (every kernel file,
the host sources plug / create / chain / scan_host / feed / state / api.cpp, csrc/group.cpp, csrc/host_design.cpp, csrc/compat.cpp: the whole C-ABI of include/irdm_hip.h) compiled with g++ against the
HIP emulation of tests/hip_emul/hip/hip_runtime.h (and, for group.cpp, the emulated RCCL of tests/hip_emul/rccl/rccl.h), so that the `-m "not gpu"` tests can drive the product end to end without a GPU
(tests/test_pipeline_emul.py).  Never loaded by the product: iridium-sniffer_amd/irdm.py loads libirdm_hip.so unless a
test points IRDM_LIB elsewhere.

Source changes made on the way (text substitutions on copies under tests/_build/emul/):
  * `extern __shared__ ... name[];` (dynamic LDS) -> a pointer to the emulation's LDS buffer
  * scan_fast.hip: `s_waitcnt` / `s_barrier` inline asm -> nothing / __syncthreads()
  * fir_reg.hip: the one `v_writelane_b32` inline asm -> hip_emul::writelane0
  * csrc/fir_mac.inc (generated gfx950 assembly) is replaced by tests/hip_emul/fir_mac.inc: the same multiply-add chains in
    plain C++, products and sums rounded separately, same order.
  * csrc/fft_bfly.inc (K1's butterflies: inline gfx950 assembly) likewise by tests/hip_emul/fft_bfly.inc, and
    csrc/fir_fma.inc (the AVX2-order decimator's fused multiply-adds and packed complex products) by tests/hip_emul/fir_fma.inc."""
import concurrent.futures
import os
import re
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "iridium-sniffer_amd", "csrc")
EMUL = os.path.join(ROOT, "tests", "hip_emul")
OUT = os.path.join(ROOT, "tests", "_build", "emul")
SO = os.path.join(ROOT, "tests", "_build", "libirdm_emul.so")
SOURCES = ["detect.hip", "scan_fast.hip", "scan_band.hip", "downmix.hip", "fir_reg.hip", "demod.hip", "bitlayer.hip",
           "plug.cpp", "create.cpp", "chain.cpp", "scan_host.cpp", "feed.cpp", "state.cpp", "api.cpp", "group.cpp", "host_design.cpp",
           "compat.cpp"]


def transform(name, text):
    text = re.sub(r"extern __shared__ __attribute__\(\(aligned\(16\)\)\) unsigned char (\w+)\[\];",
                  r"unsigned char *\1 = hip_emul::dyn_lds();", text)
    if name == "scan_fast.hip":
        text = text.replace('asm volatile("s_waitcnt vmcnt(0)" ::: "memory")', "((void)0)")
        text = text.replace('asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_barrier" ::: "memory")', "__syncthreads()")
    if name == "fir_reg.hip":
        text = text.replace('asm("v_writelane_b32 %0, %1, 0" : "=v"(r) : "s"(s), "0"(v));', "r = hip_emul::writelane0(v, s);")
    if name == "host_design.cpp":
        # (hipcc = clang has __builtin_complex in C++; g++ spells it with __real__ / __imag__)
        text = ("#define __builtin_complex(re, im) ({ float _Complex z_; __real__ z_ = (re); __imag__ z_ = (im); z_; })\n" + text)
    assert "asm(" not in text.replace('asm volatile("" ::: "memory")', "") or name in ("create.cpp", "chain.cpp", "scan_host.cpp", "feed.cpp", "state.cpp", "api.cpp", "plug.cpp"), name
    return text


def newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, sanitize=False):
    """sanitize: an AddressSanitizer build (tests/_build/libirdm_emul_asan.so; ucontext switches, which the sanitizer
    follows): out-of-bounds accesses of kernels to "device" memory, which a GPU does not report, stop the run.  Load it
    with LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0."""
    global SO, OUT
    if sanitize:
        SO = os.path.join(ROOT, "tests", "_build", "libirdm_emul_asan.so")
        OUT = os.path.join(ROOT, "tests", "_build", "emul_asan")
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(EMUL, "hip", "hip_runtime.h"),
                                                                os.path.join(EMUL, "rccl", "rccl.h"),
                                                                os.path.join(EMUL, "fir_mac.inc"), os.path.join(EMUL, "fft_bfly.inc"),
                                                                os.path.join(EMUL, "fir_fma.inc"),
                                                                os.path.abspath(__file__)]
    deps += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    if not force and os.path.exists(SO) and os.path.getmtime(SO) >= newest(deps):
        return SO
    os.makedirs(OUT, exist_ok=True)
    shutil.copy(os.path.join(EMUL, "fir_mac.inc"), os.path.join(OUT, "fir_mac.inc"))
    shutil.copy(os.path.join(EMUL, "fft_bfly.inc"), os.path.join(OUT, "fft_bfly.inc"))
    shutil.copy(os.path.join(EMUL, "fir_fma.inc"), os.path.join(OUT, "fir_fma.inc"))
    jobs = []
    for name in SOURCES:
        dst = os.path.join(OUT, name.replace(".hip", "_hip") .replace(".cpp", "_cpp") + ".cpp")
        open(dst, "w").write(transform(name, open(os.path.join(CSRC, name)).read()))
        obj = dst[:-4] + ".o"
        extra = ["-fsanitize=address", "-fno-omit-frame-pointer", "-DHIP_EMUL_UCONTEXT", "-g"] if sanitize else []
        jobs.append((obj, ["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-pthread", "-w"] + extra + ["-I" + EMUL, "-I" + OUT,
                           "-I" + CSRC, "-I" + os.path.join(ROOT, "include"), "-c", dst, "-o", obj]))
    with concurrent.futures.ThreadPoolExecutor(max_workers=6) as ex:
        for rc in ex.map(lambda j: subprocess.run(j[1], capture_output=True, text=True), jobs):
            if rc.returncode != 0:
                raise RuntimeError("emulated build failed:\n" + rc.stderr[-4000:])
    subprocess.check_call(["g++", "-shared", "-pthread"] + (["-fsanitize=address"] if sanitize else []) + ["-o", SO] + [j[0] for j in jobs])
    return SO


if __name__ == "__main__":
    import sys
    print(build(force=True, sanitize="asan" in sys.argv[1:]))
