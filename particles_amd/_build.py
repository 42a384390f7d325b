"""Builds libsmc_hip.so (gfx950) in-tree with hipcc.  No fallback: if hipcc or
the sources are missing this raises."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBPATH = os.path.join(LIBDIR, "libsmc_hip.so")
SOURCES = ["smc_api.hip", "smc_ops.hip", "smc_filter.hip", "smc_comm.hip", "smc_sort.hip", "smc_qmc.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [
    os.path.join("..", "..", "include", "smc_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
         # MFMA results straight into VGPRs (they are the next product's B operand):
         # no v_accvgpr_read/write copies
         "-mllvm", "-amdgpu-mfma-vgpr-form=1",
         # leading scalar kernel arguments preloaded into SGPRs by the command processor (kernels that take
         # only the argument block are unaffected)
         "-mllvm", "-amdgpu-kernarg-preload-count=14"]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libsmc_hip.so cannot be built")


def is_stale():
    if not os.path.exists(LIBPATH):
        return True
    t = os.path.getmtime(LIBPATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIBPATH
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [_hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIBPATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd + ["-ldl"], check=True)
    return LIBPATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
