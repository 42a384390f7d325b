"""Builds tests/emu/_build/libsmc_emu.so: the SAME kernel sources compiled with
g++ -DSMC_EMULATE against hip_emu.h.  Test infrastructure only."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "particles_amd", "csrc")
OUT = os.path.join(HERE, "_build", "libsmc_emu.so")
SOURCES = ["smc_api.hip", "smc_ops.hip", "smc_filter.hip", "smc_comm.hip", "smc_sort.hip", "smc_qmc.hip"]


def build(force=False):
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "hip_emu.h")]
    if (not force and os.path.exists(OUT)
            and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps)):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-DSMC_EMULATE",
           "-ffp-contract=off", "-I", HERE, "-Wall", "-Wno-unused-function",
           "-Wno-unused-variable", "-Wno-unknown-pragmas"]
    for s in SOURCES:
        cmd += ["-x", "c++", os.path.join(CSRC, s)]
    cmd += ["-o", OUT, "-ldl"]
    subprocess.run(cmd, check=True)
    return OUT


def build_fake_rccl(force=False):
    """tests/emu/_build/libfake_rccl.so: the RCCL test double (fake_rccl.c) the emulator build binds when
    SMC_RCCL_LIBRARY points at it."""
    src = os.path.join(HERE, "fake_rccl.c")
    out = os.path.join(HERE, "_build", "libfake_rccl.so")
    if force or not os.path.exists(out) or os.path.getmtime(src) > os.path.getmtime(out):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        tmp = out + ".tmp%d" % os.getpid()
        subprocess.run(["gcc", "-O1", "-g", "-std=gnu11", "-fPIC", "-shared", "-Wall", src, "-o", tmp], check=True)
        os.replace(tmp, out)
    return out


if __name__ == "__main__":
    print(build(force=True))
