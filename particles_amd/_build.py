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


def build(force=False, verbose=False, extra=(), out=None):
    """One hipcc -c per source, side by side (the translation units share no device symbols), then one link.
    extra: more compiler flags (ablation builds: -DSMC_...); out: another library path."""
    out = out or LIBPATH
    if not force and not extra and out == LIBPATH and not is_stale():
        return LIBPATH
    os.makedirs(os.path.dirname(out), exist_ok=True)
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    cflags = [f for f in FLAGS if f != "-shared"] + list(extra)
    with tempfile.TemporaryDirectory(prefix="smc_build_") as tmp:
        def one(src):
            obj = os.path.join(tmp, src.replace(".hip", ".o"))
            cmd = [_hipcc()] + cflags + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            return obj
        with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
            objs = list(ex.map(one, SOURCES))
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out, "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
