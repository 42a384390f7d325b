import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch.distributed as dist
import particles_amd as pa
from particles_amd import _lib
_lib.ctx()
import ctypes
libs = set()
for l in open("/proc/self/maps"):
    p = l.split()[-1]
    if any(k in p for k in ("amdhip64", "rccl", "libsmc", "hsa-runtime", "rocblas")):
        libs.add(p)
print("\n".join(sorted(libs)))
