"""sha256 over everything the device code is built from (particles_amd/csrc/*, include/smc_hip.h, the compiler flags):
what a committed profile record (profiles/traffic_<leg>.json, tools/summarise_prof.py) is stamped with, and what
bench.py / tests/test_bench.py compare it against -- a record of kernels whose source has changed since is not
reported as this tree's."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash():
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "particles_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".h", ".hip")))
    files.append(os.path.join(ROOT, "include", "smc_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    import sys
    sys.path.insert(0, ROOT)
    from particles_amd import _build
    h.update(" ".join(_build.FLAGS).encode())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
