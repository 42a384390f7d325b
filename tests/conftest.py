import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

# Nothing this suite starts may write into the read-only reference tree: test_adapter.py imports the REAL package from
# /root/reference and drives its multiSMC through loky worker processes, which do not inherit sys.dont_write_bytecode --
# they do inherit the environment (round 5 left a __pycache__/ there).
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
REAL_LIB = os.path.join(ROOT, "particles_amd", "lib", "libsmc_hip.so")


def _gpu_visible():
    """True iff the real libsmc_hip.so loads and sees a HIP device."""
    if not os.path.exists(REAL_LIB):
        return False
    try:
        n = ctypes.c_int(0)
        ctypes.CDLL(REAL_LIB).smc_device_count(ctypes.byref(n))
        return n.value > 0
    except OSError:
        return False


HAS_GPU = _gpu_visible()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")
    if not HAS_GPU:
        # GPU-less container: route the CPU suite's kernel-logic tests through
        # the fiber emulator build of the SAME kernel sources (tests/emu).
        sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
        import build_emu
        os.environ["SMC_HIP_LIBRARY"] = build_emu.build()
        os.environ["SMC_TEST_EMULATOR"] = "1"        # particles_amd refuses non-gfx950 builds otherwise


def pytest_xdist_auto_num_workers(config):
    """`-n auto` (pytest.ini): up to 8 worker processes for the emulator suite, none on a GPU box."""
    if HAS_GPU or os.environ.get("SMC_TEST_SERIAL") == "1":
        return 0
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(0, min(8, n)) if n > 1 else 0


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def has_gpu():
    return HAS_GPU


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The C half of the oracle is a build product (oracle/_build); make it."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True,
                   stdout=subprocess.DEVNULL)


def pytest_terminal_summary(terminalreporter):
    """The near-tie counts every audit observed (tests/parity_cases.log_near_ties), totalled: the log
    of a run records how often the exact integer CDF and the reference's sequential fp64 CDF chose a
    different -- certified near-tie -- ancestor."""
    try:
        import parity_cases as pc
    except Exception:
        return
    if not pc.NEAR_TIE_LOG:
        return
    tr = terminalreporter
    tr.write_sep("=", "near-ties against the reference's sequential CDF")
    tot_t = tot_d = 0
    for where, ties, draws in pc.NEAR_TIE_LOG:
        tot_t += ties
        tot_d += draws
        if ties or draws >= 1 << 20:
            tr.write_line("  %-52s %6d of %13d" % (where, ties, draws))
    tr.write_line("  TOTAL %d near-ties in %d audited ancestors (%.2e per ancestor)" % (tot_t, tot_d, tot_t / max(1, tot_d)))
    strict = [(w, t, d) for w, t, d in pc.NEAR_TIE_LOG if w.startswith("STRICT")]
    tr.write_line("  strict_ancestors=True (the reference's sequential fp64 CDF itself): %d differences in %d audited ancestors "
                  "(must be 0)" % (sum(t for _, t, _ in strict), sum(d for _, _, d in strict)))


def pytest_sessionfinish(session, exitstatus):
    """The reference tree is read-only by contract: no process of this suite may have left bytecode there."""
    ref = "/root/reference"
    if os.path.isdir(ref):
        left = [os.path.join(d, x) for d, sub, _ in os.walk(ref) for x in sub if x == "__pycache__"]
        if left:
            session.exitstatus = 1
            print("\nERROR: the suite wrote into the reference tree: %s" % left[:3])
