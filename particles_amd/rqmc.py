"""Counterpart of ``particles.rqmc`` (rqmc.py:1-22): randomised quasi-Monte Carlo points
for SQMC (``SMC(qmc=True)``, core.py:315-349).

'numpy' RNG mode: the reference's own expression -- ``scipy.stats.qmc.Sobol(d).random(N)``
behind ``safe_generate`` -- on the host (like the reference, such a run is not reproducible
from ``np.random.seed``: the engine seeds itself).  'philox' mode: the points are generated
on the device (``smc_sobol``: Joe-Kuo direction numbers, the same unscrambled sequence as
scipy's, randomised by a digital shift from the counted Philox stream) and stay there.
"""
import numpy as np

from . import _lib
from ._lib import DeviceArray, check, lib

TOL = 1e-10


def safe_generate(N, d, engine_cls):
    eng = engine_cls(d)
    u = eng.random(N)
    return 0.5 + (1.0 - TOL) * (u - 0.5)                      # rqmc.py:9-13


def sobol(N, d):
    """(N, d) scrambled Sobol' points in (0, 1) (rqmc.py:15-16)."""
    if _lib.RNG_MODE[0] == "numpy":
        from scipy.stats import qmc
        u = safe_generate(N, d, qmc.Sobol)
        return DeviceArray.from_numpy(u) if _lib.RESIDENT[0] else u
    out = DeviceArray((N, d))
    check(lib().smc_sobol(out.ctx.h, N, d, 1, 1, _lib.next_counter(), out.ptr))
    return out


def sobol_sorted(N, d):
    """``u = sobol(N, d); u[np.argsort(u[:, 0])]`` -- the only way SQMC reads its points
    (core.py:343-347) -- or None when only the generic route exists.  With device-generated points
    and N a power of two the sorted order is known in closed form (`smc_sobol_sorted`): no sort, no
    gather."""
    if _lib.RNG_MODE[0] == "numpy" or N & (N - 1):
        return None
    out = DeviceArray((N, d))
    check(lib().smc_sobol_sorted(out.ctx.h, N, d, 1, 1, _lib.next_counter(), out.ptr))
    return out


def sobol_unscrambled(N, d):
    """The first N points of the plain Sobol' sequence, from the device (tests: equals
    ``scipy.stats.qmc.Sobol(d, scramble=False).random(N)``)."""
    out = DeviceArray((N, d))
    check(lib().smc_sobol(out.ctx.h, N, d, 0, 0, 0, out.ptr))
    return out.get()
