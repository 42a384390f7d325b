"""particles_amd -- an MI355X-native SMC inner loop behind the API of
nchopin/particles (``SMC`` / ``FeynmanKac`` / ``resampling`` / ``distributions``).

All arithmetic runs in hand-written HIP kernels (libsmc_hip.so, C ABI in
include/smc_hip.h); this package is the thin Python host layer.  There is no
CPU fallback: without the built library and a visible GPU the operators raise.
"""
from ._lib import DeviceArray, seed  # noqa: F401
from .core import SMC, FeynmanKac, multiSMC  # noqa: F401
from . import hilbert, rqmc  # noqa: F401


def set_resident(flag=True):
    """Host-facing operators (``distributions``, ``resampling``) return ``DeviceArray``s
    instead of numpy arrays: a user-defined model -- ``M0 / M / logG`` or ``PX0 / PX / PY``
    written with numpy expressions -- then runs its whole step on arrays that stay in HBM
    (``DeviceArray`` supports the arithmetic and the numpy ufuncs; ``.get()`` or
    ``numpy.asarray`` brings a result to the host)."""
    from . import _lib
    _lib.RESIDENT[0] = bool(flag)

__version__ = "0.1.0"
