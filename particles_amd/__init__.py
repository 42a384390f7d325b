"""particles_amd -- an MI355X-native SMC inner loop behind the API of
nchopin/particles (``SMC`` / ``FeynmanKac`` / ``resampling`` / ``distributions``).

All arithmetic runs in hand-written HIP kernels (libsmc_hip.so, C ABI in
include/smc_hip.h); this package is the thin Python host layer.  There is no
CPU fallback: without the built library and a visible GPU the operators raise.
"""
from ._lib import DeviceArray, seed  # noqa: F401
from .core import SMC, FeynmanKac, multiSMC  # noqa: F401

__version__ = "0.1.0"
