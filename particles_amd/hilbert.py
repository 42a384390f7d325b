"""Counterpart of ``particles.hilbert`` (hilbert.py) for SQMC's hot loop: ``hilbert_sort``
(:33-58) and ``hilbert_array`` (:13-30) on the device.  For univariate particles the Hilbert
order is the order of the reals (:52-54: ``np.argsort(x, axis=0)``) -- a device-wide radix
sort; for (N, d) particles the standardise / logistic / integer-grid / Hilbert-index pipeline
runs in HIP kernels (Witham's codec restated with integer operations) before the same sort."""
import numpy as np

from . import _lib
from ._lib import DeviceArray, as_device, check, lib


def argsort(x):
    """``np.argsort(x)`` of a (N,) host or device array (int64; device in, device out)."""
    xd, host = as_device(x)
    if xd.dtype != np.float64:
        raise TypeError("argsort takes a float64 array")
    out = DeviceArray((xd.size,), np.int64, xd.ctx)
    check(lib().smc_argsort(xd.ctx.h, xd.ptr, xd.size, out.ptr))
    return out.get() if (host and not _lib.RESIDENT[0]) else out


def hilbert_array(xint):
    """Hilbert indices of N points with non-negative integer coordinates (hilbert.py:13-30)."""
    xd, host = as_device(np.ascontiguousarray(xint, dtype=np.int64) if not isinstance(xint, DeviceArray) else xint)
    N, d = xd.shape
    out = DeviceArray((N,), np.int64, xd.ctx)
    check(lib().smc_hilbert_array(xd.ctx.h, xd.ptr, N, d, out.ptr))
    return out.get() if (host and not _lib.RESIDENT[0]) else out


def hilbert_sort(x):
    """Hilbert sort of N vectors, (N,) or (N, d) (hilbert.py:33-58): argsort of the vectors'
    Hilbert indices after standardisation and a logistic map to [0,1]^d."""
    d = 1 if x.ndim == 1 else x.shape[1]
    if d == 1:
        return argsort(x)
    xd, host = as_device(x)
    N = xd.shape[0]
    out = DeviceArray((N,), np.int64, xd.ctx)
    check(lib().smc_hilbert_sort(xd.ctx.h, xd.ptr, N, d, out.ptr, None))
    return out.get() if (host and not _lib.RESIDENT[0]) else out
