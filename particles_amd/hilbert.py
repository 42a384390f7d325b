"""Counterpart of ``particles.hilbert.hilbert_sort`` (hilbert.py:33-58) for the case SQMC's
hot loop uses on the device: univariate particles, where the Hilbert order is the order of
the reals (hilbert.py:52-54: ``np.argsort(x, axis=0)``) -- a device-wide radix sort."""
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


def hilbert_sort(x):
    """Hilbert sort of N vectors (hilbert.py:33-58); (N,) or (N, 1) only."""
    d = 1 if x.ndim == 1 else x.shape[1]
    if d != 1:
        raise NotImplementedError("hilbert_sort on the device is built for d = 1 only "
                                  "(the Hilbert codec of hilbert.py:61-292 is not part of this path)")
    return argsort(x)
