"""Stand-in for numba, used ONLY to import the reference in the build container.

numba is not installed and there is no network; the reference does
``from numba import jit`` (particles/resampling.py:134, hilbert.py:6).  With
this stub the jitted functions run as plain Python, which is slower but
computes exactly the same thing.  Never shipped with, nor imported by, the
product.
"""


def jit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda f: f


njit = jit
