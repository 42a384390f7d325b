"""CPU oracle for the SMC hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (NumPy + a small C library) of the
algorithm that nchopin/particles runs inside ``SMC.__next__`` and
``resampling.py``.  It exists so that the HIP path can be checked; it is
never imported by ``particles_amd`` (the product).  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.

Parity status: the reference ships no golden vectors for this path
(SURVEY.md section 8c), so the oracle is pinned against outputs of the
reference itself, generated in the build container by
``tests/golden/make_golden.py`` (reference imported from /root/reference with
the ``oracle/numba_shim`` stub) and committed under ``tests/golden/``.
"""
