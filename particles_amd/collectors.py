"""Side-cars of the SMC step that must keep working on the device path:
``Summaries`` / ``Collector`` / default collectors ``ESSs, LogLts, Rs_flags``
and ``Moments`` (particles/collectors.py:215-317), and the history containers
the step saves into (particles/smoothing.py:164-255).
"""
from collections import deque

import numpy as np

from . import _lib


class Summaries:
    """Stores and updates summaries (collectors.py:215-231)."""

    def __init__(self, cols):
        self._collectors = [cls() for cls in default_collector_cls]
        if cols is not None:
            self._collectors.extend(col() for col in cols)
        for col in self._collectors:
            setattr(self, col.summary_name, col.summary)

    def collect(self, smc):
        for col in self._collectors:
            col.collect(smc)

    def _extend_defaults(self, ess, logLt, flags):
        """Bulk append of the three default summaries after a device-resident
        run (the values were kept per step in the device ring buffer)."""
        self.ESSs.extend(float(v) for v in ess)
        self.logLts.extend(float(v) for v in logLt)
        self.rs_flags.extend(bool(v) for v in flags)


    def _extend_moments(self, moms):
        """Bulk append for ``Moments`` collectors evaluated on the device."""
        for col in self._collectors:
            if isinstance(col, Moments):
                col.summary.extend(moms)


class Collector:
    """Base class for collectors (collectors.py:234-271)."""

    signature = {}

    @property
    def summary_name(self):
        cn = self.__class__.__name__
        return cn[0].lower() + cn[1:]

    def __init__(self, **kwargs):
        self.summary = []
        for k, v in self.signature.items():
            setattr(self, k, v)
        for k, v in kwargs.items():
            if k in self.signature.keys():
                setattr(self, k, v)
            else:
                raise ValueError(f"Collector {self.__class__.__name__}: unknown parameter {k}")

    def __call__(self):
        return self.__class__(**{k: getattr(self, k) for k in self.signature.keys()})

    def collect(self, smc):
        self.summary.append(self.fetch(smc))


class ESSs(Collector):
    summary_name = "ESSs"

    def fetch(self, smc):
        return smc.wgts.ESS


class LogLts(Collector):
    def fetch(self, smc):
        return smc.logLt


class Rs_flags(Collector):
    def fetch(self, smc):
        return smc.rs_flag


default_collector_cls = [ESSs, LogLts, Rs_flags]


class Moments(Collector):
    """Empirical moments of the particles (collectors.py:301-317); the default
    is the weighted mean and variance, evaluated on the device."""

    signature = {"mom_func": None}

    def fetch(self, smc):
        f = smc.fk.default_moments if self.mom_func is None else self.mom_func
        return f(smc.W, smc.X)


# ---- history containers (smoothing.py:141-255) -----------------------------

def generate_hist_obj(option, smc):
    if option is True:
        return ParticleHistory(smc.fk, smc.qmc)
    if option is False:
        return None
    if callable(option):
        return PartialParticleHistory(option)
    if isinstance(option, int) and option >= 0:
        return RollingParticleHistory(option)
    raise ValueError("store_history: invalid option")


class PartialParticleHistory:
    def __init__(self, func):
        self.is_save_time = func
        self.X, self.wgts = {}, {}

    def save(self, smc):
        t = smc.t
        if self.is_save_time(t):
            self.X[t] = smc.X
            self.wgts[t] = _frozen_weights(smc.wgts)


class RollingParticleHistory:
    def __init__(self, length):
        self.X = deque([], length)
        self.A = deque([], length)
        self.wgts = deque([], length)

    @property
    def N(self):
        return self.X[0].shape[0]

    @property
    def T(self):
        return len(self.X)

    def save(self, smc):
        self.X.append(smc.X)
        self.A.append(smc.A)
        self.wgts.append(_frozen_weights(smc.wgts))

    def compute_trajectories(self):
        """(T, N) genealogy (smoothing.py:209-219)."""
        Bs = [np.arange(self.N)]
        for A in list(self.A)[-1:0:-1]:
            Bs.append(A[Bs[-1]])
        Bs.reverse()
        return np.array(Bs)


class ParticleHistory(RollingParticleHistory):
    def __init__(self, fk, qmc):
        self.X, self.A, self.wgts = [], [], []
        self.fk = fk


class _LazySteps:
    """Sequence over the steps run so far whose items are fetched from the device
    history on access."""

    def __init__(self, smc, fetch):
        self._smc, self._fetch = smc, fetch

    def __len__(self):
        return self._smc._n

    def __getitem__(self, t):
        n = len(self)
        if isinstance(t, slice):
            return [self[i] for i in range(*t.indices(n))]
        if t < 0:
            t += n
        if not 0 <= t < n:
            raise IndexError("history index out of range")
        return self._fetch(t)

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class DeviceParticleHistory:
    """``ParticleHistory`` (smoothing.py:222-255: ``X``, ``A``, ``wgts`` of every
    step, ``compute_trajectories``) of a fused run: the history never leaves HBM
    until an item is read -- the device step loop writes step t into slot t of
    (T, N[, d]) arrays (``keep_history``), so ``save`` has nothing to do and
    ``SMC.run`` keeps its single asynchronous launch sequence."""

    def __init__(self, smc):
        self._smc = smc
        self.fk = smc.fk
        # (bound methods, not lambdas: the history object pickles with its filter)
        self.X = _LazySteps(smc, self._X_at)
        # hist.A[0] is the A of a filter that has not resampled yet: None (core.py:229)
        self.A = _LazySteps(smc, self._A_at)
        self.wgts = _LazySteps(smc, self._wgts_at)

    def _X_at(self, t):
        return self._smc._history(_lib.FIELD_X, t)

    def _A_at(self, t):
        return self._smc._history(_lib.FIELD_A, t) if t else None

    def _wgts_at(self, t):
        smc = self._smc
        f = _Frozen()
        f.lw = smc._history(_lib.FIELD_LW, t)
        f.W = smc._history(_lib.FIELD_W, t)
        s = smc._summ()[0, t]
        f.ESS, f.log_mean = float(s[0]), float(s[1])
        f.N = smc.N
        return f

    @property
    def N(self):
        return self._smc.N

    @property
    def T(self):
        return self._smc._n

    def save(self, smc):            # the device already did
        pass

    def compute_trajectories(self):
        """(T, N) genealogy (smoothing.py:209-219), computed on the device."""
        return self._smc._trajectories()

    def extract_one_trajectory(self):
        """A single trajectory (smoothing.py:256-269): the final state is drawn from the final
        weights, its line of ancestors is followed back on the device."""
        from . import resampling as rs
        smc = self._smc
        n = rs.multinomial_once(smc._history(_lib.FIELD_W, smc._n - 1))
        d = getattr(smc, "_d", 1)
        out = np.empty((smc._n, d))
        _lib.check(_lib.lib().smc_filter_one_trajectory(smc._f, 0, n, out.ctypes.data_as(
            _lib.P(_lib.c_dbl))))
        return [row[0] if d == 1 else row.copy() for row in out]


class DeviceRollingParticleHistory:
    """``RollingParticleHistory`` (smoothing.py:186-219) of a fused run: the k most recent
    particle systems stay in a ring of k slots in HBM (``keep_history = k``), the step loop writes
    them there, ``save`` has nothing to do.  ``X``, ``A``, ``wgts`` behave like the reference's
    deques: index 0 is the oldest resident step, -1 the newest."""

    def __init__(self, smc, length):
        self._smc, self.length = smc, int(length)
        self.fk = smc.fk
        self.X = _Window(self, self._X_at)
        self.A = _Window(self, self._A_at)
        self.wgts = _Window(self, self._wgts_at)

    def _X_at(self, t):
        return self._smc._history(_lib.FIELD_X, t)

    def _wgts_at(self, t):
        return DeviceParticleHistory._wgts_at(self, t)

    def _A_at(self, t):
        smc = self._smc
        if t == 0:
            return None                                    # core.py:229: no ancestors at t = 0
        if t - 1 < smc._n - self.length:                   # its parents' step has left the window:
            s = smc._summ()[0, t]                          # the indices themselves are still there
            if not s[4]:
                return np.arange(smc.N)
        return smc._history(_lib.FIELD_A, t)

    @property
    def N(self):
        return self._smc.N

    @property
    def T(self):
        return min(self._smc._n, self.length)

    def save(self, smc):
        pass

    def compute_trajectories(self):
        """(T, N) genealogy of the resident steps (smoothing.py:209-219), on the device."""
        out = np.empty((self.T, self.N), dtype=np.int64)
        _lib.check(_lib.lib().smc_filter_trajectories(self._smc._f, 0, out.ctypes.data_as(_lib.P(_lib.c_i64))))
        return out


class _Window:
    """deque-like view of the resident steps of a rolling device history."""

    def __init__(self, hist, fetch):
        self._h, self._fetch = hist, fetch

    def __len__(self):
        return self._h.T

    def __getitem__(self, i):
        n = len(self)
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(n))]
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("history index out of range")
        return self._fetch(self._h._smc._n - n + i)

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class _Frozen:
    pass


def _frozen_weights(w):
    """Copy-on-save: the device double-buffers its arrays, so a saved history
    entry must own host copies (the reference stores references,
    smoothing.py:204-207)."""
    f = _Frozen()
    f.lw, f.W, f.ESS, f.log_mean = w.lw, w.W, w.ESS, w.log_mean
    f.N = w.N
    return f
