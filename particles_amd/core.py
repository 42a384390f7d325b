"""Counterpart of ``particles.core``: ``FeynmanKac`` (core.py:108-197), ``SMC``
(:200-409) and ``multiSMC`` (:431-518), with the per-time-step particle loop
executed on an MI355X.

``SMC`` keeps the reference's constructor, iterator protocol and public
attributes (``X, Xp, A, wgts, aux, W, t, rs_flag, logLt, loglt, summaries,
hist, cpu_time``).  Two execution paths, both driving HIP kernels:

* fused  -- the Feynman-Kac model is a ``Bootstrap`` / ``GuidedPF`` of a model
  from the closed family (``kalman.LinearGauss``, ``StochVol``): the whole time
  loop (resample decision, resampling, propagation, weighting, evidence) runs
  on the device through ``smc_filter_*``; attributes are fetched lazily.
* generic -- any other ``FeynmanKac`` (user ``M0`` / ``M`` / ``logG`` in Python):
  the template method of core.py:369-383 is followed literally, with
  ``Weights`` and ``resampling`` served by the device operators.
"""
import ctypes
import time

import numpy as np

from . import _lib
from . import collectors
from . import hilbert
from . import resampling as rs
from . import rqmc
from ._lib import DeviceArray, check, lib


class FeynmanKac:
    """Abstract base class for Feynman-Kac models (core.py:108-197)."""

    def __init__(self, T):
        self.T = T

    def _error_msg(self, meth):
        return f"method/property {meth} missing in class {self.__class__.__name__}"

    def M0(self, N):
        raise NotImplementedError(self._error_msg("M0"))

    def M(self, t, xp):
        raise NotImplementedError(self._error_msg("M"))

    def logG(self, t, xp, x):
        raise NotImplementedError(self._error_msg("logG"))

    def Gamma0(self, u):
        """Deterministic map of u ~ U([0,1]^d) to X_0 (SQMC, core.py:157-160)."""
        raise NotImplementedError(self._error_msg("Gamma0"))

    def Gamma(self, t, xp, u):
        """Deterministic map of (xp, u) to X_t ~ M_t(xp, dx) (SQMC, core.py:162-166)."""
        raise NotImplementedError(self._error_msg("Gamma"))

    @property
    def isAPF(self):
        return hasattr(self, "logeta")             # (core.py:168-170: "logeta" in dir(self); asked twice per step)

    def done(self, smc):
        """Time to stop the algorithm (core.py:177-179)."""
        return smc.t >= self.T

    def time_to_resample(self, smc):
        """When to resample (core.py:181-183)."""
        return smc.aux.ESS < smc.N * smc.ESSrmin

    def default_moments(self, W, X):
        return rs.wmean_and_var(W, X)

    def summary_format(self, smc):
        return "t=%i: resample:%s, ESS (end of iter)=%.2f" % (smc.t, smc.rs_flag, smc.wgts.ESS)

    def _device_model(self):
        return None


class _DeviceWeights:
    """``rs.Weights`` view of one island of a fused filter (lazy downloads)."""

    def __init__(self, smc, island=0):
        self._smc, self._isl = smc, island

    @property
    def lw(self):
        return self._smc._get(_lib.FIELD_LW, self._isl)

    @property
    def W(self):
        return self._smc._get(_lib.FIELD_W, self._isl)

    @property
    def ESS(self):
        return self._smc._summ()[self._isl, -1, 0]

    @property
    def log_mean(self):
        return self._smc._summ()[self._isl, -1, 1]

    @property
    def N(self):
        return self._smc.N


_seed_counter = [0]


def _default_seed():
    """Philox key of a run created without ``seed``: a function of numpy's legacy generator
    state (so ``numpy.random.seed`` makes runs repeatable) that does NOT advance it -- the
    reference's ``SMC.__init__`` draws nothing, and a run replaying the reference's draws
    must find the stream where the reference finds it."""
    import zlib
    st = np.random.get_state()
    _seed_counter[0] += 1
    return (zlib.crc32(st[1].tobytes()) ^ (int(st[2]) * 2654435761) ^ (_seed_counter[0] << 20)) & (2 ** 31 - 1)


class SMC:
    """Particle filter / SMC algorithm (core.py:200-409), device-resident.

    Parameters are those of ``particles.SMC`` (core.py:258-268); extras:

    seed : int, optional -- Philox key of this run (default: derived from the state of
        ``numpy.random`` without advancing it)
    n_islands : int -- number of independent replicas advanced together
        (the batched form of ``multiSMC``); attributes refer to island 0,
        ``logLts_islands`` has them all
    replay : (z, u) device/host tapes of the reference's own draws, see
        ``smc_filter_set_replay`` (parity tests)
    collapsed_proposal : GuidedPF of an MVLinearGauss only -- weigh with the collapsed form of the
        optimal proposal's weight, log p(y_t | x_{t-1}) (SMC_FLAG_COLLAPSED_PROPOSAL): same
        particles, log-weights equal up to rounding, 40 % fewer matrix instructions
    strict_ancestors : resample with the reference's own sequential fp64 CDF of the filter's weights: ``A`` equals
        ``particles.resampling.inverse_cdf(su, W)`` bit for bit.  Fused univariate Bootstrap / Guided filters:
        SMC_FLAG_STRICT_ANCESTORS (two launches, verified per step); every other filter -- the
        template-method step, SQMC included (qmc=True then runs on device operators) -- wraps its resampling calls in
        ``rs.strict_mode`` (``smc_inverse_cdf_strict``)
    use_graph : replay the step sequence from hipGraphs instead of launching the kernels one by
        one (off by default: on MI355X a dependent kernel boundary costs the same either way,
        eager launches measured 2 % faster at C2 and start sooner after an idle stream)
    """

    def __init__(self, fk=None, N=100, qmc=False, resampling="systematic", ESSrmin=0.5,
                 store_history=False, verbose=False, collect=None, seed=None, n_islands=1,
                 replay=None, use_graph=False, island_offset=0, collapsed_proposal=False, strict_ancestors=False):
        self._fk_list = None
        if isinstance(fk, (list, tuple)):        # one Feynman-Kac model per island (SMC^2: one theta each)
            self._fk_list = list(fk)
            fk = self._fk_list[0]
            n_islands = len(self._fk_list)
        if resampling not in rs.rs_funcs:
            raise ValueError(f"{resampling} is not a valid resampling scheme")
        self.fk, self.N, self.qmc = fk, N, qmc
        self.resampling, self.ESSrmin, self.verbose = resampling, ESSrmin, verbose
        self.n_islands = n_islands
        self.t = 0
        self.rs_flag = False
        self._logLt = 0.0
        self.cpu_time = 0.0
        self._collapsed = bool(collapsed_proposal)
        self._strict = bool(strict_ancestors)
        if collect == "off":
            self.summaries = None
        else:
            self.summaries = collectors.Summaries(collect)
        self._user_collectors = bool(collect) and collect != "off"
        # collect=[Moments()] with the default moments: evaluated on the device after every
        # step (opts.moments), no per-step host work
        self._device_moments = (
            self._user_collectors and all(isinstance(c, collectors.Moments) and c.mom_func is None
                                          for c in collect)
            and type(fk).default_moments is FeynmanKac.default_moments)
        self.hist = collectors.generate_hist_obj(store_history, self)
        self._store_history = store_history
        if seed is None:
            seed = _default_seed()
        self.seed = seed
        self._f = None
        self._n = 0          # steps executed on the device
        self._cache = {}
        self._summ_cache = None
        model = fk._device_model() if hasattr(fk, "_device_model") else None
        if fk is not None and fk.isAPF and not self._apf_fusable(
                fk, N, resampling, replay, store_history,
                bool(collect and collect != "off" and self._device_moments)):
            model = None                       # the operator path
        if qmc and not self._sqmc_fusable(fk, N, model, replay, use_graph, strict_ancestors):
            model = None           # SQMC as the template-method step on device operators
        if qmc and model is not None and (model["kind"] == _lib.MODEL_MVLINGAUSS or N < 2048) and (
                store_history or collapsed_proposal or (collect and collect != "off" and self._device_moments)):
            model = None           # (the fused SQMC on the flat step keeps no history slots / device moments)
        self._fused = self._will_fuse(fk, qmc, resampling, model, N=N)
        self._ctor = (use_graph, island_offset, replay is not None)       # (what __setstate__ re-creates the filter with)
        if self._fused:
            # full history on the fused path stays on the device: the step loop
            # writes step t into slot t (no per-step host copies, no per-step sync)
            rolling = (isinstance(store_history, int) and not isinstance(store_history, bool)
                       and store_history >= 2)
            self._device_hist = store_history is True or rolling
            self._keep_history = int(store_history) if rolling else (1 if store_history is True else 0)
            self._create_filter(model, replay, use_graph, island_offset)
            if store_history is True:
                self.hist = collectors.DeviceParticleHistory(self)
            elif rolling:       # RollingParticleHistory on the device: a ring of k slots
                self.hist = collectors.DeviceRollingParticleHistory(self, store_history)
        else:
            if n_islands != 1:
                raise ValueError("n_islands > 1 needs a model of the fused family")
            self._wgts = rs.Weights()
            self.aux = None
            self._X = self._Xp = self._A = None

    @staticmethod
    def _apf_fusable(fk, N, resampling, replay, store_history, device_moments):
        """Fused APF: the one-launch filter (N <= 1024; no Philox multinomial there) or the two-level
        step (N > 1024; no rolling window); stock StochVol only, no device-side moments."""
        stock = getattr(fk, "_fk_kind", None) in (_lib.FK_APF, _lib.FK_APF_BOOT)
        model = fk._device_model() if hasattr(fk, "_device_model") else None
        if stock and model is not None and model["kind"] == _lib.MODEL_MVLINGAUSS:
            # MVLinearGauss: k_mv_aux in front of the flat step, any N (history slots: k_propagate_mv puts the plain weights
            # back into the previous step's slot once the resampling kernels have used the auxiliary ones)
            return True
        # (the two-level step keeps the plain log-weights in the state -- the auxiliary ones only drive the tile partials and
        #  the integer CDF -- so history slots, a rolling window and the device-side Moments see what the reference's do)
        small = N <= 1024 and not (resampling == "multinomial" and replay is None) and not device_moments
        two_level = 1024 < N <= (1 << 30)
        return stock and (small or two_level)

    @staticmethod
    def _sqmc_fusable(fk, N, model, replay=None, use_graph=False, strict=False):
        """SQMC as a fused loop (SMC_FLAG_SQMC): Bootstrap / Guided filters of the fused family, N = 2^k >= 32,
        device-generated points (N >= 2048, univariate: the two-level step; else the flat step, eager launches)."""
        if not (_lib.FUSED_SQMC[0] and _lib.RNG_MODE[0] == "philox" and model is not None
                and getattr(fk, "_fk_kind", None) in (_lib.FK_BOOTSTRAP, _lib.FK_GUIDED)
                and N <= (1 << 30) and N & (N - 1) == 0 and replay is None and not strict):
            return False
        if model["kind"] == _lib.MODEL_MVLINGAUSS:      # the flat step behind the Hilbert sort, d + 1 <= 10 Sobol' coordinates
            return 2 <= model["dx"] <= 9 and N >= 32 and not use_graph
        return (model.get("params") is not None and model.get("dx", 1) == 1 and N >= 32
                and not (use_graph and N < 2048))

    @staticmethod
    def _will_fuse(fk, qmc=False, resampling="systematic", model=False, N=None):
        """Does SMC(fk, qmc, resampling) run the fused device loop?  (One predicate for ``SMC``
        and for ``multiSMC``'s decision to batch runs as islands.)"""
        if fk is None or resampling not in _lib.SCHEMES:
            return False
        if fk.isAPF and getattr(fk, "_fk_kind", None) not in (_lib.FK_APF, _lib.FK_APF_BOOT):
            return False
        if model is False:
            model = fk._device_model() if hasattr(fk, "_device_model") else None
        if qmc and (N is None or not SMC._sqmc_fusable(fk, N, model)):
            return False
        return model is not None and (model.get("params") is not None
                                      or model["kind"] == _lib.MODEL_MVLINGAUSS)

    # ------------------------------------------------------------------ fused
    def _create_filter(self, model, replay, use_graph, island_offset, restore=None):
        """restore (unpickling): {"path_flags", "sqmc"} of the filter the state was saved from -- the verification
        switches it was created under (the environment of the RECEIVING process may differ: the slab's layout must not)
        and its point-set key and counter (the process-wide counter of the receiving process is left alone)."""
        fk = self.fk
        T = fk.T
        y = np.ascontiguousarray(np.asarray(fk.data, dtype=np.float64).reshape(T, -1))
        m = _lib.SmcModel()
        m.kind, m.fk, m.dx, m.dy = model["kind"], fk._fk_kind, model["dx"], model["dy"]
        self._d = model["dx"]
        if model.get("params") is not None:
            if self._fk_list is None:
                params = np.ascontiguousarray(np.tile(model["params"], (self.n_islands, 1)))
            else:
                rows = []
                for g in self._fk_list:
                    mg = g._device_model()
                    if mg is None or mg["kind"] != model["kind"] or g._fk_kind != fk._fk_kind or g.T != T \
                            or not np.array_equal(np.asarray(g.data, dtype=np.float64).reshape(T, -1), y):
                        raise ValueError("per-island models must be of one fused kind, with the same data")
                    # the per-step scalar of the transition / observation law rides in ONE (T,) array
                    # shared by all islands: the parameters it depends on must not vary
                    for key in ("aux", "aux_from_data"):
                        if model.get(key) is not None and not np.array_equal(
                                np.asarray(mg[key](T if key == "aux" else y)),
                                np.asarray(model[key](T if key == "aux" else y))):
                            raise ValueError("per-island models: the per-step term of this model depends on a "
                                             "parameter that differs between islands (not supported)")
                    rows.append(mg["params"])
                params = np.ascontiguousarray(np.stack(rows))
            m.params_host = params.ctypes.data_as(_lib.P(_lib.c_dbl))
            aux = None
            if model.get("aux") is not None:        # per-step term of the transition mean
                aux = np.ascontiguousarray(model["aux"](T), dtype=np.float64)
            elif model.get("aux_from_data") is not None:      # per-step term of log p(y_t | x_t)
                aux = np.ascontiguousarray(model["aux_from_data"](y), dtype=np.float64)
            if aux is not None:
                m.aux_host = aux.ctypes.data_as(_lib.P(_lib.c_dbl))
            self._keep = (y, params, aux)
        else:       # MVLinearGauss: the matrices, row-major fp64 (kalman.py:296-361)
            if self._fk_list is not None:
                raise ValueError("per-island models are not supported for MVLinearGauss (one set of "
                                 "matrices per filter)")
            mats = {k: np.ascontiguousarray(model[k], dtype=np.float64)
                    for k in ("F", "G", "covX", "covY", "mu0", "cov0")}
            for k, v in mats.items():
                setattr(m, k + "_host", v.ctypes.data_as(_lib.P(_lib.c_dbl)))
            self._keep = (y, mats)
        o = _lib.SmcFilterOpts()
        o.N, o.T, o.n_islands = self.N, T, self.n_islands
        o.scheme, o.ESSrmin, o.seed = _lib.SCHEMES[self.resampling], self.ESSrmin, self.seed
        o.rng_mode = 1 if replay is not None else 0
        o.use_graph = 1 if use_graph else 0
        o.island_offset = island_offset
        o.keep_history = self._keep_history
        o.moments = 1 if self._device_moments else 0
        o.flags = (_lib.FLAG_COLLAPSED_PROPOSAL if self._collapsed else 0) | \
                  (_lib.FLAG_STRICT_ANCESTORS if self._strict else 0) | \
                  (_lib.FLAG_SQMC if self.qmc else 0) | (restore["path_flags"] if restore else _lib.path_flags())
        self._path_flags = int(o.flags) & ~7
        self._ctx = _lib.ctx()
        h = _lib.c_vp()
        check(lib().smc_filter_create(self._ctx.h, ctypes.byref(m), ctypes.byref(o),
                                      y.ctypes.data_as(_lib.P(_lib.c_dbl)), ctypes.byref(h)))
        self._f = h
        if self.qmc:
            # the points of the operator path (rqmc.sobol / sobol_sorted): the context's key, one
            # point-set counter per time step -- the same run whichever path executes it
            if restore:              # (the saved state carries the points' key and counter: load_state puts them back)
                self._sqmc_key = restore["sqmc"]
                check(lib().smc_filter_sqmc_points(self._f, self._sqmc_key[0], self._sqmc_key[1]))
            else:
                self._sqmc_key = (self._ctx._seed & (2 ** 64 - 1), _lib._counter + 1)
                check(lib().smc_filter_sqmc_points(self._f, self._sqmc_key[0], self._sqmc_key[1]))
                _lib._counter += T
        if replay is not None:
            z, u = replay
            self._tapes = (_lib.as_device(z)[0], _lib.as_device(u)[0])
            check(lib().smc_filter_set_replay(self._f, self._tapes[0].ptr, self._tapes[1].ptr))

    def __del__(self):
        f = getattr(self, "_f", None)
        if f:
            try:
                lib().smc_filter_destroy(f)
            except Exception:
                pass
            self._f = None

    def __deepcopy__(self, memo):
        """``copy.deepcopy(pf)`` as the reference's SMC^2 does it for every duplicated theta-particle
        (smc_samplers.py:319-361 ``all_distinct``): the device filter is CLONED in its current state
        (smc_filter_clone: particles, weights, summaries, step record) -- the handle is never shared, so
        neither copy can free the other's filter.  Copies of a reference filter diverge because numpy's
        global generator moves on; here the streams are a function of (seed, island, particle, t), so
        the copy is re-keyed (smc_filter_reseed) and draws noise of its own from its next step on."""
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        shared = ("_f", "_ctx", "_tapes", "_keep")          # handles / read-only host arrays behind them
        for k, v in self.__dict__.items():
            if k in shared:
                new.__dict__[k] = v
            elif k in ("_cache", "_summ_cache"):
                new.__dict__[k] = {} if k == "_cache" else None
            else:
                new.__dict__[k] = copy.deepcopy(v, memo)
        if getattr(self, "_fused", False) and self._f:
            h = _lib.c_vp()
            new._f = None                                  # (a failing clone must not leave the source's handle here)
            check(lib().smc_filter_clone(self._f, ctypes.byref(h)))
            new._f = h
            new.seed = _default_seed()
            check(lib().smc_filter_reseed(new._f, new.seed))
        return new

    # ---- pickling: checkpoint / resume, and how multiSMC's worker processes hand their filters back
    # (core.py:415-428, utils.py:178-186).  The device filter travels as ONE host buffer -- every device array of its
    # slab plus the host-side counters (smc_filter_save_state) -- and is re-created from the same (model, options, data)
    # on the other side (smc_filter_create + smc_filter_load_state): q = pickle.loads(pickle.dumps(pf)) continues bit for
    # bit, history slots, summary ring, Philox counters and all.
    def __getstate__(self):
        d = {k: v for k, v in self.__dict__.items() if k not in ("_f", "_ctx", "_tapes", "_keep", "_cache", "_summ_cache")}
        d["_state_blob"] = None
        d["_tapes_host"] = None
        if getattr(self, "_fused", False) and self._f:
            nb = _lib.c_i64()
            check(lib().smc_filter_state_bytes(self._f, ctypes.byref(nb)))
            blob = np.empty(int(nb.value), dtype=np.uint8)
            check(lib().smc_filter_save_state(self._f, blob.ctypes.data_as(_lib.c_vp), int(nb.value)))
            d["_state_blob"] = blob
            if getattr(self, "_tapes", None) is not None:
                d["_tapes_host"] = tuple(t.get() for t in self._tapes)
        return d

    def __setstate__(self, d):
        blob, tapes = d.pop("_state_blob", None), d.pop("_tapes_host", None)
        self.__dict__.update(d)
        self._f = None
        self._cache, self._summ_cache = {}, None
        if blob is not None:
            use_graph, island_offset = self._ctor[0], self._ctor[1]
            model = self.fk._device_model()
            # re-created under the SAVED verification switches and point-set key: the environment and the process-wide
            # counters of the receiving process are neither read nor moved
            self._create_filter(model, tapes, use_graph, island_offset,
                                restore={"path_flags": d.get("_path_flags", _lib.path_flags()),
                                         "sqmc": d.get("_sqmc_key", (0, 1))})
            nb = _lib.c_i64()
            check(lib().smc_filter_state_bytes(self._f, ctypes.byref(nb)))
            if int(nb.value) != blob.nbytes:
                raise ValueError("unpickling a device filter: the saved state has %d bytes, a filter of this shape holds %d "
                                 "(another library build?)" % (blob.nbytes, int(nb.value)))
            check(lib().smc_filter_load_state(self._f, blob.ctypes.data_as(_lib.c_vp), blob.nbytes))

    def _invalidate(self):
        self._cache = {}
        self._summ_cache = None

    def _summ(self):
        """(n_islands, t, 5) per-step ESS, log_mean_w, loglt, logLt, rs_flag."""
        if self._summ_cache is None:
            out = np.zeros((self.n_islands, self._n, _lib.SUMMARY_COLS))
            if self._n:
                check(lib().smc_filter_summaries(self._f, out.ctypes.data_as(_lib.P(_lib.c_dbl))))
            self._summ_cache = out
        return self._summ_cache

    def _get(self, field, island=0):
        key = (field, island)
        if key not in self._cache:
            dt = np.int64 if field == _lib.FIELD_A else np.float64
            d = getattr(self, "_d", 1)
            shape = (self.N, d) if (d > 1 and field in (_lib.FIELD_X, _lib.FIELD_XP)) else self.N
            out = np.empty(shape, dtype=dt)
            check(lib().smc_filter_get(self._f, field, island, out.ctypes.data_as(_lib.c_vp)))
            self._cache[key] = out
        return self._cache[key]

    def _history(self, field, step, island=0):
        """State of an earlier step of a store_history=True run (device-resident
        history, smc_filter_history)."""
        dt = np.int64 if field == _lib.FIELD_A else np.float64
        d = getattr(self, "_d", 1)
        shape = (self.N, d) if (d > 1 and field in (_lib.FIELD_X, _lib.FIELD_XP)) else self.N
        out = np.empty(shape, dtype=dt)
        check(lib().smc_filter_history(self._f, field, step, island, out.ctypes.data_as(_lib.c_vp)))
        return out

    def _moments(self, island=0):
        """Per-step {'mean', 'var'} of the particles, computed on the device (opts.moments)."""
        d = getattr(self, "_d", 1)
        out = np.empty((self.n_islands, self._n, 2 * d))
        check(lib().smc_filter_moments(self._f, out.ctypes.data_as(_lib.P(_lib.c_dbl))))
        rows = out[island]
        if d == 1:
            return [{"mean": r[0], "var": r[1]} for r in rows]
        return [{"mean": r[:d].copy(), "var": r[d:].copy()} for r in rows]

    def _spacings(self, t, island=0):
        """The sorted uniforms the multinomial resampling of step t draws on the device (production
        mode): smc_filter_spacings."""
        out = np.empty(self.N)
        check(lib().smc_filter_spacings(self._f, int(t), island, out.ctypes.data_as(_lib.P(_lib.c_dbl))))
        return out

    def _trajectories(self, island=0):
        out = np.empty((self._n, self.N), dtype=np.int64)
        check(lib().smc_filter_trajectories(self._f, island,
                                            out.ctypes.data_as(_lib.P(_lib.c_i64))))
        return out

    def permute_islands(self, A):
        """theta-level resampling of whole filters (SMC^2, smc_samplers.py:319-361): island i
        continues from the state of island ``A[i]``."""
        A = np.ascontiguousarray(A, dtype=np.int64)
        if A.shape != (self.n_islands,):
            raise ValueError("permute_islands: one source island per island")
        check(lib().smc_filter_permute_islands(self._f, A.ctypes.data_as(_lib.P(_lib.c_i64))))
        if self._fk_list is not None:
            self._fk_list = [self._fk_list[int(i)] for i in A]
        self._invalidate()

    def accept_islands_from(self, other, accept):
        """PMCMC move of SMC^2 (smc_samplers.py:1129-1143): where ``accept[i]`` is true, island i
        takes over island i of ``other`` (a batch run on the proposed thetas up to the same t)."""
        acc = np.ascontiguousarray(accept, dtype=np.uint8)
        if acc.shape != (self.n_islands,):
            raise ValueError("accept_islands_from: one flag per island")
        check(lib().smc_filter_copy_islands(self._f, other._f, acc.ctypes.data_as(_lib.c_vp)))
        if self._fk_list is not None and other._fk_list is not None:
            self._fk_list = [o if a else s for s, o, a in zip(self._fk_list, other._fk_list, acc)]
        self._invalidate()

    def take_islands_from(self, other, src, dst=None):
        """Islands ``dst`` (default: 0 .. len(src) - 1) of this filter continue from the state of islands
        ``src`` of ``other`` (same model family, N and data; both at the same time index, or this one
        fresh: it is then fast-forwarded to ``other``'s).  Whole filters move packed
        (smc_filter_pack_islands / _unpack_islands), Philox streams stay with the slot."""
        src = np.ascontiguousarray(src, dtype=np.int64)
        dst = np.arange(len(src), dtype=np.int64) if dst is None else np.ascontiguousarray(dst, dtype=np.int64)
        if len(src) != len(dst):
            raise ValueError("take_islands_from: one destination per source")
        if self._n == 0 and other._n > 0:
            check(lib().smc_filter_fast_forward(self._f, other._n))
            self.t = self._n = other._n
        if self._n != other._n:
            raise ValueError("take_islands_from: the two filters are at different time steps")
        if other._n == 0 or len(src) == 0:
            return
        nb = _lib.c_i64()
        check(lib().smc_filter_island_bytes(other._f, ctypes.byref(nb)))
        buf = DeviceArray((max(1, len(src) * int(nb.value) // 8),), dtype=np.int64)
        P64 = _lib.P(_lib.c_i64)
        check(lib().smc_filter_pack_islands(other._f, src.ctypes.data_as(P64), len(src), buf.ptr))
        check(lib().smc_filter_unpack_islands(self._f, dst.ctypes.data_as(P64), len(dst), buf.ptr))
        if self._fk_list is not None and other._fk_list is not None:
            for s_, d_ in zip(src, dst):
                self._fk_list[int(d_)] = other._fk_list[int(s_)]
        self._invalidate()

    def set_state(self, X=None, lw=None, island=0):
        """Replace the particles and / or log-weights of the step just done (the reference lets
        a caller assign ``pf.X`` / ``pf.wgts`` between two steps): the summaries of that step
        and everything the next step derives from the weights are recomputed on the device."""
        if not self._fused:
            if X is not None:
                self.X = X
            if lw is not None:
                self.wgts = rs.Weights(lw=np.array(lw, dtype=np.float64))
            return
        Xc = None if X is None else np.ascontiguousarray(X, dtype=np.float64)
        lc = None if lw is None else np.ascontiguousarray(lw, dtype=np.float64)
        d = getattr(self, "_d", 1)
        if Xc is not None and Xc.size != self.N * d or lc is not None and lc.shape != (self.N,):
            raise ValueError("set_state: X must be (N[, d]) and lw (N,)")
        check(lib().smc_filter_set_state(
            self._f, island, None if Xc is None else Xc.ctypes.data_as(_lib.c_vp),
            None if lc is None else lc.ctypes.data_as(_lib.c_vp)))
        self._invalidate()

    def step_async(self, nsteps=1):
        """Enqueue ``nsteps`` time steps on the device without synchronising."""
        todo = min(nsteps, self.fk.T - self.t)
        if todo > 0:
            check(lib().smc_filter_step(self._f, todo))
            self.t += todo
            self._n += todo
            self._invalidate()
        return todo

    def sync(self):
        if self._f:
            check(lib().smc_filter_sync(self._f))

    # ------------------------------------------------------- public attributes
    @property
    def X(self):
        if not self._fused:
            return self._X
        return self._get(_lib.FIELD_X) if self._n else None

    @X.setter
    def X(self, v):
        self._X = v

    @property
    def Xp(self):
        if not self._fused:
            return self._Xp
        return self._get(_lib.FIELD_XP) if self._n > 1 else None

    @Xp.setter
    def Xp(self, v):
        self._Xp = v

    @property
    def A(self):
        if not self._fused:
            return self._A
        return self._get(_lib.FIELD_A) if self._n > 1 else None

    @A.setter
    def A(self, v):
        self._A = v

    @property
    def wgts(self):
        if not self._fused:
            return self._wgts
        return _DeviceWeights(self) if self._n else rs.Weights()

    @wgts.setter
    def wgts(self, v):
        self._wgts = v

    @property
    def W(self):
        return self.wgts.W

    @property
    def logLt(self):
        if not self._fused:
            return self._logLt
        return float(self._summ()[0, -1, 3]) if self._n else 0.0

    @logLt.setter
    def logLt(self, v):
        self._logLt = v

    @property
    def logLts_islands(self):
        """log-evidence estimate of every island (multiSMC's per-run outputs)."""
        out = np.zeros(self.n_islands)
        check(lib().smc_filter_logLt(self._f, out.ctypes.data_as(_lib.P(_lib.c_dbl))))
        return out

    def __str__(self):
        return self.fk.summary_format(self)

    # ------------------------------------------------- generic template method
    def reset_weights(self):
        """Reset weights after a resampling step (core.py:299-305)."""
        if self.fk.isAPF:
            lw = rs.log_mean_exp(self.logetat, W=self.W) - self.logetat[self.A]
            self.wgts = rs.Weights(lw=lw)
        else:
            self.wgts = rs.Weights()

    def setup_auxiliary_weights(self):
        """Auxiliary weights of the APF (core.py:307-313)."""
        if self.fk.isAPF:
            self.logetat = self.fk.logeta(self.t - 1, self.X)
            self.aux = self.wgts.add(self.logetat)
        else:
            self.aux = self.wgts

    def generate_particles(self):
        if self.qmc:                                               # core.py:315-321
            u = rqmc.sobol(self.N, self.fk.du)
            self.X = self.fk.Gamma0(u[:, 0] if self.fk.du == 1 else u)   # u.squeeze(): (N,) if du = 1
        else:
            self.X = self.fk.M0(self.N)

    def reweight_particles(self):
        self.wgts = self.wgts.add(self.fk.logG(self.t, self.Xp, self.X))   # core.py:323-324

    def resample_move(self):
        self.rs_flag = self.fk.time_to_resample(self)              # core.py:326-337
        if self.rs_flag:
            # (strict_ancestors on the template-method step: the reference's sequential CDF, as on the fused one)
            with rs.strict_mode(self._strict or rs.STRICT[0]):
                self.A = rs.resampling(self.resampling, self.aux.W, M=self.N)
            self.Xp = self.X[self.A]
            self.reset_weights()
        else:
            self.A = np.arange(self.N)
            self.Xp = self.X
        self.X = self.fk.M(self.t, self.Xp)

    def resample_move_qmc(self):
        """SQMC step (core.py:339-349): always resample; the particles are visited in Hilbert
        order (= sorted order for univariate particles), the first QMC coordinate -- sorted -- drives the
        inverse-CDF choice of ancestors, the others the move Gamma.  Sorts, gathers, inverse CDF
        and the inverse-normal-CDF move are device operators."""
        self.rs_flag = True
        us = rqmc.sobol_sorted(self.N, self.fk.du + 1)             # = u[tau] without the sort, when
        if us is not None:                                         # the order is known in closed form
            u, tau, su0 = us, None, us[:, 0]
        else:
            u = rqmc.sobol(self.N, self.fk.du + 1)
            tau = hilbert.argsort(u[:, 0])
            su0 = u[:, 0][tau]
        self.h_order = hilbert.hilbert_sort(self.X)
        self.A = self.h_order[rs.inverse_cdf(su0, self.aux.W[self.h_order], strict=True if self._strict else None)]
        self.Xp = self.X[self.A]
        if self.fk.du == 1:
            v = u[:, 1] if tau is None else u[:, 1][tau]           # u[tau, 1:].squeeze()
        elif isinstance(u, DeviceArray):
            v = DeviceArray.stack_columns([u[:, 1 + i] for i in range(self.fk.du)])
            v = v if tau is None else v[tau]
        else:
            v = u[tau, 1:]
        self.reset_weights()
        self.X = self.fk.Gamma(self.t, self.Xp, v)

    def compute_summaries(self):
        if self.t > 0:                                             # core.py:351-367
            prec_log_mean_w = self.log_mean_w
        self.log_mean_w = self.wgts.log_mean
        if self.t == 0 or self.rs_flag:
            self.loglt = self.log_mean_w
        else:
            self.loglt = self.log_mean_w - prec_log_mean_w
        self.logLt += self.loglt
        if self.verbose:
            print(self)
        if self.hist:
            self.hist.save(self)
        if self.summaries:
            self.summaries.collect(self)

    def _fused_after_step(self):
        """Per-step host work the API semantics force (verbose / history /
        user collectors read the particle system at every t)."""
        s = self._summ()[0, -1]
        self.rs_flag = bool(s[4])
        self.loglt = float(s[2])
        self.t -= 1               # collectors see smc.t = index of the step just done
        try:
            if self.verbose:
                print(self)
            if self.hist:
                self.hist.save(self)
            if self.summaries:
                self.summaries.collect(self)
        finally:
            self.t += 1

    def __next__(self):
        """One step of a particle filter (core.py:369-383)."""
        if self.fk.done(self):
            raise StopIteration
        if self._fused:
            self.step_async(1)
            self._fused_after_step()
            return
        if self.t == 0:
            self.generate_particles()
        else:
            self.setup_auxiliary_weights()
            if self.qmc:
                self.resample_move_qmc()
            else:
                self.resample_move()
        self.reweight_particles()
        self.compute_summaries()
        self.t += 1

    def next(self):
        return self.__next__()

    def __iter__(self):
        return self

    def _needs_per_step_host(self):
        host_hist = bool(self.hist) and not getattr(self, "_device_hist", False)
        host_coll = self._user_collectors and not (self._fused and self._device_moments)
        return self.verbose or host_hist or host_coll

    def _run_to_save_times(self):
        """``store_history=<callable>`` (PartialParticleHistory, smoothing.py:164-184) on the fused
        path: the steps between two save times are enqueued in one call; the host synchronises --
        and copies X and the weights, which the history has to own anyway -- only AT the save times."""
        T, first = self.fk.T, self.t
        while self.t < T:
            nxt = next((u for u in range(self.t, T) if self.hist.is_save_time(u)), None)
            self.step_async((T if nxt is None else nxt + 1) - self.t)
            if nxt is not None:
                self.hist.X[nxt] = self.X
                self.hist.wgts[nxt] = collectors._frozen_weights(self.wgts)
        self.sync()
        if self.summaries and self.t > first:
            s = self._summ()[0]
            self.summaries._extend_defaults(s[first:, 0], s[first:, 3], s[first:, 4] != 0)
            if self._device_moments:
                self.summaries._extend_moments(self._moments()[first:])
        if self._n:
            s = self._summ()[0, -1]
            self.rs_flag, self.loglt = bool(s[4]), float(s[2])

    def run(self):
        """Run until completion (core.py:391-409); sets ``cpu_time``."""
        t0 = time.perf_counter()
        partial = (self._fused and isinstance(self.hist, collectors.PartialParticleHistory)
                   and not self.verbose and not (self._user_collectors and not self._device_moments))
        if partial:
            self._run_to_save_times()
        elif self._fused and not self._needs_per_step_host():
            first = self.t
            self.step_async(self.fk.T - self.t)
            self.sync()
            if self.summaries and self.t > first:      # default collectors, in one go
                s = self._summ()[0]
                self.summaries._extend_defaults(s[first:, 0], s[first:, 3], s[first:, 4] != 0)
                if self._device_moments:
                    self.summaries._extend_moments(self._moments()[first:])
            if self._n:
                s = self._summ()[0, -1]
                self.rs_flag, self.loglt = bool(s[4]), float(s[2])
        else:
            for _ in self:
                pass
        self.cpu_time = time.perf_counter() - t0


####################################################

def multiSMC(nruns=10, nprocs=0, out_func=None, collect=None, group=None, **args):
    """Run SMC algorithms for different combinations of parameters
    (core.py:431-518, utils.py:216-269).

    Same calling convention and output format as the reference: list-valued
    (dict-valued) keyword arguments are expanded into their Cartesian product,
    each combination is run ``nruns`` times with distinct seeds, and the result
    is a list of dicts with keys ``run``, ``seed``, the varied arguments, and
    ``output`` (the SMC object, or ``out_func(smc)``).  ``nprocs`` is accepted
    for compatibility: the runs of one combination execute as islands of ONE
    device-resident filter instead of a pool of processes.

    group : ``particles_amd.distributed.Group`` -- the multi-GPU form of the reference's
        ``nprocs`` (utils.py:158-186 fans the runs out to worker processes; here one process per
        GPU was launched and EVERY rank makes this same call).  The ``nruns`` runs of each
        combination are block-partitioned over the ranks (``shard_islands``), each share runs as
        islands ``first .. first + count - 1`` of the combination's filter -- the Philox island
        word is the GLOBAL run index, so the results do not depend on the number of ranks -- and
        the outputs are gathered in run order on every rank: numeric ``out_func`` results
        (a float or equal-shaped float arrays: the log-evidences of BASELINE config C5) through
        the group's RCCL all-gather, anything else pickled over the host rendezvous.  With
        ``out_func=None`` a remote run's output is a host snapshot (``RunSnapshot``) of its SMC
        object: device memory does not travel.
    """
    import itertools

    fixed, lists, dicts = {}, {}, {}
    for k, v in args.items():
        if isinstance(v, list):
            lists[k] = v
        elif isinstance(v, dict):
            dicts[k] = v
        else:
            fixed[k] = v
    keys = list(lists) + list(dicts)
    choices = [list(enumerate(lists[k])) for k in lists] + \
              [list(dicts[k].items()) for k in dicts]
    combos = list(itertools.product(*choices)) if keys else [()]
    bw = (2 ** 32 - 1) // max(1, nruns * len(combos))           # utils.py:189-202
    seeds = np.arange(0, nruns * len(combos) * bw, bw) + np.random.randint(bw, size=nruns * len(combos))
    rank, world = (group.rank, group.world) if group is not None else (0, 1)
    if world > 1:
        seeds = group.broadcast_host(seeds)          # rank 0's draw: one job, one set of seeds
    first, count = 0, nruns
    if world > 1:
        from .distributed import shard_islands
        first, count = shard_islands(nruns, rank, world)
    results, si = [], 0
    for combo in combos:
        kw, label = dict(fixed), {}
        for k, (lab, val) in zip(keys, combo):
            kw[k] = val
            label[k] = val if k in lists else lab
        run_seeds = seeds[si:si + nruns]
        si += nruns
        fk = kw.get("fk")
        batch = (SMC._will_fuse(fk, kw.get("qmc", False), kw.get("resampling", "systematic"), N=kw.get("N", 100))
                 and not collect and not kw.get("store_history") and not kw.get("verbose")
                 and kw.get("n_islands", 1) == 1
                 and not (fk.isAPF and not SMC._apf_fusable(fk, kw.get("N", 100), kw.get("resampling", "systematic"),
                                                            None, False, False)))
        mine = []                                    # this rank's outputs, runs first .. first + count - 1
        if batch and out_func is not None:
            # islands of one filter; Philox island word = (global) run index, key = first seed
            if count > 0:
                pf = SMC(collect="off", seed=int(run_seeds[0]), n_islands=count, island_offset=first, **kw)
                pf.run()
                mine = [out_func(_IslandView(pf, r)) for r in range(count)]
            out_seeds = [int(run_seeds[0])] * nruns
        else:
            for r in range(first, first + count):
                np.random.seed(int(run_seeds[r]))       # utils.py:209-213: the seeder of each run
                pf = SMC(collect=collect, seed=int(run_seeds[r]), **kw)
                pf.run()
                mine.append(out_func(pf) if out_func is not None else (pf if world == 1 else RunSnapshot(pf)))
            out_seeds = [int(v) for v in run_seeds]
        outputs = mine if world == 1 else group.gather_outputs(mine, nruns)
        for r in range(nruns):
            d = {"run": r, "seed": out_seeds[r]}
            d.update(label)
            d["output"] = outputs[r]
            results.append(d)
    return results


class RunSnapshot:
    """Host copy of what a finished SMC run leaves behind (multiSMC over several ranks with
    ``out_func=None``: the reference pickles the SMC object back from its worker process,
    utils.py:178-186; a device-resident filter cannot travel, its results can)."""

    def __init__(self, pf):
        self.N, self.t, self.cpu_time = pf.N, pf.t, pf.cpu_time
        self.logLt = float(pf.logLt)
        self.X = np.array(pf.X)
        w = pf.wgts
        self.lw, self.W, self.ESS = np.array(w.lw), np.array(w.W), float(w.ESS)
        sm = pf.summaries
        self.summaries = None if sm is None else {k: list(getattr(sm, k).copy() if hasattr(getattr(sm, k), "copy")
                                                         else getattr(sm, k))
                                                  for k in ("ESSs", "logLts", "rs_flags") if hasattr(sm, k)}


class _IslandView:
    """What ``out_func`` sees for island r of a batched run: the SMC attributes
    of that replica."""

    def __init__(self, pf, r):
        self._pf, self._r = pf, r
        self.fk, self.N, self.t = pf.fk, pf.N, pf.t
        self.cpu_time = pf.cpu_time
        self.summaries = None

    @property
    def logLt(self):
        return float(self._pf._summ()[self._r, -1, 3])

    @property
    def X(self):
        return self._pf._get(_lib.FIELD_X, self._r)

    @property
    def A(self):
        return self._pf._get(_lib.FIELD_A, self._r)

    @property
    def wgts(self):
        return _DeviceWeights(self._pf, self._r)

    @property
    def W(self):
        return self.wgts.W

    @property
    def ESSs(self):
        return self._pf._summ()[self._r, :, 0]

    @property
    def logLts(self):
        return self._pf._summ()[self._r, :, 3]
