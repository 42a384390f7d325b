"""ctypes binding of libsmc_hip.so (include/smc_hip.h) + device arrays.

There is NO CPU fallback: if the HIP library is missing or no MI355X is
visible, every operator raises.  ``SMC_HIP_LIBRARY`` may point at another
build of the same C ABI (the CPU test-suite uses it to load the fiber emulator
of tests/emu, which exists only to exercise kernel logic without a GPU).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "lib", "libsmc_hip.so")

c_i64, c_u64, c_int, c_dbl, c_vp = (ctypes.c_int64, ctypes.c_uint64, ctypes.c_int,
                                    ctypes.c_double, ctypes.c_void_p)
c_sz = ctypes.c_size_t
P = ctypes.POINTER

MULTINOMIAL, STRATIFIED, SYSTEMATIC = 0, 1, 2
SCHEMES = {"multinomial": MULTINOMIAL, "stratified": STRATIFIED, "systematic": SYSTEMATIC}
MODEL_LINGAUSS, MODEL_STOCHVOL, MODEL_MVLINGAUSS, MODEL_GORDON, MODEL_THETALOGISTIC = 1, 2, 3, 4, 5
MODEL_SVLEVERAGE = 6
MODEL_DISCRETECOX = 7
FK_BOOTSTRAP, FK_GUIDED, FK_APF, FK_APF_BOOT = 0, 1, 2, 3
FIELD_X, FIELD_XP, FIELD_A, FIELD_LW, FIELD_W = range(5)
SUMMARY_COLS = 5
PARAM_STRIDE = 16


class SmcModel(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("fk", ctypes.c_int32),
                ("dx", ctypes.c_int32), ("dy", ctypes.c_int32),
                ("params_host", P(c_dbl)),
                ("F_host", P(c_dbl)), ("G_host", P(c_dbl)), ("covX_host", P(c_dbl)),
                ("covY_host", P(c_dbl)), ("mu0_host", P(c_dbl)), ("cov0_host", P(c_dbl)),
                ("aux_host", P(c_dbl))]


class SmcFilterOpts(ctypes.Structure):
    _fields_ = [("N", c_i64), ("T", c_i64), ("n_islands", ctypes.c_int32),
                ("scheme", ctypes.c_int32), ("ESSrmin", c_dbl), ("seed", c_u64),
                ("rng_mode", ctypes.c_int32), ("use_graph", ctypes.c_int32),
                ("island_offset", ctypes.c_int32), ("keep_history", ctypes.c_int32),
                ("moments", ctypes.c_int32), ("flags", ctypes.c_int32)]


FLAG_COLLAPSED_PROPOSAL = 1
FLAG_STRICT_ANCESTORS = 2
FLAG_SQMC = 4
# verification switches (include/smc_hip.h SMC_PATH_*): the environment variables the test-suite and
# tools/ set to force an alternative, equivalent code path; read HERE, not in the library
PATH_FLAGS = {"SMC_FLAT_CDF": 1 << 8, "SMC_TWO_LEVEL_MID": 1 << 9, "SMC_EXACT_COUNTS": 1 << 10,
              "SMC_FORCE_UNFUSED": 1 << 12, "SMC_NO_SMALL": 1 << 13, "SMC_NO_HEAVY": 1 << 15, "SMC_NO_TK": 1 << 16,
              "SMC_SPACING_3PASS": 1 << 19, "SMC_SPLIT_REDUCE": 1 << 24, "SMC_SQ_GATHER": 1 << 29, "SMC_NO_WIDE": 1 << 30,
              "SMC_STRICT_LITERAL": 1 << 6, "SMC_NO_XCD_CHUNKS": 1 << 5, "SMC_MV_DENSE": 1 << 3}


def path_flags():
    f = 0
    for name, bit in PATH_FLAGS.items():
        if os.environ.get(name):
            f |= bit
    tp = os.environ.get("SMC_SP_TPW")
    if tp and int(tp) in (1, 2, 4, 8):
        f |= int(tp) << 25
    mv = os.environ.get("SMC_MV_CHUNKS")
    if mv and int(mv) in (1, 2, 4, 8):
        f |= int(mv) << 20
    return f


# name -> (restype, argtypes): every symbol include/smc_hip.h declares
SIGNATURES = {
    "smc_device_count": (c_int, [P(c_int)]),
    "smc_ctx_create": (c_int, [c_int, c_u64, P(c_vp)]),
    "smc_ctx_destroy": (c_int, [c_vp]),
    "smc_ctx_sync": (c_int, [c_vp]),
    "smc_ctx_seed": (c_int, [c_vp, c_u64]),
    "smc_last_error": (ctypes.c_char_p, []),
    "smc_version": (ctypes.c_char_p, []),
    "smc_ctx_device_info": (c_int, [c_vp, ctypes.c_char_p, c_sz, P(c_int), P(c_u64)]),
    "smc_ctx_device_pci": (c_int, [c_vp, ctypes.c_char_p, c_sz]),
    "smc_malloc": (c_int, [c_vp, c_sz, P(c_vp)]),
    "smc_free": (c_int, [c_vp, c_vp]),
    "smc_memcpy_h2d": (c_int, [c_vp, c_vp, c_vp, c_sz]),
    "smc_memcpy_d2h": (c_int, [c_vp, c_vp, c_vp, c_sz]),
    "smc_memcpy_d2d": (c_int, [c_vp, c_vp, c_vp, c_sz]),
    "smc_memset": (c_int, [c_vp, c_vp, c_int, c_sz]),
    "smc_timer_start": (c_int, [c_vp]),
    "smc_timer_stop": (c_int, [c_vp, P(ctypes.c_float)]),
    "smc_lse_normalise": (c_int, [c_vp, c_vp, c_i64, c_vp, P(c_dbl)]),
    "smc_log_wmean_exp": (c_int, [c_vp, c_vp, c_vp, c_i64, P(c_dbl)]),
    "smc_wmean_var": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, P(c_dbl)]),
    "smc_wmean_cov": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, P(c_dbl)]),
    "smc_inverse_cdf": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp]),
    "smc_inverse_cdf_strict": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp]),
    "smc_seq_prefix_sums": (c_int, [c_vp, c_vp, c_i64, c_vp, c_int, P(c_i64)]),
    "smc_resample": (c_int, [c_vp, c_int, c_vp, c_i64, c_i64, c_vp, c_u64, c_vp]),
    "smc_uniform_spacings": (c_int, [c_vp, c_i64, c_u64, c_vp]),
    "smc_gather": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp]),
    "smc_normal_rvs": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_u64, c_i64, c_vp]),
    "smc_normal_logpdf": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_vp]),
    "smc_copy_strided": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_i64]),
    "smc_normal_ppf": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_vp]),
    "smc_argsort": (c_int, [c_vp, c_vp, c_i64, c_vp]),
    "smc_debug_sort_window_min": (c_int, [ctypes.c_longlong]),
    "smc_hilbert_array": (c_int, [c_vp, c_vp, c_i64, ctypes.c_int32, c_vp]),
    "smc_hilbert_sort": (c_int, [c_vp, c_vp, c_i64, ctypes.c_int32, c_vp, c_vp]),
    "smc_sobol": (c_int, [c_vp, c_i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_u64, c_vp]),
    "smc_sobol_sorted": (c_int, [c_vp, c_i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_u64, c_vp]),
    "smc_filter_describe": (c_int, [c_vp, ctypes.c_char_p, c_sz]),
    "smc_filter_state_bytes": (c_int, [c_vp, ctypes.POINTER(ctypes.c_int64)]),
    "smc_filter_save_state": (c_int, [c_vp, c_vp, ctypes.c_int64]),
    "smc_filter_load_state": (c_int, [c_vp, c_vp, ctypes.c_int64]),
    "smc_filter_strict_stats": (c_int, [c_vp, ctypes.c_int32, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "smc_poisson_logpmf": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, c_vp]),
    "smc_standard_normal": (c_int, [c_vp, c_u64, c_i64, c_vp]),
    "smc_uniform": (c_int, [c_vp, c_u64, c_i64, c_vp]),
    "smc_mvn_rvs": (c_int, [c_vp, c_vp, c_i64, c_dbl, P(c_dbl), c_vp, c_u64, c_i64, c_i64, c_vp]),
    "smc_mvn_logpdf": (c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_dbl, P(c_dbl), c_i64, c_i64, c_vp]),
    "smc_filter_create": (c_int, [c_vp, P(SmcModel), P(SmcFilterOpts), P(c_dbl), P(c_vp)]),
    "smc_filter_destroy": (c_int, [c_vp]),
    "smc_filter_clone": (c_int, [c_vp, P(c_vp)]),
    "smc_filter_reseed": (c_int, [c_vp, c_u64]),
    "smc_filter_sqmc_points": (c_int, [c_vp, c_u64, c_u64]),
    "smc_filter_fast_forward": (c_int, [c_vp, c_i64]),
    "smc_filter_theta_enable_sharded": (c_int, [c_vp, c_vp, c_dbl]),
    "smc_comm_allgather_f64_async": (c_int, [c_vp, c_vp, c_i64, c_vp]),
    "smc_comm_rank": (c_int, [c_vp, P(c_int), P(c_int)]),
    "smc_filter_set_replay": (c_int, [c_vp, c_vp, c_vp]),
    "smc_filter_step": (c_int, [c_vp, c_i64]),
    "smc_filter_sync": (c_int, [c_vp]),
    "smc_filter_t": (c_int, [c_vp, P(c_i64)]),
    "smc_filter_summaries": (c_int, [c_vp, P(c_dbl)]),
    "smc_filter_logLt": (c_int, [c_vp, P(c_dbl)]),
    "smc_elementwise": (c_int, [c_vp, c_int, c_vp, c_i64, c_vp, c_i64, c_dbl, c_i64, c_vp]),
    "smc_rows_matmul": (c_int, [c_vp, c_vp, c_i64, c_i64, P(c_dbl), c_i64, c_vp]),
    "smc_wquantiles": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, P(c_dbl), c_int, P(c_dbl)]),
    "smc_residual_split": (c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, P(c_i64)]),
    "smc_residual_ancestors": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp]),
    "smc_resample_ssp": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp]),
    "smc_killing_split": (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp, P(c_i64)]),
    "smc_killing_ancestors": (c_int, [c_vp, c_vp, c_vp, c_i64, c_vp]),
    "smc_filter_get": (c_int, [c_vp, c_int, c_int, c_vp]),
    "smc_filter_set_state": (c_int, [c_vp, c_int, c_vp, c_vp]),
    "smc_filter_island_bytes": (c_int, [c_vp, P(c_i64)]),
    "smc_filter_pack_islands": (c_int, [c_vp, P(c_i64), c_int, c_vp]),
    "smc_filter_unpack_islands": (c_int, [c_vp, P(c_i64), c_int, c_vp]),
    "smc_filter_theta_enable": (c_int, [c_vp, c_dbl]),
    "smc_filter_theta_state": (c_int, [c_vp, c_vp, P(c_i64), P(c_i64), c_vp]),
    "smc_filter_theta_resume": (c_int, [c_vp, c_vp]),
    "smc_filter_theta_logmeans": (c_int, [c_vp, c_vp, P(c_i64)]),
    "smc_filter_copy_islands": (c_int, [c_vp, c_vp, c_vp]),
    "smc_filter_moments": (c_int, [c_vp, P(c_dbl)]),
    "smc_filter_permute_islands": (c_int, [c_vp, P(c_i64)]),
    "smc_filter_history": (c_int, [c_vp, c_int, c_i64, c_int, c_vp]),
    "smc_filter_one_trajectory": (c_int, [c_vp, c_int, c_i64, P(c_dbl)]),
    "smc_filter_trajectories": (c_int, [c_vp, c_int, P(c_i64)]),
    "smc_filter_spacings": (c_int, [c_vp, c_i64, c_int, P(c_dbl)]),
    "smc_filter_info": (c_int, [c_vp, P(c_dbl), P(c_int)]),
    "smc_filter_profile": (c_int, [c_vp, c_int]),
    "smc_filter_kernel_ms": (c_int, [c_vp, P(c_dbl), P(c_dbl), P(c_i64)]),
    "smc_comm_unique_id": (c_int, [ctypes.c_char_p]),
    "smc_comm_create": (c_int, [c_vp, c_int, c_int, ctypes.c_char_p, P(c_vp)]),
    "smc_comm_allgather_f64": (c_int, [c_vp, c_vp, c_i64, c_vp]),
    "smc_comm_alltoallv": (c_int, [c_vp, c_vp, P(c_i64), P(c_i64), c_vp, P(c_i64), P(c_i64)]),
    "smc_comm_destroy": (c_int, [c_vp]),
}

_lib = None


def library_path():
    return os.environ.get("SMC_HIP_LIBRARY", DEFAULT_LIB)


def lib():
    """The loaded C-ABI library; raises loudly if it is not there."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError(
                "particles_amd: %s not found. Build it with "
                "`python -m particles_amd._build` (needs hipcc); there is no CPU "
                "fallback." % path)
        L = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)      # AttributeError if a declared symbol is missing
            fn.restype, fn.argtypes = res, args
        # the product only ever runs gfx950 code: a build of the C ABI for anything else (the
        # test-suite's fiber emulator) is accepted only when the test harness says so
        if b"gfx950" not in L.smc_version() and os.environ.get("SMC_TEST_EMULATOR") != "1":
            raise RuntimeError("particles_amd: %s is not a gfx950 build (%s); there is no CPU "
                               "fallback." % (path, L.smc_version().decode()))
        _lib = L
    return _lib


class SmcError(RuntimeError):
    pass


def check(rc):
    if rc == 0:
        return
    msg = lib().smc_last_error().decode()
    if rc in (1, 4):          # SMC_ERR_INVALID, SMC_ERR_SCHEME (resampling.py:477-481)
        raise ValueError(msg)
    if rc == 3:
        raise MemoryError(msg)
    raise SmcError(msg)


class Context:
    """One device + stream (smc_ctx)."""

    def __init__(self, device=0, seed=0):
        h = c_vp()
        check(lib().smc_ctx_create(int(device), int(seed) & (2 ** 64 - 1), ctypes.byref(h)))
        self.h, self.device, self._seed = h, device, int(seed)

    def seed(self, seed):
        self._seed = int(seed)
        check(lib().smc_ctx_seed(self.h, self._seed & (2 ** 64 - 1)))

    def sync(self):
        check(lib().smc_ctx_sync(self.h))

    def device_info(self):
        name = ctypes.create_string_buffer(256)
        ncu, mem = c_int(), c_u64()
        check(lib().smc_ctx_device_info(self.h, name, 256, ctypes.byref(ncu), ctypes.byref(mem)))
        return {"name": name.value.decode(), "n_cu": ncu.value, "hbm_bytes": mem.value}

    def device_pci(self):
        """PCI bus id of the device: the same GPU has the same id in every process of the node."""
        buf = ctypes.create_string_buffer(64)
        check(lib().smc_ctx_device_pci(self.h, buf, 64))
        return buf.value.decode()

    def close(self):
        if self.h:
            lib().smc_ctx_destroy(self.h)
            self.h = None


_default_ctx = None
_counter = 0


# Where host-facing operators take their random draws from: "numpy" = the host's legacy
# generator, in the reference's order (a run seeded with numpy.random.seed reproduces the
# reference's run); "philox" = the device's counter-based generator.  Device-resident inputs
# always use the device generator.  Set with particles_amd.resampling.set_rng().
RNG_MODE = ["numpy"]
# Host-facing operators return numpy arrays for numpy inputs (drop-in mode).  With
# RESIDENT[0] = True (particles_amd.set_resident) they return DeviceArrays instead: a
# user-defined model then runs its whole step in HBM (DeviceArray supports the arithmetic).
RESIDENT = [False]
# SMC(qmc=True) on the fused SQMC step (SMC_FLAG_SQMC) where it applies; False: always the operator path
FUSED_SQMC = [True]


def default_device():
    return int(os.environ.get("SMC_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def ctx():
    """Process-wide default context (device = $SMC_HIP_DEVICE / $LOCAL_RANK / 0)."""
    global _default_ctx
    if _default_ctx is None:
        n = c_int()
        check(lib().smc_device_count(ctypes.byref(n)))
        if n.value < 1:
            raise RuntimeError("particles_amd: no HIP device visible (there is no CPU fallback)")
        _default_ctx = Context(default_device() % n.value, seed=0)
    return _default_ctx


def seed(s):
    """Counterpart of ``numpy.random.seed`` for the device Philox stream
    (utils.py:209-213 seeds the global generator per run)."""
    global _counter
    ctx().seed(s)
    _counter = 0


def next_counter():
    """A fresh Philox sub-stream id for each stand-alone draw."""
    global _counter
    _counter += 1
    return _counter


def _device_array_from_host(a):
    return DeviceArray.from_numpy(a, dtype=a.dtype)


_F64 = np.dtype(np.float64)
_SCALARS = {}          # DeviceArray.scalar's cache (default context); emptied before the interpreter tears the context down
import atexit
atexit.register(_SCALARS.clear)


class DeviceArray:
    """A C-contiguous fp64 / int64 array in HBM."""

    def __init__(self, shape, dtype=np.float64, context=None):
        # (plain Python here: the template-method step of a user-defined model creates eight of these per time step,
        #  and np.prod / np.isscalar / np.dtype cost more than the allocation they describe)
        self.ctx = context or ctx()
        if type(shape) is int:
            self.shape = (shape,)
            size = shape
        else:
            self.shape = (int(shape),) if np.isscalar(shape) else tuple(int(s) for s in shape)
            size = 1
            for s in self.shape:
                size *= s
        self.dtype = _F64 if dtype is np.float64 else np.dtype(dtype)
        assert self.dtype.itemsize == 8
        self.size = size
        p = c_vp()
        check(lib().smc_malloc(self.ctx.h, size * 8, ctypes.byref(p)))
        self.ptr = p

    @classmethod
    def from_numpy(cls, a, dtype=None, context=None):
        a = np.ascontiguousarray(a, dtype=dtype or (np.int64 if np.asarray(a).dtype.kind in "iu"
                                                     else np.float64))
        out = cls(a.shape if a.ndim else (1,), a.dtype, context)
        check(lib().smc_memcpy_h2d(out.ctx.h, out.ptr, a.ctypes.data_as(c_vp), a.nbytes))
        return out

    @classmethod
    def scalar(cls, v):
        """A (1,) fp64 array holding the Python / numpy scalar ``v`` -- kept: distributions with scalar parameters
        (``Normal(loc=xp, scale=self.sigma)``, the observation of the step) ask for the same few values at every time
        step, and an upload per request was a third of the host's work per step of a user-defined model."""
        v = float(v)
        key = v if v == v else "nan"
        a = _SCALARS.get(key)
        if a is None:
            if len(_SCALARS) >= 4096:
                _SCALARS.clear()
            a = cls((1,))
            buf = ctypes.c_double(v)
            check(lib().smc_memcpy_h2d(a.ctx.h, a.ptr, ctypes.byref(buf), 8))
            _SCALARS[key] = a
        return a

    def get(self):
        out = np.empty(self.shape, dtype=self.dtype)
        check(lib().smc_memcpy_d2h(self.ctx.h, out.ctypes.data_as(c_vp), self.ptr, out.nbytes))
        return out

    def __reduce__(self):
        """Pickling moves the values through the host (the reference's arrays are numpy arrays: utils.py:178-186
        ships them between processes as such); the copy lives in the unpickling process's context."""
        return (_device_array_from_host, (self.get(),))

    def __len__(self):
        return self.shape[0]

    @property
    def ndim(self):
        return len(self.shape)

    # ---- arithmetic, so that model code written for numpy arrays (xp, x in M / logG of a
    # FeynmanKac or StateSpaceModel subclass) runs unchanged on arrays that stay in HBM
    __array_priority__ = 1000.0
    _EW = {"add": 0, "sub": 1, "mul": 2, "div": 3, "rsub": 4, "rdiv": 5, "neg": 6, "exp": 7,
           "log": 8, "sqrt": 9, "cos": 10, "sin": 11, "abs": 12, "square": 13, "pow": 14,
           "min": 15, "max": 16, "arctan": 17}

    def _ew(self, op, other=None):
        if self.dtype != np.float64:
            raise TypeError("device arithmetic is defined for float64 arrays")
        b, sb, alpha, shape = None, 0, 0.0, self.shape
        a, sa = self, 1
        if other is not None:
            if not isinstance(other, DeviceArray) and np.size(other) > 1:
                other = DeviceArray.from_numpy(np.asarray(other, dtype=np.float64), context=self.ctx)
            if isinstance(other, DeviceArray):
                if other.size == self.size:
                    b, sb = other, 1
                elif other.size == 1:
                    b, sb = other, 0
                elif self.size == 1:
                    sa, b, sb, shape = 0, other, 1, other.shape
                elif self.ndim == 2 and other.size == self.shape[1]:      # (N, d) with a (d,) row
                    b, sb = other, -self.shape[1]
                elif other.ndim == 2 and self.size == other.shape[1]:
                    sa, b, sb, shape = -other.shape[1], other, 1, other.shape
                else:
                    raise ValueError("operands could not be broadcast together")
            else:
                alpha = float(np.asarray(other).reshape(-1)[0])
        out = DeviceArray(shape, np.float64, self.ctx)
        check(lib().smc_elementwise(self.ctx.h, self._EW[op], a.ptr, sa, b.ptr if b is not None else None,
                                    sb, alpha, out.size, out.ptr))
        return out

    def __add__(self, o): return self._ew("add", o)
    def __radd__(self, o): return self._ew("add", o)
    def __sub__(self, o): return self._ew("sub", o)
    def __rsub__(self, o): return self._ew("rsub", o)
    def __mul__(self, o): return self._ew("mul", o)
    def __rmul__(self, o): return self._ew("mul", o)
    def __truediv__(self, o): return self._ew("div", o)
    def __rtruediv__(self, o): return self._ew("rdiv", o)
    def __neg__(self): return self._ew("neg")
    def __abs__(self): return self._ew("abs")

    def __pow__(self, e):
        if np.ndim(e) == 0 and e == 2:
            return self._ew("square")           # as numpy: x ** 2 is x * x
        if np.ndim(e) == 0 and e == 0.5:
            return self._ew("sqrt")
        return self._ew("pow", e)

    _UFUNCS = {"add": ("add", "add"), "subtract": ("sub", "rsub"), "multiply": ("mul", "mul"),
               "true_divide": ("div", "rdiv"), "divide": ("div", "rdiv"),
               "minimum": ("min", "min"), "maximum": ("max", "max")}
    _UNARY = {"exp": "exp", "log": "log", "sqrt": "sqrt", "cos": "cos", "sin": "sin",
              "absolute": "abs", "fabs": "abs", "negative": "neg", "square": "square",
              "arctan": "arctan"}

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        """numpy ufuncs called on a DeviceArray (np.exp(x), np.cos(x), 2.0 * x with a numpy
        scalar on the left, ...) stay on the device."""
        if method != "__call__" or kwargs:
            return NotImplemented
        name = ufunc.__name__
        if name in self._UNARY and len(inputs) == 1:
            return inputs[0]._ew(self._UNARY[name])
        if name == "power" and isinstance(inputs[0], DeviceArray):
            return inputs[0].__pow__(inputs[1])
        if name in self._UFUNCS and len(inputs) == 2:
            fwd, rev = self._UFUNCS[name]
            if isinstance(inputs[0], DeviceArray):
                return inputs[0]._ew(fwd, inputs[1])
            return inputs[1]._ew(rev, inputs[0])
        return NotImplemented

    def column(self, i):
        """``x[..., i]`` of an (N, d) array as a contiguous (N,) device array
        (distributions.py:1102)."""
        if self.ndim != 2 or self.dtype != np.float64:
            raise TypeError("column() takes a float64 (N, d) array")
        N, d = self.shape
        i = int(i) + (d if int(i) < 0 else 0)
        if not 0 <= i < d:
            raise IndexError("index %d is out of bounds for axis 1 with size %d" % (i, d))
        out = DeviceArray((N,), np.float64, self.ctx)
        check(lib().smc_copy_strided(self.ctx.h, c_vp(self.ptr.value + 8 * i), d, out.ptr, 1, N))
        return out

    @classmethod
    def stack_columns(cls, cols):
        """``np.stack(cols, axis=1)`` of (N,) device arrays (distributions.py:1106)."""
        N, d = cols[0].size, len(cols)
        out = cls((N, d), np.float64, cols[0].ctx)
        for i, c in enumerate(cols):
            if not isinstance(c, DeviceArray) or c.size != N:
                raise ValueError("all input arrays must have the same shape")
            check(lib().smc_copy_strided(out.ctx.h, c.ptr, 1, c_vp(out.ptr.value + 8 * i), d, N))
        return out

    def __getitem__(self, idx):
        """``x[A]`` with a device or host int64 index array (core.py:332 Xp = X[A]);
        ``x[..., i]`` / ``x[:, i]``: one column of an (N, d) array."""
        if isinstance(idx, tuple) and len(idx) == 2 and isinstance(idx[1], (int, np.integer)) \
                and (idx[0] is Ellipsis or idx[0] == slice(None)):
            return self.column(idx[1])
        A = idx if isinstance(idx, DeviceArray) else DeviceArray.from_numpy(
            np.ascontiguousarray(idx, dtype=np.int64), context=self.ctx)
        if A.dtype != np.int64:
            raise TypeError("DeviceArray indexing takes an int64 index array")
        d = self.size // self.shape[0]
        # (8-byte elements either way: an int64 array -- h_order[idx], core.py:344 -- is gathered
        # by the same kernel, which only copies words)
        out = DeviceArray((A.size,) + tuple(self.shape[1:]), self.dtype, self.ctx)
        check(lib().smc_gather(self.ctx.h, self.ptr, A.ptr, A.size, d, out.ptr))
        return out

    def squeeze(self):
        return self

    def __matmul__(self, M):
        """``X @ M`` for device rows ``X`` (N, d) and a small host matrix ``M`` (d, k)."""
        Mh = np.ascontiguousarray(M, dtype=np.float64)
        d = self.shape[-1] if self.ndim == 2 else 1
        if Mh.ndim != 2 or Mh.shape[0] != d:
            raise ValueError("matmul: shapes (%s) and %s do not align" % (self.shape, Mh.shape))
        N = self.size // d
        out = DeviceArray((N, Mh.shape[1]), np.float64, self.ctx)
        check(lib().smc_rows_matmul(self.ctx.h, self.ptr, N, d, Mh.ctypes.data_as(P(c_dbl)),
                                    Mh.shape[1], out.ptr))
        return out

    def __array_function__(self, func, types, args, kwargs):
        """``np.dot(xp, F.T)`` as the reference's models write it (kalman.py:339)."""
        if func in (np.dot, np.matmul) and len(args) == 2 and isinstance(args[0], DeviceArray) \
                and not isinstance(args[1], DeviceArray) and not kwargs:
            return args[0].__matmul__(args[1])
        if func is np.stack and len(args) == 1 and kwargs.get("axis", 0) == 1 \
                and all(isinstance(c, DeviceArray) and c.ndim == 1 for c in args[0]):
            return DeviceArray.stack_columns(list(args[0]))
        return NotImplemented

    def __array__(self, dtype=None, copy=None):
        a = self.get()
        return a if dtype is None else a.astype(dtype)

    def free(self):
        if getattr(self, "ptr", None):
            try:
                lib().smc_free(self.ctx.h, self.ptr)
            except Exception:
                pass
            self.ptr = None

    def __del__(self):
        self.free()


def as_device(a, dtype=None):
    """(device array, was_host) for a numpy array or a DeviceArray."""
    if isinstance(a, DeviceArray):
        return a, False
    return DeviceArray.from_numpy(a, dtype=dtype), True


def dptr(a):
    return a.ptr if a is not None else None


def host_dbl(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(P(c_dbl))
