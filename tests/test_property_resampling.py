"""Property tests of the resampling operators through the C ABI (emulator build when no GPU
is visible, the real library on a GPU box): for arbitrary weight vectors -- zeros, huge dynamic
range, ties -- the device's ancestors equal the oracle's on the exact Q62 CDF, bit for bit."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import particles_amd as pa  # noqa: F401  (conftest points it at the right library)
from oracle import smc_oracle as orc
from particles_amd import resampling as rs

pytestmark = pytest.mark.filterwarnings("ignore")


def weights(draw, n):
    kind = draw(st.sampled_from(["lognormal", "sparse", "ties", "spike", "tiny"]))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    if kind == "lognormal":
        w = np.exp(draw(st.floats(0.1, 30.0)) * rng.standard_normal(n))
    elif kind == "sparse":
        w = rng.random(n) * (rng.random(n) < 0.1)
        w[rng.integers(0, n)] += 1e-3
    elif kind == "ties":
        w = np.ones(n)
    elif kind == "spike":
        w = np.full(n, 1e-300)
        w[rng.integers(0, n)] = 1.0
    else:
        w = rng.random(n) * 1e-200 + 1e-250
    return w / w.sum()


@st.composite
def case(draw):
    n = draw(st.sampled_from([1, 2, 3, 63, 64, 65, 200, 1023, 1024, 1025, 2500]))
    m = draw(st.sampled_from([1, 2, 7, 64, 999, 1024, 3000]))
    return weights(draw, n), m, draw(st.integers(0, 2 ** 31 - 1))


@settings(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck))
@given(case())
def test_schemes_equal_q62_oracle(c):
    W, M, seed = c
    for scheme in ("systematic", "stratified", "multinomial"):
        np.random.seed(seed)
        got = rs.resampling(scheme, W, M=M)
        np.random.seed(seed)
        want = orc.resampling(scheme, W, M=M, cdf="q62")
        assert got.dtype == np.int64 and got.shape == (M,)
        assert np.array_equal(got, want), (scheme, W.shape, M)
        assert np.all(np.diff(got) >= 0) and got.min() >= 0 and got.max() < W.shape[0]
        assert np.all(W[got] > 0.0) or W.shape[0] == 1                 # zero-weight particles are never picked


@settings(max_examples=25, deadline=None, suppress_health_check=list(HealthCheck))
@given(case())
def test_weights_equal_oracle(c):
    W, _, seed = c
    rng = np.random.default_rng(seed)
    lw = np.log(np.maximum(W, 1e-320)) + 5.0 * rng.standard_normal()
    lw[W == 0.0] = -np.inf
    d, o = rs.Weights(lw=lw.copy()), orc.Weights(lw=lw.copy())
    assert np.allclose(d.W, o.W, rtol=1e-12, atol=0.0)
    assert abs(d.ESS / o.ESS - 1.0) < 1e-11 and abs(d.log_mean - o.log_mean) < 1e-11 * max(1.0, abs(o.log_mean))


@st.composite
def case2(draw):
    n = draw(st.sampled_from([2048, 4096]))
    return weights(draw, n), draw(st.integers(0, 2 ** 31 - 1))


@settings(max_examples=25, deadline=None, suppress_health_check=list(HealthCheck))
@given(case2())
def test_two_level_contract_matches_sequential_cdf(c):
    """The two-level exact CDF of the fused step loop (oracle restatement of k_ancestors2's
    contract) against the reference's sequential fp64 CDF: same ancestors except within
    rounding distance of a CDF step; never a zero-weight parent; monotone."""
    W, seed = c
    rng = np.random.default_rng(seed)
    with np.errstate(divide="ignore"):
        lw = np.log(W) + rng.normal() * 50.0              # the contract takes log-weights, any offset
    for su in ((rng.random() + np.arange(W.size)) / W.size,
               (rng.random(W.size) + np.arange(W.size)) / W.size):
        A = orc.inverse_cdf_2level(su, lw)
        ref = orc.inverse_cdf(su, W)
        assert np.all(np.diff(A) >= 0) and A.min() >= 0 and A.max() < W.size
        assert np.all(W[A] > 0.0)
        assert np.mean(A == ref) >= 0.995, np.mean(A == ref)
        assert np.max(np.abs(A - ref)) <= 2 or np.mean(A == ref) >= 0.999
