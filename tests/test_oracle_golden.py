"""The CPU oracle against the reference's own outputs (tests/golden/*.npz).

Bit-for-bit: the oracle consumes the numpy legacy generator in the same order
as the reference and evaluates the same fp64 expressions.  (no GPU needed)
"""
import numpy as np
import pytest

from oracle import smc_oracle as orc

MODELS = {
    "toy": lambda: orc.ToySSM(sigma=0.2),
    "sv": lambda: orc.StochVol(),
    "lg_adaptive": lambda: orc.LinGauss(rho=0.9, sigmaX=1.0, sigmaY=1.5),
    "lg_guided": lambda: orc.LinGauss(rho=0.9, sigmaX=1.0, sigmaY=0.2),
    "lg_apf": lambda: orc.LinGauss(rho=0.9, sigmaX=1.0, sigmaY=0.6),
    "mv4": lambda: orc.Guarniero(alpha=0.4, dx=4),
    "mv32": lambda: orc.Guarniero(alpha=0.4, dx=32),
    "gordon": lambda: orc.Gordon(),
    "theta": lambda: orc.ThetaLogistic(),
    "svlev": lambda: orc.StochVolLeverage(phi=-0.5),
    "cox": lambda: orc.DiscreteCox(mu=0.5, sigma=0.4, phi=0.9),
}

CASES = ([("toy_%s" % s, "toy", "bootstrap") for s in ("systematic", "stratified", "multinomial")]
         + [("sv_%s" % s, "sv", "bootstrap") for s in ("systematic", "stratified", "multinomial")]
         + [("lg_adaptive", "lg_adaptive", "bootstrap"), ("lg_guided", "lg_guided", "guided"),
            ("mv4_guided", "mv4", "guided"), ("mv4_boot", "mv4", "bootstrap"),
            ("mv32_guided", "mv32", "guided"), ("mv32_boot", "mv32", "bootstrap"),
            ("gordon_boot", "gordon", "bootstrap"), ("theta_boot", "theta", "bootstrap"),
            ("svlev_boot", "svlev", "bootstrap"), ("cox_boot", "cox", "bootstrap"),
            ("sv_guided", "sv", "guided"), ("sv_apf", "sv", "apf"), ("lg_apf", "lg_apf", "apf"),
            ("mv_apf", "mv4", "apf"), ("sv_apfboot", "sv", "apfboot"), ("lg_apfboot", "lg_apf", "apfboot")])


SQMC_CASES = [("sqmc_toy", lambda: orc.LinGauss(rho=0.9, sigmaX=1.0, sigmaY=0.5), "bootstrap"),
              ("sqmc_sv", lambda: orc.StochVol(), "bootstrap"),
              ("sqmc_guided", lambda: orc.LinGauss(rho=0.9, sigmaX=1.0, sigmaY=0.3), "guided"),
              ("sqmc_mv2", lambda: orc.Guarniero(alpha=0.4, dx=2), "bootstrap"),
              ("sqmc_mv3_guided", lambda: orc.Guarniero(alpha=0.4, dx=3), "guided")]


def test_hilbert_vs_reference(golden):
    """hilbert_array / hilbert_sort (hilbert.py:13-58), int64 wrap-around from d = 4 included."""
    g = golden("hilbert")
    for d in (2, 3, 5, 8):
        assert np.array_equal(orc.hilbert_array(g["xint%d" % d]), g["h%d" % d])
        assert np.array_equal(orc.hilbert_sort(g["x%d" % d]), g["order%d" % d])
    assert (g["h5"] < 0).any() and (g["h8"] < 0).any()


@pytest.mark.parametrize("case,mk,fk", SQMC_CASES)
def test_sqmc_bit_exact_vs_reference(golden, case, mk, fk):
    """SMC(qmc=True) of the reference (core.py:315-349) on its recorded Sobol' points."""
    g = golden(case)
    out = orc.run_sqmc(mk(), list(g["y"]), int(g["N"]), [g["u0"]] + list(g["u"]), fk=fk)
    assert np.array_equal(np.array(out["rs_flag"]), g["rs_flags"])
    assert np.array_equal(np.array(out["ESS"]), g["ESSs"])
    assert np.array_equal(np.array(out["logLt"]), g["logLts"])
    assert np.array_equal(out["X"], g["X"])
    assert np.array_equal(out["A"], g["A"])
    assert np.array_equal(out["lw"], g["lw"])
    assert np.array_equal(out["W"], g["W"])


@pytest.mark.parametrize("case,model,fk", CASES)
def test_filter_bit_exact_vs_reference(golden, case, model, fk):
    g = golden(case)
    np.random.seed(int(g["run_seed"]))
    out = orc.run_filter(MODELS[model](), list(g["y"]), int(g["N"]),
                         scheme=str(g["scheme"]), ESSrmin=float(g["ESSrmin"]), fk=fk)
    assert np.array_equal(np.array(out["rs_flag"]), g["rs_flags"])
    assert np.array_equal(np.array(out["ESS"]), g["ESSs"])
    assert np.array_equal(np.array(out["logLt"]), g["logLts"])
    assert out["final_logLt"] == float(g["logLt"])
    assert np.array_equal(out["X"], g["X"])
    assert np.array_equal(out["A"], g["A"])
    assert np.array_equal(out["lw"], g["lw"])
    assert np.array_equal(out["W"], g["W"])


def test_history_and_genealogy_vs_reference(golden):
    """Every step's X, A, lw, W and compute_trajectories (smoothing.py:181-219)."""
    g = golden("history")
    np.random.seed(int(g["run_seed"]))
    N = int(g["N"])
    out = orc.run_filter(MODELS["lg_adaptive"](), list(g["y"]), N, scheme=str(g["scheme"]),
                         ESSrmin=float(g["ESSrmin"]), keep=True)
    h = out["hist"]
    assert out["final_logLt"] == float(g["logLt"]) and bool(g["A0_is_none"]) and h["A"][0] is None
    assert np.array_equal(np.array(h["X"]), g["hist_X"])
    assert np.array_equal(np.array(h["lw"]), g["hist_lw"])
    assert np.array_equal(np.array(h["W"]), g["hist_W"])
    # steps that did not resample: the reference stores arange (core.py:336)
    A = [None] + [a if a is not None else np.arange(N) for a in h["A"][1:]]
    assert np.array_equal(np.array(A[1:]), g["hist_A"])
    assert 0 < int(g["rs_flags"].sum()) < len(A) - 1          # both branches are exercised
    assert np.array_equal(orc.compute_trajectories(A, N), g["trajectories"])


def test_replay_tape_reproduces_run(golden):
    g = golden("toy_stratified")
    np.random.seed(int(g["run_seed"]))
    rec = orc.RecordingRNG()
    a = orc.run_filter(orc.ToySSM(0.2), list(g["y"]), 1000, "stratified", 0.5, rng=rec)
    b = orc.run_filter(orc.ToySSM(0.2), list(g["y"]), 1000, "stratified", 0.5,
                       rng=orc.ReplayRNG(rec.tape))
    assert a["final_logLt"] == b["final_logLt"] == float(g["logLt"])
    assert np.array_equal(a["X"], b["X"])


@pytest.mark.parametrize("name,model", [("toy", "toy"), ("mv32", "mv32")])
def test_kalman_kat(golden, name, model):
    g = golden("kalman_" + name)
    ll, means = orc.kalman_loglik(MODELS[model](), list(g["y"]))
    assert ll == pytest.approx(float(g["loglik"]), rel=1e-13)
    assert np.allclose(means, g["filt_means"], rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("scheme", ["systematic", "stratified", "multinomial"])
@pytest.mark.parametrize("M", [1500, 400, 4000])
def test_resampling_schemes(golden, scheme, M):
    g = golden("resampling")
    np.random.seed(11)
    A = orc.resampling(scheme, g["W"], M=M)
    assert A.dtype == np.int64 and np.array_equal(A, g["A_%s_%d" % (scheme, M)])


def test_residual_killing(golden):
    g = golden("resampling2")
    for M in (1500, 400, 4000):
        np.random.seed(11)
        assert np.array_equal(orc.residual(g["W"], M), g["A_residual_%d" % M])
    np.random.seed(11)
    assert np.array_equal(orc.killing(g["W"], 1500), g["A_killing_1500"])
    for M in (1500, 400, 4000):
        np.random.seed(11)
        assert np.array_equal(orc.ssp(g["W"], M), g["A_ssp_%d" % M])
    np.random.seed(11)
    assert np.array_equal(orc.residual(g["W_integral"], 64), g["A_residual_integral"])
    with pytest.raises(ValueError):
        orc.killing(g["W"], 10)


def test_uniform_spacings(golden):
    np.random.seed(5)
    su = orc.uniform_spacings_from(np.random.rand(101))
    assert np.array_equal(su, golden("resampling")["spacings_100"])


def test_unknown_scheme():
    with pytest.raises(ValueError, match="not a valid resampling scheme"):
        orc.resampling("bogus", np.ones(4) / 4)


def test_weights(golden):
    g = golden("weights")
    lw = g["lw_in"].copy()
    w = orc.Weights(lw=lw)
    assert np.array_equal(lw, g["lw_after"])          # NaN -> -inf, in place
    assert np.array_equal(w.W, g["W"]) and w.ESS == g["ESS"] and w.log_mean == g["log_mean"]
    assert orc.log_sum_exp(w.lw) == g["lse"] and orc.log_mean_exp(w.lw) == g["lme"]
    assert orc.essl(w.lw) == g["essl"]
    assert np.array_equal(orc.exp_and_normalise(w.lw), g["ean"])
    assert orc.log_mean_exp(g["lw2"], W=w.W) == g["lme_w"]
    mv = orc.wmean_and_var(w.W, np.sin(np.arange(1000.0)))
    assert mv["mean"] == g["wmean"] and mv["var"] == g["wvar"]


def test_wmean_and_cov_and_structured_arrays(golden):
    """resampling.py:341-380, 420-442 restated: bit for bit against the reference's own outputs."""
    g = golden("moments_cov")
    m, c = orc.wmean_and_cov(g["W"], g["X"])
    assert np.array_equal(m, g["mean5"]) and np.array_equal(c, g["cov5"])
    m1, c1 = orc.wmean_and_cov(g["W"], g["X"][:, 0])
    assert m1 == g["mean1"] and np.array_equal(np.array(c1), g["cov1"])
    xs = np.zeros(len(g["W"]), dtype=[("a", float), ("b", float)])
    xs["a"], xs["b"] = g["sa"], g["sb"]
    mv = orc.wmean_and_var_str_array(g["W"], xs)
    assert mv["mean"]["a"] == g["sm_a"] and mv["mean"]["b"] == g["sm_b"] and mv["var"]["a"] == g["sv_a"] and mv["var"]["b"] == g["sv_b"]
    wq = orc.wquantiles_str_array(g["W"], xs, alphas=(0.1, 0.5, 0.9))
    assert np.array_equal(np.array(wq["a"]), g["sq_a"]) and np.array_equal(np.array(wq["b"]), g["sq_b"])


def test_wquantiles(golden):
    g = golden("weights")
    x = np.sin(np.arange(1000.0))
    assert np.array_equal(np.array(orc.wquantiles(g["W"], x, alphas=(0.05, 0.25, 0.5, 0.75, 0.999))), g["wq"])
    x2 = np.stack([x, np.cos(3.0 * np.arange(1000.0))], axis=1)
    assert np.array_equal(orc.wquantiles(g["W"], x2), g["wq2"])


def test_weights_edge_cases():
    with np.errstate(all="ignore"):
        w = orc.Weights(lw=np.full(5, -np.inf))
    assert np.isnan(w.W).all() and np.isnan(w.ESS) and np.isnan(w.log_mean)
    e = orc.Weights()
    assert e.N == 0 and not hasattr(e, "W")
    assert np.array_equal(e.add(np.zeros(3)).W, np.full(3, 1 / 3))


def test_distributions(golden):
    g = golden("dists")
    assert np.array_equal(orc.normal_logpdf(g["x"], g["loc"], 0.7), g["normal_logpdf"])
    assert np.array_equal(orc.normal_logpdf(np.array([0.3]), 0.0, np.exp(0.5 * g["x"])),
                          g["normal_logpdf_sv"])
    np.random.seed(10)
    assert np.array_equal(orc.normal_rvs(g["loc"], 0.7, np.random.standard_normal(64)),
                          g["normal_rvs"])
    L = np.linalg.cholesky(g["cov"])
    assert np.array_equal(orc.mvnormal_logpdf(g["x5"], g["mloc"], 1.3, L), g["mv_logpdf"])
    np.random.seed(12)
    assert np.array_equal(orc.mvnormal_rvs(g["mloc"], 1.3, L, np.random.standard_normal((40, 5))),
                          g["mv_rvs"])


# ---- the Q62 fixed-point CDF contract ------------------------------------

def test_q62_matches_sequential_on_dyadic_weights():
    """Exactly summable weights k/2^30: every summation order gives the same
    CDF, so Q62 must reproduce the reference ordering 100%."""
    rng = np.random.default_rng(0)
    N = 4096
    k = rng.integers(0, 2 ** 19, size=N)
    k[-1] += 2 ** 30 - k.sum()
    W = k / 2.0 ** 30
    assert W.sum() == 1.0
    for M in (N, 1000, 10000):
        for scheme in ("systematic", "stratified"):
            u = rng.random(orc.N_UNIFORMS[scheme](M))
            su = orc.sorted_uniforms(scheme, M, u)
            assert np.array_equal(orc.inverse_cdf(su, W), orc.inverse_cdf_q62(su, W))


@pytest.mark.parametrize("scheme", ["systematic", "stratified", "multinomial"])
def test_q62_vs_sequential_near_tie_audit(scheme):
    rng = np.random.default_rng(1)
    N = 1 << 16
    W = orc.exp_and_normalise(2.0 * rng.standard_normal(N))
    mism = 0
    for rep in range(5):
        u = rng.random(orc.N_UNIFORMS[scheme](N))
        su = orc.sorted_uniforms(scheme, N, u)
        try:
            A_seq = orc.inverse_cdf(su, W)
        except IndexError:
            continue
        A_q = orc.inverse_cdf_q62(su, W)
        n, ok = orc.audit_near_ties(su, W, A_seq, A_q)
        assert ok
        mism += n
    assert mism <= 5


def test_q62_c_matches_numpy():
    rng = np.random.default_rng(2)
    W = orc.exp_and_normalise(3.0 * rng.standard_normal(5000))
    W[7] = 0.0
    su = orc.su_stratified(3000, rng.random(3000))
    import ctypes
    A = np.empty(3000, dtype=np.int64)
    orc.clib().orc_inverse_cdf_q62(orc._dp(su), orc._dp(W), 3000, 5000,
                                   A.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    assert np.array_equal(A, orc.inverse_cdf_q62(su, W))


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    import ctypes
    for ctr, key, want in kat:
        got = [int(v) for v in orc.philox4x32_10(*ctr, *key)]
        assert tuple(got) == want
        c = (ctypes.c_uint32 * 4)(*ctr)
        k = (ctypes.c_uint32 * 2)(*key)
        o = (ctypes.c_uint32 * 4)()
        orc.clib().orc_philox4x32_10(c, k, o)
        assert tuple(o) == want


def test_philox_normals_moments():
    z = orc.philox_normals(123, 200001, t=5)
    assert z.shape == (200001,)
    assert abs(z.mean()) < 0.01 and abs(z.var() - 1) < 0.01
    assert abs((z ** 4).mean() - 3) < 0.1


def test_c_philox_filter_close_to_kalman(golden):
    g = golden("kalman_toy")
    y = np.ascontiguousarray(np.squeeze(g["y"]))
    summ = np.zeros(4 * y.size)
    ll = orc.clib().orc_toy_filter_philox(orc._dp(y), y.size, 20000, 1.0, 1.0, 0.2, 1.0,
                                          0.5, 123, orc._dp(summ))
    assert abs(ll - float(g["loglik"])) < 0.5
