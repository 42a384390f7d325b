"""Parity tests proper: the HIP path (libsmc_hip.so, through the C ABI) against
the CPU oracle on a real MI355X.  Run with ``pytest -m gpu``.

Layers: (1) operator-level bit/ulp parity at sizes the oracle finishes in
seconds; (2) the fused step loop replaying the reference's own draws for the
golden cases (the oracle reproduces the reference bit for bit on those);
(3) BASELINE.json's full sizes through size-independent properties (Kalman
exact likelihood, determinism, sortedness / offspring bounds, island
independence).
"""
import numpy as np
import pytest

import parity_cases as pc

pytestmark = pytest.mark.gpu


def test_real_library_is_loaded():
    from particles_amd import _lib
    assert b"gfx950" in _lib.lib().smc_version()
    info = _lib.ctx().device_info()
    assert info["n_cu"] > 0


def test_weights(golden):
    pc.check_weights(golden)
    pc.check_weights_edges(3000)
    pc.check_weights_edges(1 << 20)


@pytest.mark.parametrize("N,M", [(5000, 5000), (1000, 3777), (4097, 1024), (700, 1),
                                 (1 << 20, 1 << 20), (1 << 22, 1 << 20), (100003, 1 << 21)])
def test_inverse_cdf(N, M):
    pc.check_inverse_cdf(N, M)


def test_inverse_cdf_dyadic():
    pc.check_inverse_cdf_dyadic(2048, 3000)
    pc.check_inverse_cdf_dyadic(1 << 18, 1 << 18)


def test_schemes_vs_reference(golden):
    pc.check_schemes_vs_reference(golden)


@pytest.mark.parametrize("N,M", [(3000, 3000), (2049, 500), (300, 4100), (1 << 20, 1 << 20)])
def test_schemes_replay(N, M):
    pc.check_schemes_replay(N, M)


def test_schemes_philox():
    pc.check_schemes_philox(2500, 2500)
    pc.check_schemes_philox(1 << 18, (1 << 18) + 5)


def test_resampling_statistics():
    pc.check_resampling_statistics(2000, 200)


def test_wmean_and_cov(golden):
    pc.check_wmean_and_cov(golden)


def test_wquantiles(golden):
    pc.check_wquantiles(golden)


def test_residual_killing(golden):
    pc.check_residual_killing(golden)


def test_unknown_scheme():
    pc.check_unknown_scheme()


def test_gather():
    pc.check_gather(1 << 20, 1)
    pc.check_gather(50000, 32)


def test_sqmc(golden, monkeypatch):
    pc.check_sqmc(golden, monkeypatch)


def test_sqmc_fused():
    pc.check_sqmc_fused(sizes=(2048, 1 << 16), T=6, audit_sizes=(1 << 14, 1 << 20), islands_N=1 << 13)


def test_sqmc_fused_large_grid():
    """N = 2^22: 4096 tiles (k_reduce2's chunked reduction), 2048 sort tiles."""
    import numpy as np
    from particles_amd import _lib, kalman, resampling as rs, state_space_models as ssm
    import particles_amd as pa
    from oracle import smc_oracle as orc
    y = [np.array([v]) for v in (0.3, -0.2, 0.5)]
    rs.set_rng("philox")
    try:
        pa.seed(61)
        c0 = _lib._counter + 1
        pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=1 << 22, qmc=True, collect="off", store_history=True)
        assert pf._fused
        pf.run()
        pc.audit_sqmc_history(pf, lambda: orc.ToySSM(0.2), "bootstrap", y, 61, c0)
    finally:
        rs.set_rng("numpy")


def test_sqmc_fused_small():
    pc.check_sqmc_fused_small()


def test_sqmc_fused_multivariate():
    pc.check_sqmc_fused_mv(cases=((1024, 2), (4096, 3), (1 << 16, 5), (1 << 18, 9)), T=5, islands_N=1 << 12)


def test_indep_prod(golden):
    pc.check_indep_prod(golden)


def test_poisson_and_cox(golden):
    pc.check_poisson(golden)


def test_normal(golden):
    pc.check_normal(golden)
    pc.check_normal_philox(1 << 20)


def test_mvn(golden):
    pc.check_mvn(golden)
    pc.check_mvn_large(50000, 32)


@pytest.mark.parametrize("case,model,fk", [
    ("toy_systematic", "toy", "bootstrap"),      # BASELINE config C1
    ("toy_stratified", "toy", "bootstrap"),
    ("toy_multinomial", "toy", "bootstrap"),
    ("sv_systematic", "sv", "bootstrap"),        # reduced C3
    ("sv_stratified", "sv", "bootstrap"),
    ("sv_multinomial", "sv", "bootstrap"),
    ("lg_adaptive", "lg_adaptive", "bootstrap"),
    ("lg_guided", "lg_guided", "guided"),
    ("mv4_guided", "mv4", "guided"),             # reduced C4
    ("mv4_boot", "mv4", "bootstrap"),
    ("mv32_guided", "mv32", "guided"),
    ("mv32_boot", "mv32", "bootstrap"),
    ("gordon_boot", "gordon", "bootstrap"),
    ("theta_boot", "theta", "bootstrap"),
    ("svlev_boot", "svlev", "bootstrap"),
    ("cox_boot", "cox", "bootstrap"),
])
def test_filter_replay(golden, case, model, fk):
    pc.check_filter_replay(golden, case, model, fk)


@pytest.mark.parametrize("case,model", [("gordon_boot", "gordon"), ("theta_boot", "theta"),
                                        ("cox_boot", "cox")])
def test_nonlinear_models_philox(golden, case, model):
    pc.check_model_philox_vs_oracle(golden, case, model, N=50000, runs=24)


@pytest.mark.parametrize("case,N", [("toy_stratified", 1024), ("toy_stratified", 4096),
                                    ("toy_systematic", 2048), ("toy_multinomial", 1024)])
def test_filter_replay_power_of_two(golden, case, N):
    """N = 2^k takes the closed-form offspring counts (systematic and stratified)."""
    pc.check_filter_replay(golden, case, "toy", "bootstrap", T=20, N=N)


def test_two_level_adaptive(golden):
    """Steps that do not resample on the two-level path (evidence increments across them,
    core.py:355-359) and the guided filter, N = 2^k."""
    pf, o = pc.check_filter_replay(golden, "lg_adaptive", "lg_adaptive", "bootstrap", T=40, N=2048)
    assert 0 < sum(o["rs_flag"]) < 39
    pc.check_filter_replay(golden, "lg_guided", "lg_guided", "guided", T=30, N=4096)
    pc.check_filter_replay(golden, "sv_stratified", "sv", "bootstrap", T=20, N=2048)


def test_heavy_parents(monkeypatch):
    pc.check_heavy_parents(monkeypatch)


def test_wide_general(monkeypatch):
    pc.check_wide_general(monkeypatch, sizes=(1500, 3000, 17 * 1024 + 1, 10 ** 6 + 3))


def test_two_level_cdf(golden, monkeypatch):
    pc.check_describe()
    pc.check_two_level_stepwise()
    pc.check_two_level_cdf(golden, monkeypatch)
    pc.check_two_level_large(golden, monkeypatch)


def test_graph_replay_matches_direct(golden):
    pc.check_graph_replay_matches_direct(golden, N=200000)


def test_normals_on_host_time_index(golden, monkeypatch):
    pc.check_normals_on_host_t(golden, monkeypatch, sizes=(5000, 4096, 1 << 20))


def test_unfused_path(golden, monkeypatch):
    pc.check_unfused_path(golden, monkeypatch)


def test_small_filter_equals_general(golden, monkeypatch):
    pc.check_small_filter_equals_general(golden, monkeypatch)


def test_edge_sizes():
    pc.check_edge_sizes()


def test_filter_stepwise(golden):
    pc.check_filter_stepwise(golden)


def test_device_history(golden):
    pc.check_device_history(golden)
    pc.check_device_history_philox(100000, 9, golden)
    pc.check_device_history_philox(1 << 17, 9, golden)          # two-level CDF path, history slots


@pytest.mark.parametrize("N,sigmaY", [(1 << 16, 0.2), (3000, 0.2), (1024, 0.2), (1 << 18, 0.002),
                                      (1 << 14, 1e-4), (100000, 0.2)])
def test_filter_philox_vs_c(golden, N, sigmaY):
    pc.check_filter_philox_vs_c(N, 100, golden, sigmaY)


def test_islands(golden):
    pc.check_islands(1500, 10, golden)
    pc.check_islands(1 << 16, 50, golden, scheme="systematic")
    pc.check_islands(5000, 20, golden, scheme="multinomial")


def test_permute_islands(golden):
    pc.check_permute_islands(100000, golden)
    pc.check_permute_islands(1 << 14, golden, tol=0.4)          # two-level path: partials travel too


def test_collectors_and_history(golden):
    pc.check_collectors_on_fused(golden)


def test_generic_path(golden):
    pc.check_generic_path(golden)


def test_resident_user_model(golden):
    pc.check_resident_user_model(golden)


def test_apf_and_guided_generic(golden):
    pc.check_apf_and_guided_generic(golden)


def test_mv_philox_kalman():
    pc.check_mv_kalman(1 << 16, 4, "guided")
    pc.check_mv_kalman(20000, 6, "guided", scheme="stratified")
    pc.check_mv_kalman(1 << 17, 32, "guided")
    pc.check_mv_kalman(1 << 18, 8, "bootstrap")


def test_c4_mv32_guided_full_size():
    """C4: MVLinearGauss_Guarniero d=32, GuidedPF, N = 2^20: determinism and the
    exact Kalman likelihood (guided + optimal proposal => tiny MC error)."""
    import particles_amd as pa
    from oracle import smc_oracle as orc
    from particles_amd import kalman
    from particles_amd import state_space_models as ssm
    d, T, N = 32, 20, 1 << 20
    rng = np.random.RandomState(3)
    om = orc.Guarniero(alpha=0.4, dx=d)
    x = np.zeros(d)
    y = []
    for t in range(T):
        x = (om.F @ x if t else np.zeros(d)) + rng.standard_normal(d)
        y.append((x + rng.standard_normal(d)).reshape(1, d))
    ll, _ = orc.kalman_loglik(om, y)
    fk = ssm.GuidedPF(ssm=kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=d), data=y)
    a = pa.SMC(fk=fk, N=N, seed=4)
    a.run()
    b = pa.SMC(fk=fk, N=N, seed=4)
    b.run()
    assert a.logLt == b.logLt and np.array_equal(a.X, b.X) and a.X.shape == (N, d)
    assert abs(a.logLt - ll) < 0.05, (a.logLt, ll)


# ---- BASELINE.json full sizes against the ORACLE ---------------------------

def test_two_level_contract_injected_full_sizes():
    """k_ancestors2 / k_reduce2 on uploaded weights (skewed, -inf, empty tile, collapsed):
    np.array_equal with the oracle's restatement at N = 2^12 .. 2^22."""
    pc.check_two_level_injected(sizes=(1 << 12, 1 << 17, 1 << 20, 1 << 22, 3000, 100000, 10 ** 6 + 1))


def test_c2_oracle_full_size():
    """C2's N = 2^20, T = 20, replay of numpy draws: every step audited, X / lw bit-exact."""
    mk_dev, mk_orc = pc.MODELS["toy"]
    pc.check_oracle_at_size("toy", mk_dev, mk_orc, 1 << 20, 20, "systematic", 0.5)
    pc.check_oracle_at_size("toy", mk_dev, mk_orc, 1 << 20, 6, "systematic", 0.5, replay=False)
    # population sizes users type: not powers of two, the last tile ragged (general counts)
    pc.check_oracle_at_size("toy", mk_dev, mk_orc, 10 ** 6, 8, "systematic", 0.5)
    pc.check_oracle_at_size("toy", mk_dev, mk_orc, 10 ** 5, 8, "stratified", 0.7)
    pc.check_oracle_at_size("toy", mk_dev, mk_orc, 10 ** 6 + 3, 5, "multinomial", 1.0)
    pc.check_oracle_at_size("toy", mk_dev, mk_orc, 10 ** 5, 6, "systematic", 0.5, replay=False)


@pytest.mark.parametrize("scheme", ["multinomial", "stratified", "systematic"])
def test_c3_oracle_full_size(scheme):
    """C3's N = 2^22 StochVol, ESSrmin = 1, each scheme against the oracle (not each other)."""
    mk_dev, mk_orc = pc.MODELS["sv"]
    pc.check_oracle_at_size("sv", mk_dev, mk_orc, 1 << 22, 5, scheme, 1.0)


def test_c4_oracle_d32():
    """C4's d = 32 guided filter at N = 2^17 against the oracle (replayed draws)."""
    mk_dev, mk_orc = pc.MODELS["mv32"]
    pc.check_oracle_at_size("mv32", mk_dev, mk_orc, 1 << 17, 4, "systematic", 1.0, fk="guided", d=32)
    pc.check_oracle_at_size("mv32", mk_dev, mk_orc, 1 << 17, 3, "systematic", 0.5, fk="guided", d=32,
                            expect_resample=False)
    pc.check_oracle_at_size("mv32", mk_dev, mk_orc, 1 << 17, 3, "systematic", 1.0, fk="bootstrap", d=32)


@pytest.mark.parametrize("log2N,chunks", [(18, 2), (19, 4), (20, 8)])
def test_c4_oracle_d32_every_chunk_count(log2N, chunks):
    """C4 as it is benchmarked: N = 2^20 runs k_propagate_mv with 8 chunks of 256 particles per
    workgroup (2^18: 2, 2^19: 4) -- a different prefetch loop from the single chunk of N <= 2^17.
    Every particle of every step against the oracle (kalman.py:339-356, distributions.py:931-969):
    guided and bootstrap on replayed numpy draws, guided on the production Philox streams (the
    oracle restates the (particle, pair) -> counter layout), and the collapsed weight."""
    mk_dev, mk_orc = pc.MODELS["mv32"]
    N = 1 << log2N
    import particles_amd as pa
    from particles_amd import kalman, state_space_models as ssm
    y = [np.zeros((1, 32))] * 2
    probe = pa.SMC(fk=ssm.GuidedPF(ssm=kalman.MVLinearGauss_Guarniero_etal(alpha=0.4, dx=32), data=y), N=N)
    assert "[mv_chunks=%d]" % chunks in pc.describe(probe)          # what production picks at this N
    pc.check_oracle_at_size("mv32", mk_dev, mk_orc, N, 3, "systematic", 1.0, fk="guided", d=32)
    pc.check_oracle_at_size("mv32", mk_dev, mk_orc, N, 3, "systematic", 1.0, fk="bootstrap", d=32)
    pc.check_oracle_at_size("mv32", mk_dev, mk_orc, N, 3, "systematic", 1.0, fk="guided", d=32, replay=False)
    pc.check_mv_collapsed(N, 32, T=3)


def test_mv_diagonal_factors_equal_dense(monkeypatch):
    """BASELINE C4's model has G = covX = covY = I: the element-wise form of the three triangular factors against the
    dense MFMA products, bit for bit, up to the benchmarked N = 2^20 (8 chunks per workgroup)."""
    pc.check_mv_diag_equals_dense(monkeypatch, cases=((1 << 20, 32), (1 << 17, 32), (70001, 20), (1 << 16, 4), (30000, 16)), T=3)


@pytest.mark.parametrize("model,d,N", [("mvd8", 8, 1 << 16), ("mvd32", 32, 1 << 18), ("mvd32", 32, 1 << 20)])
def test_mv_dense_model_against_the_oracle(model, d, N):
    """A MVLinearGauss whose covariances and observation matrix are full: the dense MFMA products of k_propagate_mv
    (72 per 16 particles at d = 32), every particle of every step against the oracle."""
    mk_dev, mk_orc = pc.MODELS[model]
    dy = mk_orc().dy
    pc.check_oracle_at_size(model, mk_dev, mk_orc, N, 3, "systematic", 1.0, fk="guided", d=d, dy=dy)
    if N < 1 << 20:
        pc.check_oracle_at_size(model, mk_dev, mk_orc, N, 3, "systematic", 1.0, fk="bootstrap", d=d, dy=dy, replay=False)


def test_mv_collapsed_proposal():
    pc.check_mv_collapsed(1 << 14, 4)
    pc.check_mv_collapsed(1 << 17, 32)
    pc.check_mv_collapsed(1 << 15, 20, T=4)


def test_c5_oracle_islands():
    """C5's share of one GPU, 32 islands x N = 2^18, production Philox streams: islands 0, 13
    and 31 audited step by step against the oracle's Philox restatement."""
    mk_dev, mk_orc = pc.MODELS["toy"]
    pc.check_oracle_at_size("toy", mk_dev, mk_orc, 1 << 18, 8, "systematic", 0.5, replay=False,
                            n_islands=32, islands=(0, 13, 31), seed=21)


# ---- BASELINE.json full sizes: size-independent properties -----------------

def _toy_data(golden, T):
    import particles_amd as pa
    from particles_amd import kalman
    pa.seed(42)
    x, y = kalman.ToySSM(0.2).simulate(T)
    return y


def test_c2_full_size_properties(golden):
    """C2: ToySSM, N = 2^20, T = 1000, systematic."""
    import particles_amd as pa
    from oracle import smc_oracle as orc
    from particles_amd import kalman
    from particles_amd import state_space_models as ssm
    N, T = 1 << 20, 1000
    y = _toy_data(golden, T)
    ll_kalman, _ = orc.kalman_loglik(orc.ToySSM(0.2), y)
    fk = ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y)
    runs = []
    for seed in (1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12):
        pf = pa.SMC(fk=fk, N=N, seed=seed)
        pf.run()
        runs.append(pf)
    a, b, c = runs[:3]
    assert a.logLt == b.logLt and np.array_equal(a.X, b.X)      # deterministic given the seed
    assert a.logLt != c.logLt
    # against the exact likelihood (Kalman filter): 12 independent runs give the estimator's own sd; every run
    # within 5 sd, the mean within 4 standard errors (+ the -var/2 bias of a log of an unbiased estimator)
    lls = np.array([pf.logLt for pf in runs[1:]])
    sd = lls.std(ddof=1)
    assert 0.02 < sd < 0.4, sd                                   # (measured: 0.15 -- sigma_Y = 0.2 makes the weights uneven)
    assert np.all(np.abs(lls - ll_kalman) < 5.0 * sd + 0.5 * sd * sd), (lls - ll_kalman, sd)
    assert abs(lls.mean() + 0.5 * sd * sd - ll_kalman) < 4.0 * sd / np.sqrt(len(lls)), (lls.mean() - ll_kalman, sd)
    print("C2 T=1000: logLt - Kalman: mean %.4f, sd %.4f over %d runs" % (lls.mean() - ll_kalman, sd, len(lls)))
    for pf in (a, c):
        ess = np.array(pf.summaries.ESSs)
        assert np.all((ess >= 1) & (ess <= N)) and all(pf.summaries.rs_flags[1:])
        A = pf.A
        assert np.all(np.diff(A) >= 0) and A[0] >= 0 and A[-1] < N
        # systematic: offspring counts within {floor, ceil}(N W) of the parent weights
    # one more step-level check at full size: offspring bounds for systematic
    from particles_amd import resampling as rs
    W = a.W
    np.random.seed(0)
    counts = np.bincount(rs.systematic(W), minlength=N)
    assert np.all((counts >= np.floor(N * W)) & (counts <= np.ceil(N * W)))


def test_c3_stochvol_schemes_full_size():
    """C3: StochVol, N = 2^22, ESSrmin = 1, the three schemes agree within MC error."""
    import particles_amd as pa
    from particles_amd import state_space_models as ssm
    N, T = 1 << 22, 100
    pa.seed(42)
    model = ssm.StochVol()
    x, y = model.simulate(T)
    lls = {}
    for scheme in ("multinomial", "stratified", "systematic"):
        pf = pa.SMC(fk=ssm.Bootstrap(ssm=model, data=y), N=N, resampling=scheme, ESSrmin=1.0,
                    seed=7)
        pf.run()
        assert all(pf.summaries.rs_flags[1:]) and np.isfinite(pf.logLt)
        assert np.all(np.diff(pf.A) >= 0)
        lls[scheme] = pf.logLt
    v = np.array(list(lls.values()))
    assert np.ptp(v) < 0.05, lls


def test_c5_islands_full_size(golden):
    """C5 (one GPU's share): 32 islands x N = 2^18, T = 100."""
    import particles_amd as pa
    from oracle import smc_oracle as orc
    from particles_amd import kalman
    from particles_amd import state_space_models as ssm
    y = _toy_data(golden, 100)
    ll_kalman, _ = orc.kalman_loglik(orc.ToySSM(0.2), y)
    pf = pa.SMC(fk=ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y), N=1 << 18, n_islands=32,
                seed=11, collect="off")
    pf.run()
    ll = pf.logLts_islands
    assert len(set(ll.tolist())) == 32
    assert np.max(np.abs(ll - ll_kalman)) < 0.3 and abs(ll.mean() - ll_kalman) < 0.05


def test_property_resampling_on_the_gpu():
    """The hypothesis properties of tests/test_property_resampling.py against the real library."""
    import test_property_resampling as tp
    tp.test_schemes_equal_q62_oracle()
    tp.test_weights_equal_oracle()


def test_smc2_example():
    """examples/smc2_toy.py: SMC^2 assembled from the island primitives recovers sigma."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "smc2_toy", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                 "examples", "smc2_toy.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mean, sd = mod.main(T=60, Ntheta=128, Nx=256)
    assert np.isfinite(mean) and np.isfinite(sd) and abs(mean - 0.3) < 0.15


RCCL_WORKER = r"""
import os, sys, json
sys.path.insert(0, {root!r})
import numpy as np
from particles_amd.distributed import Group
grp = Group(device_collective=True)
v = grp.gather_evidence(np.array([1.5 + grp.rank, -2.0 * grp.rank]))
if grp.rank == 0:
    print("RESULT " + json.dumps({{"v": v.tolist(), "path": grp.evidence_path}}))
grp.close()
"""


def _run_ranks(world, tmp_path, extra_env):
    import json, os, socket, subprocess, sys
    from conftest import ROOT
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER.format(root=ROOT))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SMC_HIP_DEVICE="0", **extra_env)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    return procs, outs, json


def test_rccl_gather_one_rank(tmp_path):
    """The RCCL branch executes for real: world size 1 -- ncclGetUniqueId, ncclCommInitRank with
    the 128-byte id passed by value, ncclAllGather of ncclDouble on the filter's stream."""
    procs, outs, json = _run_ranks(1, tmp_path, {})
    assert procs[0].returncode == 0, outs[0]
    res = json.loads([l for l in outs[0].splitlines() if l.startswith("RESULT ")][0][7:])
    assert res["path"] == "rccl" and res["v"] == [1.5, -0.0]


def test_rccl_two_ranks_one_gpu(tmp_path):
    """Two ranks on the ONE GPU of this box: the unique id travels over the TCP rendezvous,
    both ranks call ncclCommInitRank -- RCCL either accepts (then the all-gather must be right)
    or refuses two ranks on one device; the refusal must surface as an error (no silent
    fallback), and with SMC_ALLOW_HOST_GATHER=1 as a labelled host fallback with the right values."""
    procs, outs, json = _run_ranks(2, tmp_path, {"SMC_ALLOW_HOST_GATHER": "1"})
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    res = json.loads([l for l in outs[0].splitlines() if l.startswith("RESULT ")][0][7:])
    assert res["v"] == [1.5, -0.0, 2.5, -2.0]
    assert res["path"] == "rccl" or res["path"].startswith("host-fallback: ")
    if res["path"] != "rccl":
        procs, outs, json = _run_ranks(2, tmp_path, {})
        assert all(p.returncode != 0 for p in procs)               # no opt-in: loud failure
        assert "RCCL evidence gather unavailable" in "".join(outs)


def test_multinomial_spacings_regenerated():
    pc.check_device_spacings(sizes=(2048, 3000, 1 << 14, 10 ** 6 + 1, 1 << 22))


def test_strict_ancestors_equal_the_reference_cdf():
    pc.check_strict_ancestors(sizes=(3000, 1 << 17, (1 << 18) + 333), op_N=1 << 20, op_cases=100)


def test_strict_ancestors_heavy_parents():
    for model in ("peaky", "collapsed"):
        pc.check_strict_ancestors(sizes=(9000, 1 << 18), op_cases=0, schemes=("systematic", "stratified", "multinomial"), model=model,
                                  small=False, T=5, ESSrmin=1.0)


def test_sort_window_and_fixup():
    pc.check_sort_window(sizes=(8193, 50001, (1 << 17) + 5, 1 << 20), window_min=8193)


def test_strict_verifies_every_step():
    pc.check_strict_never_leaves_the_fast_path(
        [(5000, 4, "systematic", "toy", 0.5), (3000, 8, "multinomial", "toy", 0.5), (70000, 2, "stratified", "sv", 1.0),
         (1 << 16, 2, "systematic", "peaky", 1.0)], T=1000)
    # more than 1024 tiles (k_reduce2's prefixes are the estimate in front of a tile): even and collapsed mass
    pc.check_strict_never_leaves_the_fast_path(
        [((1 << 21) + 5, 1, "systematic", "toy", 0.5), (1 << 21, 1, "stratified", "collapsed", 1.0),
         (1 << 21, 1, "multinomial", "peaky", 1.0)], T=40)


def test_models_without_a_fused_descriptor(golden):
    pc.check_models_without_descriptor(golden)


def test_pickle_resume_of_device_filters():
    pc.check_pickle_resume(sizes=(700, 3000, 1 << 18))


def test_strict_ancestors_c2_full_size():
    """The north star's literal guarantee at the size it is benchmarked on: C2's N = 2^20, the fused strict loop --
    at every resampling step A_t == inverse_cdf(su_t, W_{t-1}) of the reference (resampling.py:484-509), replayed and
    Philox draws, np.array_equal (no near-tie allowance)."""
    pc.check_strict_ancestors(sizes=(1 << 20,), op_cases=0, schemes=("systematic",), small=False, T=6, ESSrmin=0.5)


@pytest.mark.parametrize("scheme", ["systematic", "stratified", "multinomial"])
def test_strict_ancestors_c3_full_size(scheme):
    """... and at C3's: StochVol, N = 2^22 (4096 tiles: k_reduce2 in front), ESSrmin = 1, each scheme."""
    pc.check_strict_ancestors(sizes=(1 << 22,), op_cases=0, schemes=(scheme,), model="sv", small=False, T=4, ESSrmin=1.0)


def test_merged_reduce_equals_split(golden, monkeypatch):
    pc.check_merged_reduce_ab(golden, monkeypatch, sizes=(4096, 3000, 1 << 20))


def test_apf_lingauss_fused(golden):
    pc.check_apf_lingauss(golden)


def test_apf_mvlingauss_fused(golden):
    pc.check_apf_mv(golden, big=((3000, 8, "systematic", 0.7), (1 << 13, 32, "stratified", 0.8), (2048, 5, "multinomial", 0.9),
                                 (1 << 17, 32, "systematic", 0.7), (1 << 19, 16, "systematic", 0.7)))


def test_device_sort():
    pc.check_device_sort(sizes=(1, 64, 2049, 4096, 4097, 8192, 8193, 16384, 16385, 50001, 131072, 131073, (1 << 20) + 3, 1 << 22))


def test_smc2_device_theta_level():
    pc.check_smc2(Ntheta=256, Nx=512, T=60, big_Nx=(2048, 3000))


@pytest.mark.parametrize("sharded", [False, True])
def test_smc2_pinned_to_the_reference(golden, sharded):
    """Device SMC^2 (one-GPU class and the sharded class at world 1) against 24 recorded runs of the
    reference's own SMC2: evidence, posterior moments, ESS trajectory, number of moves within 3 SE."""
    pc.check_smc2_vs_reference(golden, R=48, sharded=sharded)


def test_smc2_wastefree_pinned_to_the_reference(golden):
    """The waste-free move (the reference's default, smc_samplers.py:669-684) on device filters: 32 chains of 4
    states, every state kept with its filter -- against 32 recorded runs of the reference's own waste-free SMC2."""
    pc.check_smc2_vs_reference(golden, R=48, wastefree=True)


def test_partial_history_syncs_at_save_times_only():
    pc.check_partial_history(N=100000, T=40)


def test_rolling_history_on_device():
    pc.check_rolling_history()
    pc.check_rolling_history(N=1 << 17, T=12, ks=(3,))


MIGRATE_WORKER = r"""
import os, sys, json
sys.path.insert(0, {root!r})
import numpy as np
import particles_amd as pa
from particles_amd import kalman, state_space_models as ssm
from particles_amd.distributed import Group
grp = Group(device_collective=True)
rng = np.random.RandomState(42)
y = [np.array([v]) for v in np.cumsum(rng.standard_normal(12))]
fk = ssm.Bootstrap(ssm=kalman.ToySSM(0.2), data=y)
mk = lambda: pa.SMC(fk=fk, N=4096, seed=77, n_islands=6, collect="off")
src = np.array([4, 4, 0, 5, 1, 2])
a = mk(); a.step_async(5); grp.migrate_islands(a, src); a.step_async(7)
b = mk(); b.step_async(5); b.permute_islands(src); b.step_async(7)
print("RESULT " + json.dumps({{"same": bool(np.array_equal(a.logLts_islands, b.logLts_islands)
                                               and np.array_equal(a._get(0, 3), b._get(0, 3))),
                               "path": grp.evidence_path}}))
grp.close()
"""


def test_island_migration_through_rccl(tmp_path):
    """Group.migrate_islands with the device collective: packed island states through
    smc_comm_alltoallv (grouped ncclSend / ncclRecv; world size 1: the self block) must equal
    SMC.permute_islands bit for bit, and the run continues identically."""
    import json, os, socket, subprocess, sys
    from conftest import ROOT
    script = tmp_path / "migrate_worker.py"
    script.write_text(MIGRATE_WORKER.format(root=ROOT))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), SMC_HIP_DEVICE="0")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stdout + p.stderr
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    assert res["path"] == "rccl" and res["same"]


def test_sharded_smc2_over_rccl(tmp_path):
    """ShardedSMC2 on the device: world 1 with the RCCL communicator -- the theta level replicated on the
    device, fed by ncclAllGather of the evidence increments enqueued behind every step
    (smc_filter_theta_enable_sharded), migration through smc_comm_alltoallv: the same run as the one-GPU
    class bit for bit.  Then two ranks on this box's one GPU (RCCL if it accepts two ranks per device, the
    labelled host fallback with its per-step host theta level otherwise: same decisions, weights to
    rounding); larger filters (N_x = 4096: the multi-kernel step) as well."""
    import numpy as np
    from test_distributed_cpu import _run_smc2_world
    for nx in ("128", "4096"):
        one = _run_smc2_world(1, tmp_path, SMC_TEST_RCCL="1", SMC_TEST_NX=nx)
        two = _run_smc2_world(2, tmp_path, SMC_TEST_RCCL="1", SMC_ALLOW_HOST_GATHER="1", SMC_TEST_NX=nx)
        assert one["path"] == "rccl" and one["device_theta"] and one["moves"] >= 1
        for k in ("lw", "theta", "logLt", "ESSs", "logLts", "moves", "Nx"):
            assert one["single"][k] == one[k], k
        assert two["path"] == "rccl" or two["path"].startswith("host-fallback: ")
        if two["device_theta"]:
            for k in ("lw", "theta", "logLt", "ESSs", "moves", "acc"):
                assert one[k] == two[k], k
        else:
            for k in ("theta", "moves", "acc"):
                assert one[k] == two[k], k
            assert np.allclose(one["lw"], two["lw"], rtol=0, atol=1e-9) and abs(one["logLt"] - two["logLt"]) < 1e-9
            assert np.allclose(one["ESSs"], two["ESSs"], rtol=1e-9)


def test_apf_and_guided_stochvol_fused(golden):
    pc.check_apf_fused(golden, apf2_cases=((2048, "systematic", 0.7), (4096, "stratified", 0.9),
                                         (4096, "multinomial", 0.7), (1 << 15, "systematic", 0.7), (30000, "stratified", 0.7)))


@pytest.mark.gpu
def test_sequential_prefix_sums_in_parallel(monkeypatch):
    import parity_cases as pc
    pc.check_seq_prefix_sums(sizes=(5000, 1 << 16, (1 << 18) + 77), monkeypatch=monkeypatch)


@pytest.mark.gpu
def test_auxiliary_bootstrap_fused(golden):
    import parity_cases as pc
    pc.check_apf_bootstrap(golden)


@pytest.mark.gpu
def test_strict_ancestors_on_the_operator_path(monkeypatch):
    import parity_cases as pc
    pc.check_strict_operator_path(monkeypatch, sizes=(1500, 1 << 13, 1 << 17))
