"""Kernel-logic checks WITHOUT a GPU: the kernel sources compiled for the fiber
emulator (tests/emu) are driven through the same C ABI and Python host layer at
small sizes.  This is test infrastructure for the GPU-less build container --
the parity claims proper are made by tests/test_gpu_parity.py on an MI355X.
Skipped when a GPU is visible (the real library is then loaded instead).
"""
import numpy as np
import pytest

import parity_cases as pc

pytestmark = pytest.mark.skipif(
    __import__("conftest").HAS_GPU, reason="GPU visible: covered by test_gpu_parity.py")


def test_emulator_is_what_is_loaded():
    from particles_amd import _lib
    assert b"EMULATOR" in _lib.lib().smc_version()


def test_weights(golden):
    pc.check_weights(golden)
    pc.check_weights_edges(3000)


@pytest.mark.parametrize("N,M", [(5000, 5000), (1000, 3777), (4097, 1024), (1, 50), (700, 1)])
def test_inverse_cdf(N, M):
    pc.check_inverse_cdf(N, M)


def test_reduce2_wide(monkeypatch):
    """k_reduce2w (1024 threads) against k_reduce2 at the smallest island it serves (1029 tiles, a ragged second chunk)."""
    pc.check_reduce2_wide(monkeypatch, (((1 << 20) + 4099, {}),), [np.array([0.3]), np.array([-0.2])])


def test_sort_window_and_fixup():
    pc.check_sort_window(sizes=(8193, 11003))


def test_inverse_cdf_dyadic():
    pc.check_inverse_cdf_dyadic(2048, 3000)


def test_schemes_vs_reference(golden):
    pc.check_schemes_vs_reference(golden)


@pytest.mark.parametrize("N,M", [(3000, 3000), (2049, 500), (300, 4100)])
def test_schemes_replay(N, M):
    pc.check_schemes_replay(N, M)


def test_schemes_philox():
    pc.check_schemes_philox(2500, 2500)
    pc.check_schemes_philox(1024, 3001)


def test_resampling_statistics():
    pc.check_resampling_statistics(600, 40)


def test_wmean_and_cov(golden):
    pc.check_wmean_and_cov(golden, N=20001)


def test_wquantiles(golden):
    pc.check_wquantiles(golden, N=9001)          # (the emulator runs the radix sort lane by lane)


def test_residual_killing(golden):
    pc.check_residual_killing(golden)


def test_unknown_scheme():
    pc.check_unknown_scheme()


def test_gather():
    pc.check_gather(3000, 1)
    pc.check_gather(500, 7)


def test_sqmc(golden, monkeypatch):
    pc.check_sqmc(golden, monkeypatch, philox_N=1024, philox_runs=2, philox_T=10,   # (emulated sorts are slow)
                  sorted_N=(1, 2, 64, 1024), ab_N=512, both_modes_for_all=False,
                  replay_cases=("sqmc_toy", "sqmc_mv2", "sqmc_mv3_guided"))


def test_sqmc_fused():
    pc.check_sqmc_fused(sizes=(2048,), T=4, audit_sizes=(4096,), islands_N=2048)


def test_sqmc_fused_small():
    pc.check_sqmc_fused_small(sizes=(32, 1024), T=4)


def test_sqmc_fused_multivariate():
    pc.check_sqmc_fused_mv(cases=((1024, 2), (2048, 5)), T=4, islands_N=256)


def test_indep_prod(golden):
    pc.check_indep_prod(golden)


def test_poisson_and_cox(golden):
    pc.check_poisson(golden)


def test_normal(golden):
    pc.check_normal(golden)
    pc.check_normal_philox(1001)


def test_mvn(golden):
    pc.check_mvn(golden)
    pc.check_mvn_large(300, 6)


@pytest.mark.parametrize("case,model,fk", [
    ("toy_systematic", "toy", "bootstrap"),
    ("toy_stratified", "toy", "bootstrap"),
    ("toy_multinomial", "toy", "bootstrap"),
    ("sv_systematic", "sv", "bootstrap"),
    ("lg_adaptive", "lg_adaptive", "bootstrap"),
    ("lg_guided", "lg_guided", "guided"),
    ("mv4_guided", "mv4", "guided"),
    ("mv4_boot", "mv4", "bootstrap"),
    ("mv32_guided", "mv32", "guided"),
    ("mv32_boot", "mv32", "bootstrap"),
    ("gordon_boot", "gordon", "bootstrap"),
    ("theta_boot", "theta", "bootstrap"),
    ("svlev_boot", "svlev", "bootstrap"),
    ("cox_boot", "cox", "bootstrap"),
])
def test_filter_replay(golden, case, model, fk):
    pc.check_filter_replay(golden, case, model, fk, T=12 if model.startswith("mv") else 25)


@pytest.mark.parametrize("chunks,N", [(2, 300), (4, 1100), (8, 1300)])
def test_mv_multi_chunk_loop(golden, monkeypatch, chunks, N):
    """k_propagate_mv walks `mv_chunks` chunks of 256 particles per workgroup, rows requested one
    iteration ahead and ancestor words two ahead; production picks 2 / 4 / 8 chunks only from
    N = 2^18 on (C4 runs 8), so the loop is forced here (SMC_MV_CHUNKS) and every particle of
    every step is audited against the oracle: whole and partial chunks, chunks wholly beyond N,
    more than one workgroup; guided, bootstrap and the collapsed weight."""
    monkeypatch.setenv("SMC_MV_CHUNKS", str(chunks))
    pf, _ = pc.check_filter_replay(golden, "mv32_guided", "mv32", "guided", T=4, N=N)
    assert "[mv_chunks=%d]" % chunks in pc.describe(pf)
    pf, _ = pc.check_filter_replay(golden, "mv4_boot", "mv4", "bootstrap", T=4, N=N + 1)
    assert "[mv_chunks=%d]" % chunks in pc.describe(pf)
    mk_dev, mk_orc = pc.MODELS["mv32"]
    pc.check_oracle_at_size("mv32", mk_dev, mk_orc, N, 3, "systematic", 1.0, fk="guided", d=32, replay=False)
    pc.check_mv_collapsed(N, 32, T=3)


def test_mv_diagonal_factors_equal_dense(monkeypatch):
    pc.check_mv_diag_equals_dense(monkeypatch, cases=((700, 32), (520, 20), (300, 4)), T=3)


@pytest.mark.parametrize("model,d,N", [("mvd8", 8, 600), ("mvd32", 32, 520)])
def test_mv_dense_model_against_the_oracle(model, d, N):
    """A MVLinearGauss whose covariances and observation matrix are full: the dense MFMA products of k_propagate_mv,
    every particle of every step against the oracle (guided and bootstrap, replayed and Philox draws)."""
    mk_dev, mk_orc = pc.MODELS[model]
    dy = mk_orc().dy
    pc.check_oracle_at_size(model, mk_dev, mk_orc, N, 3, "systematic", 1.0, fk="guided", d=d, dy=dy)
    pc.check_oracle_at_size(model, mk_dev, mk_orc, N, 3, "systematic", 1.0, fk="bootstrap", d=d, dy=dy, replay=False)


def test_mv_philox_kalman():
    pc.check_mv_kalman(2048, 4, "guided")
    pc.check_mv_kalman(1000, 6, "guided", scheme="stratified")


@pytest.mark.parametrize("case,model", [("gordon_boot", "gordon"), ("theta_boot", "theta"),
                                        ("cox_boot", "cox")])
def test_nonlinear_models_philox(golden, case, model):
    pc.check_model_philox_vs_oracle(golden, case, model, N=2000, runs=16)


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
    pc.check_heavy_parents(monkeypatch, T=6)


def test_wide_general(monkeypatch):
    pc.check_wide_general(monkeypatch)


def test_two_level_cdf(golden, monkeypatch):
    pc.check_describe()
    pc.check_two_level_stepwise()
    pc.check_two_level_cdf(golden, monkeypatch)


def test_graph_replay_matches_direct(golden):
    pc.check_graph_replay_matches_direct(golden, N=1500)


def test_normals_on_host_time_index(golden, monkeypatch):
    pc.check_normals_on_host_t(golden, monkeypatch, sizes=(3000, 2048), T=24)


def test_unfused_path(golden, monkeypatch):
    pc.check_unfused_path(golden, monkeypatch)


def test_small_filter_equals_general(golden, monkeypatch):
    pc.check_small_filter_equals_general(golden, monkeypatch, full=False)


def test_edge_sizes():
    pc.check_edge_sizes()


def test_filter_stepwise(golden):
    pc.check_filter_stepwise(golden)


def test_device_history(golden):
    pc.check_device_history(golden)
    pc.check_device_history_philox(1500, 9, golden)
    pc.check_device_history_philox(2048, 6, golden)             # two-level CDF path, history slots


@pytest.mark.parametrize("N,sigmaY", [(3000, 0.2), (4096, 0.2), (1024, 0.2), (8192, 0.002),
                                      (2048, 1e-4)])
def test_filter_philox_vs_c(golden, N, sigmaY):
    pc.check_filter_philox_vs_c(N, 12, golden, sigmaY)


def test_islands(golden):
    pc.check_islands(1500, 10, golden)
    pc.check_islands(1100, 6, golden, scheme="multinomial")


def test_resident_user_model(golden):
    pc.check_resident_user_model(golden)


def test_apf_and_guided_generic(golden):
    pc.check_apf_and_guided_generic(golden)


def test_permute_islands(golden):
    pc.check_permute_islands(2100, golden, tol=0.6, T=16, t0=6)  # (ragged second tile)
    pc.check_permute_islands(2048, golden, tol=0.6, T=16, t0=6)  # two-level path: partials travel too


def test_collectors_and_history(golden):
    pc.check_collectors_on_fused(golden)


def test_smc2_example():
    """examples/smc2_toy.py: SMC^2 assembled from the island primitives recovers sigma."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "smc2_toy", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                 "examples", "smc2_toy.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mean, sd = mod.main(T=12, Ntheta=6, Nx=64)
    assert np.isfinite(mean) and np.isfinite(sd) and 0.05 < mean < 2.0


def test_oracle_at_size_small():
    """check_oracle_at_size (the BASELINE-size oracle tests of the GPU suite) at emulator sizes:
    replay and Philox, the three schemes, d = 4 guided / bootstrap, islands."""
    toy, sv, mv4 = pc.MODELS["toy"], pc.MODELS["sv"], pc.MODELS["mv4"]
    pc.check_oracle_at_size("toy", *toy, 4096, 8, "systematic", 0.5)
    pc.check_oracle_at_size("toy", *toy, 2048, 5, "stratified", 0.7, replay=False)
    pc.check_oracle_at_size("sv", *sv, 3000, 4, "multinomial", 1.0)
    pc.check_oracle_at_size("sv", *sv, 2048, 4, "systematic", 1.0)
    pc.check_oracle_at_size("toy", *toy, 3001, 5, "systematic", 0.5)           # general counts, ragged last tile
    pc.check_oracle_at_size("toy", *toy, 2500, 4, "stratified", 0.7, replay=False)
    pc.check_oracle_at_size("mv4", *mv4, 1024, 3, "systematic", 1.0, fk="guided", d=4)
    pc.check_oracle_at_size("mv4", *mv4, 1024, 3, "systematic", 0.5, fk="guided", d=4, expect_resample=False)
    pc.check_oracle_at_size("toy", *toy, 2048, 4, "systematic", 0.5, replay=False, n_islands=3,
                            islands=(0, 2), seed=21)


def test_mv_collapsed_proposal():
    pc.check_mv_collapsed(2048, 4)
    pc.check_mv_collapsed(1024, 20, T=4)


def test_multinomial_spacings_regenerated():
    pc.check_device_spacings(sizes=(2048, 3000, 3001))


def test_strict_ancestors_equal_the_reference_cdf():
    pc.check_strict_ancestors(sizes=(3000,), op_N=5000, op_cases=8)


def test_strict_ancestors_more_tiles_and_models():
    """The two-launch strict step beyond a few tiles, a ragged size, the nonlinear model, every resampling step
    verified to have taken the fast path (smc_filter_strict_stats)."""
    pc.check_strict_ancestors(sizes=(9000, 8192), op_cases=0, schemes=("systematic", "multinomial"), model="sv", small=False,
                              T=4, ESSrmin=1.0)


def test_strict_ancestors_heavy_parents():
    """Strict step when a few parents (one parent) own most of the offspring: the systematic scatter search runs several
    windows per tile and carries the straddling parent across them."""
    for model in ("peaky", "collapsed"):
        pc.check_strict_ancestors(sizes=(9000, 8192), op_cases=0, schemes=("systematic", "stratified"), model=model, small=False,
                                  T=5, ESSrmin=1.0, replays=(False,))


def test_strict_verifies_every_step():
    pc.check_strict_never_leaves_the_fast_path([(3000, 4, "systematic", "toy", 0.5), (2500, 2, "multinomial", "sv", 1.0)], T=70)


def test_models_without_a_fused_descriptor(golden):
    pc.check_models_without_descriptor(golden)


def test_pickle_resume_of_device_filters():
    pc.check_pickle_resume()


def test_merged_reduce_equals_split(golden, monkeypatch):
    pc.check_merged_reduce_ab(golden, monkeypatch, sizes=(4096, 3000))


def test_apf_lingauss_fused(golden):
    pc.check_apf_lingauss(golden)


def test_apf_mvlingauss_fused(golden):
    pc.check_apf_mv(golden, big=((3000, 8, "systematic", 0.7), (2048, 5, "multinomial", 0.9)), philox_N=2048)


def test_device_sort():
    pc.check_device_sort(sizes=(1, 63, 64, 2047, 2048, 2049, 4096, 4097, 8192, 9001, 17000, 133001))   # (one / few / many sort tiles: the three forms of the scatter)


def test_smc2_device_theta_level():
    pc.check_smc2(Ntheta=32, Nx=64, big_Nx=(2048,), big_N=4, big_T=6)


def test_sharded_smc2_without_a_group_runs_the_wastefree_move():
    """ShardedSMC2(group=None, wastefree=True): the theta level is the host's (no device collective), and the base class's
    waste-free move must not ask the device theta level to resume (it did, and failed at the first move)."""
    from particles_amd import kalman, smc2
    rng = np.random.RandomState(4)
    x = np.cumsum(rng.standard_normal(10))
    y = [np.array([v]) for v in x + 0.3 * rng.standard_normal(10)]
    kw = dict(ssm_cls=lambda sigmaY: kalman.LinearGauss(rho=1.0, sigmaX=1.0, sigmaY=sigmaY, sigma0=1.0),
              prior=smc2.IndepPrior(sigmaY=("lognormal", np.log(0.8), 1.2)), data=y, init_Nx=64, N=6, seed=5, ESSrmin=0.9,
              wastefree=True, len_chain=2)
    alg = smc2.ShardedSMC2(group=None, **kw)
    assert not alg.device_theta
    alg.run()
    assert len(alg.move_times) >= 1 and np.isfinite(alg.logLt) and len(alg.theta["sigmaY"]) == 12


def test_smc2_wastefree_move():
    """The waste-free move (the reference's default, smc_samplers.py:669-684) on device filters, functionally:
    N chains x len_chain states all kept with their filters, the population assembled from the chains'
    batches (smc_filter_fast_forward + pack / unpack), evidence and posterior close to the standard move's and
    to Kalman's.  (The comparison with 32 recorded runs of the reference is a GPU test: 48 runs.)"""
    pc.check_smc2_wastefree_functional()


def test_partial_history_syncs_at_save_times_only():
    pc.check_partial_history(N=1500, T=16)


def test_rolling_history_on_device():
    pc.check_rolling_history(N=1500, T=14, ks=(2, 5))


def test_apf_and_guided_stochvol_fused(golden):
    pc.check_apf_fused(golden, apf2_cases=((2048, "systematic", 0.7), (2500, "multinomial", 0.7)))


def test_sequential_prefix_sums_in_parallel(monkeypatch):
    pc.check_seq_prefix_sums(sizes=(5000, 1 << 14), monkeypatch=monkeypatch)


def test_auxiliary_bootstrap_fused(golden):
    pc.check_apf_bootstrap(golden, big=((2048, "systematic", 0.7), (3000, "stratified", 0.8)))


def test_strict_ancestors_on_the_operator_path(monkeypatch):
    pc.check_strict_operator_path(monkeypatch, sizes=(1500, 5000))
