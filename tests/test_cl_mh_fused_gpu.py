"""The fused binnedCl Metropolis-Hastings kernels (a whole mh_run in one launch: csrc/cl_mhw_kernels.hpp,
one wave per 16 chains with everything of a step in registers, for a single data set with correlated
errors and x-factors; csrc/cl_mh_kernels.hpp, a block per 16 chains, for the other shapes) against the
step-wise engine they replace (five launches per step):
same draws, same arithmetic, so the chains must be identical bit for bit - in every proposal mode
of Proposal.py:127-164, for every shape of ClData_CosmoMC.calcLnLike (:267-315: plain, 11-point
calibration grid, analytic marginalisation, uncorrelated errors, two data sets, symmetrised-matrix
form from 128 bands up), with co-moments, traces and injected draws.  The reference's own chains
(G5) through the fused kernel: tests/test_gmh_gpu.py."""
import dataclasses

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def problem(kind):
    from oracle import refcl as RC
    if kind == "plain":                                   # x-factors, no marginalisation: 1 row per chain
        model, d, truth = RC.synth_cl_problem(nb=24, npar=6, calib=0.0, beam=0.0, lmax=700, seed=5)
        return model, [d], truth
    if kind == "calib11":                                 # 11-point calibration grid
        model, d, truth = RC.synth_cl_problem(nb=40, npar=7, calib=0.05, beam=0.0, lmax=900, seed=6)
        return model, [d], truth
    if kind == "beam11":                                  # 11-point beam grid (calib exactly 0)
        model, d, truth = RC.synth_cl_problem(nb=24, npar=5, calib=0.0, beam=0.02, lmax=700, seed=7)
        return model, [d], truth
    if kind == "grid121":                                 # 11 x 11 calibration x beam grid (:317-372): the wave kernel
        model, d, truth = RC.synth_cl_problem(nb=24, npar=6, calib=0.05, beam=0.02, lmax=700, seed=5)
        return model, [d], truth
    if kind == "grid121w":                                # ... on 5 column tiles and 20 bins
        model, d, truth = RC.synth_cl_problem(nb=72, npar=20, calib=0.05, beam=0.02, lmax=1500, seed=21)
        return model, [d], truth
    if kind == "sym128":                                  # 8 column tiles: the symmetrised matrix in the wave kernel
        model, d, truth = RC.synth_cl_problem(nb=128, npar=24, calib=0.05, beam=0.0, lmax=2400, seed=22)
        return model, [d], truth
    if kind == "analytic":                                # no x-factors: analytic calib + beam (TRIPLE rows)
        model, d, truth = RC.synth_cl_problem(nb=24, npar=6, calib=0.05, beam=0.02, lmax=700, seed=8,
                                              with_x=False)
        return model, [d], truth
    if kind == "diag":                                    # uncorrelated errors
        model, d, truth = RC.synth_cl_problem(nb=24, npar=6, calib=0.05, beam=0.0, lmax=700, seed=9)
        var = 1.0 / np.diag(d.N_inv)
        return model, [dataclasses.replace(d, N_inv=None, var=var)], truth
    if kind == "two":                                     # two data sets, different marginalisations
        model, d1, truth = RC.synth_cl_problem(nb=24, npar=6, calib=0.0, beam=0.0, lmax=700, seed=5)
        _, d2, _ = RC.synth_cl_problem(nb=40, npar=6, calib=0.05, beam=0.02, lmax=700, seed=11, with_x=False)
        return model, [d1, d2], truth
    if kind == "sym144":                                  # 9 column tiles: 8 waves, symmetrised matrix
        model, d, truth = RC.synth_cl_problem(nb=144, npar=12, calib=0.0, beam=0.0, lmax=2400, seed=12)
        return model, [d], truth
    if kind == "wide96":                                  # 6 column tiles: 8 waves, plain matrix
        model, d, truth = RC.synth_cl_problem(nb=96, npar=33, calib=0.0, beam=0.0, lmax=2400, seed=13)
        return model, [d], truth
    if kind == "p32":                                     # 32 bins: the largest co-moment triangle kept in registers
        model, d, truth = RC.synth_cl_problem(nb=64, npar=32, calib=0.0, beam=0.0, lmax=2000, seed=14)
        return model, [d], truth
    if kind == "big384":                                  # 24 column tiles, 11-point beam grid: the tables
        model, d, truth = RC.synth_cl_problem(nb=384, npar=40, calib=0.0, beam=0.02, lmax=4000, seed=16)
        return model, [d], truth                          # do not fit in LDS - the generic kernel serves it
    if kind == "densek":                                  # every band's window spans every bin: K is dense
        model, d, truth = RC.synth_cl_problem(nb=40, npar=12, calib=0.0, beam=0.0, lmax=900, seed=15)
        rng = np.random.default_rng(2)
        L = d.window.shape[2]
        w = d.window.copy()
        ell = np.arange(L, dtype=np.float64)
        for i in range(w.shape[0]):                       # broad smooth windows on top of the top hats
            w[i, 0, 2:] += 1e-3 * (1 + 0.3 * np.sin(ell[2:] / (7.0 + i))) * rng.uniform(0.5, 1.5)
        d2 = dataclasses.replace(d, window=w, win_min=np.full(w.shape[0], 2), win_max=np.full(w.shape[0], L - 1))
        return model, [d2], truth
    raise ValueError(kind)


def run(kind, fused, mode, C, track_cov, inject=None, seed=97):
    from mcmc_amd import Engine
    model, data, truth = problem(kind)
    P = model.nparam
    rng = np.random.default_rng(3)
    A = rng.normal(size=(P, P))
    cov = A @ A.T
    dg = np.sqrt(np.diag(cov))
    L = np.linalg.cholesky(cov / dg[:, None] / dg[None, :])
    block = mode != "corr_all"
    out = {}
    with Engine(seed=seed) as eng:
        eng.mh_set_fused("always" if fused else False)    # (these tests hold the kernels, not the size policy)
        eng.cl_load(data)
        eng.cl_set_binning(model.bins, model.shapefun)
        eng.mh_init_problem("cl", C, truth, np.full(P, 0.02), param_blocks=np.arange(P) if block else None,
                            track_cov=track_cov)
        assert eng.mh_fused_active == fused
        if mode != "uncorr_block":
            eng.mh_set_sqrt_matrix(L, rotate=(mode == "rotate_block"))
        if inject is None:
            eng.mh_run(37, thin=1)
            out["trace1"] = eng.mh_trace()
            eng.mh_run(45)                                # no trace; the graph path of the step-wise engine
            eng.mh_run(21, thin=3)
            out["trace2"] = eng.mh_trace()
        else:
            zn, zu = inject
            eng.mh_run(zn.shape[0], thin=1, inj_normals=zn, inj_uniforms=zu)
            out["trace1"] = eng.mh_trace()
        out["mom"] = eng.mh_moments()
        if track_cov:
            out["cov"] = eng.mh_covar(normalized=False)
            eng.mh_adapt_corr(rotate=block)               # newMCMC on the device, then more steps
            eng.mh_run(30)
            out["mom2"] = eng.mh_moments()
            out["L"] = eng.mh_sqrt_matrix()
    return out


def same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        x, y = a[k], b[k]
        if isinstance(x, dict):
            for kk in x:
                assert np.array_equal(x[kk], y[kk], equal_nan=True), (k, kk)
        elif isinstance(x, tuple):
            for i, (u, v) in enumerate(zip(x, y)):
                assert np.array_equal(u, v, equal_nan=True), (k, i)
        else:
            assert np.array_equal(x, y, equal_nan=True), k


@pytest.mark.parametrize("kind", ["plain", "calib11", "beam11", "grid121", "grid121w", "sym128", "analytic", "diag", "two",
                                  "sym144", "wide96", "densek", "big384"])
def test_fused_equals_stepwise_every_likelihood_shape(kind):
    C = 40 if kind not in ("sym144", "big384", "sym128") else 19    # not a multiple of the 16 chains of a block
    same(run(kind, True, "uncorr_block", C, False), run(kind, False, "uncorr_block", C, False))


@pytest.mark.parametrize("mode", ["uncorr_block", "corr_all", "rotate_block"])
def test_fused_equals_stepwise_every_proposal_mode(mode):
    same(run("plain", True, mode, 37, True), run("plain", False, mode, 37, True))


@pytest.mark.parametrize("kind", ["beam11", "calib11", "p32", "grid121", "sym128", "diag", "analytic", "two", "wide96"])
def test_fused_comoments_equal_stepwise(kind):
    """Running co-moments (covar(), MCMC.py:310-340) and the newMCMC stage they feed: in registers for
    the resident 4-wave kernels up to 32 parameters (odd and even counts, every form), in HBM
    otherwise (two data sets, 33 parameters) - the same numbers as the step-wise engine."""
    C = 21
    same(run(kind, True, "uncorr_block", C, True), run(kind, False, "uncorr_block", C, True))


def test_fused_equals_stepwise_with_injected_draws():
    rng = np.random.default_rng(12)
    n, C, P = 50, 5, 7
    zn = rng.standard_normal((n, C, P))
    zu = rng.uniform(size=(n, C))
    for mode in ("uncorr_block", "corr_all", "rotate_block"):
        same(run("calib11", True, mode, C, False, inject=(zn, zu)),
             run("calib11", False, mode, C, False, inject=(zn, zu)))


def test_dense_k_band_powers_match_the_oracle():
    """The compact K of the fused sampler and the k-step ranges of the batched band-power kernel on a
    data set whose windows meet every bin (no zero k-step to skip): lnL against the oracle."""
    from oracle import refcl as RC
    from mcmc_amd import Engine
    model, data, truth = problem("densek")
    rng = np.random.default_rng(6)
    q = truth[None, :] * rng.uniform(0.9, 1.1, (24, model.nparam))
    with Engine(seed=1) as eng:
        eng.cl_load(data)
        eng.cl_set_binning(model.bins, model.shapefun)
        got = eng.cl_lnlike(q)
    ref = np.array([RC.cl_lnlike(qq, model, data) for qq in q])
    np.testing.assert_allclose(got, ref, rtol=1e-12)


def test_problems_beyond_the_fused_kernels_fall_back_to_the_stepwise_engine():
    """More data sets than the fused kernels' argument block describes: the step-wise engine."""
    from oracle import refcl as RC
    from mcmc_amd import Engine
    data = []
    for k in range(5):
        model, d, truth = RC.synth_cl_problem(nb=24, npar=6, calib=0.0, beam=0.0, lmax=700, seed=5 + k)
        data.append(d)
    with Engine(seed=1) as eng:
        eng.cl_load(data)
        eng.cl_set_binning(model.bins, model.shapefun)
        eng.mh_init_problem("cl", 8, truth, np.full(6, 0.02), param_blocks=np.arange(6))
        assert not eng.mh_fused_active
        eng.mh_run(40)
        assert np.all(np.isfinite(eng.mh_moments()["lnPr"]))


def test_the_11x11_grid_runs_fused_where_that_is_the_faster_path():
    """The full calibration x beam grid (ClData_CosmoMC.py:317-372): 121 forms per chain and step.  The fused
    wave kernel holds a 16-chain tile on one CU; the step-wise engine's batched form spreads the rows over the
    chip.  Measured crossovers (cl_mhw_plan): fused from 2049 to 8192 chains, step-wise outside; "always"
    overrides; the 11-point grids stay fused at every size."""
    import os
    from oracle import refcl as RC
    from mcmc_amd import Engine
    if os.environ.get("MI355_CLMH_WAVE"):
        pytest.skip("MI355_CLMH_WAVE overrides the policy this test pins")
    model, d, truth = RC.synth_cl_problem(nb=24, npar=6, calib=0.05, beam=0.02, lmax=700, seed=5)
    model11, d11, truth11 = RC.synth_cl_problem(nb=24, npar=6, calib=0.05, beam=0.0, lmax=700, seed=5)
    with Engine(seed=1) as eng:
        eng.cl_load([d])
        eng.cl_set_binning(model.bins, model.shapefun)
        for C, want in ((40, False), (2048, False), (4096, True), (8192, True), (16384, False)):
            eng.mh_init_problem("cl", C, truth, np.full(6, 0.02), param_blocks=np.arange(6))
            assert eng.mh_fused_active == want, C
        eng.mh_set_fused("always")
        eng.mh_init_problem("cl", 40, truth, np.full(6, 0.02), param_blocks=np.arange(6))
        assert eng.mh_fused_active
        eng.mh_set_fused(True)
        eng.cl_load([d11])
        eng.cl_set_binning(model11.bins, model11.shapefun)
        for C in (40, 4096, 65536):
            eng.mh_init_problem("cl", C, truth11, np.full(6, 0.02), param_blocks=np.arange(6))
            assert eng.mh_fused_active, C


def test_comoment_update_forms_give_the_same_sums():
    """The running co-moments are kept in memory by adds that return nothing while a step has at most 2^23
    entries, by load / add / store batches beyond (gmh_comom_update).  Chain n draws from its own Philox
    stream whatever the job's size, so the first chains of a job beyond the limit (16 400 chains x 528 entries)
    must equal, bit for bit, the same chains run as a small job: fused wave kernel and step-wise engine."""
    from mcmc_amd import Engine
    model, data, truth = problem("p32")
    P = model.nparam
    out = {}
    for fused in (True, False):
        for C in (16400, 24):
            with Engine(seed=31) as eng:
                eng.mh_set_fused("always" if fused else False)
                eng.cl_load(data)
                eng.cl_set_binning(model.bins, model.shapefun)
                eng.mh_init_problem("cl", C, truth, np.full(P, 0.02), param_blocks=np.arange(P), track_cov=True)
                assert eng.mh_fused_active == fused
                eng.mh_run(40)
                eng.mh_run(23)
                out[(fused, C)] = eng.mh_covar(normalized=False)[:, :24].copy()      # [1, C, P, P]
    ref = out[(False, 24)]
    for key, got in out.items():
        assert np.array_equal(got, ref, equal_nan=True), key
    assert np.isfinite(ref).all() and np.abs(ref).max() > 0.0
