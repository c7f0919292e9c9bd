"""The fused beam sampler (csrc/beam_mh_kernels.hpp: every MH step of the chains of one map inside one launch)
against the launch-per-step engine on the same maps, seeds and draws: identical chains, moments, accept counts,
covariances, adapted widths and sqrt matrices.  Reference: MCMC/MCMC.py:225-340, MCMC/Proposal.py:127-164,
MCMC/BeamFit/*; the reference's own chain on a beam map, draw for draw, is tests/test_gmh_gpu.py::
test_beam_reference_chains (both engines)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def maps(kind):
    from oracle import refbeam as RB
    if kind == "one64":                                   # the reference's own use: one 64 x 64 map (pixels in LDS)
        x, y, d, noise, truth, prior = RB.synth_beam_maps(1, n=64, seed=4)
        return x, y, d, noise, truth, prior
    if kind == "three32":                                 # several maps, one block each
        return RB.synth_beam_maps(3, n=32, seed=2)
    if kind == "ragged":                                  # a pixel count that fills no chunk, per-pixel errors, own grids
        x, y, d, noise, truth, prior = RB.synth_beam_maps(2, n=37, seed=9)
        rng = np.random.default_rng(1)
        sig = noise * rng.uniform(0.7, 1.4, d.shape)
        xs = np.stack([x, x + 0.31])
        ys = np.stack([y, y - 0.17])
        return xs, ys, d, sig, truth, prior
    if kind == "big128":                                  # 16 384 pixels: 32 chunks, streamed from L2
        return RB.synth_beam_maps(1, n=128, seed=6)
    raise ValueError(kind)


def run(kind, fused, mode, C, track_cov, inject=False, plane=False, seed=41):
    from mcmc_amd import Engine
    x, y, d, sig, truth, prior = maps(kind)
    M, P = d.shape[0], 9
    rng = np.random.default_rng(3)
    A = rng.normal(size=(P, P))
    cov = A @ A.T
    dg = np.sqrt(np.diag(cov))
    L = np.linalg.cholesky(cov / dg[:, None] / dg[None, :])
    start = truth[:, None, :].repeat(C, 1)
    start[..., :7] *= 1.0 + 0.01 * rng.standard_normal((M, C, 7))
    start[..., 2:4] = np.clip(start[..., 2:4], prior.sig_min + 0.1, prior.sig_max - 0.1)
    if plane:
        start[..., 7] = 0.9                               # mu != 1: the plane term is evaluated
        start[..., 8] = 0.4
    sg = np.array([0.3, 0.3, 0.2, 0.2, 0.05, 0.2, 0.005, 0.02 if plane else 0.0, 0.05 if plane else 0.0])
    nblk = 9 if plane else 7
    block = mode != "corr_all" and mode != "uncorr_all"
    out = {}
    with Engine(seed=seed) as eng:
        eng.mh_set_fused("always" if fused else False)    # (the kernel, not the size policy)
        eng.beam_load(x, y, d, sig, center_min=prior.center_min, center_max=prior.center_max,
                      sig_min=prior.sig_min, sig_max=prior.sig_max)
        eng.mh_init_problem("beam", C, start, sg, param_blocks=np.arange(nblk) if block else None,
                            randomize_start=False, track_cov=track_cov)
        assert eng.mh_fused_active == fused
        if mode in ("corr_all", "corr_block", "rotate_block"):
            eng.mh_set_sqrt_matrix(L, rotate=(mode == "rotate_block"))
        if inject:
            n = 40
            zn = rng.standard_normal((n, M * C, P))
            zu = rng.uniform(size=(n, M * C))
            eng.mh_run(n, thin=1, inj_normals=zn, inj_uniforms=zu)
            out["trace1"] = eng.mh_trace()
        else:
            eng.mh_run(37, thin=1)
            out["trace1"] = eng.mh_trace()
            eng.mh_run(45)
            eng.mh_run(21, thin=3)
            out["trace2"] = eng.mh_trace()
        out["mom"] = eng.mh_moments()
        if track_cov:
            out["cov"] = eng.mh_covar(normalized=False)
            eng.mh_adapt_corr(no_correlations=False)
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


@pytest.mark.parametrize("kind,C", [("one64", 1), ("one64", 5), ("three32", 8), ("ragged", 3), ("big128", 2)])
@pytest.mark.parametrize("mode", ["uncorr_block", "uncorr_all", "corr_all", "corr_block", "rotate_block"])
def test_fused_equals_stepwise(kind, C, mode):
    a = run(kind, True, mode, C, True)
    same(a, run(kind, False, mode, C, True))
    assert a["mom"]["naccept"].sum() > 0                  # the chains did move


@pytest.mark.parametrize("mode", ["uncorr_block", "corr_all", "rotate_block"])
def test_fused_equals_stepwise_with_injected_draws_and_a_plane(mode):
    same(run("one64", True, mode, 2, False, inject=True, plane=True),
         run("one64", False, mode, 2, False, inject=True, plane=True))


def test_jobs_the_fused_kernel_leaves_to_the_stepwise_engine():
    """More than eight chains per map, or a job whose chi-square the engine runs on the matrix cores
    (another summation order): launch per step."""
    from oracle import refbeam as RB
    from mcmc_amd import Engine
    x, y, d, noise, truth, prior = RB.synth_beam_maps(3, n=32, seed=2)
    sg = np.array([0.3, 0.3, 0.2, 0.2, 0.05, 0.2, 0.005, 0.0, 0.0])
    with Engine(seed=1) as eng:
        eng.beam_load(x, y, d, noise, center_min=prior.center_min, center_max=prior.center_max,
                      sig_min=prior.sig_min, sig_max=prior.sig_max)
        eng.mh_init_problem("beam", 8, truth[:, None, :].repeat(8, 1), sg, param_blocks=np.arange(7))
        assert eng.mh_fused_active                        # 1024 pixels: two chunks
        eng.mh_init_problem("beam", 9, truth[:, None, :].repeat(9, 1), sg, param_blocks=np.arange(7))
        assert not eng.mh_fused_active
        eng.mh_set_fused("always")
        assert not eng.mh_fused_active                    # ... whatever the policy
        eng.mh_run(40)
        assert np.all(np.isfinite(eng.mh_moments()["lnPr"]))
    xm, ym, dm, noise, tr, prior = RB.synth_beam_maps(80, n=16, seed=5)
    with Engine(seed=1) as eng:
        eng.beam_load(xm, ym, dm, noise, center_min=prior.center_min, center_max=prior.center_max,
                      sig_min=prior.sig_min, sig_max=prior.sig_max)
        eng.mh_init_problem("beam", 16, tr[:, None, :].repeat(16, 1), sg, param_blocks=np.arange(7))
        assert not eng.mh_fused_active                    # 1280 (map, chain) pairs in full groups of 16
    # one 64 x 64 map - eight chunks on one CU against eight CUs - stays with the engine unless asked for
    x, y, d, noise, truth, prior = RB.synth_beam_maps(1, n=64, seed=4)
    with Engine(seed=1) as eng:
        eng.beam_load(x, y, d, noise, center_min=prior.center_min, center_max=prior.center_max,
                      sig_min=prior.sig_min, sig_max=prior.sig_max)
        eng.mh_init_problem("beam", 1, truth, sg, param_blocks=np.arange(7))
        assert not eng.mh_fused_active
        eng.mh_set_fused("always")
        assert eng.mh_fused_active
