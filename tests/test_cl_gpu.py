"""GPU parity of the binned-C_l band-power likelihood against the pinned oracle
(oracle/refcl.py) and the reference's own outputs (tests/golden/cl_g5.npz), through the C ABI."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CL_CASES = ["xfull", "xcalib", "xcalibbeam", "xbeam", "gauss_analytic", "gauss_calib", "gauss_beam",
            "gauss_diag", "x_diag_calib", "pol", "pol_gauss", "wide"]
RTOL = 1e-12          # BASELINE.json north_star: per-point log-likelihoods to 1e-12 relative


@pytest.fixture(scope="module")
def g5():
    return np.load(os.path.join(GOLD, "cl_g5.npz"))


def close(a, b, rtol=RTOL, atol=0.0):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin), (a, b)       # non-finite points are rejections
    np.testing.assert_allclose(a[fin], b[fin], rtol=rtol, atol=atol)


def mirror(g, name, data_prefix=None):
    """mcmc_amd classes for one fixture case"""
    from mcmc_amd.ClData.ClData_CosmoMC import ClData_CosmoMC
    from mcmc_amd.binnedCl.binnedClModel import binnedClModel
    pre, dp = name + "__", (data_prefix or name) + "__"
    bins = [g[pre + "bins%d" % i] for i in range(int(g[pre + "nCl"]))]
    binnedClModel.setBinning(bins, shapefun=g[pre + "shapefun"])

    def opt(k):
        a = g[dp + k] if (dp + k) in g.files else np.zeros(0)
        return None if a.size == 0 else a
    d = ClData_CosmoMC()
    d.num_points = len(g[dp + "obs"])
    d.obs = g[dp + "obs"]
    d.N_inv, d.var = opt("N_inv"), opt("var")
    d.has_corr_errors = d.N_inv is not None
    d.xfactors, d.has_xfactor = opt("xfactors"), opt("has_xfactor")
    d.has_xfactors = d.xfactors is not None and bool(np.any(d.has_xfactor))
    d.window, d.win_min, d.win_max = g[dp + "window"], g[dp + "win_min"], g[dp + "win_max"]
    d.calib_uncertainty = float(g[dp + "calib"])
    d.beam_err = opt("beam_err")
    d.beam_uncertain = d.beam_err is not None
    d.has_pol = bool(g[dp + "has_pol"]) if (dp + "has_pol") in g.files else False
    d.inc_pol = g[dp + "inc_pol"]
    return binnedClModel, d


@pytest.mark.parametrize("name", CL_CASES)
def test_cl_lnlike_matches_reference(g5, name):
    from mcmc_amd.binnedCl.binnedClLikelihood import binnedClLikelihood
    model, d = mirror(g5, name)
    pre = name + "__"
    like = binnedClLikelihood(data=[d], model=model)
    q = g5[pre + "q"]
    close(like.lnLike(q), g5[pre + "lnl"])
    close(like.lnLike(q[0]), g5[pre + "lnl"][0])
    pri = np.array([model.prior(*t) for t in q[:6]])
    assert np.array_equal(pri, g5[pre + "prior"][:6])
    # spectrum: same operations as the reference, bit for bit
    Cl = np.array([model(t)() for t in q[:2]])
    assert np.array_equal(Cl, g5[pre + "Cl"])
    # methods of the data object on explicit spectra / vectors
    bp = g5[pre + "bandpowers"]
    close(d.bandpowers(Cl[1]), bp[1], rtol=1e-13, atol=1e-13 * np.abs(bp[1]).max())
    close(like.engine.cl_bandpowers(0, q=q), bp, rtol=1e-12, atol=1e-13 * np.abs(bp).max())
    close(d.calcLnLike(Cl[1]), g5[pre + "neglnl"][1])
    close(d.calcLnLike(g5[pre + "Cl"]), g5[pre + "neglnl"][:2])
    delta = d.getdelta(bp[1])
    close(delta, g5[pre + "delta1"], rtol=1e-13, atol=1e-15)
    close(d.getquadform(g5[pre + "delta1"]), g5[pre + "quad1"])
    lin = g5[pre + "lin1"]
    close(d.getlinform(bp[1]), lin, rtol=1e-12, atol=1e-13 * np.abs(lin).max())
    close(d.getchisq(bp[1]), g5[pre + "chisq1"])


def test_cl_two_datasets(g5):
    from mcmc_amd.binnedCl.binnedClLikelihood import binnedClLikelihood
    model, d1 = mirror(g5, "xcalibbeam")
    _, d2 = mirror(g5, "xcalibbeam", data_prefix="two")
    like = binnedClLikelihood(data=[d1, d2], model=model)
    close(like.lnLike(g5["xcalibbeam__q"]), g5["two__lnl"])


@pytest.mark.parametrize("nb,npar,calib,beam,n", [(64, 32, 0.0, 0.0, 300), (64, 32, 0.05, 0.02, 40),
                                                  (256, 64, 0.05, 0.0, 70), (100, 24, 0.05, 0.02, 9),
                                                  (256, 64, 0.0, 0.0, 4096)])
def test_cl_config4_against_oracle(nb, npar, calib, beam, n):
    """Config C4 shapes: synthetic offset-lognormal data, full covariance, checked
    against the oracle on a subset of the chains."""
    from oracle import refcl as RC
    from mcmc_amd import Engine
    model, d, truth = RC.synth_cl_problem(nb=nb, npar=npar, calib=calib, beam=beam, lmax=2400)
    rng = np.random.default_rng(nb + npar)
    q = truth[None, :] * rng.uniform(0.85, 1.15, (n, npar))
    with Engine() as eng:
        eng.cl_load([d])
        eng.cl_set_binning(model.bins, model.shapefun)
        got = eng.cl_lnlike(q)
    idx = np.unique(np.linspace(0, n - 1, 12).astype(int))
    ref = np.array([RC.cl_lnlike(q[i], model, [d]) for i in idx])
    close(got[idx], ref)
    assert np.all(np.isfinite(got))


def test_cl_errors():
    from mcmc_amd import Engine, Mi355Error
    with Engine() as eng:
        with pytest.raises(Mi355Error, match="no Cl data loaded"):
            eng.cl_npar = 3
            eng.cl_lnlike(np.zeros(3))
