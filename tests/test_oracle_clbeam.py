"""The Cl and beam oracles (oracle/refcl.py, oracle/refbeam.py) against fixtures produced by the
reference itself (tests/golden/cl_g5.npz, beam_g6.npz; generator: oracle/make_golden.py)."""
import os

import numpy as np
import pytest

from oracle import refbeam as RB
from oracle import refcl as RC
from oracle import refcpu as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CL_CASES = ["xfull", "xcalib", "xcalibbeam", "xbeam", "gauss_analytic", "gauss_calib", "gauss_beam",
            "gauss_diag", "x_diag_calib", "pol", "pol_gauss", "wide"]


@pytest.fixture(scope="module")
def g5():
    return np.load(os.path.join(GOLD, "cl_g5.npz"))


@pytest.fixture(scope="module")
def g6():
    return np.load(os.path.join(GOLD, "beam_g6.npz"))


def cl_case(g, name, data_prefix=None):
    """(model, dataset) of one fixture case; data_prefix picks another dataset on the same binning."""
    pre = name + "__"
    dp = (data_prefix or name) + "__"
    bins = [g[pre + "bins%d" % i] for i in range(int(g[pre + "nCl"]))]
    model = RC.BinnedClModel(bins, g[pre + "shapefun"])
    def opt(k, empty_is_none=True):
        a = g[dp + k] if (dp + k) in g.files else np.zeros(0)
        return None if a.size == 0 else a
    ninv = opt("N_inv")
    x = opt("xfactors")
    be = opt("beam_err")
    d = RC.ClDataset(obs=g[dp + "obs"], N_inv=ninv, var=opt("var"), window=g[dp + "window"],
                     win_min=g[dp + "win_min"], win_max=g[dp + "win_max"], xfactors=x,
                     has_xfactor=opt("has_xfactor") if x is not None else None,
                     calib_uncertainty=float(g[dp + "calib"]), beam_uncertain=be is not None,
                     beam_err=be, has_pol=bool(g[dp + "has_pol"]) if (dp + "has_pol") in g.files else False,
                     inc_pol=g[dp + "inc_pol"])
    return model, d


def same(a, b, rtol=0.0):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape
    nan = np.isnan(b)
    assert np.array_equal(np.isnan(a), nan)
    if rtol == 0.0:
        assert np.array_equal(a[~nan], b[~nan]), np.max(np.abs(a[~nan] - b[~nan]))
    else:
        np.testing.assert_allclose(a[~nan], b[~nan], rtol=rtol, atol=0)


@pytest.mark.parametrize("name", CL_CASES)
def test_cl_oracle_matches_reference(g5, name):
    model, d = cl_case(g5, name)
    pre = name + "__"
    q = g5[pre + "q"]
    lnl = np.array([RC.cl_lnlike(t, model, [d]) for t in q])
    same(lnl, g5[pre + "lnl"])
    same([model.prior(t) for t in q], g5[pre + "prior"])
    with np.errstate(all="ignore"):
        same([d.calc_lnlike(model.Cl(t)) for t in q], g5[pre + "neglnl"])
        same([d.bandpowers(model.Cl(t)) for t in q], g5[pre + "bandpowers"])
        same([model.Cl(t) for t in q[:2]], g5[pre + "Cl"])
        bp1 = g5[pre + "bandpowers"][1]
        delta = d.getdelta(bp1)
        same(delta, g5[pre + "delta1"])
        same(d.getquadform(delta), g5[pre + "quad1"])
        same(d.getlinform(bp1), g5[pre + "lin1"])
        same(d.getchisq(bp1), g5[pre + "chisq1"])


def test_cl_two_datasets(g5):
    model, d1 = cl_case(g5, "xcalibbeam")
    _, d2 = cl_case(g5, "xcalibbeam", data_prefix="two")
    q = g5["xcalibbeam__q"]
    same([RC.cl_lnlike(t, model, [d1, d2]) for t in q], g5["two__lnl"])


def test_cl_linearisation_is_exact_enough(g5):
    """BP = q.K + BP0 (what the device evaluates) against the reference's own band powers."""
    for name in ("xcalibbeam", "pol", "wide"):
        model, d = cl_case(g5, name)
        K, bp0 = RC.linearise_bandpowers(model, d)
        q = g5[name + "__q"]
        bp = q @ K + bp0
        ref = g5[name + "__bandpowers"]
        assert np.max(np.abs(bp - ref)) <= 1e-13 * np.max(np.abs(ref))


def test_cl_chain_draw_for_draw(g5):
    model, d = cl_case(g5, "xcalib")
    pb = RC.cl_problem(model, [d])
    k = g5["chain__k"]
    draws = R.InjectedDraws(g5["chain__normals"], g5["chain__uniforms"])
    c0 = R.Chain(pb, np.full(model.nparam, 0.05), g5["chain__start"], draws, do_block=True)
    assert draws.k == k[0]
    c0.run(400)
    assert draws.k == k[1]
    s, l, a = c0.arrays()
    same(s, g5["chain__c0_samples"]); same(l, g5["chain__c0_lnPr"])
    assert np.array_equal(a, g5["chain__c0_accepted"])
    assert c0.naccept == int(g5["chain__c0_naccept"])
    assert np.array_equal(c0.accept, g5["chain__accept0"])
    c1 = c0.new_chain(burn=1, stride=1, rotate=True)            # MCMC.py:389-393
    assert draws.k == k[2]
    same(c1.prop.sigmas, g5["chain__c1_sigmas"]); same(c1.prop.sqrt_matrix, g5["chain__c1_sqrt"])
    c1.run(300)
    assert draws.k == k[3]
    s, l, a = c1.arrays()
    same(s, g5["chain__c1_samples"]); same(l, g5["chain__c1_lnPr"])
    assert np.array_equal(a, g5["chain__c1_accepted"])
    assert np.array_equal(c1.accept, g5["chain__accept1"])
    c2 = c1.new_chain(burn=1, stride=1, no_correlations=False, do_block=False)
    assert draws.k == k[4]
    same(c2.prop.sqrt_matrix, g5["chain__c2_sqrt"])
    c2.run(200)
    assert draws.k == k[5]
    s, l, a = c2.arrays()
    same(s, g5["chain__c2_samples"]); same(l, g5["chain__c2_lnPr"])
    assert np.array_equal(a, g5["chain__c2_accepted"])


# ---- beam --------------------------------------------------------------------------------
BEAM_MAPS = ["m64a", "m64b", "m48s", "m256"]


def beam_map(g, name):
    pre = name + "__"
    sig = g[pre + "sig"]
    return RB.BeamMap(g[pre + "x"], g[pre + "y"], g[pre + "d"], sig if sig.size > 1 else sig[0])


def beam_prior_of(g):
    return RB.BeamPrior(tuple(g["centerMin"]), tuple(g["centerMax"]), float(g["sigMin"]),
                        float(g["sigMax"]))


@pytest.mark.parametrize("name", BEAM_MAPS)
def test_beam_oracle_matches_reference(g6, name):
    m = beam_map(g6, name)
    pr = beam_prior_of(g6)
    pre = name + "__"
    th = g6[pre + "theta"]
    pk = np.array([RB.beam_package(t) for t in th])
    same(pk, g6[pre + "package"])
    same([RB.beam_prior(t, pr) for t in pk], g6[pre + "prior"])
    same([RB.beam_lnlike(t, m) for t in pk], g6[pre + "lnl"])
    pix = g6[pre + "pix"]
    with np.errstate(all="ignore"):
        mv = np.array([RB.beam_model(t, m.x, m.y)[pix] if np.ndim(RB.beam_model(t, m.x, m.y)) else
                       np.full(pix.size, np.nan) for t in pk])
    same(mv, g6[pre + "model_vals"])
    same(RB.beam_stats(m, sigcut=0.0), g6[pre + "stats"])


def test_beam_chain_draw_for_draw(g6):
    m = beam_map(g6, "m64a")
    pb = RB.beam_problem(m, beam_prior_of(g6))
    k = g6["chain__k"]
    draws = R.InjectedDraws(g6["chain__normals"], g6["chain__uniforms"])
    c0 = R.Chain(pb, g6["chain__sigmas"], g6["chain__start"], draws, do_block=True)
    assert draws.k == k[0]
    c0.run(500)
    assert draws.k == k[1]
    s, l, a = c0.arrays()
    same(s, g6["chain__c0_samples"]); same(l, g6["chain__c0_lnPr"])
    assert np.array_equal(a, g6["chain__c0_accepted"])
    assert np.array_equal(c0.accept, g6["chain__c0_accept"])
    c1 = c0.new_chain(burn=1, stride=1, no_correlations=False)
    assert draws.k == k[2]
    same(c1.prop.sigmas, g6["chain__c1_sigmas"]); same(c1.prop.sqrt_matrix, g6["chain__c1_sqrt"])
    c1.run(300)
    assert draws.k == k[3]
    s, l, a = c1.arrays()
    same(s, g6["chain__c1_samples"]); same(l, g6["chain__c1_lnPr"])
    assert np.array_equal(a, g6["chain__c1_accepted"])
    assert np.array_equal(c1.accept, g6["chain__c1_accept"])
