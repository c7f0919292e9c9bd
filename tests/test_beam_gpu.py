"""GPU parity of the BeamFit path against the pinned oracle (oracle/refbeam.py)
and the reference's own outputs (tests/golden/beam_g6.npz), through the C ABI."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
RTOL = 1e-12
MAPS = ["m64a", "m64b", "m48s", "m256"]


@pytest.fixture(scope="module")
def g6():
    return np.load(os.path.join(GOLD, "beam_g6.npz"))


def close(a, b, rtol=RTOL, atol=0.0):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin), (a, b)
    np.testing.assert_allclose(a[fin], b[fin], rtol=rtol, atol=atol)


def setup_model(g):
    from mcmc_amd.BeamFit.OffsetNormalizedBeamModel import OffsetNormalizedBeamModel as mod
    mod.centerMin, mod.centerMax = tuple(g["centerMin"]), tuple(g["centerMax"])
    mod.sigMin, mod.sigMax = float(g["sigMin"]), float(g["sigMax"])
    mod.paramBlocks, mod.nBlock = [0, 1, 2, 3, 4, 5, 6], 7
    return mod


@pytest.mark.parametrize("name", MAPS)
def test_beam_lnlike_matches_reference(g6, name):
    from mcmc_amd.BeamFit.BeamData import BeamData
    from mcmc_amd.BeamFit.NormalizedBeamLikelihood import NormalizedBeamLikelihood
    mod = setup_model(g6)
    pre = name + "__"
    sig = g6[pre + "sig"]
    data = BeamData(g6[pre + "x"], g6[pre + "y"], g6[pre + "d"], sig if sig.size > 1 else sig[0])
    like = NormalizedBeamLikelihood(data=data, model=mod)
    th = g6[pre + "theta"]
    close(like.lnLike(th), g6[pre + "lnl"])
    for t in (0, 1, 4):
        close(like.lnLike(mod.package(th[t])), g6[pre + "lnl"][t])
    pr, pk = like.engine.beam_prior_package(th)
    assert np.array_equal(pk, g6[pre + "package"])           # fmod / fabs: bit for bit
    assert np.array_equal(pr, g6[pre + "prior"])
    assert mod.prior(*mod.package(th[0])) == 1 and mod.prior(*mod.package(th[3])) == 0
    pix = g6[pre + "pix"]
    mv = np.array([mod(*mod.package(t)).atxy(data.x[pix], data.y[pix]) for t in th])
    close(mv, g6[pre + "model_vals"], rtol=1e-13)
    st = g6[pre + "stats"]
    close(data.stats(sigcut=0.0), st, rtol=1e-9)


def test_beam_many_maps_against_oracle():
    """Config C5 shape at reduced size: many maps x 16 chains in one launch, spot-checked
    against the oracle; the result must not depend on how maps are batched."""
    from oracle import refbeam as RB
    from mcmc_amd import Engine
    M, n, C = 24, 96, 16
    x, y, d, noise, truth, prior = RB.synth_beam_maps(M, n=n, seed=11)
    rng = np.random.default_rng(5)
    th = truth[:, None, :] * (1 + 0.03 * rng.standard_normal((M, C, 9)))
    th[:, :, 7] = rng.uniform(0.8, 1.0, (M, C))
    th[:, 0, :] = truth
    with Engine() as eng:
        eng.beam_load(x, y, d, noise, center_min=prior.center_min, center_max=prior.center_max,
                      sig_min=prior.sig_min, sig_max=prior.sig_max)
        got = eng.beam_lnlike(th)
        part = eng.beam_lnlike(th[5:9], m0=5)
        one = eng.beam_lnlike(th[7, 3], m0=7)
    close(part, got[5:9], rtol=1e-14)
    close(one, got[7, 3], rtol=1e-14)
    for m in (0, 7, 23):
        bm = RB.BeamMap(x, y, d[m], noise)
        ref = [RB.beam_lnlike(RB.beam_package(th[m, c]), bm) for c in range(C)]
        close(got[m], ref)


@pytest.mark.parametrize("n,C,per_pixel_sigma", [(48, 16, False), (47, 13, False), (47, 16, True)])
def test_beam_many_maps_on_the_matrix_cores(n, C, per_pixel_sigma):
    """The kernel that large jobs on one shared grid take (k_beam_chi2_mf: >= 1024 (map, chain) pairs,
    groups of 16 chains at least three quarters full): against the oracle on a few maps, and against
    the lane-per-pixel kernel (what a launch of few pairs takes) on all of them.  47 x 47 pixels leave
    a last tile of one pixel; C = 13 leaves three idle rows per tile; chains with and without a plane
    share waves; one chain has a non-finite width (its wave runs the exact model)."""
    from oracle import refbeam as RB
    from mcmc_amd import Engine
    M = 96
    x, y, d, noise, truth, prior = RB.synth_beam_maps(M, n=n, seed=21)
    rng = np.random.default_rng(9)
    th = truth[:, None, :] * (1 + 0.03 * rng.standard_normal((M, C, 9)))
    th[:, 0::2, 7] = 1.0                                              # no plane (mu = 1: the integer 0)
    th[:, 1::2, 7] = rng.uniform(0.8, 1.0, (M, C // 2))               # every other chain: a plane
    th[:, 0, :] = truth
    th[40, 5, 2] = np.inf                                             # one irregular wave: the exact model
    th[41, 4, 7] = 1.02                                               # mu > 1: NaN, as the reference
    th[42, 3, 2:4] = (0.35, 0.5)                                      # a beam 1/100 of the map wide: the monomial
    #                                                                   form would lose digits - exact model instead
    sig = noise * (1 + 0.5 * rng.random((M, n * n))) if per_pixel_sigma else noise
    with Engine() as eng:
        eng.beam_load(x, y, d, sig, center_min=prior.center_min, center_max=prior.center_max,
                      sig_min=prior.sig_min, sig_max=prior.sig_max)
        got = eng.beam_lnlike(th)                                     # 96 C >= 1024 pairs
        small = np.concatenate([eng.beam_lnlike(th[m0:m0 + 8], m0=m0) for m0 in range(0, M, 8)])
    assert not np.isfinite(got[40, 5]) and np.isnan(got[41, 4]) and np.isfinite(got).sum() == got.size - 2
    close(got, small, rtol=1e-12)
    for m in (0, 40, 42, 95):
        bm = RB.BeamMap(x, y, d[m], sig[m] if per_pixel_sigma else noise)
        ref = [RB.beam_lnlike(RB.beam_package(th[m, c]), bm) for c in range(C)]
        close(got[m], ref)


def test_beam_errors():
    from mcmc_amd import Engine, Mi355Error
    with Engine() as eng:
        with pytest.raises(Mi355Error, match="no beam maps loaded"):
            eng.beam_lnlike(np.zeros(9))
        with pytest.raises(ValueError, match="DataSizeError"):
            eng.beam_load(np.zeros(4), np.zeros(4), np.zeros((2, 5)), 1.0)
