"""The reference-shaped classes for binnedCl and BeamFit (and correlated SED proposals) driven
through MCMC / sampler() the way the reference's drivers do (MCMC/test_binnedCl.py:170-178,
BeamFit/ProcessBeamData.py:124-287)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_binnedcl_sampler_rotated():
    from test_cl_gpu import mirror
    from mcmc_amd import MCMC
    from mcmc_amd.binnedCl.binnedClLikelihood import binnedClLikelihood
    g = np.load(os.path.join(GOLD, "cl_g5.npz"))
    model, d = mirror(g, "xcalib")
    like = binnedClLikelihood(data=[d], model=model)
    start = g["chain__start"]
    P = model.nparam
    stages, ana = MCMC.sampler(like, (400, 600), np.full(P, 0.05), start, doBlock=True,
                               rotateParams=True, seed=11, verbose=False)
    assert len(stages) == 2 and stages[1].prop.rotateParams
    s1 = stages[1]
    assert s1.samples.shape == (601, P) and 0.1 < s1.naccept / 601 < 0.95
    # the chain state is consistent with the likelihood: lnPr of the last sample
    last = s1.samples[-1]
    assert abs(s1.lnPr[-1] - like.lnLike(last)) <= 1e-10 * abs(s1.lnPr[-1])
    # posterior means within a few sigma of the values the data were drawn from
    means, sd = np.array(ana[0]), np.array(ana[1])
    assert np.all(np.abs(means - start) < 6 * sd + 0.05)
    # sqrt matrix of stage 1 is a Cholesky factor of the stage-0 normalised covariance
    L = s1.prop.sqrtMatrix
    cov0 = np.corrcoef(stages[0].samples[1:].T)
    np.testing.assert_allclose(L @ L.T, cov0, rtol=1e-6, atol=1e-8)


def test_beam_mcmc_blocks_and_correlated_restart():
    from mcmc_amd import MCMC
    from mcmc_amd.BeamFit.BeamData import BeamData
    from mcmc_amd.BeamFit.NormalizedBeamLikelihood import NormalizedBeamLikelihood
    from test_beam_gpu import setup_model
    g = np.load(os.path.join(GOLD, "beam_g6.npz"))
    mod = setup_model(g)
    data = BeamData(g["m64a__x"], g["m64a__y"], g["m64a__d"], g["m64a__sig"][0])
    like = NormalizedBeamLikelihood(data=data, model=mod)
    sig = mod.package(g["chain__sigmas"])
    start = mod.package(g["m64a__truth"])
    m0 = MCMC.MCMC(like, startProposal=sig, startParams=start, doBlock=True, seed=5)
    m0.MC_append(700)
    assert m0.samples.shape == (701, 9) and m0.accept.shape == (7,)
    assert np.all(m0.samples[:, 7] == m0.samples[0, 7])        # mu, phi belong to no block
    m1 = m0.newMCMC(burn=1, stride=1, noCorrelations=False)
    m1.MC_append(400)
    assert m1.prop.sqrtMatrix.shape == (9, 9)
    lp = like.lnLike(mod.package(m1.samples[-1]))
    assert abs(m1.lnPr[-1] - lp) <= 1e-10 * abs(lp)
    truth = g["m64a__truth"]
    mean, sd = np.asarray(mod.unpackage(m1.mean(1, 1))), np.asarray(mod.unpackage(m1.stdev(1, 1)))
    free = [0, 1, 2, 3, 5, 6]
    assert np.all(np.abs(mean[free] - truth[free]) < 8 * sd[free] + 1e-3)


def test_sed_sampler_with_correlations():
    """sampler(noCorrelations=False) on an SED likelihood runs on the step-wise engine."""
    from mcmc_amd import MCMC
    from mcmc_amd.submmSED import likelihood, model
    from mcmc_amd.submmSED.data import submmData
    g2 = np.load(os.path.join(GOLD, "sed_g2.npz"))
    i = 2
    nb = int(g2["nb"][i])
    dat = submmData(g2["nu"][i, :nb], g2["d"][i, :nb], g2["sig"][i, :nb], "s2", 0.1)
    like = likelihood.SEDLikelihood_normalized(data=dat, model=model.submmModel1_normalized)
    stages, ana = MCMC.sampler(like, (800, 800), np.array((0.2, 0.3, 2.0)), np.array((1.0, 2.0, 20.0)),
                               noCorrelations=False, doBlock=True, seed=3, verbose=False)
    assert stages[0].stepwise and stages[1].prop.sqrtMatrix is not None
    assert stages[1].naccept > 50
    last = stages[1].samples[-1]
    assert abs(stages[1].lnPr[-1] - like.lnLike(last)) <= 1e-10 * abs(stages[1].lnPr[-1])
