"""The reference-shaped Python API (Data / Model / Likelihood / Proposal / MCMC) on the GPU,
driven the way the reference's own driver drives it (submmSED/driver.py:141-175, :247-249)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import refcpu as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def arp220(g2):
    """txt/Arp220.txt as the reference's reader delivers it (fixture copy of the 27 bands)."""
    from mcmc_amd.submmSED.data import submmData
    i = len(g2["nb"]) - 1
    nb = int(g2["nb"][i])
    return submmData(g2["nu"][i, :nb], g2["d"][i, :nb], g2["sig"][i, :nb], "Arp220", 0.018)


def test_arp220_reader(tmp_path, g2, arp220):
    """read_wavelength_table reproduces the Arp220 branch of readfluxes_DLC_2014 (25 um upper
    limit, rest-frame frequencies) on the table the reference ships (fixture copy)."""
    from mcmc_amd.submmSED import data
    lam = data.speed_of_light / (arp220.freq / 1.018)
    flux, sig = arp220.d.copy(), arp220.sig.copy()
    l25 = np.where(np.round(lam) == 25)[0]
    flux[l25], sig[l25] = 8.0, 0.034                       # undo the upper-limit transform
    p = tmp_path / "Arp220.txt"
    np.savetxt(p, np.stack([lam, flux, sig], 1), header="Arp220 z=0.018")
    d = data.readfluxes_Arp220(str(p))[0]
    np.testing.assert_allclose(d.freq, arp220.freq, rtol=1e-12)
    np.testing.assert_allclose(d.d, arp220.d, rtol=1e-12)
    np.testing.assert_allclose(d.sig, arp220.sig, rtol=1e-12)
    assert d.d[l25] == 0 and d.n == 27


def test_likelihood_classes_anchor_values(arp220):
    from mcmc_amd.submmSED import likelihood, model
    cases = [(model.submmModel1_normalized, likelihood.SEDLikelihood_normalized, (1, 2, 10), -54534735.92133923),
             (model.submmModel1_normalized_logA, likelihood.SEDLikelihood_normalized, (0.5, 1.5, 40), -31622129.359786913),
             (model.submmModel2_normalized, likelihood.SEDLikelihood_normalized, (1, 2, 10, 0.1, 2, 20), -54515895.05740812),
             (model.submmModel1, likelihood.SEDLikelihood1, (2, 10), 2030.3003733933579)]
    for mod, lk, th, val in cases:
        like = lk(data=arp220, model=mod)
        got = like.lnLike(th)
        assert abs(got - val) <= 1e-12 * abs(val)
        assert like(th) == got
        if lk is likelihood.SEDLikelihood_normalized:
            assert like.chi2(th) == -2 * got
    like = likelihood.SEDLikelihood1(data=arp220, model=model.submmModel1)
    like.lnLike((2, 10))
    assert like.getDerived() == [like.MLamplitude] and like.MLamplitude > 0
    # the per-data-set protocol of the reference, like.lnLike1(model_vals, data) (Likelihood.py:70-75):
    # model values from the model object, quadratic forms / chi2 from the data object, both on the
    # device, against the fused-kernel lnLike of the same point
    for mod, lk, th, val in cases:
        like = lk(data=arp220, model=mod)
        vals = mod(*th)(arp220)
        got1 = like.lnLike1(vals, arp220)
        assert abs(got1 - val) <= 1e-11 * abs(val)
    l2 = likelihood.SEDLikelihood2(data=arp220, model=model.submmModel2)
    th2 = (2.0, 10.0, 1.5, 30.0)
    whole = l2.lnLike(th2)
    amp_fused = np.array(l2.MLamplitude, dtype=float)
    part = l2.lnLike1(model.submmModel2(*th2)(arp220), arp220)
    assert abs(part - whole) <= 1e-8 * abs(whole)             # the nearly singular 2x2 system
    np.testing.assert_allclose(l2.MLamplitude, amp_fused, rtol=1e-6)


@pytest.mark.parametrize("mid", range(8))
def test_model_classes_match_golden(g2, mid):
    """prior / package / at() of every model class against the reference's values."""
    from mcmc_amd.submmSED import model
    cls = model.MODEL_CLASSES[mid]
    th = g2["theta_%d" % mid]
    assert cls.nparam == th.shape[1] and cls.nBlock == cls.nparam
    for t in range(0, th.shape[0], 5):
        assert cls.prior(*th[t]) == pytest.approx(g2["prior_%d" % mid][t], rel=4e-16)
        np.testing.assert_array_equal(cls.unpackage(cls.package(th[t])), g2["package_%d" % mid][t])
    # at(): flux at one source's bands equals the oracle's (itself pinned to the reference)
    nb = int(g2["nb"][0])
    nu = g2["nu"][0, :nb]

    class D:
        freq = nu
    for t in (0, 3, 11):
        with np.errstate(all="ignore"):
            ref = R.sed_flux(mid, th[t], nu)
        got = cls(*th[t])(D)
        ok = np.isfinite(ref)
        np.testing.assert_allclose(np.asarray(got)[ok], ref[ok], rtol=2e-14)


def test_gaussian_data_quadforms():
    from mcmc_amd.GaussianData import GaussianData, DataSizeError
    rng = np.random.default_rng(0)
    n = 5000
    d, sig, A, B = rng.normal(size=n), rng.uniform(0.5, 2, n), rng.normal(size=n), rng.normal(size=n)
    g = GaussianData(d, sig)
    assert g.lnDetN == pytest.approx(np.log(sig ** 2).sum(), rel=1e-14)
    assert g.quadform(A) == pytest.approx(np.dot(A / sig ** 2, A), rel=1e-13)
    assert g.quadform(A, B) == pytest.approx(np.dot(A / sig ** 2, B), rel=1e-11, abs=1e-11)
    assert g.chi2(A) == pytest.approx(np.dot((d - A) / sig ** 2, d - A), rel=1e-13)
    assert g.chi2() == pytest.approx(np.dot(d / sig ** 2, d), rel=1e-13)
    g1 = GaussianData(d, 0.5)                             # scalar sigma
    assert g1.chi2(A) == pytest.approx(np.dot(d - A, d - A) / 0.25, rel=1e-13)
    M = rng.normal(size=(n, 2))                           # matrix model values (submmModel2)
    np.testing.assert_allclose(g.quadform(M), np.dot(M.T / sig ** 2, M), rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(g.quadform(M, d), np.dot(M.T / sig ** 2, d), rtol=1e-11, atol=1e-9)
    with pytest.raises(DataSizeError):
        GaussianData(d, sig[:7])


def test_generic_likelihood_marginalises_the_amplitude():
    """The base Likelihood with a user model class and two GaussianData sets: lnLike, chi2,
    lnNorm, lnLikeFull, lnPost follow Likelihood.py:78-163's formulae (numpy restatement)."""
    from mcmc_amd.GaussianData import GaussianData
    from mcmc_amd.Likelihood import Likelihood, ZeroPosterior
    rng = np.random.default_rng(4)
    sets = []
    for n in (700, 1300):
        x = np.linspace(-2, 2, n)
        sig = rng.uniform(0.1, 0.3, n)
        g = GaussianData(3.0 * np.exp(-0.5 * (x - 0.2) ** 2 / 0.49) + sig * rng.normal(size=n), sig)
        g.x = x
        sets.append(g)

    class Bump(object):
        def __init__(self, mu, width):
            self.mu, self.width = mu, width

        def __call__(self, dataset):
            return np.exp(-0.5 * (dataset.x - self.mu) ** 2 / self.width ** 2) * (self.width > 0)

        @staticmethod
        def lnPrior(params):
            return -0.5 * params[0] ** 2

    like = Likelihood(data=sets[0], model=Bump)
    like.adddata(sets[1])
    assert len(like.data) == 2
    p = (0.25, 0.65)
    want = want_chi2 = want_norm = 0.0
    for g in sets:
        F = Bump(*p)(g)
        ff, fd = np.dot(F / g.sig2, F), np.dot(F / g.sig2, g.d)
        want += 0.5 * (fd * fd / ff - np.log(ff))
        want_chi2 += np.dot((g.d - F) / g.sig2, g.d - F)
        want_norm += -0.5 * ((g.n - 1) * np.log(2 * np.pi) + np.log(g.sig2).sum() +
                             np.dot(g.d / g.sig2, g.d))
        amp = fd / ff
    assert like(p) == pytest.approx(want, rel=1e-12)
    assert like.MLamplitude == pytest.approx(amp, rel=1e-12)          # of the last dataset
    assert like.chi2(p) == pytest.approx(want_chi2, rel=1e-12)
    assert like.lnNorm() == pytest.approx(want_norm, rel=1e-12)
    assert like.lnLikeFull(p) == pytest.approx(want + want_norm, rel=1e-12)
    assert like.lnPost(p) == pytest.approx(want - 0.5 * p[0] ** 2, rel=1e-12)
    assert like.lnPostFull(p) == pytest.approx(want - 0.5 * p[0] ** 2 + want_norm, rel=1e-12)
    with pytest.raises(ZeroPosterior):
        like((0.25, -1.0))                                           # model vanishes: F^T N^-1 F = 0


def test_sampler_arp220_matches_reference(arp220, g4):
    """Config C1: MCMC.sampler() - the reference's staged driver - on Arp220 with the reference's
    own draws injected: same stages, same accept counts, same adapted proposal widths, same
    returned (means, stdevs, covariance)."""
    from mcmc_amd import MCMC
    from mcmc_amd.submmSED import likelihood, model
    like = likelihood.SEDLikelihood_normalized(data=arp220, model=model.submmModel1_normalized)
    stages, ana = MCMC.sampler(like, (1000, 3000), np.array((2., 4., 4.)), np.array((1., 2., 10.)),
                               noCorrelations=True, doBlock=True, verbose=False,
                               inject=(g4["normals"], g4["uniforms"]))
    assert len(stages) == int(g4["nstage"])
    for i, st in enumerate(stages):
        assert len(st.samples) == int(g4["stage%d_n" % i])
        np.testing.assert_allclose(st.samples[::10], g4["stage%d_samples" % i], rtol=1e-12)
        assert st.lnPr[-1] == pytest.approx(float(g4["stage%d_lnPr_last" % i]), rel=1e-12)
        assert st.naccept == int(g4["stage%d_naccept" % i])
        np.testing.assert_array_equal(st.accept, g4["stage%d_accept" % i])
        np.testing.assert_allclose(st.prop.sigmas, g4["stage%d_sigmas" % i], rtol=1e-10)
    np.testing.assert_allclose(ana[0], g4["ana_means"], rtol=1e-12)
    np.testing.assert_allclose(ana[1], g4["ana_stdevs"], rtol=1e-10)
    np.testing.assert_allclose(ana[2], g4["ana_covar"], rtol=1e-8, atol=1e-10)


def test_mcmc_object_free_running_and_rhat(arp220):
    """The driver's usage (driver.py:141-175, 247-249) with on-device Philox draws: one MCMC
    object, 8 chains; R-hat from the device equals the oracle's Gelman-Rubin on the traces."""
    from mcmc_amd import MCMC, convergence
    from mcmc_amd.submmSED import likelihood, model
    like = likelihood.SEDLikelihood_normalized(data=arp220, model=model.submmModel1_normalized_logA)
    m = MCMC.MCMC(like, startProposal=np.array((0.02, 0.05, 0.5)), startParams=np.array((1.9, 1.3, 45.0)),
                  doBlock=True, nchains=8, seed=42, nMC=1500)
    assert m.samples.shape == (1501, 8, 3) and m.lnPr.shape == (1501, 8)
    assert np.all(m.naccept > 20)
    rhat = convergence.gelmanrubin_MCMC(m, burn=1)
    s = m.samples[1:]
    ref, _ = R.gelman_rubin(s.mean(axis=0), s.var(axis=0))
    np.testing.assert_allclose(rhat, ref, rtol=1e-9)
    # list-of-objects form, arbitrary burn: moments from the stored traces
    ms = [MCMC.MCMC(like, startProposal=np.array((0.02, 0.05, 0.5)), startParams=np.array((1.9, 1.3, 45.0)),
                    doBlock=True, seed=100 + i, nMC=600) for i in range(3)]
    rhat = convergence.gelmanrubin_MCMC(ms, burn=100, stride=2)
    ref, _ = R.gelman_rubin([x.samples[100::2].mean(axis=0) for x in ms],
                            [x.samples[100::2].var(axis=0) for x in ms])
    np.testing.assert_allclose(rhat, ref, rtol=1e-10)
    # newMCMC: restart at the mean with 2.4/sqrt(P) * stdev proposal widths (MCMC.py:379-385)
    std = ms[0].stdev(1, 1)
    new = ms[0].newMCMC(burn=1, stride=1, noCorrelations=True)
    np.testing.assert_allclose(new.prop.sigmas, 2.4 / np.sqrt(3) * std, rtol=1e-12)
    new.MC_append(500)
    assert len(new.samples) == 501 and new.naccept >= 1 and new.samples.shape == (501, 3)
    with pytest.raises(Exception):
        ms[0].MC_append(1)                                # its device chains moved to `new`


def test_gelman_rubin_matches_the_reference_value(g3):
    """G3 `rhat__value` is the reference's own gelmanrubin_MCMC (convergence.py:58-87) on five of
    its chains; the device kernel (mi355_gelman_rubin) on those chains' means and variances must
    return it (the fixture was only compared with the CPU oracle)."""
    from mcmc_amd import Engine
    ref = g3["rhat__value"]
    means, var = g3["rhat__means"], g3["rhat__vars"]          # [5 chains, 6 parameters]
    with Engine(seed=1) as e:
        got = e.gelman_rubin(means, var)
    # the reference sums over chains with numpy's pairwise mean/var, the device in chain order:
    # identical up to the last bits
    np.testing.assert_allclose(got, ref, rtol=1e-13)
    assert np.all(np.isfinite(ref)) and ref.shape == (6,)


def test_zero_prior_start_raises(arp220):
    from mcmc_amd import MCMC
    from mcmc_amd.submmSED import likelihood, model
    like = likelihood.SEDLikelihood_normalized(data=arp220, model=model.submmModel1_normalized)
    m = MCMC.MCMC(like, startProposal=np.array((0.1, 0.1, 1.0)), doBlock=True)
    with pytest.raises(MCMC.ZeroPriorError):
        m.setStart(np.array((1.0, 2.0, 500.0)), random=False)     # T > maxTemp
