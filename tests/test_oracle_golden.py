"""Pin the CPU oracle (oracle/refcpu.py) against fixtures produced by the reference
itself (oracle/make_golden.py).  Runs on CPU; no HIP involved."""
import numpy as np
import pytest

from conftest import G3_CASES
from oracle import refcpu as R


def gdata(g2, i):
    nb = int(g2["nb"][i])
    return R.GaussData(g2["nu"][i, :nb], g2["d"][i, :nb], g2["sig"][i, :nb])


def test_g1_planck_functions_bitexact(g1):
    beta, T, nu = g1["beta"], g1["T"], g1["nu"]
    for i, b in enumerate(beta):
        for j, t in enumerate(T):
            np.testing.assert_array_equal(R.greybody(b, t, nu, nu_norm=R.NU_B), g1["grey_norm"][i, j])
            np.testing.assert_array_equal(R.greybody(b, t, nu), g1["grey_unnorm"][i, j])
    for j, t in enumerate(T):
        np.testing.assert_array_equal(R.blackbody(t, nu, nu_norm=R.NU_B), g1["bb_norm"][j])
        np.testing.assert_array_equal(R.blackbody(t, nu), g1["bb_unnorm"][j])
    # the overflow edge: expm1 -> inf -> flux 0, not NaN (model.py:104)
    assert g1["grey_norm"][0, 0, -1] == 0.0


def test_anchor_values(g2):
    """BASELINE.md section 2 anchors on txt/Arp220.txt."""
    arp = gdata(g2, len(g2["nb"]) - 1)
    assert arp.n == 27
    assert R.sed_lnlike(R.M1N, (1, 2, 10), arp) == -54534735.92133923
    assert R.sed_lnlike(R.M1N_LOGA, (0.5, 1.5, 40), arp) == -31622129.359786913
    assert R.sed_lnlike(R.M2N, (1, 2, 10, 0.1, 2, 20), arp) == -54515895.05740812
    # last digit depends on numpy.dot's summation order: the reference reader hands BLAS
    # strided views (loadtxt(unpack=True)), the fixture stores contiguous copies
    assert R.sed_lnlike(R.M1, (2, 10), arp) == pytest.approx(2030.3003733933579, rel=1e-15)


@pytest.mark.parametrize("mid", range(8))
def test_g2_lnlike_prior_package(g2, mid):
    th = g2["theta_%d" % mid]
    nD = len(g2["nb"])
    for t in range(th.shape[0]):
        assert R.sed_prior(mid, th[t]) == g2["prior_%d" % mid][t]
        np.testing.assert_array_equal(R.sed_package(mid, th[t]), g2["package_%d" % mid][t])
    for di in range(nD):
        data = gdata(g2, di)
        for t in range(th.shape[0]):
            lnl, der = R.sed_lnlike(mid, th[t], data, return_derived=True)
            ref = g2["lnl_%d" % mid][di, t]
            if np.isnan(ref):
                assert np.isnan(lnl)
            elif di == nD - 1:          # Arp220: strided-vs-contiguous dot, see test_anchor_values
                assert lnl == pytest.approx(ref, rel=1e-14), (mid, di, t)
            else:
                assert lnl == ref, (mid, di, t)
            if mid not in (R.M1, R.M2):
                assert R.sed_chi2(mid, th[t], data) == pytest.approx(g2["chi2_%d" % mid][di, t], rel=1e-14, nan_ok=True)
            else:
                np.testing.assert_allclose(der, g2["derived_%d" % mid][di, t, :len(der)], rtol=1e-13)


@pytest.mark.parametrize("name", G3_CASES)
def test_g3_injected_chain_bitexact(g2, g3, name):
    mid = int(g3[name + "__model"])
    data = gdata(g2, int(g3[name + "__dataset"]))
    draws = R.InjectedDraws(g3[name + "__all_normals"], g3[name + "__all_uniforms"])
    ch = R.Chain(R.sed_problem(mid, data), g3[name + "__sigmas"], g3[name + "__start"], draws,
                 do_block=bool(g3[name + "__do_block"]))
    assert draws.k == int(g3[name + "__k_start"])
    nsteps = len(g3[name + "__lnPr"]) - 1
    ch.run(nsteps)
    samples, lnPr, acc = ch.arrays()
    np.testing.assert_array_equal(samples, g3[name + "__samples"])
    np.testing.assert_array_equal(lnPr, g3[name + "__lnPr"])
    np.testing.assert_array_equal(acc, g3[name + "__accepted"])
    assert ch.naccept == int(g3[name + "__naccept"])
    if ch.do_block:
        np.testing.assert_array_equal(ch.accept, g3[name + "__accept"])
    # A7 moments (burn=1, ddof=0)
    np.testing.assert_array_equal(ch.mean(1, 1), g3[name + "__mean"])
    np.testing.assert_array_equal(ch.stdev(1, 1), g3[name + "__stdev"])
    np.testing.assert_array_equal(ch.covar(1, 1), g3[name + "__covar"])
    # newMCMC: adapted sigmas, randomised restart, Cholesky factor, correlated stage
    assert draws.k == int(g3[name + "__new_kstart"])
    new = ch.new_chain(burn=1, stride=1, no_correlations=False)
    assert draws.k == int(g3[name + "__new_kend"])
    np.testing.assert_array_equal(new.prop.sigmas, g3[name + "__new_sigmas"])
    np.testing.assert_array_equal(new.samples[0], g3[name + "__new_start"])
    assert new.lnPr[0] == float(g3[name + "__new_lnPr0"])
    np.testing.assert_allclose(new.prop.sqrt_matrix, g3[name + "__new_sqrt"], rtol=0, atol=0)
    new.run(300)
    s2, l2, a2 = new.arrays()
    np.testing.assert_array_equal(s2, g3[name + "__corr_samples"])
    np.testing.assert_array_equal(l2, g3[name + "__corr_lnPr"])
    np.testing.assert_array_equal(a2, g3[name + "__corr_accepted"])


def test_g3_swap_and_fixed_block_events(g3):
    """The fixtures really contain the edge cases SURVEY 3.2 lists."""
    s = g3["m2n_block__samples"]
    assert np.all(s[:, 2] <= s[:, 5])               # package keeps T1<=T2 (model.py:439-443)
    acc = g3["m2n_fixedbeta__accept"]
    n = len(g3["m2n_fixedbeta__lnPr"]) - 1
    # sigma=0 blocks propose a zero move, which is "accepted" every time (MCMC.py:264-271)
    assert acc[1] == 1 + (n + 4) // 6 and acc[4] == 1 + (n + 1) // 6


def test_g3_gelman_rubin(g3):
    rhat, idx = R.gelman_rubin(g3["rhat__means"], g3["rhat__vars"])
    assert idx.all()
    np.testing.assert_array_equal(rhat, g3["rhat__value"])


def test_g4_sampler_arp220(g2, g4):
    """Config C1: the staged sampler() on txt/Arp220.txt, injected draws."""
    arp = gdata(g2, len(g2["nb"]) - 1)
    draws = R.InjectedDraws(g4["normals"], g4["uniforms"])
    chains, ana = R.sampler(R.sed_problem(R.M1N, arp), (1000, 3000), np.array((2., 4., 4.)),
                            np.array((1., 2., 10.)), draws, no_correlations=True, do_block=True)
    assert len(chains) == int(g4["nstage"])
    assert draws.k == len(g4["normals"])
    for i, ch in enumerate(chains):
        s, l, _ = ch.arrays()
        assert len(s) == int(g4["stage%d_n" % i])
        np.testing.assert_array_equal(s[::10], g4["stage%d_samples" % i])
        assert l[-1] == float(g4["stage%d_lnPr_last" % i])
        assert ch.naccept == int(g4["stage%d_naccept" % i])
        np.testing.assert_array_equal(ch.accept, g4["stage%d_accept" % i])
        np.testing.assert_array_equal(ch.prop.sigmas, g4["stage%d_sigmas" % i])
    np.testing.assert_array_equal(np.asarray(ana[0]), g4["ana_means"])
    np.testing.assert_array_equal(np.asarray(ana[1]), g4["ana_stdevs"])
    np.testing.assert_array_equal(np.asarray(ana[2]), g4["ana_covar"])


def test_philox_known_answer():
    """Philox4x32-10 known-answer vectors from the Random123 distribution
    (kat_vectors: counter, key -> output)."""
    z = np.zeros(4, np.uint32)
    out = R.philox4x32(z, np.zeros(2, np.uint32))
    assert [hex(int(x)) for x in out] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    f = np.full(4, 0xFFFFFFFF, np.uint32)
    out = R.philox4x32(f, np.full(2, 0xFFFFFFFF, np.uint32))
    assert [hex(int(x)) for x in out] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    c = np.array([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], np.uint32)
    k = np.array([0xa4093822, 0x299f31d0], np.uint32)
    out = R.philox4x32(c, k)
    assert [hex(int(x)) for x in out] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_philox_streams_are_standard_normal_and_uniform():
    idx = np.arange(200000)
    z = R.philox_normal(12345, 7, idx)
    u = R.philox_uniform(12345, 7, idx)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    assert abs(u.mean() - 0.5) < 0.005 and u.min() > 0 and u.max() < 1
    assert abs(np.corrcoef(z[::2], z[1::2])[0, 1]) < 0.01
