"""Parity of the HIP SED path (through the C ABI) against the reference's golden vectors and
the CPU oracle.  Needs an MI355X: run with `pytest -m gpu`."""
import numpy as np
import pytest

from conftest import G3_CASES
from oracle import refcpu as R

pytestmark = pytest.mark.gpu

RTOL_LNL = 1e-12      # BASELINE.json north_star: per-point lnL within 1e-12 relative, FP64


@pytest.fixture(scope="module")
def eng():
    from mcmc_amd import Engine
    e = Engine(device=0, seed=0x5EDC0FFEE)
    yield e
    e.close()


def assert_lnl_close(got, ref, rtol, what=""):
    got, ref = np.asarray(got), np.asarray(ref)
    nan_r, nan_g = np.isnan(ref), np.isnan(got)
    assert np.array_equal(nan_r, nan_g), "NaN pattern differs " + what
    inf_r = np.isinf(ref)
    assert np.array_equal(ref[inf_r], got[inf_r]), "inf pattern differs " + what
    ok = ~(nan_r | inf_r)
    err = np.abs(got[ok] - ref[ok]) / np.maximum(np.abs(ref[ok]), 1e-300)
    assert err.size == 0 or err.max() <= rtol, "%s max rel err %.3e > %.1e" % (what, err.max(), rtol)
    return err.max() if err.size else 0.0


def load_g2_catalogue(eng, g2, mid):
    eng.sed_load(mid, g2["nb"], g2["nu"], g2["d"], g2["sig"])


@pytest.mark.parametrize("mid", range(8))
def test_lnlike_matches_reference_golden(eng, g2, mid):
    """G2: like.lnLike for every model class on 12 ragged synthetic sources + Arp220."""
    load_g2_catalogue(eng, g2, mid)
    th = g2["theta_%d" % mid]
    nD = len(g2["nb"])
    theta = np.broadcast_to(th[None], (nD,) + th.shape).copy()
    lnl, der = eng.sed_lnlike(theta, derived=True)
    ref = g2["lnl_%d" % mid]
    # every model at the north-star tolerance.  SEDLikelihood2's fixture is the reference class
    # itself (its linalg.solve(sym_pos=True) read as assume_a="pos", oracle/make_golden.py); the
    # device solves the 2 x 2 system in the same operation order, so what remains is the few-ulp
    # difference of the band fluxes through a system whose condition number reaches ~1e6 here
    # (the ML amplitudes, not lnL, carry it: they are held to 1e-9).
    if mid == R.M2:
        # 1e-12 wherever the 2 x 2 system is conditioned better than 1e4; beyond that the bound is
        # the input rounding (one ulp of the band fluxes) times the condition number of F^T N^-1 F -
        # measured: worst 1.6e-12 at cond 1.6e4 (conditions reach 6e11 in this fixture), the reference itself moves by 1.6e-13 there between
        # NumPy's SIMD code paths (tests/golden/simd_spread.json)
        cond = np.empty(ref.shape)
        for di in range(nD):
            nbd = g2["nb"][di]
            data = R.GaussData(g2["nu"][di, :nbd], g2["d"][di, :nbd], g2["sig"][di, :nbd])
            for t in range(th.shape[0]):
                with np.errstate(all="ignore"):
                    cond[di, t] = np.linalg.cond(data.quadform(R.sed_flux(R.M2, th[t], data.freq)))
        err = np.abs(lnl - ref) / np.abs(ref)
        bound = np.maximum(RTOL_LNL, 8 * np.finfo(float).eps * cond)
        assert np.all(err <= bound), (err.max(), cond[np.unravel_index(np.argmax(err / bound), err.shape)])
        assert np.all(err[cond < 1e4] <= RTOL_LNL) and (cond < 1e4).mean() > 0.3
        worst = err.max()
        print("model 5: cond median %.1e max %.1e; worst err %.2e at cond %.1e"
              % (np.median(cond), cond.max(), worst, cond.flat[np.argmax(err)]))
    else:
        worst = assert_lnl_close(lnl, ref, RTOL_LNL, "model %d" % mid)
    print("model %d: max rel err %.2e" % (mid, worst))
    if mid == R.M1:
        assert_lnl_close(der[..., 0], g2["derived_%d" % mid][..., 0], 1e-12, "ML amplitude")
    if mid == R.M2:
        wd = assert_lnl_close(der, g2["derived_%d" % mid], 1e-9, "ML amplitudes")
        print("model %d: ML amplitudes max rel err %.2e" % (mid, wd))
    pr, pk = eng.sed_prior_package(th)
    # exp(-3 nu0) of the optically-thick prior differs by an ulp between libms
    np.testing.assert_allclose(pr, g2["prior_%d" % mid], rtol=4e-16, atol=0)
    np.testing.assert_array_equal(pr > 0, g2["prior_%d" % mid] > 0)
    np.testing.assert_array_equal(pk, g2["package_%d" % mid])


def test_anchor_values_arp220(eng, g2):
    """BASELINE.md section 2 anchors, computed by the reference on txt/Arp220.txt."""
    i = len(g2["nb"]) - 1
    for mid, th, val in [(R.M1N, (1, 2, 10), -54534735.92133923),
                         (R.M1N_LOGA, (0.5, 1.5, 40), -31622129.359786913),
                         (R.M2N, (1, 2, 10, 0.1, 2, 20), -54515895.05740812),
                         (R.M1, (2, 10), 2030.3003733933579)]:
        load_g2_catalogue(eng, g2, mid)
        got = eng.sed_lnlike(np.array(th, float).reshape(1, 1, -1), s0=i)[0, 0]
        assert abs(got - val) <= RTOL_LNL * abs(val), (mid, got, val)


@pytest.mark.parametrize("mid,C", [(R.M2N, 64), (R.M2N_LOGA, 16), (R.M1N, 7), (R.THICK, 64)])
def test_lnlike_matches_oracle_batch(eng, mid, C):
    """Larger seeded batch against the oracle; C values that do and do not divide the block."""
    S = 300
    nu, d, sig, truth = R.synth_sed_catalogue(S, seed=11)
    nb = np.full(S, 8, np.int32)
    nb[::5] = 6                                    # ragged
    eng.sed_load(mid, nb, nu, d, sig)
    rng = np.random.default_rng(5)
    P = R.MODEL_NPARAM[mid]
    if mid in (R.M2N, R.M2N_LOGA):
        th = truth[:, None, :] * rng.uniform(0.8, 1.25, (S, C, 6))
        if mid == R.M2N_LOGA:
            th[..., 0] = np.log10(th[..., 0]); th[..., 3] = np.log10(th[..., 3])
    elif mid == R.M1N:
        th = truth[:, None, :3] * rng.uniform(0.8, 1.25, (S, C, 3))
    else:
        th = np.concatenate([truth[:, None, :3] * rng.uniform(0.8, 1.25, (S, C, 3)),
                             rng.uniform(0.3, 3.0, (S, C, 1))], axis=2)
    got = eng.sed_lnlike(th)
    ref = np.empty((S, C))
    for s in range(S):
        data = R.GaussData(nu[s, :nb[s]], d[s, :nb[s]], sig[s, :nb[s]])
        for c in range(C):
            ref[s, c] = R.sed_lnlike(mid, th[s, c], data)
    worst = assert_lnl_close(got, ref, RTOL_LNL, "oracle batch model %d" % mid)
    print("model %d C=%d: max rel err %.2e" % (mid, C, worst))


def test_marginalised_model_far_above_tmax(eng):
    """submmModel1's prior has no upper temperature limit (model.py:244-255): points at T >> Tmax,
    where 1 - 2^-y of the fast path would lose bits, take the exact path and stay at 1e-12."""
    S, C = 40, 16
    nu, d, sig, truth = R.synth_sed_catalogue(S, seed=21)
    nb = np.full(S, 8, np.int32)
    eng.sed_load(R.M1, nb, nu, d, sig)
    rng = np.random.default_rng(3)
    th = np.stack([rng.uniform(0.5, 2.5, (S, C)), 10.0 ** rng.uniform(1.0, 7.5, (S, C))], axis=-1)
    got = eng.sed_lnlike(th)
    ref = np.array([[R.sed_lnlike(R.M1, th[s, c], R.GaussData(nu[s], d[s], sig[s])) for c in range(C)]
                    for s in range(S)])
    assert (th[..., 1] > 100.0).sum() > S * C // 2
    assert_lnl_close(got, ref, RTOL_LNL, "M1 at large T")


def test_empty_and_single_band_sources(eng):
    """nb=0 (no data: lnL = -0) and nb=1 sources next to a full one."""
    nu, d, sig, truth = R.synth_sed_catalogue(3, seed=3)
    nb = np.array([0, 1, 8], np.int32)
    eng.sed_load(R.M2N, nb, nu, d, sig)
    th = np.broadcast_to(truth[:, None, :], (3, 4, 6)).copy()
    got = eng.sed_lnlike(th)
    assert np.all(got[0] == 0.0)
    for s in (1, 2):
        ref = R.sed_lnlike(R.M2N, truth[s], R.GaussData(nu[s, :nb[s]], d[s, :nb[s]], sig[s, :nb[s]]))
        assert_lnl_close(got[s], np.full(4, ref), RTOL_LNL)


def _g3_setup(eng, g2, g3, name):
    mid = int(g3[name + "__model"])
    di = int(g3[name + "__dataset"])
    nb = g2["nb"][di:di + 1]
    eng.sed_load(mid, nb, g2["nu"][di:di + 1], g2["d"][di:di + 1], g2["sig"][di:di + 1])
    return mid, di


@pytest.mark.parametrize("name", G3_CASES)
def test_injected_chain_matches_reference_trace(eng, g2, g3, name):
    """G3: with the reference's own proposal draws injected, the device chain reproduces the
    reference trace step for step (accept pattern identical, samples/lnPr to rounding), its
    naccept / per-block counts, and mean/stdev with burn=1."""
    _check_g3_chain(eng, g2, g3, name, incremental=False)


@pytest.mark.parametrize("name", ["m2n_block", "m2n_fixedbeta"])
def test_injected_chain_lane_per_chain_matches_reference_trace(eng, g2, g3, name):
    """The same G3 traces with the lane = chain mapping forced: a one-chain job is served by the
    (chain, band) instantiation otherwise, and the kernel behind bench.py's `value` -
    k_mh_run<M2N, fast, plain, lane = chain, single-parameter blocks>, three waves per SIMD - would be
    held to the reference's own traces only through the lane-mapping tests."""
    eng.mh_set_lane_map(1)
    try:
        _check_g3_chain(eng, g2, g3, name, incremental=False, lane_group=1)
    finally:
        eng.mh_set_lane_map(0)


@pytest.mark.parametrize("name", ["m2n_block", "m2n_fixedbeta"])
def test_incremental_injected_chain_matches_reference_trace(eng, g2, g3, name):
    """The same G3 traces of the reference through the incremental sampler (k_mh_run_inc,
    mi355_mh_set_incremental): two-component model, one parameter per block, one with sigma = 0
    blocks."""
    eng.mh_set_incremental(True)
    try:
        _check_g3_chain(eng, g2, g3, name, incremental=True)
    finally:
        eng.mh_set_incremental(False)


def _check_g3_chain(eng, g2, g3, name, incremental, lane_group=None):
    mid, _ = _g3_setup(eng, g2, g3, name)
    P = R.MODEL_NPARAM[mid]
    do_block = bool(g3[name + "__do_block"])
    samples, lnPr, acc = g3[name + "__samples"], g3[name + "__lnPr"], g3[name + "__accepted"]
    k0 = int(g3[name + "__k_start"])
    nsteps = len(lnPr) - 1
    normals = g3[name + "__normals"][k0:k0 + nsteps].reshape(nsteps, 1, P)
    uniforms = g3[name + "__uniforms"][k0:k0 + nsteps].reshape(nsteps, 1)
    eng.mh_init(1, samples[0], g3[name + "__sigmas"],
                param_blocks=np.arange(P) if do_block else None, randomize_start=False)
    m0 = eng.mh_moments()
    assert abs(m0["lnPr"][0, 0] - lnPr[0]) <= 1e-12 * abs(lnPr[0])
    assert eng.mh_incremental_active == incremental
    if lane_group is not None:
        assert eng.mh_lane_group == lane_group
    eng.mh_run(nsteps, thin=1, inj_normals=normals, inj_uniforms=uniforms)
    tr_s, tr_l, tr_a = eng.mh_trace()
    np.testing.assert_array_equal(tr_a[:, 0, 0], acc[1:])
    np.testing.assert_allclose(tr_s[:, 0, 0, :], samples[1:], rtol=1e-13, atol=0)
    np.testing.assert_allclose(tr_l[:, 0, 0], lnPr[1:], rtol=1e-11 if mid == R.M2 else 1e-12, atol=0)
    mom = eng.mh_moments()
    assert mom["naccept"][0, 0] == int(g3[name + "__naccept"])
    if do_block:
        np.testing.assert_array_equal(mom["accept"][0, 0], g3[name + "__accept"])
    # mean()/stdev() of the reference are package()d (swapped if <T1> > <T2>); undo nothing: the
    # fixtures' chains all keep <T1> < <T2>, checked here
    np.testing.assert_allclose(R.sed_package(mid, mom["mean"][0, 0]), g3[name + "__mean"], rtol=1e-11)
    np.testing.assert_allclose(np.sqrt(mom["var"][0, 0]), g3[name + "__stdev"], rtol=1e-9, atol=1e-300)


def test_philox_streams_match_oracle(eng):
    """The device Philox4x32-10 / Box-Muller streams equal the oracle's restatement."""
    N, nsteps, npn = 37, 11, 2
    z, u = eng.rng_dump(5, nsteps, N, npn)
    steps = np.arange(5, 5 + nsteps)[:, None, None]
    chains = np.arange(N)[None, :, None]
    idx = steps * npn + np.arange(npn)[None, None, :]
    zr = R.philox_normal(eng.seed, chains, idx)
    ur = R.philox_uniform(eng.seed, chains[..., 0], steps[..., 0])
    np.testing.assert_array_equal(u, ur)          # integer pipeline + exact scaling: bit-exact
    np.testing.assert_allclose(z, zr, rtol=0, atol=5e-15)


class PhiloxChainDraws:
    """What the kernels' free-running mode consumes for one chain, as an oracle draw source:
    start draw k (k=1,2,..) uses start-stream indices (stage<<40) + k*PE + i; step t uses normal
    index t*npn + rank(param) and uniform index t."""

    def __init__(self, seed, chain, P, npn, do_block, stage, step0=0):
        self.seed, self.chain, self.P, self.npn = seed, chain, P, npn
        self.do_block, self.stage = do_block, stage
        self.PE = (P + 1) & ~1
        self.k = 1
        self.t = step0
        self.in_start = True

    def begin_steps(self):
        self.in_start = False

    def normal(self, n):
        if self.in_start:
            base = (self.stage << 40) + self.k * self.PE
            self.k += 1
            return R.philox_normal(self.seed, self.chain, base + np.arange(n), R.RNG_START)
        t = self.t
        self.t += 1
        rank = np.zeros(n, dtype=np.int64) if self.do_block else np.arange(n)
        return R.philox_normal(self.seed, self.chain, t * self.npn + rank)

    def uniform(self):
        return float(R.philox_uniform(self.seed, self.chain, self.t - 1))


@pytest.mark.parametrize("mid,do_block", [(R.M2N, True), (R.M2N, False), (R.M1N_LOGA, True)])
def test_free_running_chains_match_oracle(eng, mid, do_block):
    """Free-running Philox mode, start randomisation included: every chain equals the oracle
    run on the same counter-based draws (S=3 sources x C=5 chains, 400 steps)."""
    S, C, nsteps = 3, 5, 400
    nu, d, sig, truth = R.synth_sed_catalogue(S, seed=21)
    nb = np.array([8, 7, 8], np.int32)
    eng.sed_load(mid, nb, nu, d, sig)
    P = R.MODEL_NPARAM[mid]
    if mid == R.M2N:
        start, sigma = np.array([1, 2, 10, 0.1, 2, 20.]), np.array([0.1, 0.2, 1.5, 0.05, 0.2, 2.0])
    else:
        start, sigma = np.array([0.0, 2, 20.]), np.array([0.05, 0.1, 1.0])
    stage = eng.stage
    eng.mh_init(C, start, sigma, param_blocks=np.arange(P) if do_block else None, randomize_start=True)
    theta0 = eng.mh_moments()["theta"]
    step0 = eng.step_total
    eng.mh_run(nsteps, thin=1)
    tr_s, tr_l, tr_a = eng.mh_trace()
    mom = eng.mh_moments()
    for s in range(S):
        data = R.GaussData(nu[s, :nb[s]], d[s, :nb[s]], sig[s, :nb[s]])
        for c in range(C):
            npn = 1 if do_block else (P + 1) & ~1
            draws = PhiloxChainDraws(eng.seed, s * C + c, P, npn, do_block, stage, step0)
            ch = R.Chain(R.sed_problem(mid, data), sigma, start, draws, do_block=do_block)
            draws.begin_steps()
            np.testing.assert_allclose(theta0[s, c], ch.samples[0], rtol=1e-13)
            ch.run(nsteps)
            os_, ol, oa = ch.arrays()
            np.testing.assert_array_equal(tr_a[:, s, c], oa[1:])
            np.testing.assert_allclose(tr_s[:, s, c], os_[1:], rtol=1e-12)
            np.testing.assert_allclose(tr_l[:, s, c], ol[1:], rtol=1e-12)
            assert mom["naccept"][s, c] == ch.naccept


@pytest.mark.parametrize("mid", [R.M2N, R.M2N_LOGA])
def test_incremental_sampler_equals_full_evaluation(eng, mid):
    """k_mh_run_inc against k_mh_run on the same Philox draws: 37 ragged sources (0 to 8 bands,
    odd counts included) x 70 chains, three launches whose step counts leave the block counter
    and the draw-pair parity at every offset, wide steps so that chains swap their components
    (package()) many times.  Accept decisions identical, positions / lnPr / moments to 1e-12;
    both are held to the oracle by the tests above."""
    S, C = 37, 70
    nu, d, sig, truth = R.synth_sed_catalogue(S, seed=77)
    nb = (np.arange(S) % 9).astype(np.int32)
    nb[-3:] = 8
    eng.sed_load(mid, nb, nu, d, sig)
    if mid == R.M2N:
        start, sigma = np.array([1, 2, 10, 0.1, 2, 20.]), np.array([0.3, 0.4, 6.0, 0.2, 0.4, 8.0])
    else:
        start, sigma = np.array([0.0, 2, 10, -1.0, 2, 20.]), np.array([0.2, 0.4, 6.0, 0.3, 0.4, 8.0])
    launches = (301, 64, 235)
    out = {}
    for inc in (False, True):
        eng.set_stream_pos(1000, 3)
        eng.mh_set_incremental(inc)
        try:
            eng.mh_init(C, start, sigma, param_blocks=np.arange(6), randomize_start=True)
            assert eng.mh_incremental_active == inc
            traces = []
            for n in launches:
                eng.mh_run(n, thin=1)
                traces.append([x.copy() for x in eng.mh_trace()])
            out[inc] = (traces, eng.mh_moments(), eng.rhat())
        finally:
            eng.mh_set_incremental(False)
    swaps = 0
    for (fs, fl, fa), (is_, il, ia) in zip(out[False][0], out[True][0]):
        np.testing.assert_array_equal(ia, fa)
        np.testing.assert_allclose(is_, fs, rtol=1e-12, atol=0)
        # sources with 1 to 6 bands are fitted exactly by some points (chi2 -> 0, the residuals
        # cancel): there the 1e-12 is relative to the terms of the sum, hence the absolute floor
        np.testing.assert_allclose(il, fl, rtol=RTOL_LNL, atol=1e-15)
        swaps += int(np.sum((np.diff(fs[..., 0], axis=0) != 0) & (np.diff(fs[..., 3], axis=0) != 0)))
    assert swaps > 100, "the case is meant to exercise package() swaps"
    mf, mi = out[False][1], out[True][1]
    np.testing.assert_array_equal(mi["naccept"], mf["naccept"])
    np.testing.assert_array_equal(mi["accept"], mf["accept"])
    np.testing.assert_allclose(mi["mean"], mf["mean"], rtol=1e-11)
    np.testing.assert_allclose(mi["var"], mf["var"], rtol=1e-9, atol=1e-300)
    np.testing.assert_allclose(out[True][2], out[False][2], rtol=1e-9)


def test_incremental_falls_back_where_it_does_not_apply(eng):
    """Asked for but not applicable (all parameters in one block; a one-component model; more
    than 8 bands): mi355_mh_run evaluates in full as before and says so."""
    nu, d, sig, truth = R.synth_sed_catalogue(2, seed=5)
    nb = np.array([8, 8], np.int32)
    eng.mh_set_incremental(True)
    try:
        eng.sed_load(R.M2N, nb, nu, d, sig)
        eng.mh_init(4, truth[:, None, :].repeat(4, 1), np.full(6, 0.01), param_blocks=None,
                    randomize_start=False)
        assert not eng.mh_incremental_active
        eng.mh_init(4, truth[:, None, :].repeat(4, 1), np.full(6, 0.01), param_blocks=np.arange(6),
                    randomize_start=False)
        assert eng.mh_incremental_active
        eng.sed_load(R.M1N, nb, nu, d, sig)
        eng.mh_init(4, np.array([1.0, 2, 20]), np.full(3, 0.01), param_blocks=np.arange(3),
                    randomize_start=False)
        assert not eng.mh_incremental_active
        eng.mh_run(10)
    finally:
        eng.mh_set_incremental(False)


@pytest.mark.parametrize("C,nrows", [(64, 100), (40, 37), (100, 128), (256, 50), (2, 9)])
def test_ess_matches_bda3_restatement(C, nrows):
    """mi355_mh_ess (no reference counterpart: BASELINE.json's ESS/s) against the numpy
    restatement of the same estimator on the same trace; chain counts on either side of the
    64 / 128 / 256 slot layouts and row counts that are not multiples of the lag batch."""
    from mcmc_amd import Engine, synth
    S = 6
    nu, d, sig, _ = synth.sed_catalogue(S, seed=21)
    with Engine(seed=77) as eng:
        eng.sed_load("M2N", np.full(S, 8, np.int32), nu, d, sig)
        eng.mh_init(C, synth.M2N_START, synth.M2N_PROP_SIGMAS, param_blocks=np.arange(6))
        eng.mh_run(600)
        eng.mh_adapt()
        eng.mh_run(12 * nrows, thin=12)
        tr, _, _ = eng.mh_trace()                       # [nrows, S, C, 6]
        ess = eng.mh_ess()
    assert ess.shape == (S, 6) and tr.shape[0] == nrows
    for s in range(S):
        for p in range(6):
            ref = R.ess_bda3(tr[:, s, :, p])
            assert abs(ess[s, p] - ref) <= 1e-9 * abs(ref), (s, p, ess[s, p], ref)
    assert np.all(ess > 0.5) and np.all(ess <= nrows * C * 1.5)


@pytest.mark.parametrize("mid", [R.M1N, R.M2N])
def test_catalogue_outside_the_fast_path_domain(eng, mid):
    """The clamp-free base-2 kernels are only used when every band keeps 0.02 <= h nu/kT over the
    prior box (and the exponents below 300): a radio point (1.4 GHz: h nu/k Tmax = 7e-4) and a
    mid-infrared one (h nu/k Tmin > 300) send the whole catalogue through the exact path; same
    parity bar, and the same sources without those bands agree on the common points."""
    S, C = 40, 16
    nu, d, sig, truth = R.synth_sed_catalogue(S, seed=17)
    nu2, d2, sig2 = nu.copy(), d.copy(), sig.copy()
    nu2[:, 0] = 1.4                                   # below the 1 - 2^-y guard
    nu2[:, 1] = 3.0e4                                 # h nu / k Tmin = 480
    rng = np.random.default_rng(9)
    P = R.MODEL_NPARAM[mid]
    th = truth[:, None, :P] * rng.uniform(0.8, 1.25, (S, C, P))
    eng.sed_load(mid, np.full(S, 8, np.int32), nu2, d2, sig2)
    got = eng.sed_lnlike(th)
    ref = np.array([[R.sed_lnlike(mid, th[s, c], R.GaussData(nu2[s], d2[s], sig2[s])) for c in range(C)]
                    for s in range(S)])
    assert_lnl_close(got, ref, RTOL_LNL, "exact path model %d" % mid)
    # a short chain on that catalogue against the oracle, draw for draw
    eng.mh_init(2, truth[:, None, :P].repeat(2, axis=1), np.full(P, 0.05), param_blocks=np.arange(P),
                randomize_start=False)
    z, u = eng.rng_dump(eng.step_total, 30, S * 2, 1)
    eng.mh_run(30, thin=1)
    tr, lp, _ = eng.mh_trace()
    s, c = 3, 1
    n = s * 2 + c
    zz = np.zeros((30, P))
    for t in range(30):
        zz[t, t % P] = z[t, n, 0]
    ch = R.Chain(R.sed_problem(mid, R.GaussData(nu2[s], d2[s], sig2[s])), np.full(P, 0.05), truth[s, :P],
                 R.InjectedDraws(zz, u[:, n]), do_block=True, randomize_start=False)
    ch.run(30)
    np.testing.assert_allclose(tr[:, s, c], np.asarray(ch.samples)[1:], rtol=1e-12)
    np.testing.assert_allclose(lp[:, s, c], np.asarray(ch.lnPr)[1:], rtol=1e-11)


def test_incremental_sampler_splits_a_mixed_catalogue():
    """The incremental sampler on a catalogue with a radio point: the chain blocks that hold such a source
    run the plain per-lane kernel, all the others k_mh_run_inc beside them - the same chains as the plain
    sampler gives on the same catalogue (which the test below pins source by source)."""
    from mcmc_amd import Engine
    mid, S, C, P, nsteps = R.M2N, 70, 16, 6, 60
    nu, d, sig, truth = R.synth_sed_catalogue(S, seed=23)
    nu2 = nu.copy()
    nu2[[3, 41], 0] = 1.4
    nb = np.full(S, 8, np.int32)
    start = truth[:, None, :P].repeat(C, axis=1)
    out = []
    for inc in (True, False):
        with Engine(seed=77) as e:
            e.sed_load(mid, nb, nu2, d, sig)
            assert e.sed_fast_sources == S - 2
            e.mh_set_incremental(inc)
            e.mh_init(C, start, np.full(P, 0.03), param_blocks=np.arange(P), randomize_start=False)
            assert e.mh_incremental_active == inc
            e.mh_run(nsteps, thin=1)
            tr, lp, acc = e.mh_trace()
            e.mh_run(25)
            out.append((tr, lp, acc, e.mh_moments()))
    # (the incremental evaluation sums a chain's chi-square from cached band terms: lnPr agrees to rounding,
    # the decisions and therefore the samples are the same)
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][2], out[1][2])
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=1e-12)
    for k, v in out[0][3].items():
        np.testing.assert_allclose(v, out[1][3][k], rtol=1e-12, err_msg=k)


@pytest.mark.parametrize("mid,C", [(R.M2N, 16), (R.M1N, 100), (R.THICK, 64)])
def test_mixed_catalogue_splits_source_by_source(mid, C):
    """One radio point must not move a whole catalogue to the exact path: the sources with a band
    outside the fast kernels' domain evaluate through the exact formulas by themselves, every other
    source evaluates exactly as it does in a catalogue without them (bit for bit: same kernels, same
    draws), likelihood batches and chains alike - chains in the two side-by-side launches of
    mi355_mh_run, whether a fast source shares its block of chains with a slow one or not."""
    from mcmc_amd import Engine
    S, nsteps = 70, 40
    P = R.MODEL_NPARAM[mid]
    nu, d, sig, truth = R.synth_sed_catalogue(S, seed=23)
    slow = np.array([3, 41, 69])
    nu2 = nu.copy()
    nu2[slow[:2], 0] = 1.4                                # radio point
    nu2[slow[2], 1] = 3.0e4                               # mid-infrared point
    nb = np.full(S, 8, np.int32)
    rng = np.random.default_rng(4)
    th = truth[:, None, :P] * rng.uniform(0.8, 1.25, (S, C, P))
    start = truth[:, None, :P].repeat(C, axis=1)
    out = {}
    for tag, cat in (("mixed", nu2), ("clean", nu)):
        with Engine(seed=77) as e:
            e.sed_load(mid, nb, cat, d, sig)
            assert e.sed_fast_sources == (S - slow.size if tag == "mixed" else S)
            lnl = e.sed_lnlike(th)
            e.mh_init(C, start, np.full(P, 0.03), param_blocks=np.arange(P), randomize_start=False)
            z, u = e.rng_dump(e.step_total, nsteps, S * C, 1)
            e.mh_run(nsteps, thin=1)
            tr, lp, acc = e.mh_trace()
            e.mh_run(25)                                  # no trace
            out[tag] = (lnl, tr, lp, acc, e.mh_moments(), z, u)
    fast = np.setdiff1d(np.arange(S), slow)
    for a, b in zip(out["mixed"][:4], out["clean"][:4]):
        ax = 0 if a.ndim == 2 else 1
        assert np.array_equal(np.take(a, fast, axis=ax), np.take(b, fast, axis=ax), equal_nan=True)
    for k, v in out["mixed"][4].items():
        assert np.array_equal(v[fast], out["clean"][4][k][fast], equal_nan=True), k
    # the slow sources against the oracle: likelihood batch, and one chain of each draw for draw
    lnl, tr, lp, _, _, z, u = out["mixed"]
    ref = np.array([[R.sed_lnlike(mid, th[s, c], R.GaussData(nu2[s], d[s], sig[s])) for c in range(C)]
                    for s in slow])
    assert_lnl_close(lnl[slow], ref, RTOL_LNL, "slow sources model %d" % mid)
    for s in slow:
        c = int(s) % C
        n = s * C + c
        zz = np.zeros((nsteps, P))
        for t in range(nsteps):
            zz[t, t % P] = z[t, n, 0]
        ch = R.Chain(R.sed_problem(mid, R.GaussData(nu2[s], d[s], sig[s])), np.full(P, 0.03), truth[s, :P],
                     R.InjectedDraws(zz, u[:, n]), do_block=True, randomize_start=False)
        ch.run(nsteps)
        np.testing.assert_allclose(tr[:, s, c], np.asarray(ch.samples)[1:], rtol=1e-12)
        np.testing.assert_allclose(lp[:, s, c], np.asarray(ch.lnPr)[1:], rtol=1e-11)


@pytest.mark.parametrize("mid,nbands,G,C", [(R.M2N, 8, 8, 16), (R.M1N, 27, 32, 5), (R.THICK, 8, 8, 100)])
@pytest.mark.parametrize("ext", [False, True])
def test_lane_mappings_split_a_mixed_catalogue_alike(mid, nbands, G, C, ext):
    """A small job with a radio point under lanes = (chain, band): the chain blocks that hold such a source
    run the exact formulas band-parallel too (each lane its band, the terms added in band order), all the
    others keep the fast band-parallel kernel beside them - the same chains, bit for bit, as forced lane =
    chain gives on the same catalogue (which test_mixed_catalogue_splits_source_by_source pins to the
    oracle); plain and extended sampler."""
    from mcmc_amd import Engine
    S, nsteps = 41, 50
    P = R.MODEL_NPARAM[mid]
    nu, d, sig, truth = R.synth_sed_catalogue(S, seed=31, nb=nbands)
    nu[[2, 17, 40], 0] = 1.4
    nb = np.full(S, nbands, np.int32)
    nb[5] = nbands - 1
    th0 = _start_point(mid, truth)
    rng = np.random.default_rng(8)
    A = rng.normal(size=(P, P))
    cov = A @ A.T + P * np.eye(P)
    dg = np.sqrt(np.diag(cov))
    L = np.linalg.cholesky(cov / dg[:, None] / dg[None, :])
    out = {}
    for lane_map in (1, 2):
        with Engine(seed=11) as e:
            e.mh_set_lane_map(lane_map)
            e.sed_load(mid, nb, nu, d, sig)
            assert e.sed_fast_sources == S - 3
            start = th0[:, None, :].repeat(C, axis=1)
            if ext:
                e.mh_init_problem("sed", C, start, np.full(P, 0.02), param_blocks=np.arange(P),
                                  randomize_start=False, track_cov=True)
                e.mh_set_sqrt_matrix(L)
                assert e.mh_fused_active
            else:
                e.mh_init(C, start, np.full(P, 0.02), param_blocks=np.arange(P), randomize_start=False)
                assert e.mh_lane_group == (G if lane_map == 2 else 1)
            e.mh_run(nsteps, thin=1)
            tr = e.mh_trace()
            e.mh_run(21)
            out[lane_map] = (tr, e.mh_moments(), e.mh_covar(normalized=False) if ext else ())
    for x, y in zip(out[1][0], out[2][0]):
        assert np.array_equal(x, y, equal_nan=True)
    for k in out[1][1]:
        assert np.array_equal(out[1][1][k], out[2][1][k], equal_nan=True), k
    for x, y in zip(out[1][2], out[2][2]):
        assert np.array_equal(x, y, equal_nan=True)
    assert out[1][0][2].mean() > 0.05


def _start_point(mid, truth):
    P = R.MODEL_NPARAM[mid]
    if mid == R.M1:
        return truth[:, 1:3].copy()
    if mid == R.M2:
        return truth[:, [1, 2, 4, 5]].copy()
    th = truth[:, :P].copy()
    if mid in (R.M1N_LOGA, R.M2N_LOGA, R.THICK_LOGA):
        th[:, 0] = np.log10(th[:, 0])
    if mid == R.M2N_LOGA:
        th[:, 3] = np.log10(th[:, 3])
    if mid in (R.THICK, R.THICK_LOGA):
        th[:, 3] = 0.7
    return th


@pytest.mark.parametrize("mid", [R.M1N, R.M1N_LOGA, R.M2N, R.M2N_LOGA, R.M1, R.THICK, R.THICK_LOGA])
@pytest.mark.parametrize("nbands,G", [(8, 8), (27, 32)])
def test_lane_mappings_give_identical_chains(mid, nbands, G):
    """North_star's two lane mappings of the fused sampler - lane = chain, and lanes = (chain, band)
    with the chi-square terms added in band order - produce the same chains bit for bit: traces,
    moments, accept counts; block proposals and all-parameter proposals, free-running and injected
    draws, ragged band counts, chain counts that do not fill a block."""
    from mcmc_amd import Engine
    S, C, nsteps = 5, 7, 60
    P = R.MODEL_NPARAM[mid]
    nu, d, sig, truth = R.synth_sed_catalogue(S, seed=29, nb=nbands)
    nb = np.full(S, nbands, np.int32)
    nb[1] = nbands - 2
    nb[3] = max(1, nbands // 2)
    th0 = _start_point(mid, truth)
    rng = np.random.default_rng(8)
    zn = rng.standard_normal((25, S * C, P))
    zu = rng.uniform(size=(25, S * C))
    out = {}
    for lane_map in (1, 2):
        res = []
        with Engine(seed=2024) as e:
            e.mh_set_lane_map(lane_map)
            e.sed_load(mid, nb, nu, d, sig)
            for blocks in (np.arange(P), None):
                e.mh_init(C, th0[:, None, :].repeat(C, axis=1), np.full(P, 0.02), param_blocks=blocks,
                          randomize_start=False)
                assert e.mh_lane_group == (G if lane_map == 2 else 1)
                e.mh_run(nsteps, thin=1)
                res.append(e.mh_trace())
                e.mh_run(33)
                e.mh_run(25, thin=5, inj_normals=zn, inj_uniforms=zu)
                res.append(e.mh_trace())
                res.append(e.mh_moments())
        out[lane_map] = res
    for a, b in zip(out[1], out[2]):
        if isinstance(a, dict):
            for k in a:
                assert np.array_equal(a[k], b[k], equal_nan=True), k
        else:
            for x, y in zip(a, b):
                assert np.array_equal(x, y, equal_nan=True)
    assert out[1][0][2].mean() > 0.05                     # the chains did move


@pytest.mark.parametrize("lane_map", [1, 2])
def test_many_bands_one_chain_per_source(lane_map):
    """One chain per source (the reference's own layout: one MCMC object per catalogue entry) with
    27 bands: 64 band tables per block are 110 KB of LDS - the kernels opt in to it instead of
    refusing the catalogue.  Likelihood batch, start, free-running chains against the oracle."""
    from mcmc_amd import Engine
    S, C, P, nsteps = 150, 1, 6, 40
    nu, d, sig, truth = R.synth_sed_catalogue(S, seed=41, nb=27)
    nb = np.full(S, 27, np.int32)
    nb[::7] = 19
    rng = np.random.default_rng(3)
    th = truth[:, None, :] * rng.uniform(0.9, 1.1, (S, C, P))
    with Engine(seed=5) as e:
        e.mh_set_lane_map(lane_map)
        e.sed_load(R.M2N, nb, nu, d, sig)
        got = e.sed_lnlike(th)
        ref = np.array([[R.sed_lnlike(R.M2N, th[s, 0], R.GaussData(nu[s, :nb[s]], d[s, :nb[s]], sig[s, :nb[s]]))]
                        for s in range(S)])
        assert_lnl_close(got, ref, RTOL_LNL, "27 bands, one chain per source")
        e.mh_init(C, truth[:, None, :], np.full(P, 0.02), param_blocks=np.arange(P), randomize_start=False)
        z, u = e.rng_dump(e.step_total, nsteps, S * C, 1)
        e.mh_run(nsteps, thin=1)
        tr, lp, _ = e.mh_trace()
    for s in (0, 7, 93, 149):
        zz = np.zeros((nsteps, P))
        for t in range(nsteps):
            zz[t, t % P] = z[t, s, 0]
        ch = R.Chain(R.sed_problem(R.M2N, R.GaussData(nu[s, :nb[s]], d[s, :nb[s]], sig[s, :nb[s]])), np.full(P, 0.02),
                     truth[s], R.InjectedDraws(zz, u[:, s]), do_block=True, randomize_start=False)
        ch.run(nsteps)
        np.testing.assert_allclose(tr[:, s, 0], np.asarray(ch.samples)[1:], rtol=1e-12)
        np.testing.assert_allclose(lp[:, s, 0], np.asarray(ch.lnPr)[1:], rtol=1e-11)


def test_lane_mapping_is_picked_by_job_size():
    """Automatic choice: (chain, band) while chains x G fit one wave per SIMD, lane = chain beyond;
    lane = chain wherever the band-parallel kernel does not apply."""
    from mcmc_amd import Engine
    nu, d, sig, truth = R.synth_sed_catalogue(300, seed=2)
    nb = np.full(300, 8, np.int32)
    with Engine(seed=1) as e:
        e.sed_load(R.M2N, nb[:4], nu[:4], d[:4], sig[:4])
        e.mh_init(16, truth[0], np.full(6, 0.02), param_blocks=np.arange(6))
        assert e.mh_lane_group == 8                       # 64 chains
        e.mh_set_incremental(True)
        assert e.mh_lane_group == 1                       # the incremental sampler is lane = chain
        e.mh_set_incremental(False)
        e.sed_load(R.M2N, nb, nu, d, sig)
        e.mh_init(64, truth[0], np.full(6, 0.02), param_blocks=np.arange(6))
        assert e.mh_lane_group == 1                       # 19 200 chains x 8 lanes > 65 536
        e.sed_load(R.M2, nb[:4], nu[:4], d[:4], sig[:4])
        e.mh_init(16, truth[0, [1, 2, 4, 5]], np.full(4, 0.02), param_blocks=np.arange(4))
        assert e.mh_lane_group == 1                       # SEDLikelihood2: lane = chain only
        nu2 = nu[:4].copy()
        nu2[2, 0] = 1.4
        e.sed_load(R.M2N, nb[:4], nu2, d[:4], sig[:4])
        e.mh_init(16, truth[0], np.full(6, 0.02), param_blocks=np.arange(6))
        assert e.mh_lane_group == 8                       # a source on the exact path: its block of chains runs lane = chain beside
        nu2[:, 0] = 1.4
        e.sed_load(R.M2N, nb[:4], nu2, d[:4], sig[:4])
        e.mh_init(16, truth[0], np.full(6, 0.02), param_blocks=np.arange(6))
        assert e.mh_lane_group == 1                       # every source on the exact path


def test_sed_abi_errors_leave_the_context_usable():
    """Misuse of the SED / MH entry points returns an error with a message (never a fault), and
    the context keeps working afterwards."""
    from mcmc_amd import Engine
    import ctypes as C
    from mcmc_amd._lib import Mi355Error, check
    nu, d, sig, truth = R.synth_sed_catalogue(2, seed=1)
    nb = np.array([8, 8], np.int32)
    with Engine() as e:
        with pytest.raises(Mi355Error, match="not initialised"):
            e.mh_run(10)                                                        # nothing loaded yet
        assert e.sed_fast_sources == -1 and e.mh_lane_group == -1               # nothing to report on yet
        with pytest.raises(Mi355Error, match="lane map"):
            e.mh_set_lane_map(3)
        with pytest.raises(Mi355Error, match="outside"):
            e.sed_load(R.M2N, np.array([8, 9], np.int32), nu, d, sig)          # nb > nb_max
        bad = nu.copy()
        bad[1, 3] = -5.0
        with pytest.raises(Mi355Error, match="out of range"):
            e.sed_load(R.M2N, nb, bad, d, sig)                                  # negative frequency
        bad[1, 3] = np.nan
        with pytest.raises(Mi355Error, match="out of range"):
            e.sed_load(R.M2N, nb, bad, d, sig)
        with pytest.raises(Mi355Error, match="unknown SED model"):
            m = e.default_model(R.M2N)
            m.model = 99
            e.sed_load(R.M2N, nb, nu, d, sig, consts=m)
        e.sed_load(R.M2N, nb, nu, d, sig)
        with pytest.raises(Mi355Error, match="not initialised"):
            e.mh_run(10)
        with pytest.raises(Mi355Error, match="not initialised"):
            e.mh_moments()
        with pytest.raises(Mi355Error, match="nblock=8 outside"):
            e.mh_init(4, truth[0], np.full(6, 0.01), param_blocks=np.array([0, 1, 2, 3, 4, 7]),
                      randomize_start=False)
        e.mh_init(4, truth[0], np.full(6, 0.01), param_blocks=np.arange(6), randomize_start=False)
        with pytest.raises(Mi355Error, match="no samples"):
            e.mh_adapt()
        with pytest.raises(Mi355Error, match="negative"):
            e.mh_run(-1)
        with pytest.raises(Mi355Error):                                        # normals without uniforms
            z = np.zeros((5, 8, 6))
            check(e.lib.mi355_mh_run(e._h, 5, 1, z.ctypes.data_as(C.POINTER(C.c_double)), None))
        e.mh_run(12)
        with pytest.raises(Mi355Error, match="kept no trace"):
            e.mh_trace()
        with pytest.raises(Mi355Error, match="trace rows"):
            e.mh_ess()
        # still in working order: the chain equals a fresh run of the same stream position
        got = e.sed_lnlike(np.broadcast_to(truth[:, None, :], (2, 3, 6)).copy())
        for s in range(2):
            ref = R.sed_lnlike(R.M2N, truth[s], R.GaussData(nu[s], d[s], sig[s]))
            assert_lnl_close(got[s], np.full(3, ref), RTOL_LNL)
        assert e.mh_moments()["naccept"].shape == (2, 4)


def test_module_level_planck_functions_match_reference_grids(g1):
    """model.greybody / model.blackbody (model.py:80-111) as module-level functions over the HIP
    library, on the reference's own (beta, T, nu) grids of G1: normalised and un-normalised forms,
    T down to the prior edge, the overflowing expm1 that must give exactly 0."""
    from mcmc_amd.submmSED import model as M
    beta, T, nu = g1["beta"], g1["T"], g1["nu"]
    for j, t in enumerate(T):
        for i, b in enumerate(beta):
            assert_lnl_close(M.greybody(b, t, nu, nu_norm=M.nu_b), g1["grey_norm"][i, j], RTOL_LNL,
                             "greybody norm beta=%g T=%g" % (b, t))
            assert_lnl_close(M.greybody(b, t, nu), g1["grey_unnorm"][i, j], RTOL_LNL,
                             "greybody beta=%g T=%g" % (b, t))
        assert_lnl_close(M.blackbody(t, nu, nu_norm=M.nu_b), g1["bb_norm"][j], RTOL_LNL, "blackbody norm")
        assert_lnl_close(M.blackbody(t, nu), g1["bb_unnorm"][j], RTOL_LNL, "blackbody")
    assert M.greybody(0.0, 3.0, 60000.0, nu_norm=M.nu_b) == 0.0          # scalar in, scalar out
    # a normalisation frequency other than the module's nu_b (oracle restatement of :107-109)
    got = M.greybody(1.5, 20.0, nu, nu_norm=353.0)
    assert_lnl_close(got, R.greybody(1.5, 20.0, nu, nu_norm=353.0), RTOL_LNL)
    assert M.nu_b == 1000.0


def test_ragged_catalogue_many_chains_match_vector_oracle():
    """Waves that span sources with different band counts (C = 37 chains per source, 1 to 8 bands):
    all 7400 free-running chains of 200 sources against the vectorised oracle, 300 steps, on the
    same Philox draws.  Padded bands carry weight 0 in both."""
    from mcmc_amd import Engine
    S, C, P, nsteps = 200, 37, 6, 300
    nu, d, sig, truth = R.synth_sed_catalogue(S, seed=31)
    nb = (1 + np.arange(S) % 8).astype(np.int32)
    rng = np.random.default_rng(6)
    theta0 = truth[:, None, :] * rng.uniform(0.9, 1.1, (S, C, P))
    flat0 = theta0.reshape(-1, P)
    sw = flat0[:, 2] > flat0[:, 5]
    flat0[sw] = flat0[sw][:, [3, 4, 5, 0, 1, 2]]
    sigma = np.array([0.05, 0.1, 1.0, 0.05, 0.1, 2.5])
    with Engine(seed=0xA11CE) as eng:
        eng.sed_load(R.M2N, nb, nu, d, sig)
        eng.mh_init(C, flat0.reshape(S, C, P), sigma, param_blocks=np.arange(P), randomize_start=False)
        step0, seed = eng.step_total, eng.seed
        eng.mh_run(nsteps)
        mom = eng.mh_moments()
    live = np.arange(8)[None, :] < nb[:, None]
    nu_m = np.where(live, nu, R.NU_B)
    d_m = np.where(live, d, 0.0)
    src = np.repeat(np.arange(S), C)
    vc = R.VectorChainsM2N(nu_m[src], d_m[src], sig[src], sigma, flat0[0])
    vc.isig2 = np.where(live[src], vc.isig2, 0.0)
    vc.theta = flat0.copy()
    vc.lnPr = vc.lnlike(vc.theta)
    chains = np.arange(S * C)
    for t in range(step0, step0 + nsteps):
        z = np.zeros((S * C, P))
        z[:, vc.iblock] = R.philox_normal(seed, chains, np.full(S * C, t))
        vc.step(z, R.philox_uniform(seed, chains, np.full(S * C, t)))
    np.testing.assert_array_equal(mom["naccept"].reshape(-1), vc.naccept)
    np.testing.assert_allclose(mom["theta"].reshape(-1, P), vc.theta, rtol=1e-12)
    np.testing.assert_allclose(mom["lnPr"].reshape(-1), vc.lnPr, rtol=1e-12, atol=1e-13)
