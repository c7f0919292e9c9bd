"""The step-wise Metropolis-Hastings engine (mi355_mh_init_problem: any problem kind, correlated
and rotated proposals) against the reference's own chains (golden fixtures, injected draws), the
oracle on the device's Philox streams, and the fused SED kernel."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
RTOL = 1e-12


def load(name):
    return np.load(os.path.join(GOLD, name))


def replay(eng, g, tag, rows, nsteps, P):
    """inject rows [rows, rows+nsteps) of the fixture's draw streams, compare with chain `tag`"""
    zn = g["chain__normals"][rows:rows + nsteps, None, :P]
    zu = g["chain__uniforms"][rows:rows + nsteps, None]
    eng.mh_run(nsteps, thin=1, inj_normals=np.ascontiguousarray(zn), inj_uniforms=np.ascontiguousarray(zu))
    s, l, a = eng.mh_trace()
    ref_s, ref_l = g["chain__%s_samples" % tag], g["chain__%s_lnPr" % tag]
    ref_a = g["chain__%s_accepted" % tag]
    assert np.array_equal(a[:, 0, 0], ref_a[1:])
    np.testing.assert_allclose(s[:, 0, 0], ref_s[1:], rtol=RTOL, atol=1e-15)
    np.testing.assert_allclose(l[:, 0, 0], ref_l[1:], rtol=RTOL)
    mom = eng.mh_moments()
    assert int(mom["naccept"][0, 0]) == int(g["chain__%s_naccept" % tag])
    np.testing.assert_allclose(mom["mean"][0, 0], ref_s[1:].mean(axis=0), rtol=1e-10)
    # (numpy's two-pass variance of a column that never moved is rounding noise ~1e-28, ours is 0)
    np.testing.assert_allclose(mom["var"][0, 0], ref_s[1:].var(axis=0), rtol=1e-8,
                               atol=1e-24 * float(np.max(ref_s ** 2)))
    return mom


def cl_engine(g, name="xcalib"):
    from test_cl_gpu import mirror
    from mcmc_amd import Engine
    model, d = mirror(g, name)
    eng = Engine()
    eng.cl_load([d])
    eng.cl_set_binning(model.bins, model.shapefun)
    return eng, model


@pytest.mark.parametrize("fused", [True, False])
def test_cl_reference_chains_block_rotated_correlated(fused):
    """The reference's own binnedCl chains (G5: doBlock, then rotated, then Cholesky-correlated
    proposals, 11-point calibration marginalisation) draw for draw - through the fused one-launch
    kernel (csrc/cl_mh_kernels.hpp) and through the launch-per-step engine."""
    g = load("cl_g5.npz")
    eng, model = cl_engine(g)
    P = model.nparam
    k = g["chain__k"]
    blocks = np.arange(P)
    with eng:
        eng.mh_set_fused(fused)
        # stage 0: doBlock, uncorrelated, 400 steps of the reference's chain
        eng.mh_init_problem("cl", 1, g["chain__c0_samples"][0], np.full(P, 0.05), param_blocks=blocks,
                            randomize_start=False, track_cov=True)
        assert eng.mh_fused_active == fused
        mom = replay(eng, g, "c0", k[0], 400, P)
        assert np.array_equal(mom["accept"][0, 0], g["chain__accept0"])
        # newMCMC(rotateParams=True) on the device: sigma = 2.4/sqrt(P)*stdev, Cholesky of the
        # normalised covariance (MCMC.py:343-413) - against the reference's new proposal
        ref = g["chain__c0_samples"][1:]
        cov = eng.mh_covar(normalized=False)[0, 0]
        np.testing.assert_allclose(cov, np.cov(ref.T, bias=True), rtol=1e-8, atol=1e-18)
        eng.mh_adapt_corr(rotate=True)
        assert eng.proposal_mode == 2
        np.testing.assert_allclose(eng.mh_sigmas()[0, 0], g["chain__c1_sigmas"], rtol=1e-9)
        np.testing.assert_allclose(eng.mh_sqrt_matrix()[0, 0], g["chain__c1_sqrt"], rtol=1e-7, atol=1e-9)
        # stage 1 draw for draw: the reference's start, widths and sqrt matrix, rotated proposals
        eng.mh_init_problem("cl", 1, g["chain__c1_samples"][0], g["chain__c1_sigmas"],
                            param_blocks=blocks, randomize_start=False, track_cov=True)
        eng.mh_set_sqrt_matrix(g["chain__c1_sqrt"], rotate=True)
        mom = replay(eng, g, "c1", k[2], 300, P)
        assert np.array_equal(mom["accept"][0, 0], g["chain__accept1"])
        eng.mh_adapt_corr(no_correlations=False, do_block=False)
        assert eng.proposal_mode == 1
        np.testing.assert_allclose(eng.mh_sqrt_matrix()[0, 0], g["chain__c2_sqrt"], rtol=1e-7, atol=1e-9)
        # stage 2: all parameters at once through the Cholesky factor
        eng.mh_init_problem("cl", 1, g["chain__c2_samples"][0], g["chain__c2_sigmas"],
                            param_blocks=None, randomize_start=False)
        eng.mh_set_sqrt_matrix(g["chain__c2_sqrt"], rotate=False)
        replay(eng, g, "c2", k[4], 200, P)


@pytest.mark.parametrize("fused", [True, False])
def test_cl_reference_chain_on_the_11x11_grid(fused):
    """The reference's own chain on a data set with calibration AND beam uncertainty - the full 11 x 11
    numeric marginalisation of ClData_CosmoMC.py:317-372, config C4's shape - draw for draw: through the
    fused kernel (replica waves share out the 121 points, csrc/cl_mhw_kernels.hpp) and through the
    launch-per-step engine."""
    g = load("cl_g5.npz")
    eng, model = cl_engine(g, "xcalibbeam")
    P = model.nparam
    k = g["grid121__k"]
    with eng:
        eng.mh_set_fused("always" if fused else False)    # (one chain: the size policy would pick the step-wise engine)
        eng.mh_init_problem("cl", 1, g["grid121__c0_samples"][0], np.full(P, 0.05), param_blocks=np.arange(P),
                            randomize_start=False)
        assert eng.mh_fused_active == fused
        n = 150
        zn = g["grid121__normals"][k[0]:k[0] + n, None, :P]
        zu = g["grid121__uniforms"][k[0]:k[0] + n, None]
        eng.mh_run(n, thin=1, inj_normals=np.ascontiguousarray(zn), inj_uniforms=np.ascontiguousarray(zu))
        s, l, a = eng.mh_trace()
        assert np.array_equal(a[:, 0, 0], g["grid121__c0_accepted"][1:])
        np.testing.assert_allclose(s[:, 0, 0], g["grid121__c0_samples"][1:], rtol=RTOL, atol=1e-15)
        np.testing.assert_allclose(l[:, 0, 0], g["grid121__c0_lnPr"][1:], rtol=RTOL)
        mom = eng.mh_moments()
        assert int(mom["naccept"][0, 0]) == int(g["grid121__c0_naccept"])
        assert np.array_equal(mom["accept"][0, 0], g["grid121__accept0"])


@pytest.mark.parametrize("fused", [False, True])
def test_beam_reference_chains(fused):
    """The reference's own chains on a 64 x 64 beam map, draw for draw, through the launch-per-step engine and
    through the fused beam sampler (csrc/beam_mh_kernels.hpp)."""
    from mcmc_amd import Engine
    g = load("beam_g6.npz")
    k = g["chain__k"]
    sig = g["m64a__sig"]
    with Engine() as eng:
        eng.mh_set_fused("always" if fused else False)
        eng.beam_load(g["m64a__x"], g["m64a__y"], g["m64a__d"], sig[0], center_min=g["centerMin"],
                      center_max=g["centerMax"], sig_min=float(g["sigMin"]), sig_max=float(g["sigMax"]))
        blocks = np.arange(7)                      # ProcessBeamData.py:215-216 (mu, phi in no block)
        eng.mh_init_problem("beam", 1, g["chain__c0_samples"][0], g["chain__sigmas"],
                            param_blocks=blocks, randomize_start=False, track_cov=True)
        assert eng.mh_fused_active == fused
        mom = replay(eng, g, "c0", k[0], 500, 9)
        assert np.array_equal(mom["accept"][0, 0], g["chain__c0_accept"])
        covn = eng.mh_covar(normalized=True)[0, 0]
        eng.mh_adapt_corr(no_correlations=False)
        # x never moved in this stage: the reference's stdev of that column is rounding noise
        # (1.3e-14, so it stays in the Cholesky subset with noise correlations), ours is exactly 0
        # and the parameter drops out as a fixed one (Proposal.py:85-101).  Compare what is defined.
        moved = g["chain__c1_sigmas"] > 1e-10
        sg = eng.mh_sigmas()[0, 0]
        np.testing.assert_allclose(sg[moved], g["chain__c1_sigmas"][moved], rtol=1e-9)
        assert np.all(sg[~moved] == 0)
        L = eng.mh_sqrt_matrix()[0, 0]
        sub = np.ix_(sg > 0, sg > 0)
        np.testing.assert_allclose((L @ L.T)[sub], covn[sub], rtol=1e-10, atol=1e-12)
        assert np.all(L[~(sg > 0)] == 0) and np.all(np.triu(L, 1) == 0)
        eng.mh_init_problem("beam", 1, g["chain__c1_samples"][0], g["chain__c1_sigmas"],
                            param_blocks=blocks, randomize_start=False)
        eng.mh_set_sqrt_matrix(g["chain__c1_sqrt"], rotate=False)
        mom = replay(eng, g, "c1", k[2], 300, 9)
        assert np.array_equal(mom["accept"][0, 0], g["chain__c1_accept"])


def test_sed_stepwise_equals_fused_kernel():
    """Same seed, same Philox streams: the step-wise engine and the fused kernel must produce the
    same chains bit for bit (same likelihood code, same accept rule, same moment updates)."""
    from mcmc_amd import Engine, synth
    S, C, n1, n2 = 5, 64, 120, 80
    nu, d, sig, _ = synth.sed_catalogue(S, seed=3)
    nb = np.full(S, 8, np.int32)
    out = []
    for stepwise in (False, True):
        with Engine(seed=99) as eng:
            eng.sed_load("M2N", nb, nu, d, sig)
            if stepwise:
                eng.mh_init_problem("sed", C, synth.M2N_START, synth.M2N_PROP_SIGMAS,
                                    param_blocks=np.arange(6), randomize_start=True)
            else:
                eng.mh_init(C, synth.M2N_START, synth.M2N_PROP_SIGMAS, param_blocks=np.arange(6),
                            randomize_start=True)
            eng.mh_run(n1, thin=1)
            t1 = eng.mh_trace()
            eng.mh_adapt()
            sg = eng.mh_sigmas()
            eng.mh_run(n2, thin=1)
            t2 = eng.mh_trace()
            mom = eng.mh_moments()
            out.append((t1, sg, t2, mom, eng.rhat()))
    (a1, asg, a2, am, ar), (b1, bsg, b2, bm, br) = out
    for x, y in zip(a1 + a2, b1 + b2):
        assert np.array_equal(x, y)
    assert np.array_equal(asg, bsg)
    for key in ("mean", "var", "naccept", "accept", "lnPr", "theta"):
        assert np.array_equal(am[key], bm[key]), key
    assert np.array_equal(ar, br)


@pytest.mark.parametrize("fused", [True, False])
def test_sed_reference_correlated_stage(fused):
    """newMCMC(noCorrelations=False) + 300 Cholesky-correlated steps of the reference's own
    two-component chain (sed_g3 'm2n_block'), draw for draw - through the fused SED kernel's
    extended instantiation and through the launch-per-step engine."""
    from mcmc_amd import Engine
    g2, g3 = load("sed_g2.npz"), load("sed_g3.npz")
    name = "m2n_block"
    di = int(g3[name + "__dataset"])
    nbd = int(g2["nb"][di])
    with Engine() as eng:
        eng.mh_set_fused(fused)
        eng.sed_load("M2N", g2["nb"][di:di + 1], g2["nu"][di:di + 1], g2["d"][di:di + 1],
                     g2["sig"][di:di + 1])
        ref_s = g3[name + "__corr_samples"]
        eng.mh_init_problem("sed", 1, ref_s[0], g3[name + "__new_sigmas"], param_blocks=np.arange(6),
                            randomize_start=False)
        assert eng.mh_fused_active == fused
        eng.mh_set_sqrt_matrix(g3[name + "__new_sqrt"], rotate=False)
        k0 = int(g3[name + "__new_kend"])
        zn = g3[name + "__all_normals"][k0:k0 + 300, None, :]
        zu = g3[name + "__all_uniforms"][k0:k0 + 300, None]
        eng.mh_run(300, thin=1, inj_normals=np.ascontiguousarray(zn), inj_uniforms=np.ascontiguousarray(zu))
        s, l, a = eng.mh_trace()
    assert nbd > 0
    assert np.array_equal(a[:, 0, 0], g3[name + "__corr_accepted"][1:])
    np.testing.assert_allclose(s[:, 0, 0], ref_s[1:], rtol=RTOL)
    np.testing.assert_allclose(l[:, 0, 0], g3[name + "__corr_lnPr"][1:], rtol=RTOL)


@pytest.mark.parametrize("mode", ["uncorr_block", "corr_all", "rotate_block"])
def test_cl_free_running_philox_equals_oracle(mode):
    """Free-running device chains (Philox draws) against the oracle driven by the restated
    Philox streams: pins the stream layout of every proposal mode."""
    from oracle import refcl as RC
    from oracle import refcpu as R
    from mcmc_amd import Engine
    model, d, truth = RC.synth_cl_problem(nb=24, npar=6, calib=0.0, beam=0.0, lmax=700, seed=5)
    P, C, nsteps, seed = model.nparam, 5, 60, 4321
    rng = np.random.default_rng(8)
    A = rng.normal(size=(P, P))
    cov = A @ A.T
    dg = np.sqrt(np.diag(cov))
    L = np.linalg.cholesky(cov / dg[:, None] / dg[None, :])
    sig = np.full(P, 0.03)
    block = mode != "corr_all"
    with Engine(seed=seed) as eng:
        eng.cl_load([d])
        eng.cl_set_binning(model.bins, model.shapefun)
        eng.mh_init_problem("cl", C, truth, sig, param_blocks=np.arange(P) if block else None,
                            randomize_start=False)
        if mode != "uncorr_block":
            eng.mh_set_sqrt_matrix(L, rotate=(mode == "rotate_block"))
        eng.mh_run(nsteps, thin=1)
        s, l, a = eng.mh_trace()
    npn = 1 if block else (P + 1) & ~1
    pb = RC.cl_problem(model, [d])
    for c in (0, 3):
        draws = R.PhiloxDraws(seed, c, npn)
        ch = R.Chain(pb, sig, truth, draws, do_block=block, randomize_start=False)
        if mode != "uncorr_block":
            ch.prop.sqrt_matrix = L
            ch.prop.rotate = mode == "rotate_block"
        if mode == "uncorr_block":
            # the reference draws n normals per step and uses offset[block]; the device draws one
            # normal per step for a single-parameter block: give the oracle that draw in slot `block`
            class OneDraw(R.PhiloxDraws):
                def normal(self_, n):
                    z = super().normal(1)
                    out = np.zeros(n)
                    out[(self_.step - 1) % P] = z[0]
                    return out
            ch.draws = OneDraw(seed, c, npn)
        ch.run(nsteps)
        os_, ol, oa = ch.arrays()
        assert np.array_equal(a[:, 0, c], oa[1:])
        np.testing.assert_allclose(s[:, 0, c], os_[1:], rtol=RTOL)
        np.testing.assert_allclose(l[:, 0, c], ol[1:], rtol=1e-11)


def test_stepwise_errors():
    from mcmc_amd import Engine, Mi355Error
    with Engine() as eng:
        with pytest.raises(Mi355Error, match="no binned Cl problem"):
            eng.cl_npar = 3
            eng.mh_init_problem("cl", 2, np.ones(3), np.ones(3))


@pytest.mark.parametrize("kind", ["cl", "beam", "sed"])
def test_graph_replay_equals_direct_launches(kind):
    """Long free-running runs replay a captured hipGraph of 8 MH steps; a run that keeps a trace
    launches every step directly.  Same seed -> same chains, bit for bit."""
    from oracle import refbeam as RB
    from oracle import refcl as RC
    from mcmc_amd import Engine, synth
    res = []
    for traced in (False, True):
        with Engine(seed=2024) as eng:
            if kind == "cl":
                model, d, truth = RC.synth_cl_problem(nb=24, npar=6, calib=0.05, beam=0.0, lmax=700, seed=5)
                eng.cl_load([d])
                eng.cl_set_binning(model.bins, model.shapefun)
                eng.mh_init_problem("cl", 40, truth, np.full(6, 0.03), param_blocks=np.arange(6),
                                    track_cov=True)
            elif kind == "beam":
                x, y, dm, noise, tr, prior = RB.synth_beam_maps(3, n=32, seed=2)
                eng.beam_load(x, y, dm, noise, center_min=prior.center_min, center_max=prior.center_max,
                              sig_min=prior.sig_min, sig_max=prior.sig_max)
                sg = np.array([0.05, 0.05, 0.05, 0.05, 0.02, 0.1, 0.002, 0, 0])
                eng.mh_init_problem("beam", 8, tr[:, None, :].repeat(8, 1), sg, param_blocks=np.arange(7))
            else:
                nu, d, sig, _ = synth.sed_catalogue(4, seed=3)
                eng.sed_load("M2N", np.full(4, 8, np.int32), nu, d, sig)
                eng.mh_init_problem("sed", 32, synth.M2N_START, synth.M2N_PROP_SIGMAS,
                                    param_blocks=np.arange(6))
            eng.mh_run(77, thin=1 if traced else 0)
            if kind == "cl":
                eng.mh_adapt_corr(rotate=True)
            else:
                eng.mh_adapt()
            eng.mh_run(53, thin=1 if traced else 0)
            eng.mh_run(40, thin=1 if traced else 0)      # second replay of the same graph
            res.append(eng.mh_moments())
    for key in ("theta", "lnPr", "mean", "var", "naccept", "accept"):
        assert np.array_equal(res[0][key], res[1][key]), key


def test_loading_data_ends_the_running_sampler():
    """A load frees what the sampler (and its captured graph) points at: the next mh_run must
    fail cleanly instead of replaying a stale graph."""
    from mcmc_amd import Engine
    from mcmc_amd._lib import Mi355Error
    from oracle import refcl as RC
    model, d, truth = RC.synth_cl_problem(nb=24, npar=6, calib=0.0, beam=0.0, lmax=600)
    with Engine(seed=3) as eng:
        eng.cl_load([d])
        eng.cl_set_binning(model.bins, model.shapefun)
        eng.mh_init_problem("cl", 32, truth, np.full(model.nparam, 0.02), randomize_start=False)
        eng.mh_run(64)                                  # long enough for the graph path
        eng.cl_load([d])
        eng.cl_set_binning(model.bins, model.shapefun)
        with pytest.raises(Mi355Error, match="not initialised"):
            eng.mh_run(64)
        eng.mh_init_problem("cl", 32, truth, np.full(model.nparam, 0.02), randomize_start=False)
        eng.mh_run(64)
        assert np.all(np.isfinite(eng.mh_moments()["lnPr"]))


@pytest.mark.parametrize("kind", ["cl", "beam"])
def test_graph_replay_survives_scratch_regrowth(kind):
    """A host-side likelihood call on a batch much larger than the sampler's N chains regrows the
    Cl / beam scratch buffers whose addresses a captured graph has baked in: the next long run must
    recapture (not replay freed memory) and still equal direct launches bit for bit."""
    from oracle import refbeam as RB
    from oracle import refcl as RC
    from mcmc_amd import Engine
    rng = np.random.default_rng(4)
    res = []
    for traced in (False, True):
        with Engine(seed=77) as eng:
            if kind == "cl":
                model, d, truth = RC.synth_cl_problem(nb=24, npar=6, calib=0.05, beam=0.0, lmax=700, seed=5)
                eng.cl_load([d])
                eng.cl_set_binning(model.bins, model.shapefun)
                eng.mh_init_problem("cl", 1, truth, np.full(6, 0.03), param_blocks=np.arange(6))
                big = truth[None, :] * rng.uniform(0.9, 1.1, (3000, 6))
            else:
                x, y, dm, noise, tr, prior = RB.synth_beam_maps(2, n=32, seed=2)
                eng.beam_load(x, y, dm, noise, center_min=prior.center_min, center_max=prior.center_max,
                              sig_min=prior.sig_min, sig_max=prior.sig_max)
                sg = np.array([0.05, 0.05, 0.05, 0.05, 0.02, 0.1, 0.002, 0, 0])
                eng.mh_init_problem("beam", 2, tr[:, None, :].repeat(2, 1), sg, param_blocks=np.arange(7))
                big = tr[:, None, :].repeat(600, 1) * (1 + 0.01 * rng.standard_normal((2, 600, 9)))
                big[..., 7] = 1.0
            eng.mh_run(64, thin=1 if traced else 0)
            before = eng.cl_lnlike(big) if kind == "cl" else eng.beam_lnlike(big)
            eng.mh_run(64, thin=1 if traced else 0)
            after = eng.cl_lnlike(big) if kind == "cl" else eng.beam_lnlike(big)
            assert np.array_equal(before, after)
            res.append(eng.mh_moments())
    for key in ("theta", "lnPr", "mean", "var", "naccept", "accept"):
        assert np.array_equal(res[0][key], res[1][key]), key


def _mixed_catalogue(S=20, seed=12):
    from mcmc_amd import synth
    nu, d, sig, truth = synth.sed_catalogue(S, seed=seed)
    nu = nu.copy()
    nu[[2, 17], 0] = 1.4                                   # radio points: sources 2 and 17 take the exact path
    return nu, d, sig, truth, np.full(S, 8, np.int32)


def test_sed_mixed_catalogue_stepwise_equals_fused():
    """A catalogue with sources outside the fast kernels' domain, chains started by mi355_mh_init_problem:
    the extended fused sampler runs the blocks that hold such a source in its per-lane instantiation
    beside the fast one for the others, and the launch-per-step engine follows the per-source flag in
    k_sed_lnlike - the same chains, bit for bit, as the two side-by-side launches of the plain fused
    sampler (mi355_mh_init) on the same draws."""
    from mcmc_amd import Engine
    S, C, P = 20, 16, 6
    nu, d, sig, truth, nb = _mixed_catalogue(S)
    start = truth[:, None, :].repeat(C, axis=1)
    sg = np.full(P, 0.02)
    with Engine(seed=31) as e:
        e.sed_load("M2N", nb, nu, d, sig)
        assert e.sed_fast_sources == S - 2
        e.mh_init(C, start, sg, param_blocks=np.arange(P), randomize_start=False)
        e.mh_run(50, thin=1)
        ta = e.mh_trace()
        ma = e.mh_moments()
    for fused in (True, False):
        with Engine(seed=31) as e:
            e.sed_load("M2N", nb, nu, d, sig)
            e.mh_init_problem("sed", C, start, sg, param_blocks=np.arange(P), randomize_start=False)
            e.mh_set_fused(fused)
            assert e.mh_fused_active == fused
            e.mh_run(50, thin=1)
            tb = e.mh_trace()
            mb = e.mh_moments()
        for x, y in zip(ta, tb):
            assert np.array_equal(x, y, equal_nan=True), fused
        for k in ("theta", "lnPr", "mean", "var", "naccept", "accept"):
            assert np.array_equal(ma[k], mb[k], equal_nan=True), (k, fused)


@pytest.mark.parametrize("model,lane_map", [("M2N", 1), ("M2N", 2), ("M2N_LOGA", 1), ("M1N", 1), ("THICK", 2)])
def test_sed_in_place_instantiation_equals_stepwise_engine(model, lane_map):
    """Single-parameter blocks through mi355_mh_init run the in-place instantiation of the fused sampler
    (the moved parameter changed in th itself behind a scalar branch, package()'s exchange undone on a
    rejected lane): the same chains, bit for bit, as the launch-per-step engine on the same draws - with
    stage-0 widths, so that T1 > T2 exchanges and rejections of exchanged proposals both occur."""
    from mcmc_amd import Engine, synth
    S, C = 6, 24
    nu, d, sig, truth = synth.sed_catalogue(S, seed=8)
    nb = np.array([8, 8, 7, 8, 5, 8], np.int32)
    P = {"M2N": 6, "M2N_LOGA": 6, "M1N": 3, "THICK": 4}[model]
    start = {"M2N": synth.M2N_START, "M2N_LOGA": [0.0, 2.0, 10.0, -1.0, 2.0, 20.0], "M1N": [1.0, 2.0, 10.0],
             "THICK": [1.0, 2.0, 10.0, 1.0]}[model]
    sg = {"M2N": synth.M2N_PROP_SIGMAS, "M2N_LOGA": [1.0, 4.0, 4.0, 1.0, 4.0, 4.0], "M1N": [2.0, 4.0, 4.0],
          "THICK": [2.0, 4.0, 4.0, 0.5]}[model]
    out = []
    for stepwise in (False, True):
        with Engine(seed=19) as e:
            e.mh_set_lane_map(lane_map)
            e.sed_load(model, nb, nu, d, sig)
            if stepwise:
                e.mh_init_problem("sed", C, np.array(start), np.array(sg), param_blocks=np.arange(P),
                                  randomize_start=False)
                e.mh_set_fused(False)
                assert not e.mh_fused_active
            else:
                e.mh_init(C, np.array(start), np.array(sg), param_blocks=np.arange(P), randomize_start=False)
            e.mh_run(300, thin=1)
            e.mh_run(33, thin=1)                           # a second launch: odd stream position, block phase
            out.append((e.mh_trace(), e.mh_moments()))
    (ta, ma), (tb, mb) = out
    for x, y in zip(ta, tb):
        assert np.array_equal(x, y, equal_nan=True)
    for k in ("theta", "lnPr", "mean", "var", "naccept", "accept"):
        assert np.array_equal(ma[k], mb[k], equal_nan=True), k
    if P == 6:                                             # exchanges happened: the kept T1 <= T2 and moves were made
        th = ta[0]
        assert np.all(th[..., 2] <= th[..., 5]) and ma["naccept"].sum() > 0


@pytest.mark.parametrize("kind", ["mixed", "none_fast"])
def test_sed_extended_kernel_off_the_fast_domain(kind):
    """Correlated proposals and co-moments on a catalogue that is partly ("mixed") or wholly
    ("none_fast": a radio point in every source) outside the fast kernels' domain: the extended
    fused sampler serves it (no drop to 30 us per step) and equals the launch-per-step engine."""
    from mcmc_amd import Engine
    S, C, P = 20, 16, 6
    nu, d, sig, truth, nb = _mixed_catalogue(S)
    if kind == "none_fast":
        nu[:, 0] = 1.4
    rng = np.random.default_rng(3)
    A = rng.normal(size=(P, P))
    cov = A @ A.T + P * np.eye(P)
    dg = np.sqrt(np.diag(cov))
    L = np.linalg.cholesky(cov / dg[:, None] / dg[None, :])
    start = truth[:, None, :].repeat(C, axis=1)
    res = []
    for fused in (True, False):
        with Engine(seed=5) as e:
            e.sed_load("M2N", nb, nu, d, sig)
            assert e.sed_fast_sources == (S - 2 if kind == "mixed" else 0)
            e.mh_init_problem("sed", C, start, np.full(P, 0.02), randomize_start=False, track_cov=True,
                              param_blocks=np.arange(P) if kind == "mixed" else None)
            e.mh_set_fused(fused)
            assert e.mh_fused_active == fused
            e.mh_set_sqrt_matrix(L, rotate=(kind == "mixed"))
            e.mh_run(40, thin=2)
            res.append((e.mh_trace(), e.mh_moments(), e.mh_covar(normalized=False)))
    (ta, ma, ca), (tb, mb, cb) = res
    for x, y in zip(ta, tb):
        assert np.array_equal(x, y, equal_nan=True)
    for k in ma:
        assert np.array_equal(ma[k], mb[k], equal_nan=True), k
    assert np.array_equal(ca, cb, equal_nan=True)
    assert ma["naccept"].sum() > 0


@pytest.mark.parametrize("lane_map", [1, 2])
def test_sed_extended_kernel_sizes_its_block_to_lds(lane_map):
    """16 chains per source and 27 bands: 17 band tables per 256-lane block plus a sqrt matrix per lane
    do not fit in LDS - the extended sampler then runs with a smaller block instead of leaving the
    job to the launch-per-step engine (31 against 8 / 4 us per step), and still equals it."""
    from mcmc_amd import Engine, synth
    S, C, P = 16, 16, 6
    nu, d, sig, truth = synth.sed_catalogue(S, seed=4, nb=27)
    nb = np.full(S, 27, np.int32)
    nb[3] = 20
    rng = np.random.default_rng(9)
    A = rng.normal(size=(P, P))
    cov = A @ A.T + P * np.eye(P)
    dg = np.sqrt(np.diag(cov))
    L = np.linalg.cholesky(cov / dg[:, None] / dg[None, :])
    res = []
    for fused in (True, False):
        with Engine(seed=7) as eng:
            eng.mh_set_fused(fused)
            eng.mh_set_lane_map(lane_map)
            eng.sed_load("M2N", nb, nu, d, sig)
            eng.mh_init_problem("sed", C, truth[:, None, :].repeat(C, axis=1), np.full(P, 0.02), param_blocks=None,
                                randomize_start=False, track_cov=True)
            assert eng.mh_fused_active == fused
            eng.mh_set_sqrt_matrix(L)
            eng.mh_run(70, thin=1)
            res.append((eng.mh_trace(), eng.mh_moments(), eng.mh_covar(normalized=False)))
    for x, y in zip(res[0][0], res[1][0]):
        assert np.array_equal(x, y, equal_nan=True)
    for k in res[0][1]:
        assert np.array_equal(res[0][1][k], res[1][1][k], equal_nan=True), k
    for x, y in zip(res[0][2], res[1][2]):
        assert np.array_equal(x, y, equal_nan=True)


@pytest.mark.parametrize("model,mode", [("M2N", "uncorr_block"), ("M2N", "corr_all"), ("M2N", "corr_block"),
                                        ("M2N", "rotate_block"), ("M1N_LOGA", "corr_all"),
                                        ("THICK", "rotate_block")])
@pytest.mark.parametrize("lane_map", [1, 2])
def test_sed_fused_extended_kernel_equals_stepwise_engine(model, mode, lane_map):
    """Chains started by mi355_mh_init_problem on a SED catalogue: the fused kernel's extended
    instantiation (correlated / rotated proposals, co-moments; one launch per run) - under both
    lane mappings, lane = chain and lanes = (chain, band) - against the launch-per-step engine:
    same draws, same arithmetic, identical chains, covariance, adapted widths and sqrt matrices;
    free-running and with injected draws."""
    from mcmc_amd import Engine, synth
    S, C = 5, 37
    nu, d, sig, truth = synth.sed_catalogue(S, seed=3)
    nb = np.array([8, 6, 8, 7, 8], np.int32)
    P = {"M2N": 6, "M1N_LOGA": 3, "THICK": 4}[model]
    start = {"M2N": synth.M2N_START, "M1N_LOGA": [0.0, 2.0, 10.0], "THICK": [1.0, 2.0, 10.0, 1.0]}[model]
    sg0 = {"M2N": synth.M2N_PROP_SIGMAS, "M1N_LOGA": [0.5, 0.5, 3.0], "THICK": [0.25, 0.5, 3.0, 0.3]}[model]
    rng = np.random.default_rng(5)
    A = rng.normal(size=(P, P))
    cov = A @ A.T
    dg = np.sqrt(np.diag(cov))
    L = np.linalg.cholesky(cov / dg[:, None] / dg[None, :])
    block = mode != "corr_all"
    zn, zu = rng.standard_normal((25, S * C, P)), rng.uniform(size=(25, S * C))
    res = []
    for fused in (True, False):
        out = {}
        with Engine(seed=99) as eng:
            eng.mh_set_fused(fused)
            eng.mh_set_lane_map(lane_map)
            eng.sed_load(model, nb, nu, d, sig)
            eng.mh_init_problem("sed", C, start, sg0, param_blocks=np.arange(P) if block else None,
                                randomize_start=True, track_cov=True)
            assert eng.mh_fused_active == fused
            if mode != "uncorr_block":
                eng.mh_set_sqrt_matrix(L, rotate=(mode == "rotate_block"))
            eng.mh_run(60, thin=1)
            out["t1"] = eng.mh_trace()
            eng.mh_run(45)
            eng.mh_run(25, thin=5, inj_normals=zn, inj_uniforms=zu)
            out["t2"] = eng.mh_trace()
            out["mom"] = eng.mh_moments()
            out["cov"] = eng.mh_covar(normalized=False)
            eng.mh_adapt_corr(no_correlations=False, rotate=(mode == "rotate_block"),
                              do_block=block)
            out["sig"] = eng.mh_sigmas()
            out["L"] = eng.mh_sqrt_matrix()
            eng.mh_run(50)
            out["mom2"] = eng.mh_moments()
            out["rhat"] = eng.rhat()
        res.append(out)
    a, b = res
    for k in a:
        x, y = a[k], b[k]
        items = list(x.items()) if isinstance(x, dict) else list(enumerate(x)) if isinstance(x, tuple) else [(0, x)]
        for kk, u in items:
            v = y[kk] if isinstance(y, (dict, tuple)) else y
            assert np.array_equal(u, v, equal_nan=True), (k, kk)


def test_sed_comoment_update_forms_give_the_same_sums():
    """The extended SED sampler's co-moments in memory (block proposals: the general instantiation): adds that
    return nothing for a job of few chains, loads and stores beyond 2^23 entries per step (403 200 chains x 21).
    Chain n = s C + c has its own Philox stream: the first sources of the large job equal the same sources run
    alone, bit for bit, through the fused kernel and through the step-wise engine."""
    from mcmc_amd import Engine, synth
    S, C, P = 6300, 64, 6
    nu, d, sig, truth = synth.sed_catalogue(S, seed=17)
    nb = np.full(S, 8, np.int32)
    out = {}
    for fused in (True, False):
        for s_n in (S, 3):
            with Engine(seed=5) as eng:
                eng.mh_set_fused(fused)
                eng.sed_load("M2N", nb[:s_n], nu[:s_n], d[:s_n], sig[:s_n])
                eng.mh_init_problem("sed", C, truth[:s_n, None, :].repeat(C, axis=1), np.full(P, 0.02),
                                    param_blocks=np.arange(P), randomize_start=False, track_cov=True)
                assert eng.mh_fused_active == fused
                eng.mh_run(30)
                eng.mh_run(17)
                out[(fused, s_n)] = eng.mh_covar(normalized=False)[:3].copy()
    ref = out[(False, 3)]
    for key, got in out.items():
        assert np.array_equal(got, ref, equal_nan=True), key
    assert np.isfinite(ref).all() and np.abs(ref).max() > 0.0
