#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE ITSELF (defjaf/MCMC,
mounted read-only at /root/reference) on seeded inputs.

Run in the build container only:   python oracle/make_golden.py
The reference never travels to the GPU box; the .npz fixtures (inputs and the
reference's outputs, nothing else) do.

Import recipe: register an empty synthetic package
``MCMC`` whose __path__ points at the reference, so MCMC/__init__.py (which
eagerly imports modules with dead dependencies) is skipped, then import the
sub-modules one by one.  The attribute shims below touch no arithmetic.
"""
import hashlib
import importlib
import math
import os
import sys
import types
import zlib

import numpy as np

REF = os.environ.get("MCMC_REFERENCE", "/root/reference")
OUT = os.environ.get("MCMC_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                        "..", "tests", "golden")


def load_reference():
    if not hasattr(np, "bool8"):
        np.bool8 = np.bool_
    if not hasattr(np, "math"):
        np.math = math
    if not hasattr(np, "float"):
        np.float = float
    pb = types.ModuleType("progressbar")

    class ProgressBar:
        def start(self):
            return self

        def update(self, *_):
            pass

        def finish(self):
            pass
    pb.ProgressBar = ProgressBar
    sys.modules["progressbar"] = pb
    import operator
    if not hasattr(operator, "isSequenceType"):
        operator.isSequenceType = lambda x: hasattr(x, "__getitem__")
    import matplotlib
    matplotlib.use("Agg")
    pkg = types.ModuleType("MCMC")
    pkg.__path__ = [os.path.join(REF, "MCMC")]
    sys.modules["MCMC"] = pkg
    mods = {}
    for name in ["GaussianData", "Likelihood", "Proposal", "MCMC", "convergence",
                 "submmSED.model", "submmSED.likelihood", "submmSED.data",
                 "ClData.ClData_CosmoMC", "binnedCl.binnedClLikelihood",
                 "BeamFit.BeamModel", "BeamFit.OffsetNormalizedBeamModel",
                 "BeamFit.NormalizedBeamLikelihood", "BeamFit.BeamData"]:
        mods[name] = importlib.import_module("MCMC." + name)
    # binnedClModel.setBinning names `collections.Sequence` without importing collections
    # (binnedClModel.py:205): give the module that namespace, nothing else.
    import collections.abc
    import builtins
    builtins.collections = types.SimpleNamespace(Sequence=collections.abc.Sequence)
    mods["binnedCl.binnedClModel"] = importlib.import_module("MCMC.binnedCl.binnedClModel")
    del builtins.collections
    mods["binnedCl.binnedClModel"].collections = types.SimpleNamespace(
        Sequence=collections.abc.Sequence)
    # determinism: both constructors reseed the global RNG from OS entropy
    mods["MCMC"].seed = lambda *a, **k: None
    mods["Proposal"].seed = lambda *a, **k: None
    return mods


def ref_tree_sha():
    h = hashlib.sha256()
    for sub in ["MCMC/MCMC.py", "MCMC/Proposal.py", "MCMC/Likelihood.py", "MCMC/GaussianData.py",
                "MCMC/convergence.py", "MCMC/submmSED/model.py", "MCMC/submmSED/likelihood.py",
                "MCMC/submmSED/data.py", "txt/Arp220.txt"]:
        with open(os.path.join(REF, sub), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


class Injector:
    """Replaces Proposal.normal / MCMC.uniform by recorded streams (contract of
    oracle.refcpu.InjectedDraws)."""

    def __init__(self, normals, uniforms):
        self.normals, self.uniforms, self.k = normals, uniforms, 0

    def normal(self, loc=0, scale=1, size=None):
        if size is None:
            row = self.normals[self.k, 0]
        else:
            row = self.normals[self.k, :size].copy()
        self.k += 1
        return row

    def uniform(self, lo=0, hi=1):
        return self.uniforms[self.k - 1]


def golden_sed(R):
    model, like_mod, data_mod = R["submmSED.model"], R["submmSED.likelihood"], R["submmSED.data"]
    MC, Prop, conv = R["MCMC"], R["Proposal"], R["convergence"]
    sha = ref_tree_sha()
    rng = np.random.default_rng(1234)

    import scipy.linalg as _sla

    def solve_compat(a, b, overwrite_a=False, overwrite_b=False, sym_pos=False, **kw):
        # sym_pos=True is assume_a="pos" in SciPy >= 1.11 (the same LAPACK Cholesky solve).  The
        # overwrite hints are NOT passed on: with overwrite_b honoured LAPACK returns the solution
        # in FNid itself, which likelihood.py:72-74 reads again as the right-hand side, and the
        # class returns z.z/2 instead of its documented (FB)(BB)^-1(BF)/2 (recorded once below as
        # lnl_5_overwrite_quirk; the build implements the documented formula).
        return _sla.solve(a, b, assume_a="pos" if sym_pos else "gen", **kw)
    like_mod.linalg = types.SimpleNamespace(solve=solve_compat, det=_sla.det, LinAlgError=_sla.LinAlgError)

    classes = [model.submmModel1_normalized, model.submmModel1_normalized_logA,
               model.submmModel2_normalized, model.submmModel2_normalized_logA,
               model.submmModel1, model.submmModel2,
               model.submmModel1_opticallythick, model.submmModel1_opticallythick_logA]
    likes = [like_mod.SEDLikelihood_normalized] * 4 + [like_mod.SEDLikelihood1, like_mod.SEDLikelihood2] + \
            [like_mod.SEDLikelihood_normalized] * 2

    # ---- G1: greybody / blackbody grids ---------------------------------
    beta = np.array([0.0, 0.5, 1.37, 2.0, 3.0])
    T = np.array([3.0, 3.0000001, 7.5, 10.0, 25.0, 47.3, 100.0])
    nu = np.array([239.8, 352.7, 599.6, 856.5, 1199.2, 1873.7, 2997.9, 4996.5, 11991.7, 60000.0])
    gN = np.empty((len(beta), len(T), len(nu)))
    gU = np.empty_like(gN)
    bN = np.empty((len(T), len(nu)))
    bU = np.empty_like(bN)
    with np.errstate(all="ignore"):
        for i, b in enumerate(beta):
            for j, t in enumerate(T):
                gN[i, j] = model.greybody(b, t, nu, nu_norm=model.nu_b)
                gU[i, j] = model.greybody(b, t, nu)
        for j, t in enumerate(T):
            bN[j] = model.blackbody(t, nu, nu_norm=model.nu_b)
            bU[j] = model.blackbody(t, nu)
    np.savez(os.path.join(OUT, "sed_g1.npz"), beta=beta, T=T, nu=nu, grey_norm=gN, grey_unnorm=gU,
             bb_norm=bN, bb_unnorm=bU, ref_sha=sha)

    # ---- G2: lnLike / prior / package / chi2 for every model class --------
    # Arp220 through the reference's own reader (data.py:243-256)
    import pkg_resources
    _orig = pkg_resources.resource_filename
    pkg_resources.resource_filename = lambda p, r: os.path.join(REF, "MCMC", "submmSED", r)
    # header-only catalogue: the per-row branch of the reader trips over numpy>=2.2
    # (`if l25:` on an empty array, data.py:235); only its Arp220 branch is needed.
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".csv", delete=False) as tf:
        tf.write("Name,z,F250_Jy,E250_Jy\n")
    alld = data_mod.readfluxes_DLC_2014(tf.name)
    os.unlink(tf.name)
    pkg_resources.resource_filename = _orig
    arp = alld[-1]
    assert arp.name == "Arp220"

    # synthetic sources (ragged nb) built here and fed to the reference as submmData
    S = 12
    nbs = np.array([8, 8, 8, 8, 5, 6, 7, 8, 12, 27, 3, 8])
    nbmax = int(nbs.max())
    s_nu = np.zeros((S, nbmax)); s_d = np.zeros((S, nbmax)); s_sig = np.ones((S, nbmax))
    srcs = []
    for s in range(S):
        nb = nbs[s]
        lam = np.sort(rng.uniform(40.0, 1300.0, nb))
        z = rng.uniform(0, 0.3)
        f = 299792.458 / lam * (1 + z)
        with np.errstate(all="ignore"):
            m = model.submmModel2_normalized(rng.uniform(0.5, 2), rng.uniform(1, 2.5), rng.uniform(15, 30),
                                             rng.uniform(0.05, 0.5), rng.uniform(1, 2.5),
                                             rng.uniform(35, 60)).at_nu(f)
        sg = 0.07 * m
        d = m + sg * rng.standard_normal(nb)
        if s == 3:                         # an upper limit: d=0, sig=flux (data.py:233-237)
            sg[0] = d[0]; d[0] = 0.0
        s_nu[s, :nb] = f; s_d[s, :nb] = d; s_sig[s, :nb] = sg
        srcs.append(data_mod.submmData(f, d, sg, "s%d" % s, z))
    datasets = srcs + [arp]
    an = arp.d.size
    a_nu = np.zeros((1, nbmax)); a_d = np.zeros((1, nbmax)); a_sig = np.ones((1, nbmax))
    a_nu[0, :an] = arp.freq; a_d[0, :an] = arp.d; a_sig[0, :an] = arp.sig
    g2 = dict(nb=np.concatenate([nbs, [an]]), nu=np.concatenate([s_nu, a_nu]),
              d=np.concatenate([s_d, a_d]), sig=np.concatenate([s_sig, a_sig]), ref_sha=sha)

    def rand_theta(mid, n):
        def comp(n, logA):
            A = rng.uniform(-1.5, 1.0, n) if logA else 10 ** rng.uniform(-1.5, 1.0, n)
            return A, rng.uniform(-0.2, 3.2, n), rng.uniform(2.0, 105.0, n)
        if mid in (0, 1):
            th = np.stack(comp(n, mid == 1), 1)
        elif mid in (2, 3):
            th = np.concatenate([np.stack(comp(n, mid == 3), 1), np.stack(comp(n, mid == 3), 1)], 1)
        elif mid == 4:
            th = np.stack(comp(n, False)[1:], 1)
        elif mid == 5:
            th = np.concatenate([np.stack(comp(n, False)[1:], 1), np.stack(comp(n, False)[1:], 1)], 1)
        else:
            th = np.concatenate([np.stack(comp(n, mid == 7), 1), rng.uniform(-0.1, 4.0, (n, 1))], 1)
        if mid == 0:
            th[::7, 0] *= -1               # negative amplitudes -> prior 0
        return th

    NT = 48
    for mid, (cls, lk) in enumerate(zip(classes, likes)):
        th = rand_theta(mid, NT)
        # a few hand-picked anchors
        anchors = {0: (1, 2, 10), 1: (0.5, 1.5, 40), 2: (1, 2, 10, 0.1, 2, 20), 4: (2, 10),
                   3: (0.0, 2, 10, -1.0, 2, 20), 5: (2., 20., 2., 30.), 6: (1, 2, 10, 1.0),
                   7: (0.0, 2, 10, 1.0)}
        th[0] = anchors[mid]
        nD = len(datasets)
        lnl = np.empty((nD, NT)); chi2 = np.empty((nD, NT)); pri = np.empty(NT)
        pk = np.empty_like(th)
        der = np.full((nD, NT, 2), np.nan)
        for t in range(NT):
            pri[t] = cls.prior(*th[t])
            pk[t] = cls.unpackage(cls.package(th[t]))
        for di, dset in enumerate(datasets):
            L = lk(data=dset, model=cls)
            for t in range(NT):
                with np.errstate(all="ignore"):
                    if mid == 5:
                        # SEDLikelihood2.lnLike1 itself (likelihood.py:35-74), with scipy.linalg.solve
                        # seen through solve_compat below: the class passes sym_pos=True, which
                        # SciPy >= 1.11 spells assume_a="pos" (same LAPACK Cholesky solve)
                        try:
                            lnl[di, t] = L.lnLike(th[t])
                            der[di, t] = np.asarray(L.MLamplitude, float).ravel()
                        except like_mod.ZeroPosterior:
                            lnl[di, t] = np.nan
                        chi2[di, t] = np.nan
                        if di == 0 and t == 0:               # the overwrite quirk, for the record
                            like_mod.linalg.solve = lambda a, b, sym_pos=False, **kw: _sla.solve(
                                a, b, assume_a="pos", **kw)
                            g2["lnl_5_overwrite_quirk"] = np.array([L.lnLike(th[t]), lnl[di, t]])
                            like_mod.linalg.solve = solve_compat
                    else:
                        lnl[di, t] = L.lnLike(th[t])
                        if mid == 4:
                            der[di, t, 0] = L.MLamplitude
                            chi2[di, t] = np.nan
                        else:
                            chi2[di, t] = L.chi2(th[t])
        g2["theta_%d" % mid] = th
        g2["lnl_%d" % mid] = lnl
        g2["chi2_%d" % mid] = chi2
        g2["prior_%d" % mid] = pri
        g2["package_%d" % mid] = pk
        g2["derived_%d" % mid] = der
    np.savez(os.path.join(OUT, "sed_g2.npz"), **g2)
    print("anchor M1N Arp220:", repr(g2["lnl_0"][-1, 0]), " M2N:", repr(g2["lnl_2"][-1, 0]),
          " M1:", repr(g2["lnl_4"][-1, 0]))

    # ---- G3/G4: draw-injected chains ---------------------------------------
    def run_chain(mid, dset, start, sigmas, nsteps, do_block, seed, corr_stage=False):
        cls, lk = classes[mid], likes[mid]
        P = cls.nparam
        r = np.random.default_rng(seed)
        nrows = nsteps + 4000
        inj = Injector(r.standard_normal((nrows, P)), r.uniform(size=nrows))
        Prop.normal = inj.normal
        MC.uniform = inj.uniform
        L = lk(data=dset, model=cls)
        with np.errstate(all="ignore"):
            m = MC.MCMC(L, startProposal=np.asarray(sigmas, float), startParams=np.asarray(start, float),
                        doBlock=do_block)
            k_start = inj.k
            m.MC_append(nsteps)
        out = dict(samples=m.samples.copy(), lnPr=m.lnPr.copy(), accepted=m.accepted.astype(bool),
                   naccept=m.naccept, accept=(m.accept.copy() if do_block else np.zeros(0, int)),
                   k_start=k_start, normals=inj.normals[:inj.k].copy(), uniforms=inj.uniforms[:inj.k].copy(),
                   sigmas=np.asarray(sigmas, float), start=np.asarray(start, float))
        with np.errstate(all="ignore"):
            out["mean"] = np.asarray(m.mean(1, 1), float)
            out["stdev"] = np.asarray(m.stdev(1, 1), float)
            out["covar"] = m.covar(1, 1)
        return m, inj, out

    g3 = dict(ref_sha=sha)
    cases = [
        # name, model id, dataset index, start, sigmas, nsteps, doBlock
        ("m1n_block", 0, 0, (1, 2, 10), (0.25, 0.5, 3.0), 1500, True),
        ("m1n_arp220", 0, len(datasets) - 1, (1, 2, 10), (2, 4, 4), 2000, True),
        ("m1n_loga_noblock", 1, 1, (0, 2, 20), (0.05, 0.1, 1.0), 1500, False),
        ("m2n_block", 2, 2, (1, 2, 10, 0.1, 2, 20), (0.1, 0.2, 2.0, 0.05, 0.2, 3.0), 2000, True),
        ("m2n_fixedbeta", 2, 0, (1, 2, 10, 0.1, 2, 20), (0.1, 0, 2.0, 0.05, 0, 3.0), 1500, True),
        ("m2n_loga_noblock", 3, 4, (0, 2, 20, -1, 2, 40), (0.03, 0.05, 0.8, 0.05, 0.05, 1.5), 1500, False),
        ("m1_marg", 4, 5, (2, 10), (0.3, 2.0), 1000, True),
        ("thick", 6, 6, (1, 2, 10, 1.0), (0.2, 0.3, 2.0, 0.3), 1500, True),
        ("thick_loga", 7, 7, (0, 2, 10, 1.0), (0.1, 0.3, 2.0, 0.3), 1000, True),
    ]
    chains_for_rhat = []
    for name, mid, di, start, sig, n, blk in cases:
        m, inj, out = run_chain(mid, datasets[di], start, sig, n, blk, seed=zlib.crc32(name.encode()) % 100000)
        for k, v in out.items():
            g3["%s__%s" % (name, k)] = v
        g3["%s__model" % name] = mid
        g3["%s__dataset" % name] = di
        g3["%s__do_block" % name] = blk
        swaps = 0
        if mid in (2, 3):
            swaps = int(np.sum(np.abs(np.diff(m.samples[:, 2])) > 0))
        print("%-18s naccept=%d/%d accept=%s" % (name, m.naccept, len(m.samples), out["accept"]))
        # newMCMC (MCMC.py:343-413) from this chain: start, sigmas and correlation sqrt
        with np.errstate(all="ignore"):
            k0 = inj.k
            new = m.newMCMC(burn=1, stride=1, noCorrelations=False, nMC=None)
            g3["%s__new_sigmas" % name] = np.asarray(new.prop.sigmas, float)
            g3["%s__new_start" % name] = np.asarray(new.samples[0], float)
            g3["%s__new_lnPr0" % name] = new.lnPr[0]
            g3["%s__new_kstart" % name] = k0
            g3["%s__new_kend" % name] = inj.k
            g3["%s__new_sqrt" % name] = (np.asarray(new.prop.sqrtMatrix, float)
                                          if new.prop.sqrtMatrix is not None else np.zeros((0, 0)))
            # a correlated stage of 300 steps continuing the same draw streams
            new.MC_append(300)
            g3["%s__corr_samples" % name] = new.samples.copy()
            g3["%s__corr_lnPr" % name] = new.lnPr.copy()
            g3["%s__corr_accepted" % name] = new.accepted.astype(bool)
            g3["%s__all_normals" % name] = inj.normals[:inj.k].copy()
            g3["%s__all_uniforms" % name] = inj.uniforms[:inj.k].copy()
        if name == "m2n_block":
            # four more chains on the same source for Gelman-Rubin (convergence.py:58-87)
            chains_for_rhat.append(m)
            for c in range(4):
                mm, _, oo = run_chain(mid, datasets[di], start, sig, 600, blk, seed=900 + c)
                chains_for_rhat.append(mm)
                g3["rhat_chain%d__samples" % c] = oo["samples"]
            with np.errstate(all="ignore"):
                g3["rhat__value"] = conv.gelmanrubin_MCMC(chains_for_rhat, burn=1, stride=1)
                g3["rhat__means"] = np.array([np.asarray(c.mean(1, 1), float) for c in chains_for_rhat])
                g3["rhat__vars"] = np.array([np.power(np.asarray(c.stdev(1, 1), float), 2)
                                              for c in chains_for_rhat])
    np.savez_compressed(os.path.join(OUT, "sed_g3.npz"), **g3)

    # ---- G4: the staged sampler() on Arp220 (config C1 plumbing) ------------
    r = np.random.default_rng(77)
    nrows = 12000
    inj = Injector(r.standard_normal((nrows, 3)), r.uniform(size=nrows))
    Prop.normal = inj.normal
    MC.uniform = inj.uniform
    np.random.seed(5)   # newMCMC(random=True) would use np.random.randn; unused here
    L = like_mod.SEDLikelihood_normalized(data=arp, model=model.submmModel1_normalized)
    import io, contextlib
    with np.errstate(all="ignore"), contextlib.redirect_stdout(io.StringIO()):
        stages, ana = MC.sampler(L, (1000, 3000), np.array((2., 4., 4.)), np.array((1., 2., 10.)),
                                 noCorrelations=True, doBlock=True)
    g4 = dict(ref_sha=sha, normals=inj.normals[:inj.k].copy(), uniforms=inj.uniforms[:inj.k].copy(),
              ana_means=np.asarray(ana[0]), ana_stdevs=np.asarray(ana[1]), ana_covar=np.asarray(ana[2]),
              nstage=len(stages))
    for i, st in enumerate(stages):
        g4["stage%d_samples" % i] = st.samples[::10].copy()      # thinned to keep the fixture small
        g4["stage%d_last" % i] = st.samples[-1].copy()
        g4["stage%d_lnPr_last" % i] = st.lnPr[-1]
        g4["stage%d_naccept" % i] = st.naccept
        g4["stage%d_accept" % i] = st.accept.copy()
        g4["stage%d_sigmas" % i] = np.asarray(st.prop.sigmas, float)
        g4["stage%d_n" % i] = len(st.samples)
    np.savez_compressed(os.path.join(OUT, "sed_g4.npz"), **g4)
    print("sampler() stages:", [len(s.samples) for s in stages], "means", ana[0])
    print("wrote fixtures to", os.path.abspath(OUT), "ref sha", sha[:16])


# ---------------------------------------------------------------------------------------------
# G5: binnedCl / ClData band-power likelihood
# ---------------------------------------------------------------------------------------------
def _cl_inputs(rng, shape3, nb, lmax, nbins, pol, with_x, mixed_x, corr, calib, beam, diag=False):
    """Synthetic inputs for one dataset + binning (plain arrays; nothing of the reference)."""
    ell = np.arange(lmax + 1, dtype=np.float64)
    nCl = 3 if pol else 1
    bins = []
    for iCl, nbin in enumerate(nbins[:nCl]):
        edges = np.linspace(30 + 7 * iCl, lmax - 20 - 5 * iCl, nbin + 1).astype(int)
        bins.append(np.stack([edges[:-1], edges[1:] - 1], axis=1))      # inclusive (lmin, lmax)
    # the caller truncates the shape to the binning's own lmax (setBinning does
    # not truncate an ndarray under Py3); multipoles below the first bin keep the bare shape
    mlmax = int(max(b.max() for b in bins))
    shapefun = np.array(shape3[:nCl, :mlmax + 1], dtype=np.float64)
    window = np.zeros((nb, 3, lmax + 1))
    win_min = np.empty(nb, int)
    win_max = np.empty(nb, int)
    inc_pol = np.zeros(nb, bool)
    kind = np.zeros(nb, int)
    wedges = np.linspace(10, lmax + 40, nb + 1).astype(int)             # last windows overrun lmax
    for i in range(nb):
        lo, hi = wedges[i], wedges[i + 1] + 6
        l = np.arange(lo, hi + 1, dtype=np.float64)
        w = (l * (l + 1) / (2 * math.pi)) / (hi - lo + 1) * (1 + 0.3 * np.sin(l / 13.0 + i))
        k = (i % 3) if (pol and i >= nb // 2) else 0
        kind[i] = k
        wfull = np.zeros(lmax + 61)
        wfull[lo:hi + 1] = w
        if k == 0:
            window[i, 0] = wfull[:lmax + 1]
        else:
            inc_pol[i] = True
            window[i, k] = wfull[:lmax + 1]
            window[i, 0] = 0.02 * wfull[:lmax + 1]                      # small TT leakage term
        win_min[i], win_max[i] = lo, hi                                 # win_max may exceed lmax
    return dict(shapefun=shapefun, bins=bins, window=window, win_min=win_min, win_max=win_max,
                inc_pol=inc_pol, kind=kind, nb=nb, lmax=lmax, pol=pol, with_x=with_x,
                mixed_x=mixed_x, corr=corr, calib=calib, beam=beam, diag=diag)


def golden_cl(R):
    CD = R["ClData.ClData_CosmoMC"]
    BM = R["binnedCl.binnedClModel"]
    BL = R["binnedCl.binnedClLikelihood"]
    MC, Prop = R["MCMC"], R["Proposal"]
    import contextlib
    import io
    sha = ref_tree_sha()
    rng = np.random.default_rng(4242)
    tmp = np.fromfile(os.path.join(REF, "dat", "CarloClModel.dat"), sep=" ").reshape(-1, 6)
    ells = tmp[:, 0].astype(int)
    L = ells.max()
    shape3 = np.zeros((3, L + 1))
    shape3[0, ells] = tmp[:, 1]                       # TT   test_binnedCl.py:48-58
    shape3[1, ells] = tmp[:, 3]                       # TE
    shape3[2, ells] = tmp[:, 2]                       # EE

    def build(inp, truth):
        """reference objects for one (binning, dataset) pair"""
        with contextlib.redirect_stdout(io.StringIO()):
            BM.binnedClModel.setBinning([b for b in inp["bins"]], shapefun=inp["shapefun"],
                                        doBlock=True)
        nb = inp["nb"]
        d = CD.ClData_CosmoMC()
        d.name = "synthetic"
        d.all_l_exact = False
        d.num_points = nb
        d.window, d.win_min, d.win_max = inp["window"], inp["win_min"], inp["win_max"]
        d.has_pol = bool(inp["pol"])
        d.inc_pol = inp["inc_pol"]
        d.has_xfactors = False
        d.has_corr_errors = not inp["diag"]
        d.calib_uncertainty = inp["calib"]
        d.beam_uncertain = inp["beam"] > 0
        d.beam_err = inp["beam"] * (1 + np.arange(nb) / nb) if inp["beam"] > 0 else None
        with np.errstate(all="ignore"):
            Cl = BM.binnedClModel(truth)().copy()
            bp = np.array([d.getWinBandpower(i, Cl) for i in range(nb)])
        sig = 0.06 * np.abs(bp) + 0.01 * np.abs(bp).max()
        A = rng.normal(size=(nb, nb))
        Cn = A @ A.T
        dg = np.sqrt(np.diag(Cn))
        Cn = (1 - inp["corr"]) * np.eye(nb) + inp["corr"] * Cn / dg[:, None] / dg[None, :]
        cov = Cn * sig[:, None] * sig[None, :]
        cobs = bp + np.linalg.cholesky(cov) @ rng.normal(size=nb)
        if inp["with_x"]:
            has_x = (inp["kind"] == 0)                # offset lognormal only for TT bands
            if inp["mixed_x"]:
                has_x[::4] = False
            x = np.where(has_x, 0.3 * np.abs(bp) + 1.0, 0.0)
            f = np.where(has_x, cobs + x, 1.0)
            d.obs = np.where(has_x, np.log(np.abs(cobs + x)), cobs)
            ninv = np.linalg.inv(cov) * f[:, None] * f[None, :]     # ClData_bcp.py:192-210
            d.xfactors, d.has_xfactor, d.has_xfactors = x, has_x, True
            var = np.diag(cov) / f ** 2
        else:
            d.obs = cobs
            ninv = np.linalg.inv(cov)
            d.xfactors = d.has_xfactor = None
            var = np.diag(cov).copy()
        if inp["diag"]:
            d.var, d.N_inv = var, None
        else:
            d.N_inv, d.var = ninv, None
        return d

    g = dict(ref_sha=sha)
    cases = [
        # name              nb  lmax nbins        pol    x      mixed  corr calib beam  diag
        ("xfull",           12, 600, (8,),        False, True,  False, 0.3, 0.0,  0.0, False),
        ("xcalib",          12, 600, (8,),        False, True,  False, 0.3, 0.05, 0.0, False),
        ("xcalibbeam",      16, 700, (10,),       False, True,  True,  0.3, 0.05, 0.02, False),
        ("xbeam",           12, 600, (8,),        False, True,  False, 0.2, 0.0,  0.03, False),
        ("gauss_analytic",  12, 600, (8,),        False, False, False, 0.3, 0.05, 0.02, False),
        ("gauss_calib",     12, 600, (8,),        False, False, False, 0.3, 0.04, 0.0, False),
        ("gauss_beam",      12, 600, (8,),        False, False, False, 0.3, 0.0,  0.02, False),
        ("gauss_diag",      12, 600, (8,),        False, False, False, 0.0, 0.05, 0.02, True),
        ("x_diag_calib",    12, 600, (8,),        False, True,  True,  0.0, 0.05, 0.0, True),
        ("pol",             18, 700, (8, 4, 4),   True,  True,  False, 0.3, 0.05, 0.0, False),
        ("pol_gauss",       18, 700, (8, 4, 4),   True,  False, False, 0.2, 0.03, 0.01, False),
        ("wide",            40, 900, (24,),       False, True,  False, 0.3, 0.0,  0.0, False),
    ]
    NQ = 20
    for (name, nb, lmax, nbins, pol, wx, mx, corr, calib, beam, diag) in cases:
        inp = _cl_inputs(rng, shape3, nb, lmax, nbins, pol, wx, mx, corr, calib, beam, diag)
        npar = int(sum(nbins[:3 if pol else 1]))
        truth = rng.uniform(0.8, 1.2, npar)
        d = build(inp, truth)
        like = BL.binnedClLikelihood(data=[d], model=BM.binnedClModel)
        q = truth[None, :] * rng.uniform(0.7, 1.3, (NQ, npar))
        q[0] = truth
        q[3, 0] = -0.2                          # negative TT bin: prior 0, likelihood still defined
        q[4] = -0.8 * truth                     # BP + x <= 0 on offset-lognormal bands -> -inf / NaN
        if pol:
            q[5, nbins[0]:nbins[0] + nbins[1]] *= -1.0     # negative TE is allowed (:129)
        lnl = np.empty(NQ); pri = np.empty(NQ); neg = np.empty(NQ)
        bps = np.empty((NQ, nb)); cls_ = []
        with np.errstate(all="ignore"):
            for t in range(NQ):
                lnl[t] = like.lnLike(q[t])
                pri[t] = BM.binnedClModel.prior(*q[t])
                Cl = BM.binnedClModel(q[t])().copy()
                neg[t] = d.calcLnLike(Cl)
                bps[t] = [d.getWinBandpower(i, Cl) for i in range(nb)]
                if t < 2:
                    cls_.append(Cl)
            delta = d.getdelta(bps[1].copy())
            quad = d.getquadform(delta)
            lin = d.getlinform(bps[1].copy())
            # calcChisq (:262-265) hands getdelta a list, which fails for x-factor data;
            # getchisq on the array is what calcLnLike's marginalisation loop calls (:338)
            chisq = d.getchisq(bps[1].copy())
        pre = "%s__" % name
        for k in ("shapefun", "window", "win_min", "win_max", "inc_pol"):
            g[pre + k] = inp[k]
        for i, b in enumerate(inp["bins"]):
            g[pre + "bins%d" % i] = b
        g[pre + "nCl"] = len(inp["bins"])
        g[pre + "obs"] = d.obs
        g[pre + "N_inv"] = d.N_inv if d.N_inv is not None else np.zeros((0, 0))
        g[pre + "var"] = d.var if d.var is not None else np.zeros(0)
        g[pre + "xfactors"] = d.xfactors if d.xfactors is not None else np.zeros(0)
        g[pre + "has_xfactor"] = d.has_xfactor if d.has_xfactor is not None else np.zeros(0, bool)
        g[pre + "calib"] = calib
        g[pre + "beam_err"] = d.beam_err if d.beam_err is not None else np.zeros(0)
        g[pre + "has_pol"] = pol
        g[pre + "q"] = q
        g[pre + "lnl"] = lnl
        g[pre + "prior"] = pri
        g[pre + "neglnl"] = neg
        g[pre + "bandpowers"] = bps
        g[pre + "Cl"] = np.array(cls_)
        g[pre + "delta1"] = delta
        g[pre + "quad1"] = quad
        g[pre + "lin1"] = lin
        g[pre + "chisq1"] = chisq
        g[pre + "BPnorm"] = np.concatenate([np.asarray(b) for b in BM.binnedClModel.BPnorm])
        print("cl %-15s lnL(truth)=%r  lnL[4]=%r" % (name, lnl[0], lnl[4]))
        if name == "xcalibbeam":
            # two datasets in one likelihood: the sum over datasets (binnedClLikelihood.py:32)
            inp2 = _cl_inputs(rng, shape3, 12, lmax, nbins, False, True, False, 0.25, 0.03, 0.0)
            inp2["bins"], inp2["shapefun"] = inp["bins"], inp["shapefun"]
            d2 = build(inp2, truth)
            like2 = BL.binnedClLikelihood(data=[d, d2], model=BM.binnedClModel)
            with np.errstate(all="ignore"):
                g["two__lnl"] = np.array([like2.lnLike(q[t]) for t in range(NQ)])
            for k in ("window", "win_min", "win_max", "inc_pol"):
                g["two__" + k] = inp2[k]
            g["two__obs"], g["two__N_inv"] = d2.obs, d2.N_inv
            g["two__xfactors"], g["two__has_xfactor"] = d2.xfactors, d2.has_xfactor
            g["two__calib"] = 0.03
            # a draw-injected block stage on the full 11 x 11 calibration x beam grid (:317-372)
            r = np.random.default_rng(199)
            nrows = 400
            inj = Injector(r.standard_normal((nrows, npar)), r.uniform(size=nrows))
            Prop.normal = inj.normal
            MC.uniform = inj.uniform
            like1 = BL.binnedClLikelihood(data=[d], model=BM.binnedClModel)
            with np.errstate(all="ignore"), contextlib.redirect_stdout(io.StringIO()):
                mg = MC.MCMC(like1, startProposal=np.full(npar, 0.05), startParams=truth.copy(), doBlock=True)
                kg0 = inj.k
                mg.MC_append(150)
            g["grid121__c0_samples"] = mg.samples.copy()
            g["grid121__c0_lnPr"] = mg.lnPr.copy()
            g["grid121__c0_accepted"] = mg.accepted.astype(bool)
            g["grid121__c0_naccept"] = mg.naccept
            g["grid121__accept0"] = mg.accept.copy()
            g["grid121__k"] = np.array([kg0, inj.k])
            g["grid121__normals"] = inj.normals[:inj.k].copy()
            g["grid121__uniforms"] = inj.uniforms[:inj.k].copy()
            print("cl 11 x 11 chain: naccept", mg.naccept, "draw rows", inj.k)
        if name == "xcalib":
            # draw-injected chains: block stage, then newMCMC(rotateParams=True) and a
            # Cholesky-correlated all-parameter stage (test_binnedCl.py:177 uses rotateParams)
            r = np.random.default_rng(99)
            nrows = 4000
            inj = Injector(r.standard_normal((nrows, npar)), r.uniform(size=nrows))
            Prop.normal = inj.normal
            MC.uniform = inj.uniform
            with np.errstate(all="ignore"), contextlib.redirect_stdout(io.StringIO()):
                m0 = MC.MCMC(like, startProposal=np.full(npar, 0.05), startParams=truth.copy(),
                             doBlock=True)
                k_start = inj.k
                m0.MC_append(400)
                k1 = inj.k
                m1 = m0.newMCMC(burn=1, stride=1, rotateParams=True)
                k1s = inj.k
                m1.MC_append(300)
                k2 = inj.k
                m2 = m1.newMCMC(burn=1, stride=1, noCorrelations=False, doBlock=False)
                k2s = inj.k
                m2.MC_append(200)
            for tag, m in (("c0", m0), ("c1", m1), ("c2", m2)):
                g["chain__%s_samples" % tag] = m.samples.copy()
                g["chain__%s_lnPr" % tag] = m.lnPr.copy()
                g["chain__%s_accepted" % tag] = m.accepted.astype(bool)
                g["chain__%s_naccept" % tag] = m.naccept
                g["chain__%s_sigmas" % tag] = np.asarray(m.prop.sigmas, float)
                g["chain__%s_sqrt" % tag] = (np.asarray(m.prop.sqrtMatrix, float)
                                              if m.prop.sqrtMatrix is not None else np.zeros((0, 0)))
            g["chain__accept0"] = m0.accept.copy()
            g["chain__accept1"] = m1.accept.copy()
            g["chain__k"] = np.array([k_start, k1, k1s, k2, k2s, inj.k])
            g["chain__normals"] = inj.normals[:inj.k].copy()
            g["chain__uniforms"] = inj.uniforms[:inj.k].copy()
            g["chain__start"] = truth.copy()
            print("cl chain: naccept", m0.naccept, m1.naccept, m2.naccept, "draw rows", inj.k)
    np.savez_compressed(os.path.join(OUT, "cl_g5.npz"), **g)
    print("wrote cl_g5.npz")


# ---------------------------------------------------------------------------------------------
# G6: BeamFit 2-D Gaussian beam
# ---------------------------------------------------------------------------------------------
def golden_beam(R):
    OM = R["BeamFit.OffsetNormalizedBeamModel"]
    NL = R["BeamFit.NormalizedBeamLikelihood"]
    BD = R["BeamFit.BeamData"]
    MC, Prop = R["MCMC"], R["Proposal"]
    import contextlib
    import io
    sha = ref_tree_sha()
    rng = np.random.default_rng(606)
    mod = OM.OffsetNormalizedBeamModel
    half = 20.0
    # class-level prior state, as ProcessBeamData.setup_sampler leaves it (:149-151, :215-216)
    mod.centerMin, mod.centerMax = (-half, -half), (half, half)
    mod.sigMin, mod.sigMax = 3, 8
    mod.paramBlocks = [0, 1, 2, 3, 4, 5, 6]
    mod.nBlock = 7
    g = dict(ref_sha=sha, centerMin=np.array(mod.centerMin), centerMax=np.array(mod.centerMax),
             sigMin=float(mod.sigMin), sigMax=float(mod.sigMax))

    def make_map(n, per_pixel_sigma=False, scattered=False):
        if scattered:
            x = rng.uniform(-half, half, n * n)
            y = rng.uniform(-half, half, n * n)
        else:
            gg = np.linspace(-half, half, n)
            xx, yy = np.meshgrid(gg, gg)
            x, y = xx.ravel(), yy.ravel()
        th = np.array([rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(3, 8), rng.uniform(3, 8),
                       rng.uniform(0, math.pi), 10.0, 0.01, 1.0, 0.0])
        with np.errstate(all="ignore"):
            m = mod(*mod.package(th)).atxy(x, y)
        sig = 0.1 * (1 + 0.5 * rng.uniform(size=n * n)) if per_pixel_sigma else 0.1
        d = m + sig * rng.standard_normal(n * n)
        return x, y, d, sig, th

    NT = 16
    maps = [("m64a", 64, False, False), ("m64b", 64, True, False), ("m48s", 48, False, True),
            ("m256", 256, False, False)]
    for name, n, pps, scat in maps:
        x, y, d, sig, truth = make_map(n, pps, scat)
        data = BD.BeamData(x, y, d.copy(), sig)
        like = NL.NormalizedBeamLikelihood(data=data, model=mod)
        th = truth[None, :] * (1 + 0.05 * rng.standard_normal((NT, 9)))
        th[:, 7] = rng.uniform(0.7, 1.0, NT)          # tilted planes
        th[:, 8] = rng.uniform(-1.0, 7.5, NT)         # phi outside [0, 2pi): package reduces it
        th[:, 6] = 0.01 * rng.standard_normal(NT)
        th[0] = truth                                 # mu == 1: planexy returns 0 (:150)
        th[1, 4] = -0.4                               # negative angle -> mod pi
        th[2, 4] = 4.0
        th[3, 5] = -1.0                               # amplitude <= 0: prior 0
        th[4, 7] = -0.9                               # |mu|
        th[5, 0] = 25.0                               # centre outside the box
        th[6, 2] = 2.0                                # sigma below sigMin
        th[7, 3] = 9.5                                # sigma above sigMax
        th[8, 7] = 1.3                                # mu > 1: prior 0, sqrt(negative) -> NaN model
        th[9, 7] = 0.0                                # mu == 0: division by zero -> inf
        lnl = np.empty(NT); pri = np.empty(NT); pk = np.empty((NT, 9))
        mv = np.empty((NT, 5))
        pix = np.array([0, n + 3, (n * n) // 2, n * n - n - 2, n * n - 1])
        with np.errstate(all="ignore"):
            for t in range(NT):
                p = mod.package(th[t])
                pk[t] = mod.unpackage(p)
                pri[t] = mod.prior(*p)
                lnl[t] = like.lnLike(p)
                mv[t] = like.model_vals[0][pix]
            dcopy = BD.BeamData(x, y, d.copy(), sig)
            st = dcopy.stats(sigcut=0.0)
        pre = name + "__"
        g[pre + "x"], g[pre + "y"], g[pre + "d"] = x, y, d
        g[pre + "sig"] = np.atleast_1d(np.asarray(sig, float))
        g[pre + "truth"], g[pre + "theta"] = truth, th
        g[pre + "lnl"], g[pre + "prior"], g[pre + "package"] = lnl, pri, pk
        g[pre + "pix"], g[pre + "model_vals"] = pix, mv
        g[pre + "stats"] = np.array(st, float)
        print("beam %-5s lnL(truth)=%r" % (name, lnl[0]))
        if name == "m64a":
            r = np.random.default_rng(31)
            nrows = 1500
            inj = Injector(r.standard_normal((nrows, 9)), r.uniform(size=nrows))
            Prop.normal = inj.normal
            MC.uniform = inj.uniform
            sigmas = np.array([4.0, 4.0, 4.0, 4.0, 0.6, 2 * (d.max() - d.min()), 0.3, 0, 0]) * 0.05
            start = truth * np.array([1.05, 0.9, 1.1, 0.95, 1.0, 1.1, 1.0, 1.0, 1.0])
            with np.errstate(all="ignore"), contextlib.redirect_stdout(io.StringIO()):
                m0 = MC.MCMC(like, startProposal=mod.package(sigmas), startParams=mod.package(start),
                             doBlock=True)
                k_start = inj.k
                m0.MC_append(500)
                k1 = inj.k
                m1 = m0.newMCMC(burn=1, stride=1, noCorrelations=False)
                k1s = inj.k
                m1.MC_append(300)
            for tag, m in (("c0", m0), ("c1", m1)):
                g["chain__%s_samples" % tag] = m.samples.copy()
                g["chain__%s_lnPr" % tag] = m.lnPr.copy()
                g["chain__%s_accepted" % tag] = m.accepted.astype(bool)
                g["chain__%s_naccept" % tag] = m.naccept
                g["chain__%s_accept" % tag] = m.accept.copy()
                g["chain__%s_sigmas" % tag] = np.asarray(m.prop.sigmas, float)
                g["chain__%s_sqrt" % tag] = (np.asarray(m.prop.sqrtMatrix, float)
                                              if m.prop.sqrtMatrix is not None else np.zeros((0, 0)))
            g["chain__k"] = np.array([k_start, k1, k1s, inj.k])
            g["chain__normals"] = inj.normals[:inj.k].copy()
            g["chain__uniforms"] = inj.uniforms[:inj.k].copy()
            g["chain__start"], g["chain__sigmas"] = start, sigmas
            print("beam chain: naccept", m0.naccept, m1.naccept, "accept", m0.accept, m1.accept)
    np.savez_compressed(os.path.join(OUT, "beam_g6.npz"), **g)
    print("wrote beam_g6.npz")


# ---------------------------------------------------------------------------------------------
# G7: catalogue readers (submmSED/data.py:75-607) on excerpts of the data files the reference
# ships.  The excerpts (data, a few KB each) are written to tests/golden/catalogues/ and are the
# inputs of tests/test_catalogue_readers.py; the .npz holds what the reference's readers return
# for exactly those files.
# ---------------------------------------------------------------------------------------------
def _pack_datasets(prefix, dsets, out):
    def clean(n):                                          # str(bytes) artefact of the 'a12' fields
        n = str(n)
        if n.startswith("b'") and n.endswith("'"):
            n = n[2:-1]
        return n
    out[prefix + "/name"] = np.array([clean(d.name) for d in dsets])
    out[prefix + "/z"] = np.array([float(d.z) for d in dsets])
    out[prefix + "/nb"] = np.array([np.size(d.d) for d in dsets], dtype=np.int64)
    for key, get in (("freq", lambda d: d.freq), ("d", lambda d: d.d), ("sig", lambda d: d.sig),
                     ("freq_obs", lambda d: d.freq_obs)):
        parts = [np.ravel(np.asarray(get(d), dtype=np.float64)) for d in dsets]
        parts = [q if q.size == np.size(d.d) else np.full(np.size(d.d), np.nan)
                 for q, d in zip(parts, dsets)]
        out[prefix + "/" + key] = np.concatenate(parts) if parts else np.zeros(0)
    if dsets and getattr(dsets[0], "z_alt", None) is not None:
        out[prefix + "/z_alt"] = np.array([float(d.z_alt) for d in dsets])
        out[prefix + "/name_alt"] = np.array([str(d.name_alt) for d in dsets])


def golden_readers(R):
    import contextlib
    import io
    D = R["submmSED.data"]
    src = os.path.join(REF, "MCMC", "submmSED")
    cat = os.path.join(OUT, "catalogues")
    os.makedirs(cat, exist_ok=True)

    def excerpt(rel, dst, keep):
        with open(os.path.join(src, rel)) as f:
            lines = f.readlines()
        with open(os.path.join(cat, dst), "w") as f:
            f.writelines(keep(lines))
        return os.path.join(cat, dst)

    out = {}

    def run(prefix, fn):
        with contextlib.redirect_stdout(io.StringIO()):
            _pack_datasets(prefix, fn(), out)

    mrr = excerpt("dat/ERCSCiifsczbg.dat", "mrr_new.dat", lambda L: L[:60])
    run("mrr_default", lambda: D.readfluxes_MRR(mrr))
    run("mrr_dlc1", lambda: D.readfluxes_MRR(mrr, IRAS_ignore=[0], Planck_ignore=[3], DLC_ul=True, next0=True))
    run("mrr_dlc2", lambda: D.readfluxes_MRR(mrr, IRAS_ignore=[0], Planck_ignore=[2, 3], DLC_ul=True, next0=False))
    run("mrr_nocx", lambda: D.readfluxes_MRR(mrr, colorx=[1.0, 1.0, 1.0, 1.0], next0=False))
    mrro = excerpt("dat/ERCSCalliifscz4550850.dat", "mrr_old.dat", lambda L: L[:25])
    run("mrr_old", lambda: D.readfluxes_MRR(mrro, old_MRR=True, next0=False))

    dlc0 = excerpt("dat/submmSED.txt", "dlc_format0.txt", lambda L: L)
    run("dlc0", lambda: D.readfluxes_DLC(dlc0, format=0))

    # DLC_2014: under NumPy >= 2.2 the reference's `if l25:` raises on sources without a 25 um
    # point, so UL25=True is pinned on the rows that have one and the parsing on all rows with
    # UL25=False.  getArp220 needs pkg_resources on the real package; its branch is pinned by G2.
    herus = excerpt("dat/herus_phot.csv", "herus_phot.csv", lambda L: L)
    run("herus_noUL", lambda: D.readfluxes_DLC_2014(herus, UL25=False, getArp220=False))
    run("herus_noUL_keep157", lambda: D.readfluxes_DLC_2014(herus, UL25=False, getArp220=False, del157=False))

    def rows_with_25(L):
        hdr = L[0].rstrip("\n").split(",")
        i25, e25 = hdr.index("F25_Jy"), hdr.index("E25_Jy")
        keep = [L[0]]
        for ln in L[1:]:
            c = ln.rstrip("\n").split(",")
            if len(c) > e25 and c[i25] != "" and c[e25] != "" and float(c[e25]) > 0:
                keep.append(ln)
        return keep
    herus25 = excerpt("dat/herus_phot.csv", "herus_phot_with25.csv", rows_with_25)
    run("herus_UL", lambda: D.readfluxes_DLC_2014(herus25, UL25=True, getArp220=False))
    magdis = excerpt("dat/magdis_ulirgs.dat", "magdis_ulirgs.dat", lambda L: L)
    try:
        run("magdis_noUL", lambda: D.readfluxes_DLC_2014(magdis, UL25=False, getArp220=False))
    except Exception as e:                                 # whitespace branch reads bytes under Py3
        print("  reference readfluxes_DLC_2014 on the whitespace table:", type(e).__name__, e)

    def topcat_rows(L):
        first = next(i for i, ln in enumerate(L) if ln.startswith("+-"))
        return L[:first + 3 + 80] + [L[-1]]
    top = excerpt("dat/ercsc_iifscz.txt", "ercsc_topcat.txt", topcat_rows)
    run("topcat", lambda: D.readfluxes_ERCSC_TopCat(top))
    run("topcat_noUL", lambda: D.readfluxes_ERCSC_TopCat(top, upperlim=0))

    peel = excerpt("M31/pixelfit.dat", "peel_pixelfit.dat", lambda L: L[:16])
    run("peel", lambda: D.readfluxes_peel(peel))
    run("peel_delnu", lambda: D.readfluxes_peel(peel, delnu=[1763, 12000]))
    run("dlc3", lambda: D.readfluxes_DLC(peel, format=3, delnu=1763))
    m31 = excerpt("M31/M31Flux-v2.dat", "M31Flux-v2.dat", lambda L: L)
    run("m31", lambda: D.readfluxes_M31(m31))
    run("m31_delnu", lambda: D.readfluxes_M31(m31, delnu=[28.4, 857.0]))

    mdir = os.path.join(cat, "mortier")
    os.makedirs(mdir, exist_ok=True)
    names = sorted(os.listdir(os.path.join(src, "dat", "print_seds_mergers")))[:6]
    for n in names:
        excerpt(os.path.join("dat", "print_seds_mergers", n), os.path.join("mortier", n), lambda L: L)
    with contextlib.redirect_stdout(io.StringIO()):
        ds = sorted(D.readfluxes_mortier(mdir), key=lambda d: d.name)
    _pack_datasets("mortier", ds, out)

    # model.flux / totalflux (model.py:113-127 and the flux() methods), what driver.main stores
    # per object as MLmod.flux(nu1, nu2) with nu1, nu2 = 299792/1000, 299792/8 GHz (driver.py:83-84)
    M = R["submmSED.model"]
    nu2, nu1 = 299792. / 8.0, 299792. / 1000.0
    rng = np.random.default_rng(77)
    for cls in (M.submmModel2_normalized, M.submmModel2_normalized_logA, M.submmModel1_normalized,
                M.submmModel1_normalized_logA, M.submmModel1_opticallythick,
                M.submmModel1_opticallythick_logA):
        th = []
        for _ in range(4):
            comp = lambda: [rng.uniform(0.1, 2.0), rng.uniform(0.5, 2.5), rng.uniform(8.0, 60.0)]
            t = comp() + comp() if cls.nparam == 6 else comp() + ([rng.uniform(0.5, 3.0)] if cls.nparam == 4 else [])
            th.append(t)
        th = np.array(th)
        out["flux/%s/theta" % cls.__name__] = th
        out["flux/%s/band" % cls.__name__] = np.array([cls(*t).flux(nu1, nu2) for t in th])
        out["flux/%s/bolo" % cls.__name__] = np.array([cls(*t).flux() for t in th])

    out["ref_sha"] = np.array(ref_tree_sha())
    np.savez_compressed(os.path.join(OUT, "readers_g7.npz"), **out)
    print("G7 readers:", sorted({k.split("/")[0] for k in out if "/" in k}))



# ---------------------------------------------------------------------------------------------
# G8: ClData file formats (.dataset, .newdat, window files; ClData_CosmoMC.py:85-258,
# ClData_bcp.py:20-210) and the post-fit products of binnedClModel.py:329-503.  The reference
# ships no such data files (dat/data_list.txt points at an absent data/ tree), so the input files
# are synthetic, written here into tests/golden/clfiles/; the .npz holds what the reference's
# parsers make of them.  Py2 `string.split/lower/upper` are given back to the string module
# (attribute shims, no arithmetic).
# ---------------------------------------------------------------------------------------------
def _write_windows(wdir, name, first, bands, ncls, rng, bare_l=False):
    os.makedirs(wdir, exist_ok=True)
    for k, (lo, hi) in enumerate(bands):
        with open(os.path.join(wdir, "%s%d" % (name, first + k)), "w") as f:
            for l in range(max(lo - 3, 1), hi + 4):
                w = [np.exp(-0.5 * ((l - 0.5 * (lo + hi)) / (0.3 * (hi - lo) + 2)) ** 2) / (l if not bare_l else 1.0)]
                if ncls == 3:
                    w += [0.1 * w[0] * rng.uniform(-1, 1), 0.2 * w[0] * rng.uniform(0, 1)]
                f.write("%d %s\n" % (l, " ".join("%.10e" % x for x in w)))
            f.write("%d %s\n" % (lmax_plus(), " ".join("0.0" for _ in range(ncls))))


def lmax_plus():
    return 3600                                            # a line beyond lmax = 3500: ignored


def _spd(n, rng, scale):
    A = rng.standard_normal((n, n))
    M = A @ A.T / n + np.eye(n)
    d = np.sqrt(np.diag(M))
    return M / np.outer(d, d) * np.outer(scale, scale)


def golden_clfiles(R):
    import contextlib
    import io
    import string
    string.split = lambda s, sep=None, maxsplit=-1: s.split(sep, maxsplit)
    string.lower = lambda s: s.lower()
    string.upper = lambda s: s.upper()
    CM = R["ClData.ClData_CosmoMC"]
    BCP = importlib.import_module("MCMC.ClData.ClData_bcp")
    root = os.path.join(OUT, "clfiles")
    os.makedirs(root, exist_ok=True)
    rng = np.random.default_rng(808)
    out = {}
    attrs = ("obs", "var", "sig", "Cl", "err_min", "err_pls", "beam_err", "win_min", "win_max", "ell",
             "N_inv", "xfactors", "has_xfactor", "inc_pol")

    def pack(prefix, d, Cl):
        for a in attrs:
            if hasattr(d, a):
                out[prefix + "/" + a] = np.asarray(getattr(d, a))
        w = np.asarray(d.window)
        out[prefix + "/window_nz"] = np.argwhere(w != 0).astype(np.int32)
        out[prefix + "/window_val"] = w[w != 0]
        out[prefix + "/window_shape"] = np.array(w.shape)
        out[prefix + "/scalars"] = np.array([d.num_points, d.ncls, float(d.calib_uncertainty),
                                             float(bool(d.beam_uncertain)), float(bool(d.has_pol)),
                                             float(bool(d.has_xfactors)), float(bool(d.has_corr_errors))])
        out[prefix + "/name"] = np.array(d.name)
        out[prefix + "/Clpol_idx"] = np.array(sorted((k, v[0], v[1]) for k, v in d.Clpol_idx.items()), dtype=str)
        out[prefix + "/Cl_in"] = Cl
        with np.errstate(all="ignore"):
            out[prefix + "/lnlike"] = np.array(d.calcLnLike(Cl))
            out[prefix + "/bandpowers"] = np.array([d.getWinBandpower(i, Cl) for i in range(d.num_points)])

    ell = np.arange(3501.0)
    Dl = 5000.0 * np.exp(-((ell - 220.0) / 300.0) ** 2) + 1000.0
    ClTT = np.zeros(3501)
    ClTT[2:] = Dl[2:] * 2 * np.pi / (ell[2:] * (ell[2:] + 1))
    Cl3 = np.array([ClTT, 0.05 * ClTT, 0.02 * ClTT])

    cwd = os.getcwd()
    os.chdir(root)
    try:
        # A: .dataset, TT, correlated errors, x-factors, beam + calibration uncertainty
        bands = [(50 + 60 * k, 100 + 60 * k) for k in range(6)]
        _write_windows("winA", "expA", 1, bands, 1, rng)
        obs = np.array([Dl[(lo + hi) // 2] for lo, hi in bands]) * rng.uniform(0.9, 1.1, 6)
        err = 0.08 * obs
        x = 0.3 * obs
        Nz = np.linalg.inv(_spd(6, rng, err / (obs + x)))
        np.savetxt("expA_Ninv.dat", Nz.reshape(-1, 1), fmt="%.17e")
        np.savetxt("expA_x.dat", x.reshape(-1, 1), fmt="%.17e")
        with open("expA.dataset", "w") as f:
            f.write("# synthetic CosmoMC-format data set (tests)\nname = expA\nhas_pol = F\nnum_points = 6\n"
                    "calib_uncertainty = 0.05\nbeam_uncertainty = T\nwindow_dir = winA\n"
                    "windows_are_bandpowers = T\nwindows_are_normalized = F\n; a comment\n\n"
                    "N_inv = expA_Ninv.dat\nxfactors = expA_x.dat\n")
            for k in range(6):
                f.write("data%d = %.10e %.10e %.10e %.10e\n" % (k + 1, obs[k], 0.9 * err[k], 1.1 * err[k],
                                                               0.02 * obs[k] * (k + 1)))
        with contextlib.redirect_stdout(io.StringIO()):
            pack("A", CM.ClData_CosmoMC("expA.dataset"), ClTT)

        # B: .dataset, TT, diagonal errors, bare normalised windows, a band sub-range
        bandsB = [(300 + 100 * k, 380 + 100 * k) for k in range(7)]
        _write_windows("winB", "expB", 1, bandsB, 1, rng, bare_l=True)
        obsB = np.array([Dl[(lo + hi) // 2] for lo, hi in bandsB]) * rng.uniform(0.9, 1.1, 7)
        with open("expB.dataset", "w") as f:
            f.write("name = expB\nnum_points = 4\nfile_points = 7\nfirst_band = 3\nwindow_dir = winB\n"
                    "windows_are_bare = T\nwindows_are_normalized = T\ncalib_uncertainty = 0.0\n")
            for k in range(4):
                f.write("data%d = %.10e %.10e %.10e\n" % (k + 1, obsB[k + 2], 0.1 * obsB[k + 2], 0.12 * obsB[k + 2]))
        with contextlib.redirect_stdout(io.StringIO()):
            pack("B", CM.ClData_CosmoMC("expB.dataset"), ClTT)

        # C / D: .newdat, TT only with per-band x-factor flags; polarised with BAND_SELECTION
        def write_newdat(fn, name, npol, specs, cal, beam, ilike, cov, selection=None, fisher=False):
            with open(fn, "w") as f:
                if fisher:
                    f.write("FISHER_T_CMB\n")
                f.write(name + "\n" + " ".join(str(n) for n in npol) + "\n")
                if selection is not None:
                    f.write("BAND_SELECTION\n")
                    for a, b in selection:
                        f.write("%d %d\n" % (a, b))
                f.write("%d %.6f %.6f\n" % cal)
                f.write("%d %.6f %.6f\n" % beam)
                f.write("%d\n" % ilike)
                for typ, rows in specs:
                    f.write(typ + "\n")
                    for k, r in enumerate(rows):
                        f.write("%d " % (k + 1) + " ".join("%.10e" % v for v in r[:4]) +
                                " %d %d" % (r[4], r[5]) + (" %d" % r[6] if ilike > 1 else "") + "\n")
                    for k in range(len(rows)):
                        f.write(" ".join("%.3e" % (1.0 if i == k else 0.0) for i in range(len(rows))) + "\n")
                for row in cov:
                    f.write(" ".join("%.17e" % v for v in row) + "\n")

        bandsC = [(100 + 80 * k, 170 + 80 * k) for k in range(5)]
        _write_windows("data/windows", "expC", 1, bandsC, 1, rng)
        obsC = np.array([Dl[(lo + hi) // 2] for lo, hi in bandsC]) * rng.uniform(0.9, 1.1, 5)
        rowsC = [(obsC[k], 0.07 * obsC[k], 0.09 * obsC[k], 0.25 * obsC[k], bandsC[k][0], bandsC[k][1], k % 2)
                 for k in range(5)]
        write_newdat("expC.newdat", "expC", [5, 0, 0, 0, 0, 0], [("TT", rowsC)], (1, 1.02, 0.04),
                     (1, 10.0, 0.6), 2, _spd(5, rng, 0.08 * obsC))
        with contextlib.redirect_stdout(io.StringIO()):
            pack("C", BCP.ClData_bcp("expC.newdat"), ClTT)

        bTT = [(100 + 100 * k, 190 + 100 * k) for k in range(4)]
        bEE = [(150 + 150 * k, 290 + 150 * k) for k in range(3)]
        bTE = [(150 + 150 * k, 290 + 150 * k) for k in range(3)]
        _write_windows("data/windows", "expD", 1, bTT + bEE + bTE, 3, rng)
        oTT = np.array([Dl[(lo + hi) // 2] for lo, hi in bTT])
        oEE = 0.02 * np.array([Dl[(lo + hi) // 2] for lo, hi in bEE])
        oTE = 0.05 * np.array([Dl[(lo + hi) // 2] for lo, hi in bTE]) * np.array([1, -1, 1])
        mk = lambda o, b: [(o[k], 0.1 * abs(o[k]), 0.1 * abs(o[k]), 2.0 * abs(o[k]), b[k][0], b[k][1], 1)
                           for k in range(len(o))]
        allo = np.concatenate((oTT, oEE, oTE))
        write_newdat("expD.newdat", "expD", [4, 3, 0, 0, 3, 0],
                     [("TT", mk(oTT, bTT)), ("EE", mk(oEE, bEE)), ("TE", mk(oTE, bTE))], (1, 0.98, 0.03),
                     (0, 12.0, 0.5), 1, _spd(10, rng, 0.1 * np.abs(allo)),
                     selection=[(2, 4), (1, 2), (0, 0), (0, 0), (1, 3), (0, 0)], fisher=True)
        with contextlib.redirect_stdout(io.StringIO()):
            pack("D", BCP.ClData_bcp("expD.newdat"), Cl3)
    finally:
        os.chdir(cwd)

    # post-fit products, binnedClModel.py:329-503
    BM = R["binnedCl.binnedClModel"]
    F = _spd(7, rng, rng.uniform(0.5, 2.0, 7))
    bins = [[(2, 50), (51, 120), (121, 300)], [(2, 100), (101, 300)], [(2, 150), (151, 300)]]
    out["post/F"] = F
    out["post/Wbb"] = BM.FisherWindows(F)
    out["post/Wbb_covar"] = BM.FisherWindows(F, isCovar=True)
    out["post/WBl"] = BM.FisherWindows(F, bins=bins)
    samp = np.exp(rng.normal(5.0, 0.3, 4000)) - 40.0
    out["post/samples"] = samp
    with contextlib.redirect_stdout(io.StringIO()):
        out["post/oln_cum"] = np.asarray(BM.fitOffsetLognormal_cum(samp, do_plot=0))
        fl = BM.fitOffsetLognormal_like(samp)
    out["post/oln_like_x"] = np.asarray(fl[0])
    out["post/oln_like_f"] = np.asarray(fl[1])
    o = BM.oln(samp)
    par = np.array([5.0, 0.1, 45.0])
    out["post/oln_par"] = par
    out["post/oln_eval"] = np.array([o.chi2(*par), o.like(par), o.KSnorm(par)] + list(o.derivs(par)))
    out["post/oln_cumcurve"] = o.cum(par)
    out["ref_sha"] = np.array(ref_tree_sha())
    np.savez_compressed(os.path.join(OUT, "clfiles_g8.npz"), **out)
    print("G8 clfiles:", sorted({k.split("/")[0] for k in out if "/" in k}))



class _StubLike:
    """like / like.model of a finished chain, as getdist_ahj's consistency check uses them: lnLike
    and chi2 of a stored sample are looked up (off by `skew` to exercise the 1 % assertion)."""
    nDerived = 0

    class model:
        @staticmethod
        def prior(*p):
            return 1.0

    def __init__(self, samples, lnPr, skew=1.0):
        self.tab = {tuple(np.round(s, 12)): l for s, l in zip(samples, lnPr)}
        self.skew = skew

    def lnLike(self, p):
        return self.skew * self.tab[tuple(np.round(np.asarray(p), 12))]

    def chi2(self, p):
        return -2.0 * self.tab[tuple(np.round(np.asarray(p), 12))]


def golden_getdist(R):
    """G9: MCMC/getdist_ahj.py's numeric half (printvals, histgrid(noPlot), getallsamples,
    GaussPDF, stddev) on stub chains: return values and printed text."""
    import contextlib
    import io
    import json
    g = importlib.import_module("MCMC.getdist_ahj")
    rng = np.random.default_rng(909)
    cases = {}
    for name, (n, p, nder, skew) in {"plain": (60, 3, 0, 1.0), "derived": (45, 2, 2, 1.004),
                                      "skewed": (30, 2, 0, 1.02)}.items():
        samples = rng.normal(1.0, 2.0, (n, p))
        lnPr = -0.5 * ((samples - 1.0) ** 2).sum(axis=1) - 3.0
        derived = rng.normal(size=(n, nder)) if nder else None
        m = types.SimpleNamespace(samples=samples, lnPr=lnPr, nDerived=nder, derived=derived)
        full = np.hstack([samples, derived]) if nder else samples
        # the looked-up point is a row of samples+derived columns: key on the parameter columns
        like = _StubLike(full, lnPr, skew)
        m.like = like
        out = dict(samples=samples.tolist(), lnPr=lnPr.tolist(), nDerived=nder, skew=skew,
                   derived=derived.tolist() if nder else None)
        for fn, kw in (("printvals", {}), ("histgrid", dict(noPlot=True, burn=0.2, stride=2))):
            buf = io.StringIO()
            try:
                with contextlib.redirect_stdout(buf):
                    r = getattr(g, fn)(m, **kw)
                out[fn] = dict(ret=[float(r[0]), np.asarray(r[1]).tolist()], text=buf.getvalue())
            except AssertionError as e:
                out[fn] = dict(raises=str(e), text=buf.getvalue())
        out["getallsamples"] = g.getallsamples(m, burn=7, stride=3).tolist()
        cases[name] = out
    x = np.linspace(-3, 4, 9)
    misc = dict(x=x.tolist(), gauss=g.GaussPDF(x, 0.5, 2.0).tolist(),
                stddev=float(g.stddev(np.arange(7.0) ** 2)), key=g.key("chain_EE_3.txt"))
    with open(os.path.join(OUT, "getdist_g9.json"), "w") as f:
        json.dump(dict(ref_sha=ref_tree_sha(), cases=cases, misc=misc), f, indent=1)
    print("wrote getdist_g9.json")


def golden_beamfuncs(R):
    """G10: the object-level API of BeamFit/BeamModel.py, NormalizedBeamModel.py and the plane
    helpers of OffsetNormalizedBeamModel.py (set_Cinv scalars, gauss2d / at / atxy values, package
    round trips, planexy / plane)."""
    import json
    BM = R["BeamFit.BeamModel"]
    ONB = R["BeamFit.OffsetNormalizedBeamModel"]
    NBM = importlib.import_module("MCMC.BeamFit.NormalizedBeamModel")
    rng = np.random.default_rng(1010)
    x = rng.uniform(-20, 20, 40)
    y = rng.uniform(-20, 20, 40)
    cases = []
    for _ in range(6):
        center = (float(rng.uniform(-3, 3)), float(rng.uniform(-3, 3)))
        sigmas = (float(rng.uniform(3, 8)), float(rng.uniform(3, 8)))
        angle = float(rng.uniform(-1, 4))
        amp = float(rng.uniform(0.5, 12))
        m = BM.GaussianBeamModel2D(center, sigmas, angle)
        n = NBM.NormalizedBeamModel(center, sigmas, angle, amp)
        rho = float(rng.uniform(-0.8, 0.8))
        mxy = BM.GaussianBeamModel2D_xy(center, sigmas, rho)
        flat5 = BM.GaussianBeamModel2D.unpackage((center, sigmas, angle))
        cases.append(dict(
            center=center, sigmas=sigmas, angle=angle, amp=amp, rho_in=rho,
            angle_mod=float(m.angle), sig2_xy=[float(v) for v in m.sig2_xy], rho=float(m.rho),
            Cinv=m.Cinv.tolist(), atxy=m.atxy(x, y).tolist(), get_XYRho_s2=[float(v) for v in m.get_XYRho()[1]],
            flat5=flat5.tolist(), pack5=[list(BM.GaussianBeamModel2D.package(flat5)[0]),
                                         list(BM.GaussianBeamModel2D.package(flat5)[1]),
                                         float(BM.GaussianBeamModel2D.package(flat5)[2])],
            norm_atxy=n.atxy(x, y).tolist(),
            xy_sig2=[float(v) for v in mxy.sig2_xy], xy_Cinv=mxy.Cinv.tolist(), xy_atxy=mxy.atxy(x, y).tolist(),
            xy_flat=BM.GaussianBeamModel2D_xy.unpackage((center, sigmas, rho)).tolist(),
            planexy=np.asarray(ONB.planexy(0.93, 1.1, x, y)).tolist(),
            plane=np.asarray(ONB.plane(0.93, 1.1, types.SimpleNamespace(x=x, y=y))).tolist()))
    out = dict(ref_sha=ref_tree_sha(), x=x.tolist(), y=y.tolist(), cases=cases,
               planexy_mu1=ONB.planexy(1, 0.3, 1.0, 2.0),
               gauss2d=BM.gauss2d(x, y, 0.5, -0.25, 0.04, -0.01, 0.09).tolist())
    with open(os.path.join(OUT, "beamfuncs_g10.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote beamfuncs_g10.json")


def golden_writetab(R):
    """G11: the reference's results tables - postprocess() (driver.py:345-460) on checkpoint files
    written here from seeded stub records, then writeTab() (:478-534) - as data: the records, the
    fields of the structured arrays and the text files byte for byte.  Shims (no arithmetic): an
    empty `getdist` module so that driver.py imports (:47), and the module's `open` reads the
    checkpoints in binary mode (its text-mode open(fname) predates Python 3's pickle)."""
    import builtins
    import json
    import pickle
    import tempfile
    sys.modules.setdefault("getdist", types.ModuleType("getdist"))
    drv = importlib.import_module("MCMC.submmSED.driver")
    drv.open = lambda fn, mode="rb": builtins.open(fn, mode if "w" in mode else "rb")
    rng = np.random.default_rng(77)
    full = 6                                             # every run's chains carry the flat 6 (or fewer) params
    shapes = {0: (6, 2, 3), 1: (3, 1, 4), 3: (6, 2, 2), 4: (4, 1, 3)}   # run -> (P of the chain, nt, nobj)
    ndat = 5
    out = {"runs": {}, "ndat": ndat}
    with tempfile.TemporaryDirectory() as tmp:
        for irun, (P, nt, nobj) in shapes.items():
            recs = []
            for k in range(nobj):
                A = rng.standard_normal((P, P))
                cov = A @ A.T
                sd = np.sqrt(np.diag(cov))
                ML = float(rng.normal(-20, 5))
                recs.append(((rng.normal(size=P), sd, cov / np.outer(sd, sd)),
                             (ML, rng.normal(size=P), ML - abs(float(rng.normal()))),
                             "F%05d+%04d(z=%5.3f)" % (1234 * k + irun, 77 * k, 0.013 * k), 0.013 * k,
                             [(float(a), float(b)) for a, b in rng.uniform(0.1, 9, (ndat, 2))],
                             rng.uniform(1, 2, nt), float(rng.uniform(1, 30))))
            with builtins.open(os.path.join(tmp, "check%d.npy" % irun), "wb") as f:
                pickle.dump(recs, f)
            out["runs"][str(irun)] = {"records": [
                {"mean": r[0][0].tolist(), "sig": r[0][1].tolist(), "ncov": r[0][2].tolist(), "ML": r[1][0],
                 "MLpar": r[1][1].tolist(), "meanL": r[1][2], "name": r[2], "z": r[3], "dat": r[4],
                 "flux": r[5].tolist(), "ML_chi2": r[6]} for r in recs]}
        tabs = drv.postprocess(tmp + "/", check=True)
        for irun, tab in enumerate(tabs):
            if len(tab) == 0:
                continue
            e = out["runs"][str(irun)]
            e["fields"] = {k: tab[k].tolist() for k in ("mean", "sig", "covar", "ML", "ev", "MLpar", "MeanL",
                                                        "evMean", "dlnLike", "z", "dat", "flux", "ML_chi2")}
            e["name"] = [n.decode() for n in tab["name"]]
            for nodat in (False, True):
                fn = os.path.join(tmp, "tab%d_%d.txt" % (irun, nodat))
                drv.writeTab(tab, fn, nodat=nodat)
                with builtins.open(fn) as f:
                    e["text_nodat" if nodat else "text"] = f.read()
    out["reference_sha256"] = hashlib.sha256(
        builtins.open(os.path.join(REF, "MCMC/submmSED/driver.py"), "rb").read()).hexdigest()
    with builtins.open(os.path.join(OUT, "writetab_g11.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("writetab_g11.json: runs", sorted(out["runs"]))


def golden_setup_sampler(R):
    """G12: the beam driver's set-up, ProcessBeamData.setup_sampler (:124-216), run by the reference
    module itself on a synthetic map: class-level prior state, start point, proposal widths.  The
    module uses Python-2 implicit relative imports (`import MCMC`, `import Likelihood`, `from BeamFit
    import ...`): the names are aliased to the package's own modules in sys.modules (no arithmetic
    touched); its print() calls are silenced."""
    import contextlib
    import io
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import refbeam as RB                      # the synthetic map generator the tests use
    sys.modules.setdefault("Likelihood", R["Likelihood"])
    sys.modules.setdefault("BeamFit", importlib.import_module("MCMC.BeamFit"))
    importlib.import_module("MCMC.BeamFit.NormalizedBeamModel")
    sys.modules.setdefault("getdist", types.ModuleType("getdist"))
    PBD = importlib.import_module("MCMC.BeamFit.ProcessBeamData")
    BD = R["BeamFit.BeamData"]
    mod = R["BeamFit.OffsetNormalizedBeamModel"].OffsetNormalizedBeamModel
    cases = []
    for seed, n, scale, smm in [(8, 64, None, (3, 8)), (21, 48, 0.8, (2, 6)), (5, 40, None, (3, 8))]:
        x, y, d, noise, truth, prior = RB.synth_beam_maps(1, n=n, seed=seed)
        abs_sum = float(np.abs(d[0]).sum())
        data = BD.BeamData(x.copy(), y.copy(), d[0].copy(), noise)
        with contextlib.redirect_stdout(io.StringIO()):
            like, prop_sigmas, start = PBD.setup_sampler([data], data, useNormalizedBeam=True,
                                                         rangeScale=scale, sigminmax=smm)
        cases.append(dict(
            seed=seed, n=n, rangeScale=scale, sigminmax=list(smm),
            input_abs_sum=abs_sum, noise=float(noise),
            centerMin=[float(v) for v in mod.centerMin], centerMax=[float(v) for v in mod.centerMax],
            sigMin=float(mod.sigMin), sigMax=float(mod.sigMax),
            paramBlocks=[int(v) for v in mod.paramBlocks], nBlock=int(mod.nBlock),
            prop_sigmas=[float(v) for v in mod.unpackage(prop_sigmas)],
            start=[float(v) for v in mod.unpackage(start)],
            prior_at_start=float(mod.prior(*start)),
            lnlike_at_start=float(like.lnLike(start))))
    with open(os.path.join(OUT, "setup_sampler_g12.json"), "w") as f:
        json.dump(dict(ref_sha=ref_tree_sha(), cases=cases), f, indent=0)
    print("wrote setup_sampler_g12.json")


def main():
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["sed", "cl", "beam", "readers", "clfiles", "getdist", "beamfuncs", "writetab",
                             "setup_sampler"]
    R = load_reference()
    if "sed" in which:
        golden_sed(R)
    if "cl" in which:
        golden_cl(R)
    if "beam" in which:
        golden_beam(R)
    if "readers" in which:
        golden_readers(R)
    if "clfiles" in which:
        golden_clfiles(R)
    if "getdist" in which:
        golden_getdist(R)
    if "beamfuncs" in which:
        golden_beamfuncs(R)
    if "writetab" in which:
        golden_writetab(R)
    if "setup_sampler" in which:
        golden_setup_sampler(R)


if __name__ == "__main__":
    main()
