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
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


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
                 "submmSED.model", "submmSED.likelihood", "submmSED.data"]:
        mods[name] = importlib.import_module("MCMC." + name)
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


def main():
    os.makedirs(OUT, exist_ok=True)
    R = load_reference()
    model, like_mod, data_mod = R["submmSED.model"], R["submmSED.likelihood"], R["submmSED.data"]
    MC, Prop, conv = R["MCMC"], R["Proposal"], R["convergence"]
    sha = ref_tree_sha()
    rng = np.random.default_rng(1234)

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
                        # SEDLikelihood2.lnLike1 raises TypeError under SciPy>=1.11 (sym_pos kwarg,
                        # likelihood.py:63); evaluate its formula with the same quadforms and
                        # numpy.linalg.solve.  Flagged "restated" in DESIGN.md.
                        L.setModel(th[t])
                        mv, dd = L.model_vals[0], L.data[0]
                        FNiF = dd.quadform(mv); FNid = dd.quadform(mv, dd.d)
                        try:
                            amp = np.linalg.solve(FNiF, FNid)
                        except np.linalg.LinAlgError:
                            amp = np.array([np.nan, np.nan])
                        lnl[di, t] = 0.5 * np.dot(FNid.transpose(), amp)
                        der[di, t] = amp
                        chi2[di, t] = np.nan
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


if __name__ == "__main__":
    main()
