"""Metropolis sampler with the interface of the reference's MCMC/MCMC.py (class MCMC, sampler(),
chain_analyze()).  The propose / evaluate / accept loop runs inside one HIP kernel launch per
MC_append; chain state and moments live in HBM between calls.

Extensions over the reference: `nchains` independent chains per MCMC object (all advanced by the
same launch), `seed` for the counter-based Philox streams, and draw injection for step-exact
comparisons with the reference."""
import os

import numpy as np

from .Likelihood import ZeroPosterior  # noqa: F401
from .engine import Engine, Mi355Error


class ZeroPriorError(Exception):
    def __init__(self, *value):
        self.value = value

    def __str__(self):
        return repr(self.value)


from contextlib import contextmanager


@contextmanager
def finishing(obj):                                                  # MCMC.py:51-56
    """call obj.finish() on the way out (the reference wraps its progress bar in this)"""
    try:
        yield obj
    finally:
        obj.finish()


def _device_like(like):
    if not hasattr(like, "load_into") or not hasattr(like, "problem_kind"):
        raise NotImplementedError(
            "mcmc_amd.MCMC runs the chain on the GPU and needs a likelihood backed by the HIP "
            "library (SEDLikelihood_*, binnedClLikelihood, NormalizedBeamLikelihood); there is no "
            "CPU sampler to fall back to")
    return like


class MCMC(object):
    """Metropolis sampler.  Reference: MCMC.py:58-434."""

    def __init__(self, like, proposal=None, startProposal=None, nMC=0, startParams=None,
                 doBlock=False, nchains=1, seed=None, keep_trace=True, stepwise=None):
        """stepwise: None = the fused kernel for SED likelihoods (uncorrelated proposals only),
        the step-wise engine (any proposal of Proposal.py, covariance tracked on the device) for
        binnedCl and BeamFit; True forces the step-wise engine for SED too - needed for
        newMCMC(noCorrelations=False) / rotateParams."""
        self.like = _device_like(like)
        self.kind = like.problem_kind
        self.stepwise = (self.kind != "sed") if stepwise is None else bool(stepwise)
        if self.kind != "sed" and not self.stepwise:
            raise ValueError("the fused kernel serves SED likelihoods only")
        self.doBlock = doBlock
        self.iBlock = 0
        self.mod = like.model
        for attr in ("texNames", "paramBlocks", "nBlock", "fmtstring"):
            if hasattr(self.mod, attr):
                setattr(self, attr, getattr(self.mod, attr))
        self.data = like.data
        self.nchains = int(nchains)
        self.keep_trace = keep_trace
        if seed is None:                                   # seed(seed=None), MCMC.py:113
            seed = int.from_bytes(os.urandom(8), "little")
        self.seed = seed
        self.prop = self.mod.proposal() if proposal is None else proposal()
        if startProposal is not None:
            self.prop.setSigmas(startProposal)
        self._eng = None
        self._detached = False
        if startParams is not None:
            self.setStart(startParams)
            if nMC > 0:
                self.MC_append(nMC)

    # ---- device plumbing -----------------------------------------------------------------
    def _engine(self):
        if self._detached:
            raise Mi355Error("this MCMC object handed its device chains to newMCMC()")
        if self._eng is None:
            if len(self.like.data) != 1:
                raise NotImplementedError("one dataset per sampler object (the Engine API runs "
                                          "whole catalogues / map sets in one launch)")
            eng = Engine(seed=self.seed)
            self.like.load_into(eng)
            self._eng = eng
        return self._eng

    def _squeeze(self, a):
        """[.., S=1, C, ..] device layout -> reference layout (no chain axis when nchains==1)."""
        a = np.asarray(a)
        return a[..., 0, :] if self.nchains == 1 else a

    def _refresh_counts(self):
        mom = self._eng.mh_moments()
        self.naccept = int(mom["naccept"][0, 0]) if self.nchains == 1 else mom["naccept"][0].copy()
        if self.doBlock:
            self.accept = mom["accept"][0, 0].copy() if self.nchains == 1 else mom["accept"][0].copy()
        self.prev = self.like.model.package(mom["theta"][0, 0]) if self.nchains == 1 else mom["theta"][0]
        self.prev_lnPr = float(mom["lnPr"][0, 0]) if self.nchains == 1 else mom["lnPr"][0].copy()
        return mom

    # ---- reference API -------------------------------------------------------------------------
    def setStart(self, params, random=True, verbose=True):
        """MCMC.py:138-187.  random=True redraws the start from the proposal until prior>0."""
        eng = self._engine()
        paramarr = np.asarray(self.like.model.unpackage(params), dtype=np.float64)
        self.nparams = paramarr.shape[-1]
        blocks = np.asarray(self.like.model.paramBlocks, dtype=np.int32) if self.doBlock else None
        try:
            if self.stepwise:
                eng.mh_init_problem(self.kind, self.nchains, paramarr, self.prop.sigmas,
                                    param_blocks=blocks, randomize_start=random, track_cov=True)
                if getattr(self.prop, "sqrtMatrix", None) is not None:
                    eng.mh_set_sqrt_matrix(self.prop.sqrtMatrix, rotate=self.prop.rotateParams)
            else:
                if getattr(self.prop, "sqrtMatrix", None) is not None:
                    raise NotImplementedError("correlated proposals need MCMC(..., stepwise=True)")
                eng.mh_init(self.nchains, paramarr, self.prop.sigmas, param_blocks=blocks,
                            randomize_start=random)
        except Mi355Error as e:
            if "ZeroPriorError" in str(e):
                raise ZeroPriorError(params)
            raise
        mom = self._refresh_counts()
        th0 = mom["theta"][0]                               # [C, P]
        self.samples = th0[None, 0] if self.nchains == 1 else th0[None]
        self.lnPr = mom["lnPr"][0, :1].copy() if self.nchains == 1 else mom["lnPr"][0][None]
        self.accepted = np.ones_like(self.lnPr, dtype=bool)
        self.nDerived = getattr(self.like, "nDerived", 0)
        self.iBlock = 0

    def getStart(self):
        return self.like.model.package(self.samples[0])

    def MC_append(self, nMC=1, inj_normals=None, inj_uniforms=None):
        """MCMC.py:196-221: nMC more steps for every chain in one kernel launch."""
        eng = self._engine()
        if inj_normals is not None:
            inj_normals = np.asarray(inj_normals, np.float64).reshape(nMC, self.nchains, self.nparams)
            inj_uniforms = np.asarray(inj_uniforms, np.float64).reshape(nMC, self.nchains)
        eng.mh_run(nMC, thin=1 if self.keep_trace else 0, inj_normals=inj_normals,
                   inj_uniforms=inj_uniforms)
        if self.keep_trace:
            s, l, a = eng.mh_trace()
            self.samples = np.concatenate((self.samples, self._squeeze(s[:, 0])
                                           if self.nchains == 1 else s[:, 0]))
            self.lnPr = np.concatenate((self.lnPr, l[:, 0, 0] if self.nchains == 1 else l[:, 0]))
            self.accepted = np.concatenate((self.accepted, a[:, 0, 0] if self.nchains == 1 else a[:, 0]))
        self._refresh_counts()
        if self.doBlock:
            self.iBlock = (self.iBlock + nMC) % self.like.model.nBlock

    def sample(self):
        """MCMC.py:225-286: one step; returns (params, lnPr, accepted, derived)."""
        self.MC_append(1)
        acc = self.accepted[-1] if self.keep_trace else None
        return self.prev, self.prev_lnPr, acc, None

    # ---- chain statistics (MCMC.py:290-340) -----------------------------------------------------
    def _device_moments_valid(self, burn, stride):
        eng = self._eng
        return (eng is not None and not self._detached and burn == 1 and stride == 1 and
                eng.nsamples > 0 and (not self.keep_trace or eng.nsamples == len(self.lnPr) - 1))

    def mean(self, burn=0, stride=1):
        if self._device_moments_valid(burn, stride):
            m = self._eng.mh_moments()["mean"][0]
            self.means = m[0] if self.nchains == 1 else m
        else:
            self.means = np.asarray(self.samples)[burn::stride].mean(axis=0)
        if self.nchains == 1:
            return self.like.model.package(self.means)
        return np.array([self.like.model.package(m) for m in self.means])

    def stdev(self, burn=0, stride=1):
        if self._device_moments_valid(burn, stride):
            v = self._eng.mh_moments()["var"][0]
            self.stdevs = np.sqrt(v[0] if self.nchains == 1 else v)
        else:
            self.stdevs = np.asarray(self.samples)[burn::stride].std(axis=0)
        if self.nchains == 1:
            return self.like.model.package(self.stdevs)
        return np.array([self.like.model.package(s) for s in self.stdevs])

    def covar(self, burn=0, stride=1, unNormalized=False, stdevs=None, params=None):
        if self.nchains != 1:
            raise NotImplementedError("covar() is per chain: use nchains=1")
        if params is None:
            params = list(range(self.nparams))
        if self.stepwise and self._device_moments_valid(burn, stride) and stdevs is None:
            # running co-moments of the device chain; normalised, as the reference does, by the
            # *packaged* stdevs (MCMC.py:321-333)
            sd = np.asarray(self.like.model.unpackage(self.stdev(burn, stride)), np.float64)
            cov = self._eng.mh_covar(normalized=False)[0, 0]
            self.covars = np.zeros((self.nparams, self.nparams))
            ix = np.ix_(params, params)
            with np.errstate(all="ignore"):
                self.covars[ix] = (cov if unNormalized else cov / np.outer(sd, sd))[ix]
            return self.covars
        means = self.like.model.unpackage(self.mean(burn, stride))
        if stdevs is None:
            stdevs = self.like.model.unpackage(self.stdev(burn, stride))
        chain = np.asarray(self.samples)[burn::stride]
        self.covars = np.zeros((self.nparams, self.nparams))
        with np.errstate(all="ignore"):
            for i in params:
                for j in params:
                    c = ((chain[:, i] - means[i]) * (chain[:, j] - means[j])).mean()
                    self.covars[i, j] = c if unNormalized else c / (stdevs[i] * stdevs[j])
        return self.covars

    def newMCMC(self, burn=0, stride=1, fac=None, nMC=None, noCorrelations=False, doBlock=None,
                rotateParams=False, random=False, start_draws=None):
        """MCMC.py:343-413: new chain(s) started at the present mean with proposal widths
        fac*stdev (fac = 2.4/sqrt(nparam) by default); this object keeps its trace, the device
        state moves to the new object."""
        correlated = rotateParams or not noCorrelations
        if correlated and not self.stepwise:
            raise NotImplementedError(
                "correlated / rotated proposals run on the step-wise engine: create the sampler "
                "with MCMC(..., stepwise=True) (sampler() does so when noCorrelations=False), or "
                "pass noCorrelations=True as submmSED/driver.py:173 does")
        if correlated and not self._device_moments_valid(burn, stride):
            raise NotImplementedError("correlated restarts use the device's running covariance: "
                                      "burn=1, stride=1 (what sampler() passes, MCMC.py:515-520)")
        if correlated and start_draws is not None:
            raise NotImplementedError("start_draws is a test hook of the uncorrelated path")
        if doBlock is None:
            doBlock = self.doBlock
        eng = self._engine()
        new = MCMC.__new__(MCMC)
        new.__dict__.update({k: v for k, v in self.__dict__.items()
                             if k not in ("samples", "lnPr", "accepted", "means", "stdevs", "covars")})
        new.prop = self.mod.proposal()
        new.doBlock = doBlock
        if start_draws is not None:
            # test hook: the restart is randomised with caller-supplied normals (one row of
            # nparam per call, redrawn until prior>0: MCMC.py:151-159) instead of Philox
            npar = self.like.model.nparam
            f = 2.4 / np.sqrt(npar) if fac is None else fac
            start = np.asarray(self.mean(burn, stride), np.float64)
            std = np.asarray(self.stdev(burn, stride), np.float64)
            if random:
                start = start + start_draws() * std
            sig = np.asarray(self.like.model.package(f * std), np.float64)
            while True:
                th = self.like.model.package(start + start_draws() * sig)
                if self.like.model.prior(*th) > 0:
                    break
            blocks = np.asarray(self.like.model.paramBlocks, dtype=np.int32) if doBlock else None
            self._reinit(eng, th, sig, blocks, False)
        elif correlated:
            eng.mh_adapt_corr(fac, random_restart=random, no_correlations=False, rotate=rotateParams,
                              do_block=doBlock)
        elif self.stepwise and self._device_moments_valid(burn, stride):
            eng.mh_adapt_corr(fac, random_restart=random, no_correlations=True, do_block=doBlock)
        elif self._device_moments_valid(burn, stride) and doBlock == self.doBlock:
            eng.mh_adapt(fac, random_restart=random)
        else:
            npar = self.like.model.nparam
            f = 2.4 / np.sqrt(npar) if fac is None else fac
            start = np.asarray(self.mean(burn, stride), np.float64)
            std = np.asarray(self.stdev(burn, stride), np.float64)
            if random:
                # host-side restart offsets from a generator of their own, keyed by (seed, stage):
                # the device's normal stream at step_total is what the next MC_append consumes
                rng = np.random.default_rng([self.seed & 0xFFFFFFFFFFFFFFFF, int(eng.stage), 0x5EED])
                start = start + rng.standard_normal(start.shape) * std
            sig = np.array([self.like.model.package(f * s) for s in np.atleast_2d(std)])
            blocks = np.asarray(self.like.model.paramBlocks, dtype=np.int32) if doBlock else None
            self._reinit(eng, np.atleast_2d(start)[None], sig[None], blocks, True)
        self._detached = True
        new._eng, new._detached = eng, False
        mom = new._refresh_counts()
        sig = eng_sigmas(eng)
        new.prop.setSigmas(self.like.model.package(sig[0, 0]))
        new.prop.sigmas_all = sig[0]
        if correlated:
            Ls = eng.mh_sqrt_matrix()[0]
            new.prop.sqrtMatrix = Ls[0] if self.nchains == 1 else Ls
            new.prop.rotateParams = rotateParams
        th0 = mom["theta"][0]
        new.samples = th0[None, 0] if self.nchains == 1 else th0[None]
        new.lnPr = mom["lnPr"][0, :1].copy() if self.nchains == 1 else mom["lnPr"][0][None]
        new.accepted = np.ones_like(new.lnPr, dtype=bool)
        new.iBlock = 0
        if nMC is not None:
            new.MC_append(nMC)
        return new

    def _reinit(self, eng, theta, sigma, blocks, randomize):
        if self.stepwise:
            eng.mh_init_problem(self.kind, self.nchains, theta, sigma, param_blocks=blocks,
                                randomize_start=randomize, track_cov=True)
        else:
            eng.mh_init(self.nchains, theta, sigma, param_blocks=blocks, randomize_start=randomize)

    def copyMCMC(self):                                   # MCMC.py:415-423
        new = MCMC(self.like, doBlock=self.doBlock, nchains=self.nchains)
        new.prop = self.prop
        new.setStart(self.getStart(), random=False)
        return new

    def __getstate__(self):                               # MCMC.py:425-432
        odict = self.__dict__.copy()
        odict["mod"] = self.mod.__name__
        for k in ("prop", "like", "_eng"):
            odict.pop(k, None)
        return odict


def eng_sigmas(eng):
    """Current proposal widths of every chain, [S, C, P] (after newMCMC they differ per chain)."""
    return eng.mh_sigmas()


def chain_analyze(chain, params=None):
    """(means, stdevs, normalised covariance) of a [nsamples, nparam] array.  MCMC.py:445-477."""
    chain = np.asarray(chain)
    n = chain.shape[1]
    if params is None:
        params = list(range(n))
    means = [chain[:, i].mean() for i in range(n)]
    stdevs = [chain[:, i].std() for i in range(n)]
    covar = np.zeros((n, n))
    with np.errstate(all="ignore"):
        for i in params:
            for j in params:
                covar[i, j] = (((chain[:, i] - means[i]) * (chain[:, j] - means[j])).mean() /
                               (stdevs[i] * stdevs[j]))
    return means, stdevs, covar


def sampler(like, nMC, prop_sigmas, start_params, plotter=None, fac=None, noCorrelations=False,
            doBlock=False, rotateParams=False, randomrestart=False, seed=None, verbose=True,
            inject=None):
    """Staged sampling, MCMC.py:479-575: chain k restarts from chain k-1's moments (burn = one
    sample, stride 1), a stage is repeated while it produced no usable statistics.
    `inject` (optional) = (normals [n,P], uniforms [n]) consumed stage after stage (test hook)."""
    stages = []
    if rotateParams:
        noCorrelations = False
    noCorr1 = noCorrelations
    nresample, nresample_max = 0, 10
    params = np.where(np.asarray(like.model.unpackage(prop_sigmas)) > 0)[0]
    ana = None
    burnfrac = stride = None
    cursor = [0]

    def append(s, n):
        if inject is None:
            s.MC_append(n)
        else:
            k = cursor[0]
            s.MC_append(n, inj_normals=inject[0][k:k + n], inj_uniforms=inject[1][k:k + n])
            cursor[0] = k + n

    for isamp, nMC1 in enumerate(nMC):
        if isamp == 0:
            new_s = MCMC(like, startProposal=prop_sigmas, nMC=0, startParams=None, doBlock=doBlock,
                         seed=seed, stepwise=True if not noCorrelations else None)
            if inject is None:
                new_s.setStart(start_params)
            else:                                   # randomise the start with the injected rows
                sig0 = np.asarray(like.model.unpackage(prop_sigmas), np.float64)
                while True:
                    cursor[0] += 1
                    th = like.model.package(np.asarray(start_params, np.float64) +
                                            inject[0][cursor[0] - 1] * sig0)
                    if like.model.prior(*th) > 0:
                        break
                new_s.setStart(th, random=False)
        else:
            draws = None
            if inject is not None:
                def draws():
                    cursor[0] += 1
                    return inject[0][cursor[0] - 1]
            new_s = stages[isamp - 1].newMCMC(burn=nMC[isamp - 1] // burnfrac, stride=stride, nMC=None,
                                              fac=fac, noCorrelations=noCorr1,
                                              rotateParams=rotateParams, random=randomrestart,
                                              start_draws=draws)
        burnfrac = nMC1
        while True:
            append(new_s, nMC1)
            ntot = len(new_s.samples)
            if verbose:
                print("done with chain %d. naccept=%d/%d (%f)" % (isamp, new_s.naccept, ntot,
                                                                 new_s.naccept / ntot))
                if new_s.doBlock:
                    print("per block: ", new_s.accept)
            noCorr1 = True if new_s.naccept < 10 else noCorrelations
            if new_s.naccept == 0:
                continue
            stride = 1
            ana = chain_analyze(new_s.samples[(ntot // burnfrac)::stride, :], params=params)
            means = np.array(ana[0])[params]
            stdvs = np.array(ana[1])[params]
            eps = 1.e-9
            with np.errstate(invalid="raise", divide="ignore"):
                try:
                    if max(stdvs / means) < eps:
                        nresample += 1
                        if nresample > nresample_max:
                            break
                        continue
                except (ZeroDivisionError, FloatingPointError):
                    idx = abs(means) < 1.e-11
                    if max(stdvs[idx]) < eps:
                        continue
            break
        stages.append(new_s)
        if plotter is not None:
            plotter(new_s)
    return stages, ana
