"""Binned C_l model with the interface of the reference's binnedClModel
(MCMC/binnedCl/binnedClModel.py:44-277): the parameters are the q_b factors multiplying a shape
spectrum in bins of l.  Class-level state (binning, shape) as in the reference; the spectrum, the
prior and - through binnedClLikelihood - the band powers are evaluated on the GPU."""
import math

import numpy as np

from .. import Proposal
from ..engine import Engine

positive_corr = True                                                 # :42


def bin_spectrum(bins, llCl):                                        # :268-277
    """Bin D_l = l(l+1)C_l/2pi with the logarithmic-integral weights (set-up helper)."""
    out = np.zeros(len(bins), dtype=np.float64)
    for i, b in enumerate(bins):
        ells = np.arange(b[0], b[1] + 1, dtype=np.float64)
        w = (ells + 0.5) / ells / (ells + 1.0)
        out[i] = (llCl[b[0]:b[1] + 1] * w).sum() / w.sum()
    return out


class binnedClModel(object):
    nCl = 0
    _version = 0          # bumped by setBinning: device copies of older binnings are stale
    _eng = None

    def __init__(self, Cb):
        self.Cb = np.asarray(Cb)

    @classmethod
    def _engine(cls):
        """engine holding only the binning (spectrum / prior evaluation without data)"""
        if cls._eng is None or cls._eng_version != cls._version:
            from ..ClData.ClData_CosmoMC import ClData_CosmoMC
            eng = Engine()
            L = cls.lmax + 1
            dummy = ClData_CosmoMC(num_points=1, obs=np.zeros(1), var=np.ones(1),
                                   window=np.zeros((1, 3, L)), win_min=np.zeros(1, np.int32),
                                   win_max=np.zeros(1, np.int32))
            eng.cl_load([dummy])
            eng.cl_set_binning(cls.bins, cls.shapefun)
            cls._eng, cls._eng_version = eng, cls._version
        return cls._eng

    def __call__(self):                                              # :62-86
        cls = type(self)
        cls.Cl = cls._engine().cl_model(self.Cb)
        return cls.Cl

    @classmethod
    def bandpowers(cls, qb):                                         # :88-101
        ret, ibin = [], 0
        for i in range(cls.nCl):
            ret.append(np.asarray(qb[ibin:ibin + cls.nbins[i]]) * cls.BPnorm[i])
            ibin += cls.nbins[i]
        return ret

    def BP(self, qb=None):                                           # :106-112
        return self.bandpowers(self.Cb if qb is None else qb)

    @classmethod
    def ClCovar(cls, covar):                                         # :114-120
        norm = np.asarray(np.concatenate(cls.BPnorm))
        return np.asarray(covar) * norm * norm.reshape(cls.nparam, 1)

    @classmethod
    def prior(cls, *C_b):                                            # :125-138
        if not positive_corr:
            return 1
        return int(cls._engine().cl_prior(np.asarray(C_b, dtype=np.float64)))

    @classmethod
    def setBinning(cls, bins, shapefun=None, doBlock=True, nCl=None):   # :145-252
        """bins[iCl] = [(lmin0, lmax0), ...] inclusive; shapefun[iCl] = D_l for l = 0..lmax
        (None: flat D_l = 1; scalar: flat D_l = scalar)."""
        bins = [np.asarray(b, dtype=np.int64).reshape(-1, 2) for b in bins]
        if nCl is None:
            nCl = len(bins)
        if nCl > 3:
            raise Exception("Error in shapes of bins, shapefun, giving nCl=%d" % nCl)
        cls.nCl = nCl
        cls.bins = bins
        cls.lmax = int(max(b.max() for b in bins))
        cls.lmin = int(min(b.min() for b in bins))
        if shapefun is None or np.isscalar(shapefun):
            flat = 1.0 if shapefun is None else float(shapefun)
            sf = np.full((nCl, cls.lmax + 1), flat)
            cls.BPnorm = [np.full(len(b), flat) for b in bins]
        else:
            sf = np.atleast_2d(np.asarray(shapefun, dtype=np.float64))[:, :cls.lmax + 1]
            if sf.shape != (nCl, cls.lmax + 1):
                raise ValueError("shapefun must cover l = 0..%d for each of the %d spectra" %
                                 (cls.lmax, nCl))
            cls.BPnorm = [bin_spectrum(b, s) for (b, s) in zip(bins, sf)]
        cls.shapefun = sf
        ell = np.arange(cls.lmax + 1, dtype=np.float64)
        ell[0] = 1.0
        cls.ellnorm = 2 * math.pi / (ell * (ell + 1.0))
        cls.ellnorm[0] = 0.0
        cls.ellctr = [b.mean(axis=1) for b in bins]
        cls.nbins = [len(b) for b in bins]
        cls.nparam = int(sum(cls.nbins))
        cls.Cltype = np.concatenate([[i] * n for i, n in enumerate(cls.nbins)]).astype(int)
        if doBlock:
            cls.paramBlocks = np.arange(cls.nparam)
            cls.nBlock = cls.nparam
        else:
            cls.paramBlocks = None
            cls.nBlock = 0
        cls.texNames = ["q_{%d}" % i for i in range(cls.nparam)]
        cls.fmtstring = " ".join(["%.3f"] * cls.nparam)
        cls._version += 1

    def package(Cb):
        return np.asarray(Cb)

    def unpackage(Cb):
        return np.asarray(Cb)

    proposal = Proposal.GenericGaussianProposal(package=package, unpackage=unpackage)
    unpackage = staticmethod(unpackage)
    package = staticmethod(package)


# ---------------------------------------------------------------------------------------------
# post-fit products (host-side, on the samples of a finished chain).  binnedClModel.py:329-503
# ---------------------------------------------------------------------------------------------
class oln(object):
    """offset-lognormal description of the samples C of one band power: z = ln(C + x) Gaussian
    with mean zbar and variance sigz2.  :329-387"""

    def __init__(self, C):
        self.C = C

    def chi2(self, zbar, sigz2, x):
        return ((np.log(self.C + x) / zbar - 1) ** 2).mean() * zbar ** 2 / sigz2

    @staticmethod
    def like1(par, C):
        zbar, sigz2, x = par
        return (np.log(2 * np.pi * sigz2) + (np.log(C + x) / zbar - 1) ** 2 * zbar ** 2 / sigz2 +
                2 * np.log(C + x))

    def like(self, par):
        """-2 ln P(par | samples); inf outside sigz2 > 0, C + x > 0"""
        zbar, sigz2, x = par
        if sigz2 < 0 or x + min(self.C) < 0:
            return np.inf
        return np.log(2 * np.pi * sigz2) + self.chi2(zbar, sigz2, x) + 2 * (np.log(self.C + x)).mean()

    def derivs(self, par):
        zbar, sigz2, x = par
        lz = np.log(self.C + x)
        return np.array([(1.0 / sigz2) * (1.0 - self.chi2(zbar, sigz2, x)),
                         (2.0 / sigz2) * (zbar - lz.mean()),
                         (2.0 / sigz2) * ((lz - zbar) / (self.C + x)).mean() + (2.0 / (self.C + x)).mean()])

    def cum(self, par, bins=None):
        """cumulative distribution at `bins` (default: the sorted samples; a scalar = bin width)"""
        from scipy import special
        zbar, sigz2, x = par
        if bins is None:
            bins = np.sort(self.C)
        elif np.isscalar(bins):
            bins = np.arange(self.C.min(), self.C.max(), bins)
        s = math.sqrt(2 * sigz2)
        return 0.5 * (special.erf((zbar - np.log(x)) / s) - special.erf((zbar - np.log(bins + x)) / s))

    def KSnorm(self, par):
        return max(abs(self.cum(par) - np.linspace(0, 1, len(self.C))))


def _oln_start(samples):
    x_0 = 1.1 * abs(min(samples))
    lz = np.log(samples + x_0)
    zbar_0 = lz.mean()
    return np.array([zbar_0, ((lz - zbar_0) ** 2).mean(), x_0])


def fitOffsetLognormal_cum(samples, full_output=0, do_plot=0, verbose=False):
    """(zbar, sigz2, x) minimising the Kolmogorov-Smirnov distance between the cumulative sample
    distribution and the offset lognormal (Nelder-Mead, the reference's tolerances).  :390-420
    Plotting is not rebuilt."""
    from scipy import optimize
    o = oln(samples)
    par_0 = _oln_start(samples)
    f = optimize.fmin(o.KSnorm, par_0, maxfun=100000, maxiter=100000, xtol=0.00001, ftol=0.00001,
                      disp=verbose)
    return f


fitOffsetLognormal = fitOffsetLognormal_cum                          # :422


def fitOffsetLognormal_like(samples, full_output=0):
    """maximum-likelihood offset lognormal with L-BFGS-B and analytic derivatives; returns
    scipy's (x, f, info) triple.  :424-450"""
    from scipy import optimize
    o = oln(samples)
    return optimize.fmin_l_bfgs_b(o.like, _oln_start(samples), fprime=o.derivs)


def FisherWindows(F, bins=None, isCovar=False):
    """Effective band-power windows from the inverse covariance F of the bins (isCovar: F is the
    covariance).  Without bins: W_bb' = F_bb' / F_bb.  With bins (per spectrum lists of (lmin,
    lmax)): W[B, spectrum, l] = F_Bb / (lmax_b - lmin_b) / F_BB for l in bin b.  :454-503"""
    fish = np.linalg.inv(F) if isCovar else np.asarray(F)
    nbin = fish.shape[0]
    if bins is None:
        return (fish.T / fish.diagonal()).T
    top = max(np.array(b).max() for b in bins)
    WBl = np.zeros((nbin, 3, top + 1), dtype=np.float64)
    for ib in range(nbin):
        jb = 0
        for ispec, spec in enumerate(bins):
            for b in spec:
                WBl[ib, ispec, b[0]:b[1] + 1] = fish[ib, jb] / (b[1] - b[0]) / fish[ib, ib]
                jb += 1
    return WBl


class unphys(Exception):                                             # :507 (never raised there)
    pass
