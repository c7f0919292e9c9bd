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
