"""NumPy restatement of the reference's binned-C_l band-power likelihood.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Parity status: PINNED by
tests/golden/cl_g5.npz, produced by oracle/make_golden.py from the reference's own
binnedClModel / ClData_CosmoMC classes (attributes injected, the file parsers are Py2-only).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

NUM_CLS = 3          # TT, TE, EE                       ClData_CosmoMC.py:38
HALFSTEPS = 5        # ClData_CosmoMC.py:40


def marge_weights(halfsteps=HALFSTEPS):
    """initNumericalMarge, ClData_CosmoMC.py:42-47."""
    step = np.arange(-halfsteps, halfsteps + 1, 1.0, dtype=np.float64)
    w = np.exp(-(step * 3 / float(halfsteps)) ** 2 / 2)
    return sum(w), w


MARGENORM, MARGEWEIGHTS = marge_weights()


# ---------------------------------------------------------------------------
# binnedClModel: binnedCl/binnedClModel.py
# ---------------------------------------------------------------------------
@dataclass
class BinnedClModel:
    """State that binnedClModel.setBinning (:145-252) puts on the class."""
    bins: List[np.ndarray]            # per spectrum: [nbin, 2] inclusive (lmin, lmax)
    shapefun: np.ndarray              # [nCl, lmax+1]  D_l = l(l+1)C_l/2pi shape

    def __post_init__(self):
        self.bins = [np.asarray(b, dtype=np.int64).reshape(-1, 2) for b in self.bins]
        self.nCl = len(self.bins)
        self.lmax = int(max(b.max() for b in self.bins))
        self.lmin = int(min(b.min() for b in self.bins))
        self.shapefun = np.asarray(self.shapefun, np.float64)[:, :self.lmax + 1]   # App. B quirk
        ell = np.arange(self.lmax + 1, dtype=np.float64)
        ell[0] = 1.0
        self.ellnorm = 2 * math.pi / (ell * (ell + 1.0))                            # :216-219
        self.ellnorm[0] = 0.0
        self.nbins = [len(b) for b in self.bins]
        self.nparam = int(sum(self.nbins))
        self.Cltype = np.concatenate([[i] * n for i, n in enumerate(self.nbins)]).astype(int)
        self.param_blocks = np.arange(self.nparam)
        self.nblock = self.nparam

    def Cl(self, q):
        """binnedClModel.__call__, :62-86 -> [3, lmax+1] (rows beyond nCl stay zero)."""
        q = np.asarray(q, np.float64)
        out = np.zeros((3, self.lmax + 1))
        ibin = 0
        for iCl in range(self.nCl):
            out[iCl] = self.shapefun[iCl].copy()
            Cb = q[ibin:ibin + self.nbins[iCl]]
            for Cbin, rng in zip(Cb, self.bins[iCl]):
                out[iCl, rng[0]:rng[1] + 1] *= Cbin
            ibin += self.nbins[iCl]
            out[iCl] *= self.ellnorm
        return out

    def prior(self, q):
        """:125-132 (positive_corr=True): auto-spectra bins must be >= 0."""
        q = np.asarray(q)
        return 0 if np.any(q[self.Cltype != 1] < 0) else 1


# ---------------------------------------------------------------------------
# ClData_CosmoMC: ClData/ClData_CosmoMC.py:262-446
# ---------------------------------------------------------------------------
@dataclass
class ClDataset:
    obs: np.ndarray                   # [nb]: z = ln(C+x) where has_xfactor, else C
    N_inv: Optional[np.ndarray]       # [nb, nb] (has_corr_errors) ...
    var: Optional[np.ndarray]         # ... or [nb] variances
    window: np.ndarray                # [nb, NUM_CLS, lmax_w+1]
    win_min: np.ndarray
    win_max: np.ndarray
    xfactors: Optional[np.ndarray] = None
    has_xfactor: Optional[np.ndarray] = None
    calib_uncertainty: float = 0.0
    beam_uncertain: bool = False
    beam_err: Optional[np.ndarray] = None
    has_pol: bool = False
    inc_pol: Optional[np.ndarray] = None

    def __post_init__(self):
        self.num_points = len(self.obs)
        self.has_corr_errors = self.N_inv is not None
        self.has_xfactors = self.xfactors is not None and bool(np.any(self.has_xfactor))

    def win_bandpower(self, i, Cl):                                     # :422-446
        ClTT = Cl[0] if np.ndim(Cl) == 2 else Cl
        maxl = min([len(ClTT), self.win_max[i] + 1])
        lo = self.win_min[i]
        bp = np.dot(ClTT[lo:maxl], self.window[i, 0, lo:maxl])
        if self.has_pol and self.inc_pol[i]:
            for iCl in range(1, NUM_CLS):
                bp += np.dot(Cl[iCl, lo:maxl], self.window[i, iCl, lo:maxl])
        return bp

    def bandpowers(self, Cl):
        return np.array([self.win_bandpower(i, Cl) for i in range(self.num_points)])

    def getdelta(self, BP):                                             # :377-398
        if self.has_xfactors:
            diffs = np.empty(self.num_points)
            idx = self.has_xfactor
            ndx = np.logical_not(idx)
            zth = BP[idx] + self.xfactors[idx]
            zth[zth <= 0.0] = 0
            with np.errstate(invalid="ignore", divide="ignore"):
                diffs[idx] = self.obs[idx] - np.log(zth)
            diffs[ndx] = self.obs[ndx] - BP[ndx]
            return diffs
        return self.obs - BP

    def getlinform(self, vec):                                          # :407-412
        return np.dot(self.N_inv, vec) if self.has_corr_errors else vec / self.var

    def getquadform(self, vec):                                         # :415-420
        if self.has_corr_errors:
            return np.dot(vec, np.dot(self.N_inv, vec))
        return sum(vec ** 2 / self.var)

    def getchisq(self, BP):                                             # :400-405
        return self.getquadform(self.getdelta(BP))

    def calib_marge_chisq(self, bandpowers):                            # :317-372
        hs = HALFSTEPS
        bandpowers = np.array(bandpowers)
        chisqcalib = np.empty(2 * hs + 1)
        chisq = np.empty(2 * hs + 1)
        hrange = list(range(-hs, hs + 1))
        with np.errstate(all="ignore"):
            for ibeam in hrange:
                bbp = bandpowers.copy()
                if self.beam_uncertain:
                    bbp *= 1 + self.beam_err * ibeam * 3 / hs
                for i in hrange:
                    calib = 1 + self.calib_uncertainty * i * 3 / hs
                    chisq[i + hs] = self.getchisq(calib * bbp)
                minchisq = min(chisq)
                exparg = -(chisq - minchisq) / 2
                chisqcalib[ibeam + hs] = -2 * np.log(sum(MARGEWEIGHTS * np.exp(exparg)) / MARGENORM) + minchisq
                if not self.beam_uncertain:
                    return chisqcalib[ibeam + hs]
            minchisq = min(chisqcalib)
            exparg = -(chisqcalib - minchisq) / 2
            return -2 * np.log(sum(MARGEWEIGHTS * np.exp(exparg)) / MARGENORM) + minchisq

    def calc_lnlike(self, Cl):
        """calcLnLike (:267-315): returns MINUS ln L = chisq/2."""
        denom = 1
        bandpowers = self.bandpowers(Cl)
        with np.errstate(all="ignore"):
            if self.has_xfactors and (self.calib_uncertainty > 1.0e-4 or self.beam_uncertain):
                chisq = self.calib_marge_chisq(bandpowers)
            else:
                diffs = self.getdelta(bandpowers)
                chisq = self.getquadform(diffs)
                if self.calib_uncertainty > 1.0e-4 or self.beam_uncertain:
                    tmp = self.getlinform(bandpowers)
                    chi2op = np.dot(diffs, tmp)
                    chi2pp = np.dot(bandpowers, tmp)
                    if self.beam_uncertain:
                        beam = self.beam_err * bandpowers
                        tmp = self.getlinform(beam)
                        chi2dd = np.dot(beam, tmp)
                        chi2pd = np.dot(bandpowers, tmp)
                        chi2od = np.dot(diffs, tmp)
                    if self.calib_uncertainty > 1.0e-4:
                        wpp = 1 / (chi2pp + 1 / self.calib_uncertainty ** 2)
                        chisq -= wpp * chi2op ** 2
                        denom /= wpp * self.calib_uncertainty ** 2
                    else:
                        wpp = 0
                    if self.beam_uncertain:
                        wdd = 1 / (chi2dd - wpp * chi2pd ** 2 + 1)
                        chisq -= wdd * (chi2od - wpp * chi2op * chi2pd) ** 2
                        denom /= wdd
            if denom != 1:
                chisq += np.log(denom)
        return chisq / 2


def cl_lnlike(q, model: BinnedClModel, datasets: Sequence[ClDataset]) -> float:
    """binnedClLikelihood.lnLike, binnedCl/binnedClLikelihood.py:26-32."""
    Cl = model.Cl(q)
    return float(sum([-d.calc_lnlike(Cl) for d in datasets]))


# ---------------------------------------------------------------------------
# linearisation used by the device path: BP = q.K + BP0 (exact, BP is linear in q)
# ---------------------------------------------------------------------------
def linearise_bandpowers(model: BinnedClModel, d: ClDataset):
    """K[b,i] = d BP_i / d q_b, BP0_i = BP_i at q=0 minus ... (the un-binned multipoles keep
    the shape, :77).  Evaluated with the model itself so it is the same arithmetic."""
    npar = model.nparam
    bp0 = d.bandpowers(model.Cl(np.zeros(npar)))
    K = np.empty((npar, d.num_points))
    for b in range(npar):
        e = np.zeros(npar)
        e[b] = 1.0
        K[b] = d.bandpowers(model.Cl(e)) - bp0
    return K, bp0


# ---------------------------------------------------------------------------
# synthetic config C4: flat top-hat windows on a shape spectrum
# ---------------------------------------------------------------------------
def synth_cl_problem(nb=64, npar=32, seed=7, lmax=2000, with_x=True, calib=0.05, beam=0.02,
                     corr=0.3, shape=None):
    rng = np.random.default_rng(seed)
    ell = np.arange(lmax + 1, dtype=np.float64)
    if shape is None:
        shape = 5000.0 * np.exp(-((ell - 220.0) / 600.0) ** 2) + 800.0 / (1 + (ell / 800.0) ** 2)
    shape = np.asarray(shape, np.float64)[:lmax + 1].copy()
    shape[:2] = 0.0
    edges = np.linspace(2, lmax + 1, npar + 1).astype(int)
    bins = [np.stack([edges[:-1], edges[1:] - 1], axis=1)]
    model = BinnedClModel(bins, shape[None, :])
    wedges = np.linspace(30, lmax - 10, nb + 1).astype(int)
    window = np.zeros((nb, NUM_CLS, lmax + 1))
    win_min = np.empty(nb, int)
    win_max = np.empty(nb, int)
    for i in range(nb):
        lo, hi = wedges[i], wedges[i + 1] + 5          # slightly overlapping top hats in D_l
        hi = min(hi, lmax)
        l = ell[lo:hi + 1]
        w = (l * (l + 1) / (2 * math.pi)) / (hi - lo + 1) * (1 + 0.2 * np.sin(l / 17.0))
        window[i, 0, lo:hi + 1] = w
        win_min[i], win_max[i] = lo, hi
    truth = rng.uniform(0.8, 1.2, npar)
    probe = ClDataset(np.zeros(nb), None, np.ones(nb), window, win_min, win_max)
    bp_true = probe.bandpowers(model.Cl(truth))
    sig = 0.05 * bp_true
    A = rng.normal(size=(nb, nb))
    Cn = A @ A.T
    dg = np.sqrt(np.diag(Cn))
    Cn = Cn / dg[:, None] / dg[None, :]
    Cn = (1 - corr) * np.eye(nb) + corr * Cn          # SPD correlation, |rho| <= corr off-diagonal
    cov = Cn * sig[:, None] * sig[None, :]
    cobs = bp_true + np.linalg.cholesky(cov) @ rng.normal(size=nb)
    if with_x:
        x = 0.3 * bp_true
        has_x = np.ones(nb, bool)
        has_x[::7] = False                             # a few Gaussian bands among lognormal ones
        obs = np.where(has_x, np.log(cobs + x), cobs)
        # z-space inverse covariance: N^-1_z = (C+x) N^-1 (C+x)   (ClData_bcp.py:192-210)
        f = np.where(has_x, cobs + x, 1.0)
        ninv = np.linalg.inv(cov) * f[:, None] * f[None, :]
    else:
        x, has_x, obs, ninv = None, None, cobs, np.linalg.inv(cov)
    d = ClDataset(obs, ninv, None, window, win_min, win_max, xfactors=x, has_xfactor=has_x,
                  calib_uncertainty=calib, beam_uncertain=beam > 0,
                  beam_err=(beam * (1 + ell[wedges[:-1]] / lmax)) if beam > 0 else None)
    return model, d, truth


def cl_problem(model: BinnedClModel, datasets: Sequence[ClDataset]):
    """What MCMC.MCMC needs of binnedClLikelihood + binnedClModel, as an oracle.refcpu.Problem."""
    from .refcpu import Problem
    return Problem(nparam=model.nparam, lnlike=lambda q: cl_lnlike(q, model, datasets),
                   prior=model.prior, package=lambda q: np.asarray(q, np.float64),
                   param_blocks=list(model.param_blocks), nblock=model.nblock)
