"""NumPy restatement of the reference's 2-D Gaussian beam fit.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Parity status: PINNED by
tests/golden/beam_g6.npz, produced by oracle/make_golden.py from the reference's
OffsetNormalizedBeamModel / NormalizedBeamLikelihood / BeamData classes.

Flat parameter vector (OffsetNormalizedBeamModel.unpackage, :123-127):
    (x, y, sigma1, sigma2, angle mod pi, amplitude, offset, |mu|, phi mod 2pi)
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

NPARAM = 9


def gauss2d(x, y, x0, y0, cxx, cxy, cyy):                   # BeamModel.py:288-290
    dx = x - x0
    dy = y - y0
    return np.exp(-0.5 * (dx * dx * cxx + dy * dy * cyy + 2 * dx * dy * cxy))


def cinv_majminang(sig1, sig2, angle):
    """setParameters_MajMinAng + set_Cinv, BeamModel.py:55-71,128-136 -> (Cxx, Cxy, Cyy)."""
    c = np.cos(angle)
    s = np.sin(angle)
    s12 = sig1 * sig1
    s22 = sig2 * sig2
    sig2x, sig2y = c * c * s12 + s * s * s22, s * s * s12 + c * c * s22
    rho = c * s * (s22 - s12) / np.sqrt(sig2x * sig2y)
    det = sig2x * sig2y * (1.0 - rho * rho)
    return np.array([sig2y, -rho * np.sqrt(sig2x * sig2y), sig2x], np.float64) / det


def planexy(mu, phi, x, y):                                 # OffsetNormalizedBeamModel.py:146-152
    if mu == 1:
        return 0
    return np.sqrt(1 / mu / mu - 1) * (x * np.cos(phi) + y * np.sin(phi))


def beam_package(flat):
    """package() then unpackage() of a flat vector (:123-133): angle mod pi, |mu|, phi mod 2pi."""
    f = np.array(flat, dtype=np.float64, copy=True)
    f[4] = f[4] % math.pi
    f[7] = np.fabs(f[7])
    f[8] = f[8] % (2 * math.pi)
    return f


def beam_model(flat, x, y):
    """OffsetNormalizedBeamModel(*package(flat)).at(data), :53-82.  Note the reference builds the
    shape from the *raw* angle (cos/sin of `angle`, BeamModel.py:64) - package has already reduced it."""
    x0, y0, s1, s2, ang, amp, off, mu, phi = flat
    cxx, cxy, cyy = cinv_majminang(s1, s2, ang)
    return amp * (gauss2d(x, y, x0, y0, cxx, cxy, cyy) + off + planexy(mu, phi, x - x0, y - y0))


@dataclass
class BeamPrior:
    """Class-level prior state (BeamModel.py:41-43 + ProcessBeamData.py:149-151)."""
    center_min: Tuple[float, float]
    center_max: Tuple[float, float]
    sig_min: float
    sig_max: float


def beam_prior(flat, pr: BeamPrior):
    """OffsetNormalizedBeamModel.prior -> GaussianBeamModel2D.prior (:93-101, BeamModel.py:165-187)."""
    x0, y0, s1, s2, ang, amp, off, mu, phi = flat
    if amp <= 0:
        return 0
    if mu < -1 or mu > 1:
        return 0
    if (x0 < pr.center_min[0] or y0 < pr.center_min[1] or x0 > pr.center_max[0] or
            y0 > pr.center_max[1]):
        return 0
    if s1 < pr.sig_min or s2 < pr.sig_min:
        return 0
    if max(s1, s2) > pr.sig_max:
        return 0
    return 1


@dataclass
class BeamMap:
    x: np.ndarray
    y: np.ndarray
    d: np.ndarray
    sig: np.ndarray          # scalar (size 1) or per pixel

    def chi2(self, vec):                                     # GaussianData.py:47,64
        r = self.d - vec
        sig2 = self.sig * self.sig
        return np.dot(r.transpose() / sig2, r)


def beam_lnlike(flat, m: BeamMap) -> float:
    """NormalizedBeamLikelihood.lnLike1 = -chi2/2, NormalizedBeamLikelihood.py:9-16."""
    with np.errstate(all="ignore"):
        return float(-0.5 * m.chi2(beam_model(flat, m.x, m.y)))


def synth_beam_maps(M, n=256, seed=3, noise=0.1, half=20.0):
    """Config C5: M maps of n x n pixels on (-half, half)^2."""
    rng = np.random.default_rng(seed)
    g = np.linspace(-half, half, n)
    xx, yy = np.meshgrid(g, g)
    x, y = xx.ravel(), yy.ravel()
    truth = np.empty((M, NPARAM))
    d = np.empty((M, n * n))
    for m in range(M):
        th = np.array([rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(3, 8), rng.uniform(3, 8),
                       rng.uniform(0, math.pi), 10.0, 0.01, 1.0, 0.0])
        truth[m] = th
        d[m] = beam_model(th, x, y) + noise * rng.standard_normal(n * n)
    prior = BeamPrior((-half, -half), (half, half), 3.0, 8.0 + 4.0)
    return x, y, d, noise, truth, prior


def beam_stats(m: BeamMap, sigcut=None, do_abs=False):
    """BeamData.stats, BeamData.py:35-71: data-weighted <x>, <y>, var x, var y.  (The reference
    subtracts the mean from self.d in place, :45-52, so its later likelihood calls see the shifted
    map; this function is pure - the product's BeamData.stats reproduces the shift, pinned by G12.)"""
    dd = np.abs(m.d) if do_abs else m.d.copy()
    dd = dd - dd.mean()
    if sigcut is not None:
        idx = np.nonzero(dd > sigcut * m.sig)
        dd, xx, yy = dd[idx], m.x[idx], m.y[idx]
    else:
        xx, yy = m.x, m.y
    norm = dd.sum()
    ex = (xx * dd).sum() / norm
    ey = (yy * dd).sum() / norm
    ex2 = (xx * xx * dd).sum() / norm
    ey2 = (yy * yy * dd).sum() / norm
    return ex, ey, ex2 - ex * ex, ey2 - ey * ey


def beam_problem(m: BeamMap, prior: BeamPrior, param_blocks=(0, 1, 2, 3, 4, 5, 6), nblock=7):
    """What MCMC.MCMC needs of NormalizedBeamLikelihood + OffsetNormalizedBeamModel
    (ProcessBeamData.py:139-141, :215-216), as an oracle.refcpu.Problem."""
    from .refcpu import Problem
    return Problem(nparam=NPARAM, lnlike=lambda th: beam_lnlike(th, m),
                   prior=lambda th: beam_prior(th, prior), package=beam_package,
                   param_blocks=list(param_blocks), nblock=nblock)
