"""Base of the beam models: an elliptical 2-D Gaussian given by centre, (sigma_major, sigma_minor)
and position angle, with the class-level prior state (centre box, sigma limits) that
ProcessBeamData.setup_sampler fills in.  Interface of the reference's GaussianBeamModel2D
(MCMC/BeamFit/BeamModel.py:32-225) as far as OffsetNormalizedBeamModel - the class the samplers
use - and its driver need it; beam values and the prior are evaluated by the HIP library.  The
(sigma_x, sigma_y, rho) parameterisation and NormalizedBeamModel of the reference are not
rebuilt: nothing on the likelihood path uses them."""
import math

import numpy as np

from .. import Proposal
from ..engine import default_engine


def _coordinates(first, second):
    """(xs, ys) from a data set with .x / .y, or from one or two coordinate sequences"""
    if hasattr(first, "x") and hasattr(first, "y"):
        return np.asarray(first.x, float), np.asarray(first.y, float)
    return np.asarray(first, float), np.asarray(first if second is None else second, float)


def inverse_covariance(sigmas, angle):
    """packed (xx, xy, yy) inverse covariance of the ellipse: R diag(1/s1^2, 1/s2^2) R^T in the
    reference's angle convention (BeamModel.py:55-71 with :128-136 give the same three numbers
    through sigma_x, sigma_y and rho)"""
    c, s = math.cos(angle), math.sin(angle)
    w1, w2 = 1.0 / (sigmas[0] * sigmas[0]), 1.0 / (sigmas[1] * sigmas[1])
    return np.array([c * c * w1 + s * s * w2, c * s * (w2 - w1), s * s * w1 + c * c * w2])


class GaussianBeamModel2D(object):
    centerMin = None                 # class-level prior state (BeamModel.py:41-43;
    centerMax = None                 # sigMin only appears in ProcessBeamData.py:150)
    sigMax = None
    sigMin = None

    nparam = 5
    fmtstring = "(%.3f %.3f) (%.3f %.3f) %.3f"
    paramBlocks = list(range(5))
    nBlock = 5
    texNames = [r"x", r"y", r"$\sigma_1$", r"$\sigma_2$", r"$\alpha$", r"$A$"]

    def __init__(self, center, sigmas, angle):
        self.center = tuple(center)
        self.sigmas = tuple(sigmas)
        self.angle = angle % math.pi
        self.Cinv = inverse_covariance(self.sigmas, angle)

    def atxy(self, x, y):                                            # :157-161
        return gauss2d(x, y, self.center[0], self.center[1], *self.Cinv)

    def at(self, data):                                              # :153-155
        return self.atxy(data.x, data.y)

    __call__ = at

    @classmethod
    def prior(cls, center, sigmas, angle):                           # :165-187
        """0 or 1: centre inside the box and sigMin <= sigmas <= sigMax, tested by the device on
        the flat vector (mi355_beam_prior_package)"""
        from .OffsetNormalizedBeamModel import OffsetNormalizedBeamModel as ONB
        flat = np.array([center[0], center[1], sigmas[0], sigmas[1], angle, 1.0, 0.0, 1.0, 0.0])
        return int(ONB._prior_engine_for(cls).beam_prior_package(flat)[0])

    @classmethod
    def setsigMax(cls, xs, ys=None):                                 # :75-86
        """largest extent of the coordinates along either axis"""
        xs, ys = _coordinates(xs, ys)
        cls.sigMax = max(np.ptp(xs), np.ptp(ys))

    @classmethod
    def setxyRange(cls, xs, ys=None, scale=None):                    # :88-116
        """centre box = bounding box of the coordinates, optionally scaled about its middle"""
        xs, ys = _coordinates(xs, ys)
        lo = np.array([xs.min(), ys.min()])
        hi = np.array([xs.max(), ys.max()])
        if scale is not None:
            mid, half = 0.5 * (hi + lo), 0.5 * (hi - lo)
            lo, hi = mid - scale * half, mid + scale * half
        cls.centerMin, cls.centerMax = (lo[0], lo[1]), (hi[0], hi[1])

    @classmethod
    def prior_state(cls):
        """what the device needs of the class-level prior (snapshot per load)"""
        return dict(center_min=cls.centerMin, center_max=cls.centerMax, sig_min=cls.sigMin,
                    sig_max=cls.sigMax)

    def unpackage(param_seqs):                                       # :190-193
        (x, y), (s1, s2), ang = param_seqs
        return np.array([x, y, s1, s2, ang % math.pi], dtype=np.float64)

    def package(params_flat):                                        # :195-198
        f = params_flat
        return ((f[0], f[1]), (f[2], f[3]), f[4] % math.pi)

    proposal = Proposal.GenericGaussianProposal(package=package, unpackage=unpackage)
    unpackage = staticmethod(unpackage)
    package = staticmethod(package)


def gauss2d(x, y, x0, y0, Cinv_xx, Cinv_xy, Cinv_yy):               # :288-290
    """exp(-(dx^2 Cxx + dy^2 Cyy + 2 dx dy Cxy)/2) on the device (mi355_beam_gauss2d)"""
    xa = np.asarray(x, np.float64)
    out = default_engine().beam_gauss2d(np.atleast_1d(xa), np.atleast_1d(np.asarray(y, np.float64)),
                                        x0, y0, Cinv_xx, Cinv_xy, Cinv_yy)
    return out if xa.ndim else float(out[0])
