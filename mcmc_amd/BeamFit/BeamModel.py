"""The reference's GaussianBeamModel2D / GaussianBeamModel2D_xy / gauss2d
(MCMC/BeamFit/BeamModel.py:32-290): an un-normalised 2-D Gaussian beam, parameterised by
(sigma_major, sigma_minor, angle) or (sigma_x, sigma_y, rho); class-level prior box and sigma
limits; parameter blocks.  Beam values (at / atxy / gauss2d) and the prior are evaluated by the
HIP library; the per-object scalars (sig2_xy, rho, the packed C^-1 of set_Cinv) are set on the
host in the reference's operation order.  OffsetNormalizedBeamModel is the class the samplers
use: there the whole shape arithmetic lives in the kernels (csrc/beam_kernels.hpp)."""
import math

import numpy as np
from numpy import cos, sin, sqrt

from .. import Proposal
from ..engine import default_engine

separate_sigma12 = False                                             # :26


class GaussianBeamModel2D(object):
    centerMin = None                                                 # :41-43
    centerMax = None
    sigMax = None
    sigMin = None        # the reference only defines it in setup_sampler (ProcessBeamData.py:150)

    nparam = 5
    fmtstring = "(%.3f %.3f) (%.3f %.3f) %.3f"
    paramBlocks = [0, 1, 2, 3, 4]
    nBlock = max(paramBlocks) + 1
    texNames = [r"x", r"y", r"$\sigma_1$", r"$\sigma_2$", r"$\alpha$", r"$A$"]

    def setParameters_MajMinAng(self, center, sigmas, angle):        # :55-71
        """set the parameters from x, y, sigma_major, sigma_minor, angle"""
        self.center = center
        self.sigmas = sigmas
        self.angle = angle % math.pi
        c = cos(angle)
        s = sin(angle)
        s12 = sigmas[0] * sigmas[0]
        s22 = sigmas[1] * sigmas[1]
        self.sig2_xy = (c * c * s12 + s * s * s22, s * s * s12 + c * c * s22)
        self.rho = c * s * (s22 - s12) / sqrt(self.sig2_xy[0] * self.sig2_xy[1])
        self.set_Cinv()

    __init__ = setParameters_MajMinAng

    def setParameters_XYRho(self, center, sigma_xy, rho):            # :120-126
        """set the parameters from x, y, sig_x, sig_y, rho = correlation coefficient"""
        self.center = center
        self.sigmas = None
        self.angle = None
        self.sig2_xy = np.array(sigma_xy, np.float64) ** 2
        self.rho = rho
        self.set_Cinv()

    def set_Cinv(self):                                              # :128-136
        """the packed inverse correlation matrix (xx, xy, yy)"""
        sig2x, sig2y = self.sig2_xy
        det = sig2x * sig2y * (1.0 - self.rho * self.rho)
        self.Cinv = np.array([sig2y, -self.rho * sqrt(sig2x * sig2y), sig2x], np.float64) / det

    def get_XYRho(self):                                             # :138-143 (sigma^2, not sigma)
        return self.center, self.sig2_xy, self.rho

    def get_MajMinAng(self):                                         # :145-151
        return self.center, self.sigmas, self.angle

    def at(self, data):                                              # :153-155
        return gauss2d(data.x, data.y, self.center[0], self.center[1], *self.Cinv)

    def atxy(self, x, y):                                            # :157-161
        return gauss2d(x, y, self.center[0], self.center[1], *self.Cinv)

    __call__ = at

    @classmethod
    def prior(cls, center, sigmas, angle):                           # :165-187
        """the unnormalised prior (0 or 1): centre inside the box, sigMin <= sigmas <= sigMax; the
        device evaluates the same tests on a flat vector (mi355_beam_prior_package)"""
        from .OffsetNormalizedBeamModel import OffsetNormalizedBeamModel as ONB
        eng = ONB._prior_engine_for(cls)
        flat = np.array([center[0], center[1], sigmas[0], sigmas[1], angle, 1.0, 0.0, 1.0, 0.0])
        return int(eng.beam_prior_package(flat)[0])

    def startfrom(self, data=None, random=None):                     # :205-225
        """a random starting point inside the class-level centre box when `random` is given (the
        reference draws it with an unimported `uniform` and drops it; here it is returned)"""
        if random is None:
            return None
        dx = (self.centerMin[0], self.centerMax[0])
        dy = (self.centerMin[1], self.centerMax[1])
        u = np.random.uniform
        return ((u(*dx), u(*dy)), (u(0, (dx[1] - dx[0]) / 5), u(0, (dy[1] - dy[0]) / 5)),
                self._start_last())

    @staticmethod
    def _start_last():
        return np.random.uniform(0, math.pi / 2)                     # the angle

    @classmethod
    def setsigMax(cls, xmax, ymax=None):                             # :75-86
        try:
            cls.sigMax = max(max(xmax.x) - min(xmax.x), max(xmax.y) - min(xmax.y))
        except (NameError, AttributeError):
            if ymax is None:
                ymax = xmax
            cls.sigMax = max(max(xmax) - min(xmax), max(ymax) - min(ymax))

    @classmethod
    def setxyRange(cls, xrng, yrng=None, scale=None):                # :88-116
        try:
            cls.centerMin = min(xrng.x), min(xrng.y)
            cls.centerMax = max(xrng.x), max(xrng.y)
        except (NameError, AttributeError):
            if yrng is None:
                yrng = xrng
            cls.centerMin = (min(xrng), min(yrng))
            cls.centerMax = (max(xrng), max(yrng))
        if scale is not None:
            xr = 0.5 * (cls.centerMax[0] - cls.centerMin[0])
            yr = 0.5 * (cls.centerMax[1] - cls.centerMin[1])
            x0 = 0.5 * (cls.centerMax[0] + cls.centerMin[0])
            y0 = 0.5 * (cls.centerMax[1] + cls.centerMin[1])
            cls.centerMin = x0 - scale * xr, y0 - scale * yr
            cls.centerMax = x0 + scale * xr, y0 + scale * yr

    @classmethod
    def prior_state(cls):
        """what the device needs of the class-level prior (snapshot per load)"""
        return dict(center_min=cls.centerMin, center_max=cls.centerMax, sig_min=cls.sigMin,
                    sig_max=cls.sigMax)

    def unpackage(param_seqs):                                       # :190-193
        xy, sig12, ang = param_seqs
        return np.array([xy[0], xy[1], sig12[0], sig12[1], ang % math.pi], dtype=np.float64)

    def package(params_flat):                                        # :195-198
        return (tuple(params_flat[0:2]), tuple(params_flat[2:4]), params_flat[4] % math.pi)

    proposal = Proposal.GenericGaussianProposal(package=package, unpackage=unpackage)
    unpackage = staticmethod(unpackage)
    package = staticmethod(package)


class GaussianBeamModel2D_xy(GaussianBeamModel2D):
    """like GaussianBeamModel2D, but with sig_x, sig_y, rho as parameters (:229-285)"""
    texNames = [r"x", r"y", r"$\sigma_x$", r"$\sigma_y$", r"$\rho$", r"$A$"]

    def __init__(self, center, sigmas, rho):
        super(GaussianBeamModel2D_xy, self).setParameters_XYRho(center, sigmas, rho)

    @classmethod
    def prior(cls, center, sigmas, rho):                             # :237-248
        if not super(GaussianBeamModel2D_xy, cls).prior(center, sigmas, 0):
            return 0
        if rho < -1 or rho > 1:
            return 0
        return 1

    @staticmethod
    def _start_last():
        return np.random.uniform(-1, 1)                              # rho (:280-282)

    def unpackage_xy(param_seqs):                                    # :251-254
        xy, sigxy, rho = param_seqs
        return np.array([xy[0], xy[1], sigxy[0], sigxy[1], rho], dtype=np.float64)

    def package_xy(params_flat):                                     # :256-258
        return (tuple(params_flat[0:2]), tuple(params_flat[2:4]), params_flat[4])

    proposal = Proposal.GenericGaussianProposal(package=package_xy, unpackage=unpackage_xy)
    unpackage = staticmethod(unpackage_xy)
    package = staticmethod(package_xy)


def gauss2d(x, y, x0, y0, Cinv_xx, Cinv_xy, Cinv_yy):               # :288-290
    """exp(-(dx^2 Cxx + dy^2 Cyy + 2 dx dy Cxy)/2) on the device (mi355_beam_gauss2d)"""
    xa = np.asarray(x, np.float64)
    out = default_engine().beam_gauss2d(np.atleast_1d(xa), np.atleast_1d(np.asarray(y, np.float64)), x0, y0,
                                        Cinv_xx, Cinv_xy, Cinv_yy)
    return out if xa.ndim else float(out[0])
