"""Class-level state and helpers of the reference's GaussianBeamModel2D
(MCMC/BeamFit/BeamModel.py:32-227): prior box, sigma limits, parameter blocks.  The shape
arithmetic (sigma_1, sigma_2, angle -> packed C^-1, gauss2d) lives in the HIP kernels
(csrc/beam_kernels.hpp); OffsetNormalizedBeamModel is the class the samplers use."""
import math

import numpy as np

from .. import Proposal

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
