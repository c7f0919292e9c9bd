"""A 2-D Gaussian beam with an amplitude, interface of the reference's NormalizedBeamModel
(MCMC/BeamFit/NormalizedBeamModel.py:24-95): parameters ((x, y), (sigma_1, sigma_2), angle,
amplitude); model values = amplitude * gauss2d on the device.  (The samplers of the reference use
OffsetNormalizedBeamModel, which adds the offset and the gradient plane.)"""
import math

import numpy as np

from .. import Proposal
from .BeamModel import GaussianBeamModel2D as G2D

use_xy = False                                                       # :17


class NormalizedBeamModel(G2D):
    fmtstring = G2D.fmtstring + " %.3f"
    nparam = 6

    def __init__(self, center, sigmas, angle, amplitude):            # :36-46
        super(NormalizedBeamModel, self).__init__(center, sigmas, angle)
        self.amplitude = amplitude
        self.norm = amplitude                                        # (the reference's "CHECK THIS")

    def at(self, data):                                              # :48-49
        return self.norm * G2D.at(self, data)

    def atxy(self, x, y):                                            # :51-52
        return self.norm * G2D.atxy(self, x, y)

    __call__ = at

    @classmethod
    def prior(cls, center, sigmas, angle, amplitude):                # :56-63: any amplitude
        return super(NormalizedBeamModel, cls).prior(center, sigmas, angle)

    # The reference's use_xy = False branch (:78-87) reads an undefined name in unpackage and
    # does not reduce the angle; these are the (sigma_1, sigma_2, angle) forms its use_xy = True
    # branch spells out (:67-76), which is what the five-parameter base class does too.
    def unpackage(param_seqs):
        xy, sig12, ang, amp = param_seqs
        return np.array([xy[0], xy[1], sig12[0], sig12[1], ang % math.pi, amp])

    def package(params_flat):
        return (tuple(params_flat[0:2]), tuple(params_flat[2:4]), params_flat[4] % math.pi,
                params_flat[5])

    proposal = Proposal.GenericGaussianProposal(package=package, unpackage=unpackage)
    unpackage = staticmethod(unpackage)
    package = staticmethod(package)
