"""Normalised 2-D Gaussian beam with offset and gradient plane, interface of the reference's
OffsetNormalizedBeamModel (MCMC/BeamFit/OffsetNormalizedBeamModel.py:31-152).  Parameters are the
reference's structured tuple ((x, y), (sigma_1, sigma_2), angle, amplitude, (offset, mu, phi));
model values, prior and package arithmetic are evaluated by libmi355mcmc.so."""
import math

import numpy as np

from .. import Proposal
from ..engine import default_engine
from .BeamModel import GaussianBeamModel2D as G2D

use_xy = False                                                       # :22


class OffsetNormalizedBeamModel(G2D):
    problem_kind = "beam"
    fmtstring = G2D.fmtstring + " %.3f (%.3f %.3f %.3f)"
    locBlk = [0, 1, 2, 3]
    paramBlocks = (G2D.paramBlocks + [G2D.nBlock + l for l in locBlk])   # :45-47
    nBlock = max(paramBlocks) + 1
    texNames = G2D.texNames + [r"$O$", r"$\mu$", r"$\phi$"]
    nparam = 9

    def __init__(self, center, sigmas, angle, amplitude, plane, round=None):   # :53-76
        self.center, self.sigmas, self.angle = center, sigmas, angle % math.pi
        self.amplitude = self.norm = amplitude
        self.offset, self.offset_mu, self.offset_phi = plane
        self._flat = np.array([center[0], center[1], sigmas[0], sigmas[1], angle, amplitude,
                               plane[0], plane[1], plane[2]], dtype=np.float64)

    def at(self, data):                                              # :78-82
        return self.atxy(data.x, data.y)

    def atxy(self, x, y):                                            # :84-88
        x = np.asarray(x, np.float64)
        return default_engine().beam_model(self._flat, x, y).reshape(x.shape)

    __call__ = at

    _prior_eng = None
    _prior_key = None

    @classmethod
    def _prior_engine(cls):
        """engine holding a snapshot of the class-level prior state (reloaded when it changes)"""
        return OffsetNormalizedBeamModel._prior_engine_for(cls)

    _prior_engines = {}

    @staticmethod
    def _prior_engine_for(cls):
        """the same for any class of the GaussianBeamModel2D family (each has its own class-level
        prior state)"""
        from ..engine import Engine
        key = str(cls.prior_state())
        have = OffsetNormalizedBeamModel._prior_engines.get(cls)
        if have is None or have[1] != key:
            if have is not None:
                have[0].close()
            eng = Engine()
            eng.beam_load(np.zeros(1), np.zeros(1), np.zeros((1, 1)), 1.0, **cls.prior_state())
            have = OffsetNormalizedBeamModel._prior_engines[cls] = (eng, key)
        return have[0]

    @classmethod
    def prior(cls, center, sigmas, angle, amplitude, plane):         # :93-101, BeamModel.py:165-187
        flat = np.array([center[0], center[1], sigmas[0], sigmas[1], angle, amplitude, plane[0],
                         plane[1], plane[2]], dtype=np.float64)
        return int(cls._prior_engine().beam_prior_package(flat)[0])

    def unpackage(param_seqs):                                       # :123-127
        xy, sig12, ang, amp, plane = param_seqs
        return np.array([xy[0], xy[1], sig12[0], sig12[1], ang % math.pi, amp, plane[0],
                         np.fabs(plane[1]), plane[2] % (2 * math.pi)])

    def package(params_flat):                                        # :129-133
        return (tuple(params_flat[0:2]), tuple(params_flat[2:4]), params_flat[4] % math.pi,
                params_flat[5], (params_flat[6], np.fabs(params_flat[7]),
                                 params_flat[8] % (2 * math.pi)))

    proposal = Proposal.GenericGaussianProposal(package=package, unpackage=unpackage)
    unpackage = staticmethod(unpackage)
    package = staticmethod(package)


def planexy(mu, phi, x, y):                                          # :146-152
    """z coordinate of a plane with normal vector (mu = cos(theta), phi) at (x, y).  Host-side
    helper with the reference's operation order; inside the likelihood the plane term is part of
    the beam kernels (csrc/beam_device.hpp)."""
    if mu == 1:
        return 0
    return np.sqrt(1 / mu / mu - 1) * (x * np.cos(phi) + y * np.sin(phi))


def plane(mu, phi, data):                                            # :156-167 ("Not used!")
    if mu == 1:
        return 0
    return np.sqrt(1 - mu * mu) / mu * (data.x * np.cos(phi) + data.y * np.sin(phi))
