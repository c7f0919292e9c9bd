"""Gaussian proposal with the interface of the reference's MCMC/Proposal.py:18-204.

In this implementation the draws happen inside the MH kernel (Philox4x32-10 + Box-Muller); a
proposal object is the host-side description of the density - widths, optional Cholesky factor
of the normalised covariance, rotation flag - that MCMC hands to the device.  getNewParams is
kept for API compatibility and is itself served by the device generator."""
import os

import numpy as np


def GenericGaussianProposal(package=None, unpackage=None):
    """Class factory, Proposal.py:18-200: returns a proposal *class* bound to a model's
    package / unpackage functions."""
    _package, _unpackage = package, unpackage

    class GenericGaussianClass(object):
        def __init__(self, sigmas=None, package=_package, unpackage=_unpackage, matrix=None,
                     isSqrt=False, rotateParams=False):
            self.package = package if package is not None else (lambda x: x)
            self.unpackage = unpackage if unpackage is not None else (lambda x: x)
            self.sqrtMatrix = None
            if sigmas is not None:
                self.setSigmas(sigmas)
            if matrix is not None:
                if not isSqrt:
                    self.setNormalizedMatrix(matrix)
                else:
                    self.setNormalizedMatrixFromSqrt(matrix)
            self.rotateParams = rotateParams
            # the reference reseeds from OS entropy per instance (Proposal.py:64): every instance
            # gets its own window of the device generator's stream
            self._draws = int.from_bytes(os.urandom(4), "little") << 20

        def setSigmas(self, sigmas):                      # Proposal.py:114-117
            self.sigmas = np.asarray(self.unpackage(sigmas), dtype=np.float64)
            self.n = len(self.sigmas)
            self.subset = np.where(self.sigmas > 0)[0]

        def getSigmas(self):
            return self.package(self.sigmas)

        def setNormalizedMatrix(self, matrix):            # Proposal.py:81-101
            """Cholesky factor of the normalised covariance restricted to the parameters with
            sigma>0 (np.linalg.LinAlgError propagates to the caller, as in the reference)."""
            sub = self.subset
            mat2 = np.asarray(matrix, dtype=np.float64)[np.ix_(sub, sub)]
            L = np.linalg.cholesky(mat2)
            self.sqrtMatrix = np.zeros((self.n, self.n))
            self.sqrtMatrix[np.ix_(sub, sub)] = L

        def setNormalizedMatrixFromSqrt(self, matrix):    # Proposal.py:106-109
            self.sqrtMatrix = np.asarray(matrix, dtype=np.float64)

        def getNewParams(self, prevParams, block=None):   # Proposal.py:127-164
            from .engine import default_engine
            eng = default_engine()
            n = self.n
            z, _ = eng.rng_dump(self._draws, 1, 1, n)     # n device normals for this call
            self._draws += 1
            z = z[0, 0]
            new = np.array(self.unpackage(prevParams), dtype=np.float64, copy=True)
            if self.rotateParams:
                offset = sum(self.sqrtMatrix[:, j] * z[i] for i, j in enumerate(block))
                new += offset * self.sigmas
            else:
                offset = (z if self.sqrtMatrix is None else self.sqrtMatrix @ z) * self.sigmas
                if block is None:
                    new += offset
                else:
                    new[block] += offset[block]
            return self.package(new)

        def densityRatio(self, old, new):                 # symmetric: Proposal.py:184-190
            return 1

        def lndensityRatio(self, old, new):               # Proposal.py:192-198
            return 0

    return GenericGaussianClass


def makeProposal(proposal=GenericGaussianProposal, package=None, unpackage=None):
    return proposal(package=package, unpackage=unpackage)
