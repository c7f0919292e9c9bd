"""Gaussian data container with the interface of the reference's MCMC/GaussianData.py:7-72.
The quadratic forms run on the GPU (mi355_gauss_quadform); there is no host fallback."""
import numpy as np

from .engine import default_engine


class DataSizeError(Exception):
    def __init__(self, *value):
        self.value = value

    def __str__(self):
        return repr(self.value)


class GaussianData(object):
    """(d, sigma) with diagonal noise N = diag(sigma^2).  Reference: GaussianData.py:13-28."""

    def __init__(self, data, sigma):
        self.d = np.asarray(data, np.float64)
        self.n = self.d.size
        self.sig = np.asarray(sigma, np.float64)
        self.nsig = self.sig.size
        self.sig2 = self.sig * self.sig
        if self.nsig == self.n:
            self.lnDetN = (np.log(self.sig2)).sum()
        elif self.nsig == 1:
            self.lnDetN = self.n * np.log(self.sig2)
        else:
            raise DataSizeError(self.n, self.nsig)
        self.data_chi2 = None

    def quadform(self, A=None, B=None):
        """A^T N^-1 B, A^T N^-1 A or d^T N^-1 d.  Reference: GaussianData.py:31-53.
        A may be [n] or [n, k] (k model columns, as submmModel2 returns): the result is then a
        scalar, a [k] vector (B a vector) or a [k, k] matrix."""
        eng = default_engine()
        if A is None and B is None:
            if self.data_chi2 is None:
                self.data_chi2 = eng.gauss_quadform(self.d.ravel(), self.sig2)
            return self.data_chi2
        if A is None:
            raise NotImplementedError
        A = np.asarray(A, np.float64)
        if A.ndim == 1:
            return eng.gauss_quadform(A.ravel(), self.sig2, None if B is None else np.ravel(B))
        cols = np.ascontiguousarray(A.reshape(self.n, -1).T)        # [k, n]
        k = cols.shape[0]
        if B is None:
            out = np.empty((k, k))
            for j in range(k):
                out[:, j] = eng.gauss_quadform(cols, self.sig2, cols[j])
            return out
        return eng.gauss_quadform(cols, self.sig2, np.ravel(B))

    def chi2(self, vec=None):
        """(d - vec)^T N^-1 (d - vec).  Reference: GaussianData.py:55-64."""
        if vec is None:
            return self.quadform()
        return self.quadform(self.d - vec)
