"""SED likelihoods with the interface of the reference's MCMC/submmSED/likelihood.py.

lnLike(params) is one call of the fused HIP kernel (model + Gaussian chi-square);
lnLike_batch(thetas) evaluates many parameter vectors per launch."""
import numpy as np

from .. import Likelihood
from ..Likelihood import ZeroPosterior  # noqa: F401
from ..engine import Engine, MODEL_IDS
from . import model as _model

dolog10 = False   # likelihood.py:14
dodet = False     # likelihood.py:15

_MARGINALISED = (MODEL_IDS["M1"], MODEL_IDS["M2"])


from .data import catalogue_arrays  # noqa: E402,F401  (packs ragged data sets for mi355_sed_load)


class _SEDLikelihood(Likelihood.Likelihood):
    """Common device plumbing.  A likelihood over a *list* of datasets means, as in the
    reference (Likelihood.py:99-105), the same parameters for every dataset and the sum of the
    per-dataset log-likelihoods."""

    def setdata(self, data):
        super().setdata(data)
        self._eng = None

    def setmodel_class(self, model):
        super().setmodel_class(model)
        self._eng = None

    problem_kind = "sed"

    def load_into(self, eng):
        """load this likelihood's data sets as the SED catalogue of `eng`"""
        nb, nu, dd, sg = catalogue_arrays(self.data)
        eng.sed_load(self.model.model_id, nb, nu, dd, sg, consts=_model.device_constants())

    def engine(self):
        if self._eng is None:
            eng = Engine()
            self.load_into(eng)
            self._eng = eng
        return self._eng

    def lnLike_batch(self, thetas, derived=False):
        """thetas [C, P] -> sum over datasets of lnLike, [C] (and ML amplitudes [S, C, 2])."""
        eng = self.engine()
        th = np.asarray(thetas, dtype=np.float64).reshape(-1, self.model.nparam)
        rep = np.broadcast_to(th[None], (eng.S,) + th.shape)
        out = eng.sed_lnlike(rep, derived=derived)
        if derived:
            return out[0].sum(axis=0), out[1]
        return out.sum(axis=0)

    def lnLike(self, params):                             # Likelihood.py:99-105
        th = np.asarray(self.model.unpackage(params), dtype=np.float64)
        marg = self.model.model_id in _MARGINALISED
        if marg:
            lnl, der = self.lnLike_batch(th[None], derived=True)
            amp = der[0, 0]
            self.MLamplitude = amp[0] if self.model.model_id == MODEL_IDS["M1"] else amp.copy()
            if self.model.model_id == MODEL_IDS["M2"] and np.isnan(lnl[0]):
                raise ZeroPosterior                        # singular 2x2: likelihood.py:64-65
        else:
            lnl = self.lnLike_batch(th[None])
            self.MLamplitude = None
        return float(lnl[0])

    def chi2(self, params):                               # Likelihood.py:117-123
        if self.model.model_id in _MARGINALISED:
            return super().chi2(params)
        return -2.0 * self.lnLike(params)                 # lnLike1 = -chi2/2 exactly


class SEDLikelihood2(_SEDLikelihood):
    """two greybodies, both amplitudes marginalised: lnL = 1/2 b^T M^-1 b.  likelihood.py:18-88"""
    nDerived = 2
    derivedTexNames = [r"$A_1$", r"$A_2$"]

    def getDerived(self, *params):
        return np.log10(self.MLamplitude) if dolog10 else self.MLamplitude

    def lnLike1(self, model_vals=None, data=None):        # likelihood.py:35-74
        """one data set, model values given ([n, 2]: the two un-normalised greybodies): the
        quadratic forms on the device (mi355_gauss_quadform), the 2x2 solve here (the reference
        calls scipy's solve with a keyword SciPy no longer has and raises TypeError)"""
        FNiF = data.quadform(model_vals)
        FNid = data.quadform(model_vals, data.d)
        try:
            self.MLamplitude = np.linalg.solve(FNiF, FNid)
        except np.linalg.LinAlgError:
            raise ZeroPosterior
        return 0.5 * np.dot(FNid.transpose(), self.MLamplitude)


class SEDLikelihood1(_SEDLikelihood):
    """one greybody, amplitude marginalised: lnL = 1/2 (F N^-1 d)^2 / (F N^-1 F).
    likelihood.py:93-123"""
    nDerived = 1
    derivedTexNames = [r"$A$"]

    def getDerived(self, *params):
        return [np.log10(self.MLamplitude)] if dolog10 else [self.MLamplitude]

    def lnLike1(self, model_vals=None, data=None):        # likelihood.py:99-110
        FNiF = data.quadform(model_vals)
        FNid = data.quadform(model_vals, data.d)
        self.MLamplitude = FNid / FNiF
        return 0.5 * FNid * self.MLamplitude


class SEDLikelihood_normalized(_SEDLikelihood):
    """explicit amplitudes: lnL = -chi2/2.  likelihood.py:128-164"""
    nDerived = 0

    def lnLike1(self, model_vals=None, data=None):        # likelihood.py:141-155
        return -0.5 * data.chi2(model_vals)

    def lnNorm1(self, data=None):                         # likelihood.py:158-164
        return -0.5 * (data.n * np.log(2 * np.pi) + data.lnDetN)
