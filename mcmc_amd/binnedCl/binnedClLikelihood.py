"""Likelihood of binned C_l given band-power data sets, interface of the reference's
binnedClLikelihood (MCMC/binnedCl/binnedClLikelihood.py:7-32).  lnLike(q) is one device call:
band powers through the precomputed bin->band matrix, offset-lognormal deltas, the
full-covariance quadratic form on the FP64 matrix cores and the calibration/beam
marginalisation (ClData_CosmoMC.calcLnLike, :267-315), summed over the data sets."""
import numpy as np

from .. import Likelihood
from ..engine import Engine


class binnedClLikelihood(Likelihood.Likelihood):
    problem_kind = "cl"

    def __init__(self, data=None, model=None, device=0):
        self._device = device
        self._eng = None
        self._key = None
        super().__init__(data, model)

    def setdata(self, data):
        super().setdata(data)
        self._eng = None

    @property
    def engine(self):
        """Engine with this likelihood's data sets and the model's current binning loaded."""
        key = (id(self.model), getattr(self.model, "_version", 0), len(self.data))
        if self._eng is None or self._key != key:
            eng = Engine(device=self._device)
            self.load_into(eng)
            self._eng, self._key = eng, key
        return self._eng

    def load_into(self, eng):
        eng.cl_load(self.data)
        eng.cl_set_binning(self.model.bins, self.model.shapefun)

    def lnLike1(self, model_vals, data):                             # :12-14
        return -data.calcLnLike(model_vals)

    def lnNorm1(self, data=None):
        return 0.0

    def setModel(self, params):                                      # :21-24
        self.this_model = self.model(params)
        self.model_vals = self.this_model()

    def lnLike(self, params):                                        # :26-32
        """params [npar] -> float; [n, npar] -> array of n log-likelihoods (one launch)."""
        return self.engine.cl_lnlike(np.asarray(params, dtype=np.float64))

    def chi2(self, params):
        """-2 lnLike: the chi-square the summaries of getdist_ahj print.  (The reference inherits
        Likelihood.chi2, which asks each data set for a chi2() that ClData does not have, so its
        band-power runs end in AttributeError at that print, test_binnedCl.py:244-249.)"""
        return -2.0 * self.lnLike(params)
