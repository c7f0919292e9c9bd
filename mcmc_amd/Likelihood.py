"""Likelihood base class with the interface of the reference's MCMC/Likelihood.py:14-175.

Protocol kept from the reference, because user model classes are written against it:
  * the model is a *class*; `model(*params)` builds an instance, calling the instance with a
    dataset returns the model values on that dataset's grid;
  * the data are a list of dataset objects with `quadform`, `chi2`, `d`, `n`, `lnDetN`
    (GaussianData.py); a single dataset is wrapped into a list of one;
  * the amplitude of the model is marginalised analytically, and its maximum-likelihood value
    for the last dataset evaluated is left in `MLamplitude`.
The dataset objects of this package evaluate their quadratic forms on the GPU; the built-in
SED / C_l / beam likelihoods override lnLike with one fused kernel call instead."""
import math


class ZeroPosterior(Exception):
    """raised where the likelihood (or prior) vanishes; carries the offending value(s)"""

    def __init__(self, *value):
        self.value = value

    def __str__(self):
        return repr(self.value)


def isiterable(obj):
    try:
        iter(obj)
    except TypeError:
        return False
    return True


def _as_list(data):
    if data is None:
        return None
    return list(data) if isiterable(data) else [data]


class Likelihood(object):
    nDerived = 0

    def __init__(self, data=None, model=None):
        self.setdata(data)
        self.setmodel_class(model)

    # ---- configuration (Likelihood.py:50-68) ---------------------------------------------------
    def setmodel_class(self, model):
        self.model, self.this_model, self.model_vals = model, None, None

    def setdata(self, data):
        self.data = _as_list(data)
        self.model_vals = None

    def adddata(self, dataset):
        # the reference's first branch names an undefined variable (Likelihood.py:57); the intent,
        # starting the list with this dataset, is what is done here
        if self.data is None:
            self.setdata(dataset)
        else:
            self.data += _as_list(dataset)

    def setModel(self, params):                            # Likelihood.py:70-75
        inst = self.model(*params)
        self.this_model = inst
        self.model_vals = [inst(dataset) for dataset in self.data]
        self.MLamplitude = None

    # ---- one dataset ------------------------------------------------------------------------
    def lnLike1(self, model_vals=None, data=None):
        """ln p(d | params) with the amplitude A of the model F marginalised (flat prior), up to
        terms free of the parameters:  (F^T N^-1 d)^2 / (2 F^T N^-1 F) - ln(F^T N^-1 F) / 2
        (Likelihood.py:78-97)"""
        ff = data.quadform(model_vals)
        fd = data.quadform(model_vals, data.d)
        if ff <= 0:
            raise ZeroPosterior(ff)
        amp = fd / ff
        self.MLamplitude = amp
        return 0.5 * (fd * amp - math.log(ff))

    def chi2_1(self, model_vals=None, data=None):          # Likelihood.py:107-114
        return data.chi2(model_vals)

    def lnNorm1(self, data=None):
        """the parameter-free part: -[(n-1) ln 2 pi + ln det N + d^T N^-1 d] / 2
        (Likelihood.py:129-138)"""
        return -0.5 * ((data.n - 1) * math.log(2.0 * math.pi) + (data.lnDetN + data.chi2()))

    # ---- all datasets ---------------------------------------------------------------------------
    def _over_datasets(self, params, term):
        self.setModel(params)
        total = 0.0
        for vals, dataset in zip(self.model_vals, self.data):
            total = total + term(vals, dataset)
        return total

    def lnLike(self, params):                              # Likelihood.py:99-105
        return self._over_datasets(params, self.lnLike1)

    def chi2(self, params):                                # Likelihood.py:117-123
        return self._over_datasets(params, self.chi2_1)

    def __call__(self, params):
        return self.lnLike(params)

    def lnNorm(self):                                      # Likelihood.py:140-144
        total = 0.0
        for dataset in self.data:
            total = total + self.lnNorm1(dataset)
        return total

    def lnLikeFull(self, params):                          # Likelihood.py:146-150
        return self.lnLike(params) + self.lnNorm()

    def lnPost(self, params):                              # Likelihood.py:152-156
        return self.lnLike(params) + self.model.lnPrior(params)

    def lnPostFull(self, params):
        """Likelihood.py:158-163 takes no argument and names an undefined `params`; here the
        parameters are passed"""
        return self.lnPost(params) + self.lnNorm()
