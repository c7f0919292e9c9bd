"""Likelihood base class with the interface of the reference's MCMC/Likelihood.py:14-175.

The generic base keeps the reference's protocol for arbitrary user model classes (a model is a
*class*; `model(*params)(data)` returns model values; the amplitude is marginalised
analytically) with the quadratic forms on the GPU.  The built-in SED / C_l / beam likelihoods
override lnLike with one fused kernel call."""
import numpy as np


class ZeroPosterior(Exception):
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


class Likelihood(object):
    nDerived = 0

    def __init__(self, data=None, model=None):
        self.setdata(data)
        self.setmodel_class(model)

    def setmodel_class(self, model):                      # Likelihood.py:50-53
        self.model = model
        self.this_model = None
        self.model_vals = None

    def adddata(self, dataset):                           # Likelihood.py:55-61
        if self.data is None:
            self.setdata(dataset)
        elif isiterable(dataset):
            self.data.extend(dataset)
        else:
            self.data.append(dataset)

    def setdata(self, data):                              # Likelihood.py:63-68
        if data is None:
            self.data = None
        elif isiterable(data):
            self.data = list(data)
        else:
            self.data = [data]
        self.model_vals = None

    def setModel(self, params):                           # Likelihood.py:70-75
        self.this_model = self.model(*params)
        self.model_vals = [self.this_model(d) for d in self.data]
        self.MLamplitude = None

    def lnLike1(self, model_vals=None, data=None):        # Likelihood.py:78-97
        FNiF = data.quadform(model_vals)
        FNid = data.quadform(model_vals, data.d)
        if FNiF <= 0:
            raise ZeroPosterior(FNiF)
        self.MLamplitude = FNid / FNiF
        return 0.5 * (FNid * self.MLamplitude - np.log(FNiF))

    def lnLike(self, params):                             # Likelihood.py:99-105
        self.setModel(params)
        return sum([self.lnLike1(v, d) for (v, d) in zip(self.model_vals, self.data)])

    def chi2_1(self, model_vals=None, data=None):         # Likelihood.py:107-114
        return data.chi2(model_vals)

    def chi2(self, params):                               # Likelihood.py:117-123
        self.setModel(params)
        return sum([self.chi2_1(v, d) for (v, d) in zip(self.model_vals, self.data)])

    def __call__(self, params):
        return self.lnLike(params)

    def lnNorm1(self, data=None):                         # Likelihood.py:129-138
        norm = (data.n - 1) * np.log(2 * np.pi)
        norm2 = data.lnDetN + data.chi2()
        return -0.5 * (norm + norm2)

    def lnNorm(self):                                     # Likelihood.py:140-144
        return sum(map(self.lnNorm1, self.data))

    def lnLikeFull(self, params):                         # Likelihood.py:146-150
        return self.lnLike(params) + self.lnNorm()
