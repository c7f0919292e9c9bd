"""Sub-mm SED greybody models with the class interface of the reference's
MCMC/submmSED/model.py (a model is a *class*: nparam, paramBlocks, nBlock, texNames, fmtstring,
prior(), package(), unpackage(), proposal, startfrom(); instances are callable on a data object).

Everything numerical - flux, prior box, T1<=T2 canonicalisation - is evaluated by the HIP
library; the module-level constants below (the reference's "compile-time" flags,
model.py:13,55-61) are captured into the device handle when a catalogue is loaded."""
import numpy as np

from .. import Proposal
from ..engine import Engine, MODEL_IDS

speed_of_light = 299792.458   # micron GHz      model.py:13
singleParams = True            # model.py:15

h_over_k = 0.04799237          # K/GHz           model.py:55
prefac = 1.0e-9                # model.py:56
nu_b = 1000.0                  # GHz             model.py:57
minTemp, maxTemp = 3.0, 100.0  # model.py:59
minb, maxb = 0., 3.            # model.py:61


def device_constants():
    """The module constants in the layout of mi355_sed_model (captured at handle creation)."""
    return dict(h_over_k=h_over_k, nu_b=nu_b, prefac=prefac, Tmin=minTemp, Tmax=maxTemp,
                bmin=minb, bmax=maxb)


_ENGINES = {}


def _model_engine(model_id):
    """Engine with a one-band dummy catalogue, for prior / package / flux calls that need no
    data.  Re-created when the module constants change (class-level state must not go stale)."""
    key = (model_id,) + tuple(sorted(device_constants().items()))
    eng = _ENGINES.get(key)
    if eng is None:
        while len(_ENGINES) >= 16:                          # (distinct constants are rare: bounded anyway)
            _ENGINES.pop(next(iter(_ENGINES))).close()
        eng = Engine()
        one = np.ones((1, 1))
        eng.sed_load(model_id, np.ones(1, np.int32), one * nu_b, one, one, consts=device_constants())
        _ENGINES[key] = eng
    return eng


def _flux_engine(model_id, nu_norm):
    """Engine for the module-level Planck functions: the normalisation frequency of a call is the
    nu_b of its device constants (model.py:107-109 normalises at the argument nu_norm)."""
    global nu_b
    if not nu_norm:
        return _model_engine(model_id)
    saved = nu_b
    try:
        nu_b = float(nu_norm)
        return _model_engine(model_id)
    finally:
        nu_b = saved


def greybody(beta, T, nu, nu_norm=False):
    """Greybody flux at frequency nu [GHz] for temperature T [K], evaluated by the HIP library
    (mi355_sed_flux): (nu/nu_norm)^(3+beta) expm1(h nu_norm/kT)/expm1(h nu/kT) when nu_norm is
    given, prefac nu_b^-beta nu^(3+beta)/expm1(h nu/kT) otherwise; an overflowing expm1 gives 0
    as in the reference.  model.py:97-111"""
    nu_arr = np.asarray(nu, dtype=np.float64)
    if nu_norm:
        out = _flux_engine("M1N", nu_norm).sed_flux(np.array([1.0, beta, T]), nu_arr.reshape(-1))
    else:
        out = _flux_engine("M1", False).sed_flux(np.array([beta, T], np.float64), nu_arr.reshape(-1))
    return out.reshape(nu_arr.shape) if nu_arr.ndim else float(out[0])


def blackbody(T, nu, nu_norm=False):
    """Blackbody flux: the greybody with beta = 0 (nu_b^-0 = 1 and the exponent 3+0 are exact, so
    this is model.py:80-94 operation for operation)."""
    return greybody(0.0, T, nu, nu_norm=nu_norm)


def _greybody_host(beta, T, nu, nu_norm=False):
    """Integrand of totalflux only (post-processing of a finished fit, adaptive quadrature on the
    host; the likelihood path never comes here).  model.py:97-111"""
    x = h_over_k * nu / T
    with np.errstate(over="ignore"):
        if nu_norm:
            return (nu / nu_norm) ** (3 + beta) * np.expm1(h_over_k * nu_norm / T) / np.expm1(x)
        return prefac * nu_b ** (-beta) * nu ** (3 + beta) / np.expm1(x)


def totalflux(beta, T, nu1=None, nu2=None, nu_norm=False):
    """Total flux of a greybody over (nu1, nu2) in GHz, or the analytic bolometric value of the
    un-normalised greybody when no limits are given.  model.py:113-127"""
    from scipy import integrate, special
    if nu1 is None and nu2 is None:
        return prefac * nu_b ** (-beta) * (T / h_over_k) ** (4 + beta) * \
            special.gamma(4 + beta) * special.zeta(4 + beta, 1)
    return integrate.quad(lambda nu: _greybody_host(beta, T, nu, nu_norm=nu_norm), nu1, nu2)[0]


def startfrom_generic(start, stds, posidx=(), random=True):   # model.py:63-64
    return start


def _blocks(nparam):
    return list(range(nparam)) if singleParams else [0] * nparam


class _SedModel(object):
    """Shared machinery; the concrete classes below carry the reference's names."""
    model_id = None
    nparam = 0
    _start = ()
    _stds = ()
    _posidx = ()

    def __init__(self, *params):
        if len(params) != self.nparam:
            raise TypeError("%s takes %d parameters" % (type(self).__name__, self.nparam))
        self.params = np.asarray(params, dtype=np.float64)

    def at_nu(self, nu):
        nu_arr = np.asarray(nu, dtype=np.float64)
        out = _model_engine(self.model_id).sed_flux(self.params, nu_arr.reshape(-1))
        if out.ndim == 2:                                   # submmModel2: [nnu, 2]
            return out.reshape(nu_arr.shape + (2,)) if nu_arr.ndim else out[0]
        return out.reshape(nu_arr.shape) if nu_arr.ndim else float(out[0])

    def at(self, data):
        return self.at_nu(data.freq)

    __call__ = at

    @classmethod
    def prior(cls, *params):
        pr, _ = _model_engine(cls.model_id).sed_prior_package(np.asarray(params, np.float64))
        return float(pr)

    @staticmethod
    def unpackage(param_seqs):
        return np.asarray(param_seqs, dtype=np.float64)

    @classmethod
    def startfrom(cls, data=None, random=None):
        cls.start_stds = cls._stds
        cls.start_params = startfrom_generic(cls._start, cls._stds, cls._posidx, random=random)
        return cls.start_params


def _make_package(model_id):
    def package(params_flat):
        _, pk = _model_engine(model_id).sed_prior_package(np.asarray(params_flat, np.float64))
        return pk
    return package


def _finish(cls):
    """Attach nBlock / package / proposal the way the reference class bodies do."""
    cls.paramBlocks = _blocks(cls.nparam)
    cls.nBlock = max(cls.paramBlocks) + 1
    cls.fmtstring = "%.3f " * cls.nparam
    pk = _make_package(cls.model_id)
    cls.proposal = Proposal.GenericGaussianProposal(package=pk, unpackage=_SedModel.unpackage)
    cls.package = staticmethod(pk)
    return cls


@_finish
class submmModel2(_SedModel):
    """two un-normalised greybodies, amplitudes marginalised (SEDLikelihood2).  model.py:130-215"""
    model_id = MODEL_IDS["M2"]
    nparam = 4
    texNames = [r"$\beta_1$", r"$T_1$", r"$\beta_2$", r"$T_2$"]
    _start, _stds, _posidx = (2., 20., 2., 30.), (0.5, 6.0, 0.5, 4.0), (1, 3)

    def __init__(self, b1, T1, b2, T2):
        super().__init__(b1, T1, b2, T2)
        self.b1, self.T1, self.b2, self.T2 = b1, T1, b2, T2

    def at_nu(self, nu):                                  # [2, nnu], model.py:176-178
        return super().at_nu(nu).T

    def at(self, data):                                   # [nnu, 2], model.py:180-182
        return super().at_nu(data.freq)

    __call__ = at


@_finish
class submmModel1(_SedModel):
    """one un-normalised greybody, amplitude marginalised (SEDLikelihood1).  model.py:218-279"""
    model_id = MODEL_IDS["M1"]
    nparam = 2
    texNames = [r"$\beta$", "$T$"]
    _start, _stds, _posidx = (2.0, 10.0), (0.5, 3.0), (1,)

    def __init__(self, b, T):
        super().__init__(b, T)
        self.b, self.T = b, T


@_finish
class submmModel2_normalized(_SedModel):
    """A1 g(b1,T1) + A2 g(b2,T2), normalised at nu_b.  model.py:363-468"""
    model_id = MODEL_IDS["M2N"]
    nparam = 6
    texNames = [r"$A_1$", r"$\beta_1$", r"$T_1$", r"$A_2$", r"$\beta_2$", r"$T_2$"]
    _start, _stds = (1., 2., 5., 1., 2., 10.), (0.25, 0.5, 5.0, 0.25, 0.5, 5.0)
    _posidx = (0, 2, 3, 5)

    def __init__(self, A1, b1, T1, A2, b2, T2):
        super().__init__(A1, b1, T1, A2, b2, T2)
        self.A1, self.b1, self.T1, self.A2, self.b2, self.T2 = A1, b1, T1, A2, b2, T2

    def flux(self, nu1=None, nu2=None):                   # model.py:401-404
        """flux between the given frequencies for each component"""
        return np.asarray([self.A1 * totalflux(self.b1, self.T1, nu1, nu2, nu_norm=nu_b),
                           self.A2 * totalflux(self.b2, self.T2, nu1, nu2, nu_norm=nu_b)])


@_finish
class submmModel2_normalized_logA(_SedModel):
    """10**logA1 g(b1,T1) + 10**logA2 g(b2,T2).  model.py:470-526"""
    model_id = MODEL_IDS["M2N_LOGA"]
    nparam = 6
    texNames = [r"$\log A_1$", r"$\beta_1$", r"$T_1$", r"$\log A_2$", r"$\beta_2$", r"$T_2$"]
    _start, _stds = submmModel2_normalized._start, submmModel2_normalized._stds
    _posidx = (0, 2, 3, 5)

    def __init__(self, logA1, b1, T1, logA2, b2, T2):
        super().__init__(logA1, b1, T1, logA2, b2, T2)
        self.A1, self.A2 = 10.0 ** logA1, 10.0 ** logA2
        self.b1, self.T1, self.b2, self.T2 = b1, T1, b2, T2

    flux = submmModel2_normalized.flux                    # inherited in the reference


@_finish
class submmModel1_normalized(_SedModel):
    """A g(b,T).  model.py:529-616"""
    model_id = MODEL_IDS["M1N"]
    nparam = 3
    texNames = [r"$A$", r"$\beta$", r"$T$"]
    _start, _stds, _posidx = (1., 2., 10.), (0.25, 0.5, 3.0), (0, 2)

    def __init__(self, A, b, T):
        super().__init__(A, b, T)
        self.A, self.b, self.T = A, b, T

    def flux(self, nu1=None, nu2=None):                   # model.py:555-557
        """As the reference: integrates the UN-normalised greybody (no nu_norm), although the
        amplitude is defined at nu_b ("at least one of these is printed incorrectly",
        driver.py:526)."""
        return np.array([self.A * totalflux(self.b, self.T, nu1, nu2)])


@_finish
class submmModel1_normalized_logA(_SedModel):
    """10**logA g(b,T).  model.py:748-769"""
    model_id = MODEL_IDS["M1N_LOGA"]
    nparam = 3
    texNames = [r"$\log A$", r"$\beta$", r"$T$"]
    _start, _stds, _posidx = (1., 2., 10.), (0.25, 0.5, 3.0), (0, 2)

    def __init__(self, logA, b, T):
        super().__init__(logA, b, T)
        self.A, self.b, self.T = 10.0 ** logA, b, T

    flux = submmModel1_normalized.flux                    # inherited in the reference


@_finish
class submmModel1_opticallythick(_SedModel):
    """A (1-exp(-tau))/(1-exp(-tau_b)) B_nu(T), tau=(nu/nu_0/1000)**b.  model.py:620-705"""
    model_id = MODEL_IDS["THICK"]
    nparam = 4
    texNames = [r"$A$", r"$\beta$", r"$T$", r"$\nu_0$"]
    nu_rescale = 1000.0
    nu_renorm = True
    _start, _stds, _posidx = (1., 2., 10., 1.0), (0.25, 0.5, 3.0, 0.3), (0, 2, 3)

    def __init__(self, A, b, T, nu_0):
        super().__init__(A, b, T, nu_0)
        self.A, self.b, self.T, self.nu_0 = A, b, T, nu_0

    def flux(self, nu1=None, nu2=None):                   # model.py:669-671 (optically thin integrand)
        return np.array([self.A * totalflux(self.b, self.T, nu1, nu2, nu_norm=nu_b)])


@_finish
class submmModel1_opticallythick_logA(_SedModel):
    """model.py:709-746.  As in the reference, this class derives its at() from the plain
    normalised greybody (nu_0 enters only through the prior exp(-3 nu_0))."""
    model_id = MODEL_IDS["THICK_LOGA"]
    nparam = 4
    texNames = [r"log $A$", r"$\beta$", r"$T$", r"$\nu_0$"]
    nu_rescale = 1000.0
    nu_renorm = True
    _start, _stds, _posidx = (1., 2., 10.), (0.25, 0.5, 3.0), (0, 2)   # inherited in the reference

    def __init__(self, logA, b, T, nu_0):
        super().__init__(logA, b, T, nu_0)
        self.A, self.b, self.T, self.nu_0 = 10.0 ** logA, b, T, nu_0

    flux = submmModel1_normalized.flux                    # derives from submmModel1_normalized


MODEL_CLASSES = [submmModel1_normalized, submmModel1_normalized_logA, submmModel2_normalized,
                 submmModel2_normalized_logA, submmModel1, submmModel2,
                 submmModel1_opticallythick, submmModel1_opticallythick_logA]
