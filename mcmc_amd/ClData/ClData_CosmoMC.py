"""Band-power data set with the interface of the reference's ClData_CosmoMC
(MCMC/ClData/ClData_CosmoMC.py:51-446).  The object is the host-side description (the attributes
the reference's readDataset fills in); every method that computes - band powers, deltas, linear
and quadratic forms, chi-square, the calibration/beam marginalisation and calcLnLike - runs in
libmi355mcmc.so on the GPU.  The .dataset/.newdat/window file parsers of the reference are not
part of the hot path and are not rebuilt: construct the object and assign the arrays."""
import numpy as np

from ..engine import Engine

num_cls = 3      # TT, TE, EE                      ClData_CosmoMC.py:38
lmax = 3500      # :39
halfsteps = 5    # :40


def initNumericalMarge(halfsteps):
    """:42-47 (host mirror of the weights the device uses)."""
    step = np.arange(-halfsteps, halfsteps + 1, 1.0, dtype=np.float64)
    margeweights = np.exp(-(step * 3 / float(halfsteps)) ** 2 / 2)
    return sum(margeweights), margeweights


margenorm, margeweights = initNumericalMarge(halfsteps)


class MyErr(Exception):
    pass


class ClData_CosmoMC(object):
    _FIELDS = ("obs", "var", "N_inv", "xfactors", "has_xfactor", "window", "win_min", "win_max",
               "beam_err", "inc_pol", "num_points", "has_corr_errors", "has_xfactors",
               "calib_uncertainty", "beam_uncertain", "has_pol")

    def __init__(self, file=None, device=0, **attrs):
        if file is not None:
            raise NotImplementedError("the .dataset parser (ClData_CosmoMC.py:85-258) is host I/O "
                                      "outside the hot path: assign the arrays instead")
        self.name = "ClData"
        self.all_l_exact = False
        self.has_pol = False
        self.has_xfactors = False
        self.has_corr_errors = False
        self.calib_uncertainty = 0.0
        self.beam_uncertain = False
        self.beam_err = None
        self.inc_pol = None
        self._device = device
        for k, v in attrs.items():
            setattr(self, k, v)

    # any change to the data invalidates the device copy
    def __setattr__(self, name, value):
        object.__setattr__(self, name, value)
        if name in ClData_CosmoMC._FIELDS:
            object.__setattr__(self, "_eng", None)

    def _engine(self):
        eng = getattr(self, "_eng", None)
        if eng is None:
            if self.all_l_exact:
                raise MyErr("All_l_exact not implemented")          # :276-277
            eng = Engine(device=self._device)
            eng.cl_load([self])
            object.__setattr__(self, "_eng", eng)
        return eng

    # ---- reference methods -----------------------------------------------------------------
    def getWinBandpower(self, i, Cl):                                # :422-446
        Cl = np.asarray(Cl, dtype=np.float64)
        if Cl.ndim not in (1, 2):
            raise MyErr("Cl shape wrong in getWinBandpower:", Cl.shape)
        return float(self._engine().cl_bandpowers(0, Cl=Cl)[0, i])

    def bandpowers(self, Cl):
        """all bands at once (the list comprehension of :280)"""
        return self._engine().cl_bandpowers(0, Cl=np.asarray(Cl, dtype=np.float64))[0]

    def getdelta(self, BP):                                          # :377-398
        return self._engine().cl_data_op(0, 0, BP)

    def getlinform(self, vec):                                       # :407-412
        return self._engine().cl_data_op(0, 1, vec)

    def getquadform(self, vec):                                      # :415-420
        return self._engine().cl_data_op(0, 2, vec)

    def getchisq(self, BP):                                          # :400-405
        return self._engine().cl_data_op(0, 3, BP)

    def calcChisq(self, Cl):                                         # :262-265
        return self.getchisq(self.bandpowers(Cl))

    def getCalibMargexChisq(self, bandpowers):                       # :317-372
        if not (self.has_xfactors and (self.calib_uncertainty > 1.0e-4 or self.beam_uncertain)):
            raise MyErr("getCalibMargexChisq needs x-factors and a calibration or beam uncertainty")
        return 2 * self._engine().cl_data_op(0, 4, bandpowers)

    def calcLnLike(self, Cl):                                        # :267-315
        """-ln(likelihood(data | Cl)) (yes, negative), Cl [nCl, n_ell] or [n_ell];
        a leading batch axis evaluates many spectra in one launch."""
        return self._engine().cl_calc_lnlike(0, Cl)
