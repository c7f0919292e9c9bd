"""Band-power data set with the interface of the reference's ClData_CosmoMC
(MCMC/ClData/ClData_CosmoMC.py:51-446).  The object is the host-side description (the attributes
the reference's readDataset fills in); every method that computes - band powers, deltas, linear
and quadratic forms, chi-square, the calibration/beam marginalisation and calcLnLike - runs in
libmi355mcmc.so on the GPU.  readDataset / readWindow are the host-side file parsers of the
CosmoMC `.dataset` format (ClData_CosmoMC.py:85-258); the `.newdat` format lives in ClData_bcp."""
import math
import operator
import os
from functools import reduce

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
    def __init__(self, *value):
        self.value = value

    def __str__(self):
        return repr(self.value)


class ParamSyntaxError(MyErr):
    pass


class ClDataError(MyErr):
    pass


class Nonint32Ell(MyErr):
    pass


class ArraySizeError(MyErr):
    pass


def parseline(line):
    """`param = Value With Whitespace` -> ('param', 'Value With Whitespace'); (None, None) for
    blank lines and lines starting with ';' or '#'.  :481-495"""
    line = line.strip()
    if len(line) == 0 or line[0] in ";#":
        return None, None
    parts = line.split("=", 1)
    if len(parts) != 2:
        raise ParamSyntaxError(line)
    return parts[0].strip(), parts[1].strip()


def readparams(file=None):
    """parameter file -> dict, keys lower-cased.  :462-478"""
    out = {}
    with open(file) as fp:
        for line in fp:
            key, val = parseline(line)
            if key is not None:
                out[key.lower()] = val
    return out


def str2bool(s):
    """'T', 'True', 'true' ... -> True; booleans pass through.  :497-502"""
    if s is True or s is False:
        return s
    return s[0].upper() == "T"


def readMatrix(filename, *shape):
    """ASCII matrix.  No shape: rows as they are.  One dimension n: the first n rows (one value
    per line).  More: the file must hold exactly prod(shape) non-empty LINES - i.e. one value per
    line - or ArraySizeError is raised, as the reference counts rows, not values.  :504-529"""
    rows = []
    with open(filename) as f:
        for line in f:
            vals = [float(v) for v in line.split()]
            if vals:
                rows.append(vals)
    if shape == ():
        return np.array(rows)
    if len(shape) == 1:
        return np.reshape(np.array(rows)[0:shape[0]], shape)
    if len(rows) == int(reduce(operator.mul, shape, 1.0)):
        return np.reshape(np.array(rows), shape)
    raise ArraySizeError(filename, shape, rows)


class ClData_CosmoMC(object):
    _FIELDS = ("obs", "var", "N_inv", "xfactors", "has_xfactor", "window", "win_min", "win_max",
               "beam_err", "inc_pol", "num_points", "has_corr_errors", "has_xfactors",
               "calib_uncertainty", "beam_uncertain", "has_pol")

    honour_N_inv = False       # see readDataset

    def __init__(self, file=None, device=0, window_dir=None, **attrs):
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
        if file is not None:
            self.readDataset(file, window_dir=window_dir)

    # ---- file parsers (host I/O) -------------------------------------------------------------
    def _alloc_points(self):
        n = self.num_points
        self.obs = np.empty(n)
        self.err_min = np.empty(n)
        self.err_pls = np.empty(n)
        self.beam_err = np.empty(n)
        self.window = np.zeros((n, self.ncls, lmax + 1))
        self.inc_pol = np.zeros(n, dtype=bool)
        self.win_min = np.empty(n, dtype=np.int32)
        self.win_max = np.empty(n, dtype=np.int32)
        self.ell = np.empty(n, dtype=np.int32)

    def readDataset(self, file, window_dir=None):
        """CosmoMC `.dataset` parameter file, :85-196.  window_dir / N_inv / xfactors paths are
        used as written in the file (relative to the working directory, as the reference) unless
        `window_dir` is given.  Differences from the reference, both where it raises: has_pol data
        sets work (the reference never allocates inc_pol, :248) and a band sub-range of a
        correlated data set inverts with numpy.linalg.inv (the reference calls the non-existent
        la.inverse, :180)."""
        par = readparams(file)
        self.name = par["name"]
        self.has_pol = str2bool(par.get("has_pol", False))
        self.all_l_exact = str2bool(par.get("all_l_exact", False))
        if self.all_l_exact:
            raise MyErr("All_l_exact not implemented")
        self.num_points = int(par.get("num_points", 0))
        self.calib_uncertainty = float(par.get("calib_uncertainty", 0.0))
        self.beam_uncertain = str2bool(par.get("beam_uncertainty", False))
        if window_dir is None:
            window_dir = par.get("window_dir")
        are_bare = str2bool(par.get("windows_are_bare", False))
        are_bandpowers = str2bool(par.get("windows_are_bandpowers", True))
        are_normalized = str2bool(par.get("windows_are_normalized", False))
        file_points = int(par.get("file_points", self.num_points))
        first_band = int(par.get("first_band", 1))
        if first_band + self.num_points > file_points + 1:
            raise ClDataError("first_band=", first_band, " +num_points=", self.num_points,
                              " > file_points=", file_points)
        first_band -= 1
        if self.has_pol:
            self.ncls = num_cls
            self.Clpol_idx = {}
        else:
            self.ncls = 1
            self.Clpol_idx = {"TT": [0, self.num_points]}
        self._alloc_points()
        if str2bool(par.get("bandpowers", False)):
            raise NotImplementedError
        for i in range(self.num_points):
            vals = tuple(float(e) for e in par.get("data" + str(i + 1)).split())
            if self.beam_uncertain:
                self.obs[i], self.err_min[i], self.err_pls[i], self.beam_err[i] = vals
            else:
                self.obs[i], self.err_min[i], self.err_pls[i] = vals
                self.beam_err[i] = 0.0
            self.readWindow(window_dir, i, i + first_band, are_bare, are_bandpowers, are_normalized)
            self.ell[i] = self.lmean(i)
        if self.beam_uncertain:
            self.beam_err = self.beam_err / self.obs
        self.sig = (self.err_min + self.err_pls) / 2
        self.var = self.sig ** 2
        self.Cl = self.obs.copy()
        # The reference looks up 'N_inv' in a dict whose keys readparams has lower-cased (:171 vs
        # :476), so it never finds the entry and treats every .dataset as uncorrelated.  Same
        # here unless honour_N_inv is set on the class or the instance.
        ninv_file = par.get("n_inv", None) if self.honour_N_inv else None
        self.has_corr_errors = ninv_file is not None
        if self.has_corr_errors:
            mat = readMatrix(ninv_file, file_points, file_points)
            if self.num_points != file_points:
                mat = np.linalg.inv(mat)[first_band:first_band + self.num_points,
                                         first_band:first_band + self.num_points]
                self.N_inv = np.linalg.inv(mat)
            else:
                self.N_inv = mat[0:self.num_points, 0:self.num_points]
        xfile = par.get("xfactors", None)
        self.has_xfactors = xfile is not None
        if self.has_xfactors:                              # data is ln(C+x), variance sig^2/(C+x)^2
            tmp_x = readMatrix(xfile, self.num_points + first_band)
            self.has_xfactor = np.ones(self.num_points, dtype=bool)
            self.xfactors = tmp_x[first_band:first_band + self.num_points]
            self.var = self.var / (self.obs + self.xfactors) ** 2
            self.obs = np.log(self.obs + self.xfactors)

    def readWindow(self, dir, i, iline, are_bare, are_bandpowers, are_normalized):
        """window file <dir>/<name><iline+1>: lines `l W_TT [W_TE W_EE]`, 2 <= l <= lmax kept.
        Not bare: W_l *= l.  Band-power windows: *= (l+1/2) over [win_min, win_max], the TT row
        normalised to sum_l W_l/(l(l+1)) = 1 unless already normalised, everything / 2 pi.
        :198-258"""
        win = self.window[i]
        ll = None
        with open("".join([dir, "/", self.name, str(iline + 1)])) as fp:
            for line in fp:
                cells = line.split()
                ll = cells[0]
                ell = int(float(ll))
                if math.fabs(ell - float(ll)) > 1.0e-4:
                    raise Nonint32Ell(i, ll)
                if 2 <= ell <= lmax:
                    win[:, ell] = [float(w) for w in cells[1:self.ncls + 1]]
        if not are_bare:
            win[:, :] *= np.arange(lmax + 1)
        nz = np.where(win)[1]
        if len(nz):
            self.win_min[i], self.win_max[i] = nz.min(), nz.max()
        else:
            self.win_min[i], self.win_max[i] = 0, 0
        lo, hi = int(self.win_min[i]), int(self.win_max[i]) + 1
        if are_bandpowers:
            ellwin = np.arange(lo, hi)
            win[:, lo:hi] *= (ellwin + 0.5)
            if not are_normalized:
                IW = sum(win[0, lo:hi] / (ellwin * (ellwin + 1.0)))
                win[0, lo:hi] /= IW
            win /= 2.0 * math.pi
        if self.has_pol:
            self.inc_pol[i] = np.any(win[1:, :] != 0)

    def lmean(self, band):                                 # :448-457
        lo, hi = int(self.win_min[band]), int(self.win_max[band]) + 1
        w = self.window[band, 0, lo:hi]
        return np.dot(list(range(lo, hi)), w) / sum(w)

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
