"""`.newdat` band-power files (the format of C. Contaldi's B03 / CosmoMC releases) with the
interface of the reference's ClData_bcp (MCMC/ClData/ClData_bcp.py:15-210).  Host-side parser;
everything it fills in is consumed by mi355_cl_load.

File layout:
    name                              (optionally preceded by a line FISHER_T_CMB)
    n_TT n_EE n_BB n_EB n_TE n_TB
    [BAND_SELECTION + six lines `first last` (first = 0: spectrum not used)]
    has_calib  calib  calib_err
    has_beam   beam_width  beam_sigma
    ilike                             (0 Gaussian, 1 all offset-lognormal, 2 per band)
    per spectrum present: a type line (TT, EE, ...), n lines
        `k obs err- err+ x_factor l_min l_max [use_x]`, then n lines of correlation matrix (skipped)
    covariance matrix of all file bands
Windows are read from <window_dir>/<name><k> (band-power windows, normalised, not bare).

The parser works in three passes over the file's lines - header, band tables, covariance - and
then applies the calibration and the offset-lognormal transform to whole arrays."""
import numpy as np

from . import ClData_CosmoMC
from .ClData_CosmoMC import num_cls, lmax  # noqa: F401

_NSPEC = 6                                                 # TT EE BB EB TE TB


class _Lines(object):
    """the file's lines with a cursor"""

    def __init__(self, path):
        with open(path) as f:
            self.rows = f.read().split("\n")
        self.at = 0

    def take(self):
        row = self.rows[self.at] if self.at < len(self.rows) else ""
        self.at += 1
        return row.strip()

    def rest(self):
        out = self.rows[self.at:]
        self.at = len(self.rows)
        return out


def _header(src):
    """name, Fisher flag, bands per spectrum, selected (first, last) 0-based per spectrum,
    calibration (value, error), beam (uncertain, width, sigma), likelihood type"""
    name = src.take()
    fisher = name == "FISHER_T_CMB"
    if fisher:
        name = src.take()
    counts = [int(tok) for tok in src.take().split()]
    row = src.take()
    if row == "BAND_SELECTION":
        picked = []
        for _ in range(_NSPEC):
            first, last = (int(tok) for tok in src.take().split())
            picked.append((first - 1, last - 1) if first != 0 else (-1, -1))   # -1: none taken
        row = src.take()
        pol = _last_used(picked) > 0
    else:
        picked = [(0, n - 1) for n in counts]
        pol = sum(counts[1:]) > 0
    cal = [float(tok) for tok in row.split()[1:3]]
    beam = src.take().split()
    ilike = int(src.take().split()[0])
    return dict(name=name, fisher=fisher, counts=counts, picked=picked, has_pol=pol, cal=cal[0],
                cal_err=cal[1], beam_uncertain=int(beam[0]) != 0, beam_width=float(beam[1]),
                beam_sigma=float(beam[2]), ilike=ilike)


def _last_used(picked):
    """index of the last spectrum with a selection (the reference's has_pol after BAND_SELECTION
    is `i > 0` of the last selected spectrum, ClData_bcp.py:64-68)"""
    used = [i for i, rng in enumerate(picked) if rng[0] >= 0]
    return used[-1] if used else 0


class ClData_bcp(ClData_CosmoMC.ClData_CosmoMC):
    default_window_dir = "data/windows"                    # ClData_bcp.py:136

    def readDataset(self, file, window_dir=None):
        src = _Lines(file)
        hd = _header(src)
        self.name = hd["name"]
        self.has_pol = hd["has_pol"]
        self.all_l_exact = False
        self.calib_uncertainty = hd["cal_err"]
        self.beam_uncertain = hd["beam_uncertain"]
        self.ncls = num_cls if self.has_pol else 1
        counts, picked = hd["counts"], hd["picked"]
        self.num_points = sum(hi - lo + 1 for lo, hi in picked if lo >= 0)
        self._alloc_points()

        # ---- band tables ---------------------------------------------------------------------
        wdir = self.default_window_dir if window_dir is None else window_dir
        n = self.num_points
        xfac = np.empty(n)
        edges = np.empty((n, 2))
        self.has_xfactors = hd["ilike"] > 0
        if self.has_xfactors:
            self.has_xfactor = np.ones(n, dtype=bool)
        self.Clpol_idx = {}
        taken = []                                         # file index of every band kept
        slot = seen = 0
        for spec in range(_NSPEC):
            if counts[spec] == 0:
                continue
            label = src.take()[0:2]
            begin = slot
            lo, hi = picked[spec]
            for k in range(counts[spec]):
                cells = src.take().split()
                if lo <= k <= hi and lo >= 0:
                    vals = [float(tok) for tok in cells[1:7]]
                    self.obs[slot], self.err_min[slot], self.err_pls[slot], xfac[slot] = vals[:4]
                    edges[slot] = vals[4:6]
                    if hd["ilike"] > 1:
                        self.has_xfactor[slot] = int(cells[7])
                    self.readWindow(wdir, slot, seen, False, True, True)
                    self.ell[slot] = edges[slot].sum() / 2.0
                    taken.append(seen)
                    slot += 1
                seen += 1
            self.Clpol_idx[label] = [begin, slot]
            for _ in range(counts[spec]):                  # correlation matrix: not used
                src.take()

        # ---- covariance of all file bands, rows that are not numbers skipped -------------------
        numbers = []
        for row in src.rest():
            try:
                numbers.extend(float(tok) for tok in row.split())
            except ValueError:
                pass
        nfile = sum(counts)
        cov = np.array(numbers).reshape(nfile, nfile)[np.ix_(taken, taken)]

        # ---- calibration, beam error, offset-lognormal variables --------------------------------
        c2 = hd["cal"] ** 2
        self.has_corr_errors = True
        self.beam_err = np.abs(np.exp(-self.ell * (self.ell + 1) * 1.526e-8 * 2.0 * hd["beam_sigma"] *
                                      hd["beam_width"]) - 1.0)
        self.sig = (self.err_pls + self.err_min) / 2.0
        self.obs = self.obs * c2
        self.Cl = self.obs.copy()
        self.sig = self.sig * c2
        self.var = self.sig ** 2
        cov = cov * hd["cal"] ** 4
        if hd["fisher"]:
            cov = cov * (2.725 ** 4 * 1.0e24)
        if self.has_xfactors:                              # z = ln(C + x) for the flagged bands
            self.xfactors = c2 * xfac
            shift = np.where(self.has_xfactor, self.obs + self.xfactors, 1.0)
            cov = cov / shift[:, None]                     # rows first, then columns, as the
            cov = cov / shift[None, :]                     # reference's element loop divides
            flagged = self.has_xfactor
            self.var[flagged] /= shift[flagged] ** 2
            self.obs[flagged] = np.log(shift[flagged])
        self.N_inv = np.linalg.inv(cov)
