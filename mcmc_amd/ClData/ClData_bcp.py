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
Windows are read from <window_dir>/<name><k> (band-power windows, normalised, not bare)."""
import numpy as np

from . import ClData_CosmoMC
from .ClData_CosmoMC import num_cls, lmax  # noqa: F401


class ClData_bcp(ClData_CosmoMC.ClData_CosmoMC):
    default_window_dir = "data/windows"                    # ClData_bcp.py:136

    def readDataset(self, file, window_dir=None):
        if window_dir is None:
            window_dir = self.default_window_dir
        with open(file) as fp:
            self._read(fp, window_dir)

    def _read(self, fp, window_dir):
        self.name = fp.readline().strip()
        fisher_T_cmb = self.name == "FISHER_T_CMB"
        if fisher_T_cmb:
            self.name = fp.readline().strip()
        npol = [int(s) for s in fp.readline().split()]
        self.has_pol = sum(npol[1:]) > 0
        self.all_l_exact = False
        file_points = sum(npol)
        self.num_points = file_points
        minmax = np.empty((2, 6), dtype=np.int32)
        line = fp.readline().strip()
        if line == "BAND_SELECTION":
            self.num_points = 0
            self.has_pol = False
            for i in range(6):
                minmax[:, i] = [int(m) for m in fp.readline().split()]
                if minmax[0, i] != 0:
                    self.num_points += minmax[1, i] - minmax[0, i] + 1
                    self.has_pol = i > 0
                else:
                    minmax[1, i] = 0
            line = fp.readline().strip()
        else:
            for i in range(6):
                minmax[:, i] = [1, npol[i]]
        minmax -= 1
        cal, self.calib_uncertainty = [float(e) for e in line.split()[1:3]]
        cells = fp.readline().split()
        beam_width, beam_sigma = [float(e) for e in cells[1:3]]
        self.beam_uncertain = int(cells[0]) != 0
        self.ncls = num_cls if self.has_pol else 1
        self._alloc_points()
        n = self.num_points
        used_bands = []
        tmp_x = np.empty(n)
        lb = np.empty((2, n))
        ilike = int(fp.readline().split()[0])
        self.has_xfactors = ilike > 0
        if ilike > 0:
            self.has_xfactor = np.ones(n, dtype=bool)
        use_i = file_i = 0
        self.Clpol_idx = {}
        for k in [k for k in range(6) if npol[k] != 0]:
            ch_type = fp.readline().strip()[0:2]
            self.Clpol_idx[ch_type] = [use_i, use_i]
            for i in range(npol[k]):
                cells = fp.readline().split()
                if minmax[0, k] <= i <= minmax[1, k]:
                    self.Clpol_idx[ch_type][1] += 1
                    used_bands.append(file_i)
                    (self.obs[use_i], self.err_min[use_i], self.err_pls[use_i], tmp_x[use_i],
                     lb[0, use_i], lb[1, use_i]) = [float(e) for e in cells[1:7]]
                    if ilike > 1:
                        self.has_xfactor[use_i] = int(cells[7])
                    self.readWindow(window_dir, use_i, file_i, False, True, True)
                    self.ell[use_i] = sum(lb[:, use_i]) / 2.0
                    use_i += 1
                file_i += 1
            for i in range(npol[k]):                       # the correlation matrix is not used
                fp.readline()
        self.has_corr_errors = True
        vals = []
        for line in fp.readlines():                        # lines that are not numbers are skipped
            try:
                vals.append([float(e) for e in line.split()])
            except ValueError:
                pass
        mat = np.concatenate(vals).reshape(file_points, file_points)
        self.N_inv = mat[np.ix_(used_bands, used_bands)].copy()
        self.beam_err = np.abs(np.exp(-self.ell * (self.ell + 1) * 1.526e-8 * 2.0 * beam_sigma *
                                      beam_width) - 1.0)
        self.sig = (self.err_pls + self.err_min) / 2.0
        self.obs = self.obs * cal ** 2
        self.Cl = self.obs.copy()
        self.sig = self.sig * cal ** 2
        self.var = self.sig ** 2
        self.N_inv *= cal ** 4
        if fisher_T_cmb:
            self.N_inv *= 2.725 ** 4 * 1.0e24
        if self.has_xfactors:                              # to z = ln(C+x): covariance, obs, var
            self.xfactors = cal ** 2 * tmp_x[0:n]
            for i in range(n):
                for j in range(n):
                    if self.has_xfactor[i]:
                        self.N_inv[i, j] /= self.obs[i] + self.xfactors[i]
                    if self.has_xfactor[j]:
                        self.N_inv[i, j] /= self.obs[j] + self.xfactors[j]
            idx = self.has_xfactor
            self.var[idx] /= (self.obs[idx] + self.xfactors[idx]) ** 2
            self.obs[idx] = np.log(self.obs[idx] + self.xfactors[idx])
        self.N_inv = np.linalg.inv(self.N_inv)
