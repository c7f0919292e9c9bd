"""ClData file formats (.dataset, .newdat, window files) and the post-fit products of
binnedClModel against the reference's own parsers / functions (tests/golden/clfiles_g8.npz,
oracle/make_golden.py `clfiles`, run on the synthetic files in tests/golden/clfiles/)."""
import os

import numpy as np
import pytest

from mcmc_amd.ClData import ClData as CD, ClData_CosmoMC as CM, ClData_bcp as BCP
from mcmc_amd.binnedCl import binnedClModel as BM

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "golden", "clfiles")
G = np.load(os.path.join(HERE, "golden", "clfiles_g8.npz"))

ATTRS = ("obs", "var", "sig", "Cl", "err_min", "err_pls", "beam_err", "win_min", "win_max", "ell",
         "N_inv", "xfactors", "has_xfactor", "inc_pol")


@pytest.fixture
def in_root(monkeypatch):
    monkeypatch.chdir(ROOT)                                # the files name their windows relatively


def compare(prefix, d):
    sc = G[prefix + "/scalars"]
    assert (d.num_points, d.ncls) == (int(sc[0]), int(sc[1]))
    assert float(d.calib_uncertainty) == sc[2]
    assert [bool(d.beam_uncertain), bool(d.has_pol), bool(d.has_xfactors), bool(d.has_corr_errors)] == \
        [bool(v) for v in sc[3:7]]
    assert d.name == str(G[prefix + "/name"])
    for a in ATTRS:
        key = prefix + "/" + a
        if a == "inc_pol" and not d.has_pol:               # np.empty garbage in the reference (never read)
            continue
        if key in G.files:
            ref = G[key]
            got = np.asarray(getattr(d, a))
            assert got.shape == ref.shape, a
            if a == "N_inv" and prefix in ("C", "D"):      # la.inv of the same matrix: LAPACK, 1 ulp level
                np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max(), err_msg=a)
            else:
                np.testing.assert_array_equal(got, ref, err_msg=a)
    w = np.asarray(d.window)
    assert list(w.shape) == list(G[prefix + "/window_shape"])
    np.testing.assert_array_equal(np.argwhere(w != 0), G[prefix + "/window_nz"])
    np.testing.assert_array_equal(w[w != 0], G[prefix + "/window_val"])
    ref_idx = sorted((str(k), int(a), int(b)) for k, a, b in G[prefix + "/Clpol_idx"])
    assert sorted((k, v[0], v[1]) for k, v in d.Clpol_idx.items()) == ref_idx


def test_dataset_correlated_xfactors_beam(in_root):
    compare("A", CM.ClData_CosmoMC("expA.dataset"))
    compare("A", CD.ClData("expA.dataset"))               # factory picks the class by extension
    # the reference never finds its 'N_inv' key (case mismatch): uncorrelated unless asked
    d = CM.ClData_CosmoMC("expA.dataset", honour_N_inv=True)
    assert d.has_corr_errors and d.N_inv.shape == (6, 6)
    np.testing.assert_array_equal(d.N_inv.ravel(), np.loadtxt("expA_Ninv.dat"))


def test_dataset_subrange_bare_normalised_windows(in_root):
    compare("B", CM.ClData_CosmoMC("expB.dataset"))


def test_newdat_tt_per_band_xfactors(in_root):
    d = CD.ClData("expC.newdat")
    assert isinstance(d, BCP.ClData_bcp)
    compare("C", d)


def test_newdat_polarised_band_selection_fisher(in_root):
    compare("D", BCP.ClData_bcp("expD.newdat"))


def test_window_dir_override_and_list(tmp_path):
    d = CM.ClData_CosmoMC(os.path.join(ROOT, "expB.dataset"), window_dir=os.path.join(ROOT, "winB"))
    compare("B", d)
    d = BCP.ClData_bcp(os.path.join(ROOT, "expC.newdat"), window_dir=os.path.join(ROOT, "data", "windows"))
    compare("C", d)
    lst = tmp_path / "data_list.txt"
    lst.write_text("# list\n\n%s\n" % os.path.join(ROOT, "expB.dataset"))
    got = CD.getClData(str(lst), verbose=False, no_pol=True, window_dir=os.path.join(ROOT, "winB"))
    assert len(got) == 1 and got[0].has_pol is False and got[0].has_pol_really is False


def test_param_file_helpers(tmp_path):
    assert CM.parseline("  a = b c = d ") == ("a", "b c = d")
    assert CM.parseline("; x") == (None, None) and CM.parseline("") == (None, None)
    with pytest.raises(CM.ParamSyntaxError):
        CM.parseline("no equals sign")
    assert CM.str2bool("t") and CM.str2bool("True") and not CM.str2bool("F") and CM.str2bool(True) is True
    f = tmp_path / "m.dat"
    f.write_text("1\n2\n3\n4\n")
    np.testing.assert_array_equal(CM.readMatrix(str(f), 2, 2), [[1, 2], [3, 4]])
    np.testing.assert_array_equal(CM.readMatrix(str(f), 3), [1, 2, 3])
    f.write_text("1 2\n3 4\n")                             # the reference counts lines, not values
    with pytest.raises(CM.ArraySizeError):
        CM.readMatrix(str(f), 2, 2)
    assert CM.readMatrix(str(f)).shape == (2, 2)
    b = tmp_path / "bins.txt"
    b.write_text("2 50 1\n51 100 1\n2 100 2\n2 150 3\n")
    assert CD.readbins(str(b)) == [[(2, 50), (51, 100)], [(2, 100)], [(2, 150)]]


def test_fisher_windows():
    F = G["post/F"]
    bins = [[(2, 50), (51, 120), (121, 300)], [(2, 100), (101, 300)], [(2, 150), (151, 300)]]
    np.testing.assert_array_equal(BM.FisherWindows(F), G["post/Wbb"])
    np.testing.assert_allclose(BM.FisherWindows(F, isCovar=True), G["post/Wbb_covar"], rtol=1e-12)
    np.testing.assert_array_equal(BM.FisherWindows(F, bins=bins), G["post/WBl"])


def test_offset_lognormal_fits():
    samp = G["post/samples"]
    o = BM.oln(samp)
    par = G["post/oln_par"]
    got = np.array([o.chi2(*par), o.like(par), o.KSnorm(par)] + list(o.derivs(par)))
    np.testing.assert_allclose(got, G["post/oln_eval"], rtol=1e-13)
    np.testing.assert_allclose(o.cum(par), G["post/oln_cumcurve"], rtol=1e-13, atol=1e-16)
    assert o.like(np.array([5.0, -1.0, 45.0])) == np.inf
    # same optimiser, same start, same objective: the same iterates
    np.testing.assert_allclose(BM.fitOffsetLognormal_cum(samp), G["post/oln_cum"], rtol=1e-9)
    x, f, info = BM.fitOffsetLognormal_like(samp)
    np.testing.assert_allclose(x, G["post/oln_like_x"], rtol=1e-8)
    np.testing.assert_allclose(f, G["post/oln_like_f"], rtol=1e-12)
    assert BM.fitOffsetLognormal is BM.fitOffsetLognormal_cum
