"""Results tables of the catalogue driver (postprocess / writeTab, driver.py:345-534) against the
reference's own output: tests/golden/writetab_g11.json holds seeded stub records, the structured
arrays the reference's postprocess() made of them and the text its writeTab() wrote (made by
`oracle/make_golden.py writetab`, which imports the reference's driver with an empty `getdist`
stub).  Fields bit for bit, files byte for byte."""
import json
import os

import numpy as np
import pytest

from mcmc_amd.submmSED import driver as drv


def fake_records(n, P, ndat, nt, seed=0):
    rng = np.random.default_rng(seed)
    recs = []
    for k in range(n):
        A = rng.standard_normal((P, P))
        cov = A @ A.T
        sd = np.sqrt(np.diag(cov))
        ncov = cov / np.outer(sd, sd)
        ML = float(rng.normal())
        recs.append(((rng.normal(size=P), sd, ncov), (ML, rng.normal(size=P), ML - abs(rng.normal())),
                     "F%05d+%04d(z=%5.3f)" % (k, k, 0.01 * k), 0.01 * k,
                     [(float(a), float(b)) for a, b in rng.uniform(1, 2, (ndat, 2))],
                     rng.uniform(1, 2, nt), float(rng.uniform(1, 30))))
    return recs


@pytest.fixture(scope="module")
def g11():
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "writetab_g11.json")) as f:
        return json.load(f)


def golden_records(run):
    return [((np.array(r["mean"]), np.array(r["sig"]), np.array(r["ncov"])),
             (r["ML"], np.array(r["MLpar"]), r["meanL"]), r["name"], r["z"],
             [tuple(x) for x in r["dat"]], np.array(r["flux"]), r["ML_chi2"]) for r in run["records"]]


@pytest.mark.parametrize("irun", [0, 1, 3, 4])
def test_results_table_and_text_equal_the_reference(g11, irun, tmp_path):
    run = g11["runs"][str(irun)]
    tab = drv.results_table(golden_records(run), irun)
    assert list(tab["name"]) == run["name"]
    for k, want in run["fields"].items():
        np.testing.assert_array_equal(tab[k], np.array(want), err_msg=k)
    f = tmp_path / "tab.txt"
    drv.writeTab(tab, str(f))
    assert f.read_text() == run["text"]
    drv.writeTab(tab, str(f), nodat=True)                  # the reference drops the names, not the columns
    assert f.read_text() == run["text_nodat"]


def test_results_table_columns_and_evidence():
    recs = fake_records(4, 6, 7, 2)
    tab = drv.results_table(recs, 0)                      # 2-component, beta fixed: A1 T1 A2 T2
    ix = [0, 2, 3, 5]
    assert tab["mean"].shape == (4, 4) and tab["covar"].shape == (4, 4, 4) and tab["dat"].shape == (4, 7, 2)
    for k, r in enumerate(recs):
        np.testing.assert_array_equal(tab[k]["mean"], r[0][0][ix])
        np.testing.assert_array_equal(tab[k]["MLpar"], r[1][1][ix])
        cov = r[0][2][:, ix][ix]
        np.testing.assert_array_equal(tab[k]["covar"], cov)
        hdet, norm = 0.5 * np.linalg.det(cov), 4 * 0.5 * np.log(2 * np.pi)   # driver.py:436
        assert tab[k]["ev"] == r[1][0] + hdet + norm and tab[k]["evMean"] == r[1][2] + hdet + norm
        assert tab[k]["dlnLike"] == r[1][0] - r[1][2]
        assert tab[k]["name"] == r[2][:21]


def test_writetab_layout_and_roundtrip(tmp_path):
    recs = fake_records(5, 3, 4, 1, seed=2)
    tab = drv.results_table(recs, 1)
    f = tmp_path / "tab1.txt"
    drv.writeTab(tab, str(f), plain=True)
    lines = f.read_text().splitlines()
    assert len(lines) == 6
    ncol = 2 + 3 * 3 + 3 + 2 * 4 + 1 + 1
    assert len(lines[0]) == 22 * ncol                     # "%21s " per header cell
    hdr = [lines[0][22 * k:22 * k + 21].strip() for k in range(ncol)]
    assert hdr[:3] == ["Name", "z", "ML param 0"] and hdr[-1] == "ML_chi2" and hdr[-2] == "greybody flux 0"
    assert hdr[14:16] == ["flux 0", "sigflux 0"]
    for ln, r in zip(lines[1:], tab):
        cells = ln.split()
        assert len(cells) == ncol and cells[0] == r["name"]
        vals = np.array([float(c) for c in cells[1:]])
        want = np.concatenate(([r["z"]], r["MLpar"], r["mean"], r["sig"], [r["dlnLike"], r["ev"], r["evMean"]],
                               r["dat"].ravel(), r["flux"], [r["ML_chi2"]]))
        np.testing.assert_array_equal(vals, want)         # repr() round-trips float64 exactly
    g = tmp_path / "nodat.txt"
    drv.writeTab(drv.results_table(recs, 1, nodat=True), str(g), nodat=True, plain=True)
    assert len(g.read_text().splitlines()[1].split()) == ncol - 8


def test_checkpoint_roundtrip_and_postprocess(tmp_path):
    recs = fake_records(3, 6, 5, 2, seed=5)
    fn = str(tmp_path / "check3.npz")
    drv.save_records(fn, recs)
    back = drv.load_records(fn)
    for a, b in zip(recs, back):
        np.testing.assert_array_equal(a[0][0], b[0][0])
        np.testing.assert_array_equal(a[0][2], b[0][2])
        assert a[1][0] == b[1][0] and a[2] == b[2] and a[4] == b[4] and a[6] == b[6]
    tabs = drv.postprocess([None, None, None, fn, None])
    assert len(tabs) == 5 and len(tabs[3]) == 3 and tabs[0] == []
    drv.writeTabAll(tabs, str(tmp_path / "out"), ext=".txt")
    assert (tmp_path / "out3.txt").exists() and not (tmp_path / "out0.txt").exists()


def test_model_selection_matches_reference_pairs():
    from mcmc_amd.submmSED import model as M, likelihood as L
    assert drv.select_model(True, True) == (M.submmModel1_normalized_logA, L.SEDLikelihood_normalized)
    assert drv.select_model(False, True, linear=True) == (M.submmModel2_normalized, L.SEDLikelihood_normalized)
    assert drv.select_model(True, False) == (M.submmModel1, L.SEDLikelihood1)
    assert drv.select_model(False, False) == (M.submmModel2, L.SEDLikelihood2)
    assert drv.select_model(False, True, opticallyThick=True)[0] is M.submmModel1_opticallythick_logA


def test_readtab_reads_what_writetab_wrote(tmp_path, capsys):
    """submmSED/readtab.readtab (readtab.py:11-23) on a table of driver.writeTab: one field per
    column, the numbers exactly as written."""
    from mcmc_amd.submmSED.readtab import readtab
    recs = fake_records(5, 6, 3, 2, seed=4)
    tab = drv.results_table(recs, 0)
    f = str(tmp_path / "t.txt")
    drv.writeTab(tab, f, plain=True)
    got = readtab(f)
    out = capsys.readouterr().out
    npar = tab["MLpar"].shape[1]
    assert got.shape == (5,) and "z" in got.dtype.names and "ML_chi2" in got.dtype.names
    np.testing.assert_array_equal(got["z"], tab["z"])
    for k in range(npar):
        np.testing.assert_array_equal(got["ML_param_%d" % k], tab["MLpar"][:, k])
        np.testing.assert_array_equal(got["Mean_param_%d" % k], tab["mean"][:, k])
        np.testing.assert_array_equal(got["sigma_param_%d" % k], tab["sig"][:, k])
    np.testing.assert_array_equal(got["ML_chi2"], tab["ML_chi2"])
    np.testing.assert_array_equal(got["flux_1"], tab["dat"][:, 1, 0])
    np.testing.assert_array_equal(got["greybody_flux_0"], tab["flux"][:, 0])
    assert np.all(np.isnan(got["Name"]))
    assert "ML_chi2 = %f +- %f" % (tab["ML_chi2"].mean(), tab["ML_chi2"].std()) in out
