"""The band-power run driver (mcmc_amd/binnedCl/driver.py, interface of the reference's
MCMC/test_binnedCl.py main / getlike / getsamples) end to end on the GPU: a .newdat data set of
the G8 fixtures, a synthetic shape spectrum file, two sampler stages, the products and the files
the reference's run writes."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "golden", "clfiles")


def shape_file(path, lmax=600):
    l = np.arange(2, lmax + 1)
    tt = 2.5 * (5000.0 * np.exp(-((l - 220.0) / 300.0) ** 2) + 1500.0)
    rows = np.stack([l, tt, 0.02 * tt, 0.1 * tt, 0 * tt, 0 * tt], axis=1)
    np.savetxt(path, rows, fmt="%.8e")


def test_band_power_run_and_its_products(tmp_path, monkeypatch, capsys):
    from mcmc_amd.ClData import ClData as CD
    from mcmc_amd.binnedCl import driver
    from mcmc_amd.binnedCl.binnedClModel import binnedClModel as mod
    monkeypatch.chdir(ROOT)
    d = CD.ClData("expC.newdat", window_dir="data/windows")
    sf = str(tmp_path / "shape.dat")
    shape_file(sf)
    prefix = str(tmp_path / "run_")
    edges = [[90, 180, 260, 340, 420, 500]]
    res = driver.main(nMC=(500, 700), gridPlot=False, no_pol=True, data=[d], bins=edges,
                      rotateParams=True, prefix=prefix, shapefile=sf, seed=4)
    out = capsys.readouterr().out
    stages, ana = res
    assert len(stages) == 2 and mod.nparam == 5 and mod.nbins == [5]
    s = stages[-1]
    assert s.samples.shape == (701, 5) and s.naccept > 50
    lp = s.like.lnLike(s.samples[-1])
    assert abs(s.lnPr[-1] - lp) <= 1e-10 * abs(lp)
    pr = s.products
    np.testing.assert_array_equal(pr["ell"][0], [134.5, 219.5, 299.5, 379.5, 459.5])
    # files: what was written is what the run computed
    bp = np.loadtxt(prefix + ".bp")
    assert bp.shape == (5, 4)
    np.testing.assert_array_equal(bp[:, 0], pr["ell"][0])
    np.testing.assert_array_equal(bp[:, 1], pr["mean"][0])
    np.testing.assert_array_equal(bp[:, 2], pr["stdv"][0])
    np.testing.assert_array_equal(bp[:, 3], pr["stdv"][0])
    np.testing.assert_allclose(np.loadtxt(prefix + ".covar"), pr["Clcovar"], atol=5.1e-7)
    corr = np.loadtxt(prefix + ".corr")
    np.testing.assert_allclose(corr, pr["corr"], atol=5.1e-7)
    np.testing.assert_allclose(np.diag(pr["corr"]), 1.0, rtol=1e-12)
    for i in range(5):
        w = np.loadtxt(prefix + str(i + 1))
        assert w.shape == (500, 4)
        np.testing.assert_array_equal(w[:, 1], pr["WBl"][i, 0])
        np.testing.assert_array_equal(w[:, 0], np.arange(500))
        assert np.all(w[:90, 1:] == 0) and np.all(w[:, 2:] == 0)         # TT only, nothing below the bins
        np.testing.assert_allclose(w[edges[0][i]:edges[0][i + 1], 1],
                                   1.0 / (edges[0][i + 1] - 1 - edges[0][i]), rtol=1e-14)   # own bin: F_BB / F_BB
    # band powers = mean q times the bin's shape normalisation, and they sit near the data
    np.testing.assert_array_equal(pr["mean"][0], mod.bandpowers(s.mean())[0])
    # printed: bins, the summary of printvals, the final table
    assert "bins:" in out and "Max ln likelihood" in out
    last_lines = [ln for ln in out.strip().splitlines()[-5:]]
    assert last_lines[0].split()[0] == "134" and len(last_lines[0].split()) == 3
    # all stages' samples
    assert driver.getsamples(res).shape == (5, 501 + 701)


def test_single_bin_likelihood_scan(tmp_path, monkeypatch):
    from mcmc_amd.ClData import ClData as CD
    from mcmc_amd.binnedCl import driver
    monkeypatch.chdir(ROOT)
    d = CD.ClData("expC.newdat", window_dir="data/windows")
    sf = str(tmp_path / "shape.dat")
    shape_file(sf)
    data, like, likearr = driver.getlike(ibins=[1, 2], data=[d], bins=[90, 260, 420, 600], shapefile=sf,
                                         verbose=False)
    assert data[0] is d and likearr.shape == (100,)
    assert likearr.max() == 1.0 and np.all(np.isfinite(likearr)) and likearr[0] < 1e-3
    peak = np.linspace(0, 5, 100)[likearr.argmax()]
    assert 0.2 < peak < 4.9                                               # an interior maximum
