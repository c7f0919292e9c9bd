"""Data sets parsed from .dataset / .newdat files go through mi355_cl_load and reproduce the
reference's band powers and calcLnLike on the same files (golden G8)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "golden", "clfiles")
G = np.load(os.path.join(HERE, "golden", "clfiles_g8.npz"))


@pytest.mark.parametrize("prefix,fname,wdir", [
    ("A", "expA.dataset", "winA"), ("B", "expB.dataset", "winB"),
    ("C", "expC.newdat", "data/windows"), ("D", "expD.newdat", "data/windows")])
def test_parsed_dataset_lnlike(prefix, fname, wdir, monkeypatch):
    from mcmc_amd.ClData import ClData as CD
    monkeypatch.chdir(ROOT)                                # the files name their companions relatively
    d = CD.ClData(fname, window_dir=wdir)
    Cl = G[prefix + "/Cl_in"]
    bp = d.bandpowers(Cl)
    ref_bp = G[prefix + "/bandpowers"]
    np.testing.assert_allclose(bp, ref_bp, rtol=1e-12, atol=1e-13 * np.abs(ref_bp).max())
    got = float(np.asarray(d.calcLnLike(Cl)).reshape(-1)[0])
    np.testing.assert_allclose(got, float(G[prefix + "/lnlike"]), rtol=1e-11)
