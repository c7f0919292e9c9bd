"""mcmc_amd/getdist_ahj.py against the reference's MCMC/getdist_ahj.py: return values and printed
text on the stub chains of tests/golden/getdist_g9.json (oracle/make_golden.py getdist ran the
reference on them), and the two places where the reference raises and this module works."""
import contextlib
import io
import json
import os
import types

import numpy as np
import pytest

from mcmc_amd import getdist_ahj as G

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def g9():
    with open(os.path.join(HERE, "golden", "getdist_g9.json")) as f:
        return json.load(f)


class StubLike:
    """like / like.model of a finished chain: lnLike and chi2 of a stored sample are looked up."""
    nDerived = 0

    class model:
        @staticmethod
        def prior(*p):
            return 1.0

    def __init__(self, rows, lnPr, skew):
        self.tab = {tuple(np.round(s, 12)): l for s, l in zip(rows, lnPr)}
        self.skew = skew

    def lnLike(self, p):
        return self.skew * self.tab[tuple(np.round(np.asarray(p), 12))]

    def chi2(self, p):
        return -2.0 * self.tab[tuple(np.round(np.asarray(p), 12))]


def chain(case):
    samples, lnPr = np.array(case["samples"]), np.array(case["lnPr"])
    derived = np.array(case["derived"]) if case["derived"] is not None else None
    m = types.SimpleNamespace(samples=samples, lnPr=lnPr, nDerived=case["nDerived"], derived=derived)
    rows = np.hstack([samples, derived]) if derived is not None else samples
    m.like = StubLike(rows, lnPr, case["skew"])
    return m


@pytest.mark.parametrize("name", ["plain", "derived", "skewed"])
@pytest.mark.parametrize("fn,kw", [("printvals", {}), ("histgrid", dict(noPlot=True, burn=0.2, stride=2))])
def test_best_sample_and_printed_summary(g9, name, fn, kw):
    case = g9["cases"][name]
    exp = case[fn]
    buf = io.StringIO()
    if "raises" in exp:                       # stored lnPr and a fresh evaluation differ by > 1 %
        with pytest.raises(AssertionError) as e, contextlib.redirect_stdout(buf):
            getattr(G, fn)(chain(case), **kw)
        assert str(e.value) == exp["raises"]
    else:
        with contextlib.redirect_stdout(buf):
            ret = getattr(G, fn)(chain(case), **kw)
        assert float(ret[0]) == exp["ret"][0]
        np.testing.assert_array_equal(ret[1], exp["ret"][1])
    assert buf.getvalue() == exp["text"]


def test_getallsamples_gauss_stddev_key(g9):
    for case in g9["cases"].values():
        np.testing.assert_array_equal(G.getallsamples(chain(case), burn=7, stride=3), case["getallsamples"])
    m = g9["misc"]
    np.testing.assert_array_equal(G.GaussPDF(np.array(m["x"]), 0.5, 2.0), m["gauss"])
    assert G.stddev(np.arange(7.0) ** 2) == m["stddev"]
    assert G.key("chain_EE_3.txt") == m["key"]
    arr = np.arange(24.0).reshape(12, 2)
    np.testing.assert_array_equal(G.getallsamples(arr, burn=2), arr[2:])       # bare arrays pass through


def test_where_the_reference_raises(tmp_path, g9):
    """Fractional burn in getallsamples (the reference reads an undefined name, getdist_ahj.py:39)
    and convertSampleFile (float row count under Python 3, :314)."""
    case = g9["cases"]["plain"]
    m = chain(case)
    n = len(m.samples)
    np.testing.assert_array_equal(G.getallsamples(m, burn=0.25), m.samples[int(0.25 * n):])
    f = tmp_path / "chain_1.txt"
    f.write_text("2 1.5 0.1 0.2 0.3\n1 2.5 1.1 1.2 1.3\n3 0.5 2.1 2.2 2.3\n")
    lnl, s = G.convertSampleFile(str(f))
    np.testing.assert_array_equal(lnl, [-1.5, -1.5, -2.5, -0.5, -0.5, -0.5])
    assert s.shape == (6, 3)
    np.testing.assert_array_equal(s[[0, 2, 5]], [[0.1, 0.2, 0.3], [1.1, 1.2, 1.3], [2.1, 2.2, 2.3]])
    assert G.histgrid(np.zeros((4, 2))) is None                                # bare array, no lnLike
