"""Catalogue readers (mcmc_amd/submmSED/data.py) against what the reference's own readers
(submmSED/data.py:75-607) return for the same files: tests/golden/readers_g7.npz was produced by
oracle/make_golden.py `readers` running the reference on the excerpts in tests/golden/catalogues/
(data files the reference ships, cut to a few KB)."""
import os

import numpy as np
import pytest

from mcmc_amd.submmSED import data as D

HERE = os.path.dirname(os.path.abspath(__file__))
CAT = os.path.join(HERE, "golden", "catalogues")
G = np.load(os.path.join(HERE, "golden", "readers_g7.npz"))


def cat(name):
    return os.path.join(CAT, name)


def check(prefix, dsets, rtol=0.0, sort=False, strip_names=False):
    if sort:
        dsets = sorted(dsets, key=lambda d: d.name)
    names = [str(n) for n in G[prefix + "/name"]]
    if strip_names:
        names = [n.strip() for n in names]
    assert [d.name for d in dsets] == names
    nb = G[prefix + "/nb"]
    assert [int(np.size(d.d)) for d in dsets] == list(nb)
    np.testing.assert_allclose([float(d.z) for d in dsets], G[prefix + "/z"], rtol=rtol, atol=0)
    off = np.concatenate(([0], np.cumsum(nb)))
    for key, get in (("freq", lambda d: d.freq), ("d", lambda d: d.d), ("sig", lambda d: d.sig)):
        ref = G[prefix + "/" + key]
        for i, d in enumerate(dsets):
            np.testing.assert_allclose(np.ravel(get(d)), ref[off[i]:off[i + 1]], rtol=rtol, atol=0,
                                       err_msg="%s %s %s" % (prefix, d.name, key))
    # what GaussianData derives from them
    for d in dsets:
        np.testing.assert_array_equal(d.sig2, np.asarray(d.sig) ** 2)


# The reference reads the fixed-width MRR fields as float32 ('f' formats) and, under NumPy 2,
# keeps float32 through 1e-3*s/cx; the same text parsed to float64 agrees to float32 rounding.
MRR_RTOL = 4e-7


@pytest.mark.parametrize("prefix,kw", [
    ("mrr_default", dict()),
    ("mrr_dlc1", dict(IRAS_ignore=[0], Planck_ignore=[3], DLC_ul=True, next0=True)),
    ("mrr_dlc2", dict(IRAS_ignore=[0], Planck_ignore=[2, 3], DLC_ul=True, next0=False)),
    ("mrr_nocx", dict(colorx=[1.0, 1.0, 1.0, 1.0], next0=False)),
])
def test_mrr_new_format(prefix, kw):
    check(prefix, D.readfluxes_MRR(cat("mrr_new.dat"), **kw), rtol=MRR_RTOL, strip_names=True)


def test_mrr_old_format_and_next0_error():
    check("mrr_old", D.readfluxes_MRR(cat("mrr_old.dat"), old_MRR=True, next0=False), rtol=MRR_RTOL,
          strip_names=True)
    with pytest.raises(ValueError):                       # the reference: no field named 'next'
        D.readfluxes_MRR(cat("mrr_old.dat"), old_MRR=True)
    a = D.readfluxes_MRR(cat("mrr_new.dat"), colorx=None, next0=False)
    b = D.readfluxes_MRR(cat("mrr_new.dat"), colorx=[1.0] * 4, next0=False)
    assert all(np.array_equal(x.d, y.d) for x, y in zip(a, b))


def test_dlc_formats():
    check("dlc0", D.readfluxes_DLC(cat("dlc_format0.txt"), format=0))
    check("dlc3", D.readfluxes_DLC(cat("peel_pixelfit.dat"), format=3, delnu=1763))


def test_dlc_2014_csv():
    f = cat("herus_phot.csv")
    check("herus_noUL", D.readfluxes_DLC_2014(f, UL25=False, getArp220=False))
    check("herus_noUL_keep157", D.readfluxes_DLC_2014(f, UL25=False, getArp220=False, del157=False))
    check("herus_UL", D.readfluxes_DLC_2014(cat("herus_phot_with25.csv"), UL25=True, getArp220=False))
    # sources without a 25 um point: the reference's `if l25:` was False on the empty index
    # array before NumPy 2.2 (it raises there now); same result as UL25=False for them
    full = D.readfluxes_DLC_2014(f, UL25=True, getArp220=False)
    no25 = [d for d in full if not np.any(np.round(D.speed_of_light / d.freq_obs) == 25)]
    ref = {d.name: d for d in D.readfluxes_DLC_2014(f, UL25=False, getArp220=False)}
    assert no25 and all(np.array_equal(d.d, ref[d.name].d) for d in no25)
    with pytest.raises(FileNotFoundError):
        D.readfluxes_DLC_2014(f, arp220_file="/nonexistent/Arp220.txt")


def test_dlc_2014_whitespace_table():
    check("magdis_noUL", D.readfluxes_DLC_2014(cat("magdis_ulirgs.dat"), UL25=False, getArp220=False))


def test_ul25_quirk_first_column():
    # a 25 um point in the first flux column is left alone (index 0 is falsy): data.py:234-237
    lam = np.array([25.0, 60.0]); f = np.array([1.0, 2.0]); s = np.array([0.1, 0.2])
    f2, s2 = D._upper_limit_25(lam, f.copy(), s.copy())
    assert np.array_equal(f2, f) and np.array_equal(s2, s)
    f3, s3 = D._upper_limit_25(lam[::-1].copy(), f[::-1].copy(), s[::-1].copy())
    assert f3[1] == 0 and s3[1] == 1.0


def test_topcat():
    check("topcat", D.readfluxes_ERCSC_TopCat(cat("ercsc_topcat.txt")))
    check("topcat_noUL", D.readfluxes_ERCSC_TopCat(cat("ercsc_topcat.txt"), upperlim=0))
    ds = D.readfluxes_ERCSC_TopCat(cat("ercsc_topcat.txt"))
    np.testing.assert_array_equal([d.z_alt for d in ds], G["topcat/z_alt"])
    assert [d.name_alt for d in ds] == [str(n) for n in G["topcat/name_alt"]]
    dd = D.readfluxes_ERCSC_TopCat(cat("ercsc_topcat.txt"), delete_upperlim=True)
    assert len(dd) == len(ds) and any(d.name.endswith("D") for d in dd)
    assert all(np.all(d.d / d.sig > 2.0) and d.freq.shape == d.d.shape for d in dd)


def test_peel_m31_mortier():
    check("peel", D.readfluxes_peel(cat("peel_pixelfit.dat")))
    check("peel_delnu", D.readfluxes_peel(cat("peel_pixelfit.dat"), delnu=[1763, 12000]))
    check("m31", D.readfluxes_M31(cat("M31Flux-v2.dat")))
    check("m31_delnu", D.readfluxes_M31(cat("M31Flux-v2.dat"), delnu=[28.4, 857.0]))
    check("mortier", D.readfluxes_mortier(cat("mortier")), sort=True)


def test_catalogue_arrays_ragged():
    ds = D.readfluxes_DLC_2014(cat("herus_phot.csv"), UL25=False, getArp220=False)
    nb, nu, d, sig = D.catalogue_arrays(ds)
    assert nb.dtype == np.int32 and nu.shape == (len(ds), nb.max()) and len(set(nb)) > 1
    for s, x in enumerate(ds):
        assert np.array_equal(nu[s, :nb[s]], x.freq) and np.array_equal(d[s, :nb[s]], x.d)
        assert np.all(sig[s, nb[s]:] == 1.0) and np.all(d[s, nb[s]:] == 0.0)


def test_model_flux_matches_reference():
    """model.flux(nu1, nu2) / flux() (model.py:113-127 and the per-class flux methods, quirks
    included: the one-component classes integrate the un-normalised greybody)."""
    from mcmc_amd.submmSED import model as M
    nu2, nu1 = 299792. / 8.0, 299792. / 1000.0
    for cls in (M.submmModel2_normalized, M.submmModel2_normalized_logA, M.submmModel1_normalized,
                M.submmModel1_normalized_logA, M.submmModel1_opticallythick,
                M.submmModel1_opticallythick_logA):
        th = G["flux/%s/theta" % cls.__name__]
        band = np.array([cls(*t).flux(nu1, nu2) for t in th])
        bolo = np.array([cls(*t).flux() for t in th])
        np.testing.assert_allclose(band, G["flux/%s/band" % cls.__name__], rtol=1e-12)
        np.testing.assert_allclose(bolo, G["flux/%s/bolo" % cls.__name__], rtol=1e-12)
