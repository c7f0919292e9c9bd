"""Accuracy of the library's own FP64 transcendental kernels (mcmc_amd/csrc/fastmath.hpp)
against mpmath, in units in the last place, on the domains the SED models use."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
mp = pytest.importorskip("mpmath")
mp.mp.prec = 200


@pytest.fixture(scope="module")
def eng():
    from mcmc_amd import Engine
    e = Engine(device=0)
    yield e
    e.close()


def ulp_err(got, exact):
    """max |got - exact| / ulp(exact) with exact a list of mpf."""
    worst = 0.0
    for g, e in zip(got, exact):
        ef = float(e)
        if ef == 0.0 or not np.isfinite(ef):
            assert g == ef
            continue
        u = np.spacing(abs(ef))
        worst = max(worst, float(abs(mp.mpf(float(g)) - e) / mp.mpf(float(u))))
    return worst


def test_expm1_planck_arguments(eng):
    rng = np.random.default_rng(0)
    # x = h nu / kT over the prior box: 1e-3 .. 690, plus tiny / negative / clamped edges
    x = np.concatenate([10 ** rng.uniform(-3, np.log10(689.0), 4000), rng.uniform(-39, 0, 500),
                        [1e-300, 1e-20, 0.0, 0.3465, 0.3467, 689.999, -39.9]])
    got = eng.math_probe(1, x)
    w = ulp_err(got, [mp.expm1(mp.mpf(float(v))) for v in x])
    print("expm1: max %.2f ulp" % w)
    assert w <= 2.5          # 2.01 measured with the degree-9 core polynomial (1.68 with degree 10)
    # beyond the clamp the value saturates (flux underflows to ~1e-299 instead of exactly 0)
    assert eng.math_probe(1, np.array([700.0, 1e6]))[0] == eng.math_probe(1, np.array([690.0]))[0]


def test_exp_and_exp10(eng):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-680, 680, 3000), rng.uniform(-1, 1, 1000), [0.0]])
    w = ulp_err(eng.math_probe(0, x), [mp.exp(mp.mpf(float(v))) for v in x])
    print("exp: max %.2f ulp" % w)
    assert w <= 1.5
    y = rng.uniform(-290, 290, 3000)
    w = ulp_err(eng.math_probe(4, y), [mp.power(10, mp.mpf(float(v))) for v in y])
    print("exp10: max %.2f ulp" % w)
    assert w <= 1.5


def test_pow_with_precomputed_log(eng):
    """base^e with ln(base) stored as a double-double (what mi355_sed_load precomputes):
    bases nu/nu_b in 0.05..60, exponents 3+beta in 3..6."""
    rng = np.random.default_rng(2)
    base = 10 ** rng.uniform(np.log10(0.05), np.log10(60), 4000)
    e = rng.uniform(3.0, 6.0, 4000)
    lhi = np.empty_like(base)
    llo = np.empty_like(base)
    for i, b in enumerate(base):
        l = mp.log(mp.mpf(float(b)))
        lhi[i] = float(l)
        llo[i] = float(l - mp.mpf(lhi[i]))
    w = ulp_err(eng.math_probe(2, e, lhi, llo),
                [mp.power(mp.mpf(float(b)), mp.mpf(float(x))) for b, x in zip(base, e)])
    print("pow: max %.2f ulp" % w)
    assert w <= 1.5
    # the base-2 variant the SED fast path uses (log2 of the base stored as a double-double)
    for i, b in enumerate(base):
        l = mp.log(mp.mpf(float(b)), 2)
        lhi[i] = float(l)
        llo[i] = float(l - mp.mpf(lhi[i]))
    w = ulp_err(eng.math_probe(8, e, lhi, llo),
                [mp.power(mp.mpf(float(b)), mp.mpf(float(x))) for b, x in zip(base, e)])
    print("pow2: max %.2f ulp" % w)
    assert w <= 1.5


def test_newton_division(eng):
    rng = np.random.default_rng(3)
    a = 10 ** rng.uniform(-100, 100, 5000) * rng.choice([-1, 1], 5000)
    b = 10 ** rng.uniform(-100, 100, 5000)
    got = eng.math_probe(3, a, b)
    ref = a / b
    ulps = np.abs(got - ref) / np.spacing(np.abs(ref))
    print("div: max %.2f ulp, inexact fraction %.2e" % (ulps.max(), np.mean(ulps > 0)))
    assert ulps.max() <= 1.0


def test_box_muller_kernels(eng):
    """ln on (0,1) and sin/cos(2 pi u): the short kernels behind the proposal normals and the
    offset-lognormal deltas of the Cl path."""
    rng = np.random.default_rng(4)
    u = np.concatenate([rng.uniform(0, 1, 4000), 10 ** rng.uniform(-17, 0, 1000),
                        10 ** rng.uniform(0, 6, 1000), [2.0 ** -54, 0.5, 0.7071067811865476, 1.0]])
    u = u[u > 0]
    w = ulp_err(eng.math_probe(5, u), [mp.log(mp.mpf(float(v))) for v in u])
    print("log: max %.2f ulp" % w)
    assert w <= 2.5
    v = np.concatenate([rng.uniform(0, 1, 4000), [0.0, 0.125, 0.25, 0.375, 0.5, 0.625, 0.75, 0.875]])
    sn = eng.math_probe(6, v)
    cs = eng.math_probe(7, v)
    es = [mp.sin(2 * mp.pi * mp.mpf(float(t))) for t in v]
    ec = [mp.cos(2 * mp.pi * mp.mpf(float(t))) for t in v]
    # absolute accuracy is what the normals need (the exact values cross zero)
    assert max(abs(float(mp.mpf(float(a)) - b)) for a, b in zip(sn, es)) <= 3e-16
    assert max(abs(float(mp.mpf(float(a)) - b)) for a, b in zip(cs, ec)) <= 3e-16
    # the table-driven kernels of the Box-Muller transform itself (log_bm, sincos_bm: 64-row table):
    # the logarithm on the range of the 53-bit uniforms, relatively accurate up to u -> 1; sine and
    # cosine to the same absolute bound, from the table in LDS (14) and in device memory (15)
    uu = np.concatenate([rng.uniform(0, 1, 6000), 1.0 - 10 ** rng.uniform(-16, -1, 1500), 2.0 ** -rng.uniform(0, 54, 1500),
                         (np.arange(64, 129) / 128.0)[:-1], np.nextafter((np.arange(65, 129) / 128.0), 0),
                         [2.0 ** -54, 1.0 - 2.0 ** -54, 0.5, np.nextafter(0.5, 0)]])
    uu = uu[(uu > 0) & (uu < 1)]
    w = ulp_err(eng.math_probe(13, uu), [mp.log(mp.mpf(float(t))) for t in uu])
    print("log_bm: max %.2f ulp" % w)
    assert w <= 1.5
    vv = np.concatenate([v, np.arange(0, 129) / 128.0, np.nextafter(np.arange(1, 129) / 128.0, 0), [1.0 - 2.0 ** -54]])
    vv = vv[vv < 1]
    es = [mp.sin(2 * mp.pi * mp.mpf(float(t))) for t in vv]
    ec = [mp.cos(2 * mp.pi * mp.mpf(float(t))) for t in vv]
    ds = max(abs(float(mp.mpf(float(a)) - b)) for a, b in zip(eng.math_probe(14, vv), es))
    dc = max(abs(float(mp.mpf(float(a)) - b)) for a, b in zip(eng.math_probe(15, vv), ec))
    print("sincos_bm: max abs error %.2e / %.2e" % (ds, dc))
    assert ds <= 2.5e-16 and dc <= 2.5e-16


def test_exp2_kernels_of_the_band_loop(eng):
    """2^y as the SED band loop and the beam pixel loop evaluate it: the table-driven kernel
    (exp2_tab: 2048-entry table, degree-2 polynomial, integer exponent arithmetic) and the plain
    degree-11 chain (exp2_fast), on the whole range the fast path admits, at the table's grid
    points k/64 and next to the rounding boundaries (k + 1/2)/64."""
    rng = np.random.default_rng(9)
    k = rng.integers(-64 * 900, 64 * 900, 600)
    y = np.concatenate([rng.uniform(-1000, 1000, 4000), rng.uniform(-1, 1, 2000), k / 64.0,
                        np.nextafter((k + 0.5) / 64.0, 0), np.nextafter((k + 0.5) / 64.0, 1e9),
                        [0.0, -0.0, 1e-300, -1e-300, 999.9, -999.9, 1 / 64, -1 / 64]])
    exact = [mp.power(2, mp.mpf(float(v))) for v in y]
    w = ulp_err(eng.math_probe(9, y), exact)
    print("exp2_tab: max %.2f ulp" % w)
    assert w <= 1.4
    w = ulp_err(eng.math_probe(10, y), exact)
    print("exp2_fast: max %.2f ulp" % w)
    assert w <= 1.2
    np.testing.assert_array_equal(eng.math_probe(9, np.arange(-64.0, 65.0)), 2.0 ** np.arange(-64.0, 65.0))


def test_box_muller_radius_and_uniforms(eng):
    """sqrt on [0, 75] (the radius sqrt(-2 ln u)) and the 53-bit uniform of two Philox words, which
    has to be the stream definition's ((hi<<21 | lo>>11) + 0.5) 2^-53 to the bit."""
    rng = np.random.default_rng(6)
    x = np.concatenate([rng.uniform(0, 75, 6000), 10 ** rng.uniform(-16, 0, 2000), [1.0, 2.0, 4.0, 73.47]])
    w = ulp_err(eng.math_probe(11, x), [mp.sqrt(mp.mpf(float(v))) for v in x])
    print("sqrt: max %.2f ulp" % w)
    assert w <= 1.0
    assert eng.math_probe(11, np.array([0.0]))[0] == 2.0 ** -500
    hi = np.concatenate([rng.integers(0, 2 ** 32, 20000, dtype=np.uint64),
                         np.array([0, 0, 2 ** 32 - 1, 2 ** 32 - 1, 2 ** 31, 2 ** 31], np.uint64)])
    lo = np.concatenate([rng.integers(0, 2 ** 32, 20000, dtype=np.uint64),
                         np.array([0, 2 ** 32 - 1, 0, 2 ** 32 - 1, 2 ** 11, 2 ** 11 - 1], np.uint64)])
    want = (((hi << np.uint64(21)) | (lo >> np.uint64(11))).astype(np.float64) + 0.5) * 2.0 ** -53
    got = eng.math_probe(12, hi.astype(np.float64), lo.astype(np.float64))
    np.testing.assert_array_equal(got, want)
