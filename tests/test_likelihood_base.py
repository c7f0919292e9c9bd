"""Host logic of the Likelihood base class (Likelihood.py:14-175 of the reference): dataset list
handling, the amplitude-marginalised sum and its error behaviour, with a stub dataset whose
quadratic forms are numpy (the product's dataset classes evaluate theirs on the GPU; see
tests/test_api_gpu.py::test_generic_likelihood_marginalises_the_amplitude)."""
import numpy as np
import pytest

from mcmc_amd.Likelihood import Likelihood, ZeroPosterior, isiterable


class StubData(object):
    def __init__(self, d, sig):
        self.d, self.sig2, self.n = np.asarray(d, float), np.asarray(sig, float) ** 2, len(d)
        self.lnDetN = np.log(self.sig2).sum()

    def quadform(self, A=None, B=None):
        A = self.d if A is None else A
        return np.dot(A / self.sig2, A if B is None else B)

    def chi2(self, vec=None):
        return self.quadform() if vec is None else self.quadform(self.d - vec)


class Line(object):
    def __init__(self, slope):
        self.slope = slope

    def __call__(self, dataset):
        return self.slope * np.arange(dataset.n, dtype=float)

    @staticmethod
    def lnPrior(params):
        return -abs(params[0])


def _sets():
    rng = np.random.default_rng(0)
    return [StubData(rng.normal(size=n) + np.arange(n), rng.uniform(0.5, 1.5, n)) for n in (6, 9, 4)]


def test_dataset_list_handling():
    a, b, c = _sets()
    like = Likelihood(model=Line)
    assert like.data is None
    like.adddata(a)                                       # first dataset starts the list
    like.adddata([b, c])                                  # a sequence extends it
    assert like.data == [a, b, c]
    like.setdata(b)                                       # a single dataset is a list of one
    assert like.data == [b] and like.model_vals is None
    like.setdata((a, c))
    assert like.data == [a, c]
    assert isiterable([a]) and not isiterable(a)


def test_sum_over_datasets_and_ml_amplitude():
    sets = _sets()
    like = Likelihood(data=sets, model=Line)
    want = norm = chi2 = 0.0
    for g in sets:
        F = 0.7 * np.arange(g.n)
        ff, fd = np.dot(F / g.sig2, F), np.dot(F / g.sig2, g.d)
        want += 0.5 * (fd ** 2 / ff - np.log(ff))
        chi2 += np.dot((g.d - F) / g.sig2, g.d - F)
        norm += -0.5 * ((g.n - 1) * np.log(2 * np.pi) + g.lnDetN + np.dot(g.d / g.sig2, g.d))
    assert like([0.7]) == pytest.approx(want, rel=1e-14)
    assert like.MLamplitude == pytest.approx(fd / ff, rel=1e-14)
    assert len(like.model_vals) == 3 and isinstance(like.this_model, Line)
    assert like.chi2([0.7]) == pytest.approx(chi2, rel=1e-14)
    assert like.lnNorm() == pytest.approx(norm, rel=1e-14)
    assert like.lnLikeFull([0.7]) == pytest.approx(want + norm, rel=1e-14)
    assert like.lnPost([0.7]) == pytest.approx(want - 0.7, rel=1e-14)
    assert like.lnPostFull([0.7]) == pytest.approx(want - 0.7 + norm, rel=1e-14)


def test_vanishing_model_raises_zero_posterior():
    like = Likelihood(data=_sets(), model=Line)
    with pytest.raises(ZeroPosterior) as err:
        like([0.0])
    assert err.value.value == (0.0,)


def test_small_public_helpers_of_the_reference_modules():
    """convergence.gelmanrubin (ragged chains; the reference's raises NameError), MCMC.finishing,
    data.toFloat."""
    import numpy as np
    from mcmc_amd import MCMC, convergence
    from mcmc_amd.submmSED import data
    rng = np.random.default_rng(3)
    chains = [rng.normal(0.1 * k, 1.0 + 0.1 * k, (200 + 30 * k, 3)) for k in range(4)]
    r = convergence.gelmanrubin(chains, burn=20, thin=2)
    cs = [c[20::2] for c in chains]
    w1 = np.mean([c.var(axis=0, ddof=1) for c in cs], axis=0)
    w0 = np.mean([c.var(axis=0) for c in cs], axis=0)
    b = np.var([c.mean(axis=0) for c in cs], axis=0, ddof=1)
    np.testing.assert_allclose(r, np.sqrt((w0 + b) / w1), rtol=1e-14)
    assert r.shape == (3,) and np.all(r > 0.99)

    class Bar:
        done = False

        def finish(self):
            self.done = True
    with pytest.raises(KeyError):
        with MCMC.finishing(Bar()) as b_:
            raise KeyError("x")
    assert b_.done
    assert data.toFloat("1.5") == 1.5 and np.isnan(data.toFloat(""))
