"""What the callers of the sampler read off its chains: best sample, posterior means and widths.

Host-side mirror of the numeric half of the reference's MCMC/getdist_ahj.py (same names, same
arguments, same printed lines, same return values), for the chains `mcmc_amd.MCMC.sampler` returns
or plain [nsample, nparam] arrays.  The drawing half of that module (matplotlib histogram grids)
is a user interface and is not rebuilt: `histgrid` does the part the SED driver relies on
(submmSED/driver.py:182-184) and returns without drawing.

Where the reference raises, this module works instead (each case is covered by
tests/test_getdist.py): a fractional `burn` in `getallsamples` (getdist_ahj.py:39 reads an
undefined name) and the row count of `convertSampleFile` (getdist_ahj.py:314 divides to a float
under Python 3).
"""
import numpy as np

__all__ = ["GaussPDF", "getallsamples", "printvals", "histgrid", "stddev", "convertSampleFile", "key"]


def GaussPDF(x, mean=0, stdev=1):
    """Normal density, operation order of getdist_ahj.py:27-28."""
    return np.exp(-0.5 * (x - mean) ** 2 / stdev ** 2) / np.sqrt(2 * np.pi) / stdev


def getallsamples(MCMC, burn=0, stride=1, derived=True):
    """samples (+ derived parameters as extra columns) of a sampler object, or the array itself;
    rows burn::stride, a burn in (0,1) being a fraction of the chain (getdist_ahj.py:30-41)."""
    rows = getattr(MCMC, "samples", None)
    if rows is None:
        rows = MCMC                                  # a plain [nsample, nparam] array
    elif derived and MCMC.nDerived:
        rows = np.hstack([rows, MCMC.derived])
    first = int(burn * len(rows)) if 0 < burn < 1 else burn
    return rows[first::stride]


class _Best:
    """The best sample of a chain and the re-evaluated posterior there.  `of()` returns None when
    there is no lnLike to rank by (a bare array and no `lnLike` argument)."""

    def __init__(self, chain, rows, lnlike):
        at = int(np.argmax(lnlike))
        self.lnlike, self.params = max(lnlike), rows[at]
        like = chain.like
        self.chi2 = like.chi2(self.params)
        self.again = like.lnLike(self.params) + np.log(like.model.prior(*self.params))

    @classmethod
    def of(cls, chain, rows, lnlike):
        if lnlike is None:
            lnlike = getattr(chain, "lnPr", None)
        return None if lnlike is None else cls(chain, rows, lnlike)

    def check(self):
        # the stored lnPr against a fresh evaluation, to the reference's 1 % (:60-62, :194-196)
        if not abs(1. - self.again / self.lnlike) < 0.01:
            raise AssertionError("different likelihood calculations differ: %f != %f"
                                 % (self.again, self.lnlike))

    def result(self):
        return (self.lnlike, self.params)


def printvals(MCMC, params=None, lnLike=None, derived=True):
    """Best sample, then mean +- stdev (ddof=0) beside the best value of each parameter in
    `params` (default: all); returns (maxlnLike, maxLikeParams).  Printed lines and return value
    of getdist_ahj.py:44-76."""
    rows = getallsamples(MCMC, derived=derived)
    best = _Best.of(MCMC, rows, lnLike)
    if best is None:                               # (the reference goes on and fails with NameError, :70)
        raise ValueError("printvals needs lnLike (or a sampler object with .lnPr)")
    print("chi2 at max = %f" % (best.chi2,))
    best.check()
    print('Max ln likelihood %f at parameters:' % best.lnlike)
    print(best.params)
    columns = range(rows.shape[1]) if params is None else params
    for col, at_best in zip(columns, best.params):
        print('mean = %g +- %g | ML = %g' % (rows[:, col].mean(), rows[:, col].std(), at_best))
    return best.result()


def histgrid(MCMC, params=None, nbins=30, labels=None, lnLike=None, quiet=False, derived=True,
             maxlnLike=[], noPlot=False, burn=0, stride=1):
    """The numeric part of getdist_ahj.histgrid (:175-205): the best sample over ALL samples (burn
    and stride only select what would be drawn), the consistency check, the same printed lines;
    returns (maxlnLike, maxLikeParams), or None for a bare array without lnLike.  Nothing is
    drawn, whatever `noPlot` says."""
    best = _Best.of(MCMC, getallsamples(MCMC, derived=derived), lnLike)
    if best is None:
        return None
    print("chi2 at max = %f" % best.chi2)
    print('Max ln likelihood %f at parameters:' % best.lnlike)
    best.check()
    print(best.params)
    return best.result()


def stddev(arr):
    """ddof=1 ... as the reference writes it (getdist_ahj.py:291-292): std * sqrt((n-1)/n)."""
    return arr.std() * np.sqrt((arr.size - 1) / arr.size)


def convertSampleFile(filename):
    """CosmoMC-style chain file -> (lnLike [n], samples [n, npar]); one line per distinct point:
    `repetitions  -lnlike  parameter values...`, each point repeated `repetitions` times
    (getdist_ahj.py:295-317)."""
    lnLike, rows = [], []
    with open(filename, 'r') as fil:
        for lin in fil:
            cols = lin.split()
            if not cols:
                continue
            nsamp = int(cols[0])
            lnLike.extend([-float(cols[1])] * nsamp)
            rows.extend([[float(c) for c in cols[2:]]] * nsamp)
    return np.array(lnLike), np.array(rows)


def key(filename):
    return filename.rsplit('_', 1)[0]
