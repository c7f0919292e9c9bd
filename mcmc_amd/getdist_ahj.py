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
    try:
        s = MCMC.samples
        if derived and MCMC.nDerived:
            s = np.hstack([s, MCMC.derived])
    except AttributeError:
        s = MCMC
    if 0 < burn < 1:
        burn = int(burn * len(s))
    return s[burn::stride]


def _max_like(MCMC, s, lnLike):
    """The best sample and the reference's consistency check of the stored lnPr against a fresh
    evaluation (getdist_ahj.py:54-65, :188-199); returns (lnLike, maxlnLike, maxLikeParams)."""
    try:
        if lnLike is None:
            lnLike = MCMC.lnPr
    except AttributeError:
        lnLike = None
    if lnLike is None:
        return None, None, None
    maxlnLike = max(lnLike)
    maxLikeParams = s[np.argmax(lnLike)]
    maxLProb = MCMC.like.lnLike(maxLikeParams) + np.log(MCMC.like.model.prior(*maxLikeParams))
    maxLchi2 = MCMC.like.chi2(maxLikeParams)
    return lnLike, maxlnLike, (maxLikeParams, maxLProb, maxLchi2)


def printvals(MCMC, params=None, lnLike=None, derived=True):
    """Print the best sample and mean +- stdev (ddof=0) of every parameter; returns
    (maxlnLike, maxLikeParams).  getdist_ahj.py:44-76."""
    s = getallsamples(MCMC, derived=derived)
    lnLike, maxlnLike, best = _max_like(MCMC, s, lnLike)
    if lnLike is not None:
        maxLikeParams, maxLProb, maxLchi2 = best
        print("chi2 at max = %f" % (maxLchi2,))
        assert abs(1. - maxLProb / maxlnLike) < 0.01, \
            "different likelihood calculations differ: %f != %f" % (maxLProb, maxlnLike)
        print('Max ln likelihood %f at parameters:' % maxlnLike)
        print(maxLikeParams)
    else:
        # the reference goes on to use maxLikeParams and fails with NameError (:70)
        raise ValueError("printvals needs lnLike (or a sampler object with .lnPr)")
    if params is None:
        params = range(s.shape[1])
    for param, ML in zip(params, maxLikeParams):
        s1 = s.T[param]
        print('mean = %g +- %g | ML = %g' % (s1.mean(), s1.std(), ML))
    return (maxlnLike, maxLikeParams)


def histgrid(MCMC, params=None, nbins=30, labels=None, lnLike=None, quiet=False, derived=True,
             maxlnLike=[], noPlot=False, burn=0, stride=1):
    """The numeric part of getdist_ahj.histgrid (:175-205): the best sample over ALL samples (burn
    and stride only select what would be drawn), the consistency check, the same printed lines;
    returns (maxlnLike, maxLikeParams), or None for a bare array without lnLike.  Nothing is
    drawn, whatever `noPlot` says."""
    s = getallsamples(MCMC, derived=derived)
    lnLike, maxlnLike, best = _max_like(MCMC, s, lnLike)
    if lnLike is None:
        return None
    maxLikeParams, maxLProb, maxLchi2 = best
    print("chi2 at max = %f" % maxLchi2)
    print('Max ln likelihood %f at parameters:' % maxlnLike)
    assert abs(1. - maxLProb / maxlnLike) < 0.01, \
        "different likelihood calculations differ: %f != %f" % (maxLProb, maxlnLike)
    print(maxLikeParams)
    return (maxlnLike, maxLikeParams)


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
