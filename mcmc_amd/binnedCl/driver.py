"""The band-power run: bins and shape spectrum -> binnedClModel, data sets -> binnedClLikelihood,
the staged sampler, and the products of a finished run (band powers, their covariance and
correlation, Fisher band-power windows, optional offset-lognormal x factors).  Interface of the
reference's run script MCMC/test_binnedCl.py (`main` :33-336, `getlike` :340-449, `getsamples`
:452-455); its figures are not rebuilt, everything it computes, prints and writes is.

Sampling runs on the device (step-wise engine: `MCMC.sampler` with the Cl likelihood kernels);
the products are read off the last stage's samples on the host, as in the reference.
Parity: the reference script imports hotshot at module level and cannot be imported, so `main`
is parity unpinned as a whole; every function it calls (setBinning, bandpowers, ClCovar, covar,
FisherWindows, fitOffsetLognormal, sampler, printvals) is pinned on its own (G3, G5, G8, G9)."""
import numpy as N
from numpy import array, ones, zeros, float64, sqrt

from .. import MCMC, getdist_ahj
from ..ClData import ClData
from ..ClData.ClData import readbins
from .binnedClLikelihood import binnedClLikelihood
from .binnedClModel import FisherWindows, binnedClModel, fitOffsetLognormal

filename = "data_list.txt"                                           # :27


def _read_shapes(shapefile):
    """l(l+1)C_l/2pi of the fiducial model, columns l TT EE TE .. (:47-58)"""
    tmp = N.fromfile(shapefile, sep=" ")
    tmp.shape = -1, 6
    ells = N.int_(tmp.T[0])                       # does not usually start at l = 0
    max_ell = max(ells)
    llClTT = N.zeros(max_ell + 1)
    llClEE = N.zeros(max_ell + 1)
    llClTE = N.zeros(max_ell + 1)
    llClTT[ells] = tmp.T[1]
    llClEE[ells] = tmp.T[2]
    llClTE[ells] = tmp.T[3]
    return llClTT, llClTE, llClEE


def _edges_to_bins(binlist):
    """[lo0, lo1, ..., end] -> [(lo0, lo1-1), ...]: "non-pythonic: beginning and end" (:117-121)"""
    out = []
    for edges in binlist:
        edges = list(edges)
        out.append([(b, edges[i + 1] - 1) for i, b in enumerate(edges[:-1])])
    return out


def main(nMC=(1000,), gridPlot=True, testshape=True, no_pol=False, data=None, bins=None,
         rotateParams=True, binfile=None, prefix=None, filename=filename, get_x=False,
         shapefile="CarloClModel.dat", seed=None, verbose=True):
    """-> what MCMC.sampler returns: (list of stages, (means, stdevs, normalised covariance)).
    `bins`: per spectrum (TT[, TE, EE]) a list of bin edges [lo0, lo1, ..., end]; `binfile`: bins
    in Contaldi's format instead; `prefix`: write prefix.bp[te|ee], .covar, .corr and the window
    files prefix1, prefix2, ...; `gridPlot` only selects which summary is printed."""
    N.set_printoptions(precision=4, linewidth=150, suppress=True)
    mod = binnedClModel
    llClTT, llClTE, llClEE = _read_shapes(shapefile)
    if data is None:
        data = ClData.getClData(filename, no_pol=no_pol)
    if binfile is None:
        binsTT = [101, 201, 301, 401, 501, 601, 701, 801, 901, 1031, 1151, 1271, 1391, 1511, 1651,
                  1851, 2051]
        binsTE = [164, 326, 488, 650, 812, 974, 1136, 1540]
        binsEE = [164, 326, 488, 650, 812, 974, 1136, 1540]
        binlist = [binsTT, binsTE, binsEE] if not no_pol else [binsTT]
        if bins is not None:
            binlist = bins
        bins = _edges_to_bins(binlist)
    else:
        bins = readbins(binfile)
        if no_pol:
            bins = [bins[0]]
    shapelist = array([llClTT, llClTE, llClEE]) if not no_pol else array([llClTT])
    nbins = [len(b) for b in bins]
    npar = sum(nbins)
    if not testshape:
        raise NotImplementedError("main(testshape=False): the reference's branch reads an undefined "
                                  "name (test_binnedCl.py:141)")
    llClshape = shapelist.copy()
    start_params = ones(shape=(npar,), dtype=float64)
    prop_sigmas = zeros(shape=(npar,), dtype=float64) + 0.1
    prop_sigmas[nbins[0]:] = 1.0

    mod.setBinning(bins, shapefun=llClshape)
    ell = [array(b).mean(axis=1) for b in bins]
    if verbose:
        print('bins:')
        print(ell)
        print(mod.nbins)
    like = binnedClLikelihood(data=data, model=mod)
    fac = 2.4 / sqrt(len(ell))      # the reference's "equiv to fac=None": len(ell) = number of spectra
    retval = MCMC.sampler(like, nMC, prop_sigmas, start_params, fac=fac, noCorrelations=True,
                          doBlock=True, rotateParams=rotateParams, seed=seed, verbose=verbose)

    s = retval[0][-1]
    mean = mod.bandpowers(s.mean())
    stdv = mod.bandpowers(s.stdev())
    covar = s.covar(unNormalized=True)
    Clcovar = mod.ClCovar(covar)
    corr = s.covar()
    WBl = FisherWindows(Clcovar, bins=bins, isCovar=True)
    x = []
    if get_x:
        for samples in s.samples.T:
            x.append(fitOffsetLognormal(samples, full_output=0, do_plot=0))

    ibin = 0
    if prefix is not None:
        for es, ms, ss, suf in zip(ell, mean, stdv, [".bp", ".bpte", ".bpee"]):
            with open(prefix + suf, "w") as f:
                for ell1, mean1, stdv1 in zip(es, ms, ss):
                    if get_x:
                        print(ell1, mean1, stdv1, stdv1, x[ibin], file=f)
                        ibin += 1
                    else:
                        print(ell1, mean1, stdv1, stdv1, file=f)
        with open(prefix + ".covar", "w") as f:
            N.savetxt(f, Clcovar, fmt="%f")
        with open(prefix + ".corr", "w") as f:
            N.savetxt(f, corr, fmt="%f")
        for iwin, win in enumerate(WBl):
            with open(prefix + str(iwin + 1), "w") as f:
                for l, Wl in enumerate(win.T):
                    print(l, *[Wl[k] for k in range(len(Wl))], file=f)

    if verbose:
        if gridPlot:
            getdist_ahj.histgrid(s)
        else:
            getdist_ahj.printvals(s)
        for l, m, e in zip(ell, mean, stdv):
            for l1, m1, s1 in zip(l, m, e):
                print('%d %f %f' % (int(l1), m1, s1))
    retval[0][-1].products = dict(ell=ell, mean=mean, stdv=stdv, Clcovar=Clcovar, corr=corr, WBl=WBl, x=x)
    return retval


def getlike(ibins=[1], data=None, bins=None, filename=filename, shapefile="CarloClModel.dat",
            verbose=True):
    """The likelihood as a function of the power in single bins, the others at the shape
    spectrum: -> (data, like, likearr of the last scanned bin).  :340-449"""
    mod = binnedClModel
    llClTT, _, _ = _read_shapes(shapefile)
    if data is None:
        data = ClData.getClData(filename, no_pol=True)
    if bins is None:
        bins = [1, 1851, 1951, len(llClTT) - 1]
    bins = _edges_to_bins([bins])[0]
    npar = len(bins)
    start_params = ones(shape=(npar,), dtype=float64)
    if verbose:
        print('Start:')
        print(start_params)
    mod.setBinning([bins], shapefun=llClTT)        # one spectrum (the reference passes the flat list)
    ell = [(b[0] + b[1]) / 2 for b in bins]
    if verbose:
        print('bins:')
        print(ell)
    like = binnedClLikelihood(data=data, model=mod)
    likearr = None
    for ibin in ibins:
        bps = N.linspace(0, 5 * start_params[ibin], 100)
        lnlikearr = N.zeros_like(bps)
        for i, bp in enumerate(bps):
            pars = start_params.copy()
            pars[ibin] = bp
            lnlikearr[i] = like.lnLike(pars)
        lnlikearr[N.logical_not(N.isfinite(lnlikearr))] = min(lnlikearr[N.isfinite(lnlikearr)]) - 10.0
        lnlikearr = lnlikearr - max(lnlikearr)
        likearr = N.exp(lnlikearr)
        avg = (bps * likearr).sum() / likearr.sum()
        if verbose:
            print('<par>=%f' % (avg))
    return data, like, likearr


def getsamples(res):
    """the samples of all stages of a main() run, [nparam, nsample] (:452-455)"""
    return (N.concatenate([s.samples for s in res[0]])).T
