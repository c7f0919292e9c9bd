"""The beam-fit driver: set up likelihood, prior box, starting point and proposal widths from a
map, and run the staged sampler.  Interface of the reference's MCMC/BeamFit/ProcessBeamData.py
(`setup_sampler` :124-236, `sample1beam` :255-287); its plotting helpers (plot, plotMod, plotter,
regrid, get_likelihood_grid: matplotlib user interface) are not rebuilt.

useNormalizedBeam=True - OffsetNormalizedBeamModel + NormalizedBeamLikelihood - is the
configuration the reference's beam runs use (and the only one of the two whose likelihood the
reference can evaluate: the plain GaussianBeamModel2D branch pairs an un-normalised beam with the
amplitude-marginalising base Likelihood); the device path implements that configuration."""
import numbers

import numpy as np

from .. import MCMC
from . import BeamModel, NormalizedBeamLikelihood, OffsetNormalizedBeamModel


class DataSizeError(Exception):                                      # :30-31
    pass


def setup_sampler(data, xyrange, useNormalizedBeam=False, rangeScale=None, sigminmax=(3, 8),
                  verbose=True):
    """-> (like, prop_sigmas, start_params) for the maps in `data` (a list of BeamData), the
    centre prior spanning `xyrange` (a data set, or sequences of coordinates).  :124-236"""
    np.set_printoptions(precision=4, linewidth=150, suppress=True)
    if not useNormalizedBeam:
        raise NotImplementedError(
            "setup_sampler(useNormalizedBeam=False): GaussianBeamModel2D with the amplitude-"
            "marginalising base Likelihood has no device path; use useNormalizedBeam=True "
            "(OffsetNormalizedBeamModel + NormalizedBeamLikelihood, the reference's beam runs)")
    mod = OffsetNormalizedBeamModel.OffsetNormalizedBeamModel
    like = NormalizedBeamLikelihood.NormalizedBeamLikelihood(data=data, model=mod)
    use_xy = OffsetNormalizedBeamModel.use_xy

    mod.setxyRange(xyrange, scale=rangeScale)     # class variables: the prior of all instances
    mod.sigMin, mod.sigMax = sigminmax
    like.setdata(data)                            # (the device snapshot of the prior follows)
    if verbose:
        print('setting sigMin, sigMax=', mod.sigMin, mod.sigMax)
    dx = (mod.centerMin[0], mod.centerMax[0])
    dy = (mod.centerMin[1], mod.centerMax[1])
    delx = dx[1] - dx[0]
    dely = dy[1] - dy[0]
    if verbose:
        print('center min, max=', mod.centerMin, mod.centerMax)

    prop_sigmas = ((delx / 10, dely / 10), (delx / 10, dely / 10), 0.6)
    if use_xy:
        prop_sigmas = ((delx / 10, dely / 10), (delx / 10, dely / 10), 0.2)
    if verbose:
        print("averaging over all data for starting point")
    stats = data[0].stats(sigcut=0.0)             # device: k_beam_stats
    startsigs = np.minimum([np.sqrt(stats[2]), np.sqrt(stats[3])], mod.sigMax)
    startsigs = np.maximum(startsigs, mod.sigMin)
    start_params = ((stats[0], stats[1]), startsigs, 0)

    # amplitude: a proposal wide enough to "jump the barrier" (:199-206)
    x = data[0].x - stats[0]
    y = data[0].y - stats[1]
    d = data[0].d
    dctr = d[np.logical_and(abs(x) < startsigs[0], abs(y) < startsigs[1])]
    prop_sigmas += 2 * (max(dctr) - min(dctr)),
    prop_sigmas += (0.3, 0, 0),                   # offset, gradient [cos(theta), phi]
    start_params += (max(data[0].d) - min(data[0].d)) / 2,
    start_params += (0, 1, 0),

    mod.paramBlocks = [0, 1, 2, 3, 4, 5, 6]       # offset only: mu, phi stay fixed (:220-221)
    mod.nBlock = 7
    if verbose:
        print(("Starting point:  " + mod.fmtstring) % tuple(mod.unpackage(start_params)))
        print(("Starting sigmas: " + mod.fmtstring) % tuple(mod.unpackage(prop_sigmas)))
    return like, prop_sigmas, start_params


def sample1beam(like, prop_sigmas, start_params, nMC=(1000, 1000), fac=None, plotRes=None,
                noCorrelations=False, rotateParams=False, doBlock=False, **kw):
    """the staged sampler on one beam: (list of MCMC stages, (means, stdevs, normalised
    covariance) of the last).  :255-279 (rotateParams is accepted and, as there, not passed on)"""
    if isinstance(nMC, numbers.Number):
        nMC = (nMC, nMC)
    if plotRes is not None:
        raise NotImplementedError("sample1beam(plotRes=...): plotting is not rebuilt")
    return MCMC.sampler(like, nMC, prop_sigmas, start_params, fac=fac,
                        noCorrelations=noCorrelations, doBlock=doBlock, **kw)
