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


# proposal width of the orientation parameter: position-angle form / (x, y) form of the model
_ANGLE_STEP = {False: 0.6, True: 0.2}


def _start_and_steps(mod, bmap, angle_step):
    """Starting point and proposal widths of one map, in the model's packaged layout
    (centre(2), widths(2), orientation, amplitude, (offset, cos theta, phi)).  What each entry is
    made of is the reference's recipe (:160-216): a tenth of the prior box for centre and widths;
    the map's own moments, the widths held inside [sigMin, sigMax]; half the data range for the
    amplitude, and twice the range of the pixels within one width of the centre as its step - wide
    enough to cross from the no-beam mode to the beam."""
    box = np.subtract(mod.centerMax, mod.centerMin)
    tenth = (box[0] / 10, box[1] / 10)
    cx, cy, vx, vy = bmap.stats(sigcut=0.0)[:4]                       # device: k_beam_stats
    widths = np.clip(np.sqrt([vx, vy]), mod.sigMin, mod.sigMax)
    core = bmap.d[(np.abs(bmap.x - cx) < widths[0]) & (np.abs(bmap.y - cy) < widths[1])]
    start = ((cx, cy), widths, 0, (bmap.d.max() - bmap.d.min()) / 2, (0, 1, 0))
    steps = (tenth, tenth, angle_step, 2 * (core.max() - core.min()), (0.3, 0, 0))
    return start, steps


def setup_sampler(data, xyrange, useNormalizedBeam=False, rangeScale=None, sigminmax=(3, 8),
                  verbose=True):
    """-> (like, prop_sigmas, start_params) for the maps in `data` (a list of BeamData), the
    centre prior spanning `xyrange` (a data set, or sequences of coordinates).  Interface and
    results of the reference's :124-236 (pinned by tests/golden/setup_sampler_g12.json); the prior
    box, the width limits and the parameter blocks are class state of the model, as there."""
    if not useNormalizedBeam:
        raise NotImplementedError(
            "setup_sampler(useNormalizedBeam=False): GaussianBeamModel2D with the amplitude-"
            "marginalising base Likelihood has no device path; use useNormalizedBeam=True "
            "(OffsetNormalizedBeamModel + NormalizedBeamLikelihood, the reference's beam runs)")
    np.set_printoptions(precision=4, linewidth=150, suppress=True)
    mod = OffsetNormalizedBeamModel.OffsetNormalizedBeamModel
    like = NormalizedBeamLikelihood.NormalizedBeamLikelihood(data=data, model=mod)
    mod.setxyRange(xyrange, scale=rangeScale)
    mod.sigMin, mod.sigMax = sigminmax
    mod.paramBlocks, mod.nBlock = list(range(7)), 7               # the gradient (mu, phi) stays fixed
    like.setdata(data)                                            # device snapshot of the prior
    start_params, prop_sigmas = _start_and_steps(mod, data[0], _ANGLE_STEP[bool(OffsetNormalizedBeamModel.use_xy)])
    if verbose:
        print('setting sigMin, sigMax=', mod.sigMin, mod.sigMax)
        print('center min, max=', mod.centerMin, mod.centerMax)
        print("averaging over all data for starting point")
        for title, pars in (("Starting point:  ", start_params), ("Starting sigmas: ", prop_sigmas)):
            print((title + mod.fmtstring) % tuple(mod.unpackage(pars)))
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
