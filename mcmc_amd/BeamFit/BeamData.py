"""Beam-map data container with the interface of the reference's BeamData
(MCMC/BeamFit/BeamData.py:11-92): GaussianData plus pixel positions.  stats() runs on the GPU."""
import numpy as np

from ..GaussianData import DataSizeError, GaussianData


class BeamData(GaussianData):
    def __init__(self, xarr, yarr, data, sigma, cts=None):           # :18-29
        GaussianData.__init__(self, data, sigma)
        xarr, yarr = np.asarray(xarr), np.asarray(yarr)
        if self.n != yarr.size or self.n != xarr.size:
            raise DataSizeError(self.n, xarr.size, yarr.size)
        self.x = np.asarray(xarr, np.float64)
        self.y = np.asarray(yarr, np.float64)
        if cts is not None:
            self.cts = cts
        self.version = 0          # bumped when stats() shifts the map: device copies are reloaded

    def stats(self, sigcut=None, do_abs=False):                      # :35-71
        """(<x>, <y>, var x, var y) weighted by the mean-subtracted data.  As in the reference, the
        subtraction happens IN PLACE when do_abs is false (`dd = self.d; dd -= offset`, :45-52, with
        the author's own "IS THIS ACTUALLY SHIFTING THE self.d???"): every likelihood evaluation
        after setup_sampler sees the mean-subtracted map (and so does the caller's array, if it was
        float64 already).  Pinned by G12 (tests/golden/setup_sampler_g12.json)."""
        from ..engine import Engine
        if not do_abs:
            self.d -= self.d.mean()
            self.version += 1
        with Engine() as eng:
            eng.beam_load(self.x.ravel(), self.y.ravel(), self.d.ravel(), self.sig)
            return eng.beam_stats(0, sigcut=sigcut, do_abs=do_abs)


def BeamSim(beam_model, n, sigma=1.0, amplitude=1.0, xarr=None, yarr=None, data=None, xrng=None,
            yrng=None, rng=None):
    """Simulated beam data, :75-92 (numpy Generator instead of the global MT19937)."""
    rng = np.random.default_rng() if rng is None else rng
    xrng = (-1, 1) if xrng is None else xrng
    yrng = (-1, 1) if yrng is None else yrng
    if xarr is None:
        xarr = rng.uniform(xrng[0], xrng[1], n)
    if yarr is None:
        yarr = rng.uniform(yrng[0], yrng[1], n)
    if data is None and sigma is not None:
        data = amplitude * beam_model.atxy(xarr, yarr) + rng.normal(0, sigma, n)
    return BeamData(xarr, yarr, data, sigma)
