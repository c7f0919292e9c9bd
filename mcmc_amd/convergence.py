"""Gelman-Rubin diagnostics with the interface of the reference's MCMC/convergence.py:58-87.

gelmanrubin_MCMC keeps the reference definition: W = mean over chains of the ddof=0 variances,
B = ddof=1 variance of the chain means (not scaled by the chain length), Rhat = sqrt((W+B)/W)
over the parameters with W>0.  The reference's gelmanrubin() (convergence.py:9-53) raises
NameError (`stride`) and is not provided."""
import numpy as np

from .engine import default_engine


def gelmanrubin_MCMC(MCMCs, burn=0, stride=1, verbose=False):
    """MCMCs: a list of MCMC objects (one chain each), or a single MCMC object with nchains>1
    whose chains already live side by side on the device (then burn=1, stride=1 is served by the
    on-device moments and mi355_rhat)."""
    if not isinstance(MCMCs, (list, tuple)):
        m = MCMCs
        if m._device_moments_valid(burn, stride):
            rhat = m._eng.rhat()[0]
            return rhat[~np.isnan(rhat)]
        means = np.asarray(m.mean(burn, stride))
        var = np.power(np.asarray(m.stdev(burn, stride)), 2)
    else:
        means = np.array([np.asarray(m.mean(burn, stride)) for m in MCMCs])
        var = np.array([np.power(np.asarray(m.stdev(burn, stride)), 2) for m in MCMCs])
    rhat = default_engine().gelman_rubin(means, var)
    return rhat[~np.isnan(rhat)]
