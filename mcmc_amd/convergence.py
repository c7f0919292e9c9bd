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


def gelmanrubin(chains_complete, burn=0, thin=1, verbose=False):
    """Gelman-Rubin R-hat of a list of chains [n_sample_c, n_param] (lengths may differ), the
    ragged-chain form of MCMC/convergence.py:9-53: within-chain variance debiased (ddof=1) in the
    denominator, ddof=0 in the numerator, between-chain variance of the chain means with ddof=1.
    The reference's function raises NameError (it slices with an undefined `stride`, :34); the
    argument it documents, `thin`, is used here.  Host-side diagnostic on arrays the caller
    already holds (parity unpinned for that reason; tests/test_likelihood_base.py checks the
    formula); the samplers' own R-hat is gelmanrubin_MCMC above / mi355_rhat on the device."""
    n_chain = len(chains_complete)
    n_samples = [chain.shape[0] for chain in chains_complete]
    n_params = [chain.shape[1] for chain in chains_complete]
    n_param = n_params[0]
    assert all(n == n_param for n in n_params[1:])
    if verbose:
        print("n_chain = %d" % n_chain)
        print("n_param = %d" % n_param)
        print("n_samples = ", n_samples)
        print("Burn-in fraction = %f; thinning = %d" % (burn, thin))
    chains = [np.asarray(c)[burn::thin] for c in chains_complete]
    within = np.array([np.var(c, axis=0, ddof=1) for c in chains]).mean(axis=0)
    within_term = np.array([np.var(c, axis=0, ddof=0) for c in chains]).mean(axis=0)
    between_term = np.array([np.mean(c, axis=0) for c in chains]).var(axis=0, ddof=1)
    return np.sqrt((within_term + between_term) / within)
