"""Convergence on the C2 workload (two-component greybody, 8 bands, 64 chains per source): with
the reference's adaptive staging (newMCMC: sigma = fac * stdev, Cholesky factor of the chain
covariance, MCMC.py:343-413; Proposal.py:150-154) the chains of a source agree - Gelman-Rubin
R-hat (convergence.py:58-87) below 1.1 - after 5e4 steps.  Runs in the fused SED kernel's extended
instantiation; the single-parameter block updates of the production driver need ~2e5 steps for the
same (tools/rhat_explore.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_correlated_stages_converge_on_c2_sources():
    from mcmc_amd import Engine, synth
    S, C = 256, 64
    nu, d, sig, _ = synth.sed_catalogue(S)
    with Engine(seed=11) as eng:
        eng.sed_load("M2N", np.full(S, 8, np.int32), nu, d, sig)
        eng.mh_init_problem("sed", C, synth.M2N_START, synth.M2N_PROP_SIGMAS, param_blocks=np.arange(6),
                            randomize_start=True, track_cov=True)
        assert eng.mh_fused_active
        eng.mh_run(2000)                                   # stage 0: the driver's widths, doBlock
        eng.mh_adapt_corr(no_correlations=True)
        eng.mh_run(4000)
        for n in (2000, 10000):                            # correlated stages, all parameters at once
            eng.mh_adapt_corr(no_correlations=False, do_block=False, fac=0.6)
            eng.mh_run(n)
        eng.mh_adapt_corr(no_correlations=False, do_block=False, fac=0.6)
        eng.mh_track_cov(False)
        eng.mh_run(50000)
        rhat = eng.rhat()
        acc = eng.mh_moments()["naccept"].mean() / 50001.0
    worst = np.nanmax(rhat, axis=1)
    print("median R-hat %.3f, median of the worst parameter %.3f, sources below 1.1: %.2f, acceptance %.2f"
          % (np.nanmedian(rhat), np.nanmedian(worst), (worst < 1.1).mean(), acc))
    assert np.nanmedian(rhat) < 1.1
    assert np.nanmedian(worst) < 1.1
    assert (worst < 1.1).mean() > 0.5
    assert 0.05 < acc < 0.6
