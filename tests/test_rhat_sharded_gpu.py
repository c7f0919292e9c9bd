"""The device side of the chain-sharded Gelman-Rubin (mi355_rhat_partial_async / _finish_async,
mcmc_amd/distributed.py): three contexts on one MI355X play three ranks that hold the same sources
and their own chains; the sum over contexts stands in for the RCCL all-reduce (whose arithmetic
tests/test_distributed_cpu.py covers over gloo).  Reference: MCMC/convergence.py:58-87 over the
union of the chains."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_chain_sharded_rhat_kernels_equal_the_formula_over_all_chains():
    import torch
    from mcmc_amd import Engine, synth
    from mcmc_amd import distributed as D
    S, P, ranks = 257, 6, 3
    C_of = [24, 24, 8]                                        # ranks need not hold equally many chains
    nu, d, sig, _ = synth.sed_catalogue(S, seed=4)
    nb = np.full(S, nu.shape[1], np.int32)
    dev = torch.device("cuda", 0)
    engs = []
    try:
        for r in range(ranks):
            e = Engine(device=0, seed=0xABC + 7919 * r)
            e.sed_load("M2N", nb, nu, d, sig)
            e.mh_init(C_of[r], synth.M2N_START, synth.M2N_PROP_SIGMAS, param_blocks=np.arange(P),
                      randomize_start=True)
            e.mh_run(300)
            engs.append(e)
        Ctot = sum(C_of)

        def summed(pass_, gmean, shape):
            tot = torch.zeros(shape, dtype=torch.float64, device=dev)
            for e in engs:
                out = torch.empty(shape, dtype=torch.float64, device=dev)
                torch.cuda.synchronize(dev)
                e.rhat_partial(pass_, out.data_ptr(), gmean.data_ptr() if gmean is not None else None)
                e.sync()
                tot += out
            return tot

        s0 = summed(0, None, (P, S))
        gmean = s0 / float(Ctot)
        s1 = summed(1, gmean, (2, P, S))
        out = torch.empty((P, S), dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)
        engs[0].rhat_finish(s1.data_ptr(), Ctot, out.data_ptr())
        engs[0].sync()
        got = out.t().cpu().numpy()

        moms = [e.mh_moments() for e in engs]
        mean = np.concatenate([m["mean"] for m in moms], axis=1)      # [S, Ctot, P]
        var = np.concatenate([m["var"] for m in moms], axis=1)
        W = var.mean(axis=1)
        B = mean.var(axis=1, ddof=1)
        ref = np.sqrt((W + B) / W)
        np.testing.assert_allclose(got, ref, rtol=1e-10)
        # one rank: the sharded path equals mi355_rhat
        one = D.engine_rhat(engs[1]).cpu().numpy()
        np.testing.assert_allclose(one, engs[1].rhat(), rtol=1e-12)
    finally:
        for e in engs:
            e.close()


@pytest.mark.parametrize("C", [2, 7, 16, 17, 32, 33, 64, 100, 130])
def test_rhat_over_chain_counts_on_either_side_of_the_lane_groups(C):
    """k_rhat_partial gives a source 16, 32 or 64 lanes by its chain count (two or four sources share a load
    below 33 / 17 chains) and a wave several sources; source counts that do not fill the last wave.
    Reference: MCMC/convergence.py:71-85 on the chains' means and variances."""
    from mcmc_amd import Engine, synth
    S, P = 203, 6
    nu, d, sig, _ = synth.sed_catalogue(S, seed=6)
    nb = np.full(S, nu.shape[1], np.int32)
    with Engine(seed=0x51 + C) as e:
        e.sed_load("M2N", nb, nu, d, sig)
        e.mh_init(C, synth.M2N_START, synth.M2N_PROP_SIGMAS, param_blocks=np.arange(P), randomize_start=True)
        e.mh_run(150)
        got = e.rhat()
        mom = e.mh_moments()
    W = mom["var"].mean(axis=1)
    B = mom["mean"].var(axis=1, ddof=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        ref = np.where(W > 0.0, np.sqrt((W + B) / W), np.nan)          # :81-85 (a parameter that never moved)
    np.testing.assert_allclose(got, ref, rtol=1e-10, equal_nan=True)
    assert np.isfinite(got).mean() > 0.9
