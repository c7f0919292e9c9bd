"""mi355_rhat_noalloc_async (the entry point that allocates nothing: intermediates in a fixed
device array of the library) against mi355_rhat, through the C ABI: bit for bit, on two calls in
a row, the second with a smaller input, and the size limit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(eng, S, C, nsteps, seed):
    import torch
    from mcmc_amd import synth
    nu, d, sig, _ = synth.sed_catalogue(S, seed=seed)
    eng.sed_load("M2N", np.full(S, 8, np.int32), nu, d, sig)
    eng.mh_init(C, synth.M2N_START, synth.M2N_PROP_SIGMAS, param_blocks=np.arange(6))
    eng.mh_run(nsteps)
    want = eng.rhat()                                      # [S, P] on the host (allocating path)
    out = torch.full((6, S), -1.0, dtype=torch.float64, device="cuda")
    eng.rhat_noalloc(out.data_ptr())
    eng.sync()
    got = out.cpu().numpy().T
    assert np.all(np.isfinite(want)) and np.all(want >= 1.0)
    assert np.array_equal(got, want)
    return want


def test_noalloc_rhat_equals_rhat_twice_second_call_smaller():
    from mcmc_amd import Engine
    with Engine(seed=31) as eng:
        big = _run(eng, 300, 16, 150, seed=4)
        small = _run(eng, 41, 8, 90, seed=9)               # smaller input: stale scratch beyond it
        assert big.shape == (300, 6) and small.shape == (41, 6)


def test_noalloc_rhat_rejects_sizes_beyond_the_fixed_scratch():
    import torch
    from mcmc_amd import Engine, Mi355Error, synth
    S = 1048576 // 6 + 1
    nu, d, sig, _ = synth.sed_catalogue(8, seed=1)
    rep = (S + 7) // 8
    with Engine(seed=2) as eng:
        eng.sed_load("M2N", np.full(S, 8, np.int32), np.tile(nu, (rep, 1))[:S], np.tile(d, (rep, 1))[:S],
                     np.tile(sig, (rep, 1))[:S])
        eng.mh_init(2, synth.M2N_START, synth.M2N_PROP_SIGMAS, param_blocks=np.arange(6))
        eng.mh_run(4)
        out = torch.empty((6, S), dtype=torch.float64, device="cuda")
        with pytest.raises(Mi355Error, match="fixed scratch holds"):
            eng.rhat_noalloc(out.data_ptr())
        r = eng.rhat()                                     # the allocating entry point has no limit
        assert r.shape == (S, 6) and np.isfinite(r).any()  # (NaN where no chain of a source moved)
