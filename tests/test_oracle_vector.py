"""The vectorised CPU sampler of the bench's "best-effort CPU" line equals the one-chain-at-a-time
restatement (oracle/refcpu.Chain, itself pinned to the reference by tests/test_oracle_golden.py)
draw for draw."""
import numpy as np

from oracle import refcpu as R

SIG = np.array([0.3, 0.4, 3.0, 0.1, 0.4, 5.0])
START = np.array([1.0, 2.0, 10.0, 0.1, 2.0, 20.0])


def test_vector_chains_equal_scalar_chains():
    S, steps = 5, 240
    nu, d, sig, _ = R.synth_sed_catalogue(S)
    rng = np.random.default_rng(3)
    z = rng.standard_normal((steps, S, 6))
    u = rng.uniform(size=(steps, S))
    vc = R.VectorChainsM2N(nu, d, sig, SIG, START)
    tr = np.empty((steps, S, 6))
    for t in range(steps):
        vc.step(z[t], u[t])
        tr[t] = vc.theta
    nswaps = 0
    for s in range(S):
        ch = R.Chain(R.sed_problem(R.M2N, R.GaussData(nu[s], d[s], sig[s])), SIG, START,
                     R.InjectedDraws(z[:, s], u[:, s]), do_block=True, randomize_start=False)
        ch.run(steps)
        ref = np.asarray(ch.samples)[1:]
        np.testing.assert_allclose(tr[:, s], ref, rtol=1e-12, atol=0)
        assert ch.naccept == vc.naccept[s]
        np.testing.assert_allclose(vc.lnPr[s], ch.lnPr[-1], rtol=1e-11)
        nswaps += int(np.sum(np.abs(np.diff(ref[:, 2])) > 8))
    assert vc.naccept.min() > 10


def test_vector_rate_worker_runs():
    done, t = R.vector_m2n_rate((0, 4, 8, 0.2, SIG, START))
    assert done > 0 and t > 0
