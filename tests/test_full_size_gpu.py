"""BASELINE.json's configurations at their full sizes on one MI355X, through the C ABI.

The oracle cannot run these sizes in seconds, so each case checks the whole result through
properties that do not depend on the size (every chain is its own Markov chain keyed by its
global id; a likelihood does not depend on what else is in the batch) and the oracle itself on
a strided sample of the units that includes the first and the last one.
  C2  submmSED two-component greybody, 1e5 sources x 64 chains
  C3  one GPU's slice of 1e6 sources x 256 chains over 8 GPUs: 1e6 sources x 32 chains
  C4  binnedCl offset-lognormal, full covariance, 4096 chains, nb=256, 11 x 11 marginalisation
  C5  BeamFit, 1e4 maps of 256 x 256 pixels, 16 chains per map
"""
import numpy as np
import pytest

from oracle import refcpu as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("S,C", [(100000, 64), (1000000, 32)], ids=["C2", "C3-slice-of-one-GPU"])
def test_sed_full_size_chains_match_oracle_on_a_sample(S, C):
    """C2 (1e5 sources x 64 chains) and what one of 8 GPUs holds of C3 in the chain-sharded layout
    (all 1e6 sources x 32 of the 256 chains: 3.2e7 chains, 7.3 GB of chain state).
    Free-running chains, 20 block cycles: every 9973rd chain (first and last included)
    equals the vectorised oracle (tests/test_oracle_vector.py pins it to the scalar restatement)
    on the same Philox draws - decisions identical, position / lnPr to 1e-12 - and its running
    moments; R-hat of all sources equals the NumPy formula on the device's own moments."""
    from mcmc_amd import Engine, synth
    P, nsteps = 6, 120
    nu, d, sig, truth = synth.sed_catalogue(S)
    nb = np.full(S, nu.shape[1], np.int32)
    rng = np.random.default_rng(11)
    theta0 = truth[:, None, :] * rng.uniform(0.9, 1.1, (S, C, P))
    theta0 = theta0.reshape(S * C, P)
    swap = theta0[:, 2] > theta0[:, 5]
    theta0[swap] = theta0[swap][:, [3, 4, 5, 0, 1, 2]]
    theta0 = theta0.reshape(S, C, P)
    sigma = np.array([0.02, 0.05, 0.6, 0.02, 0.05, 1.5])
    pick = np.unique(np.concatenate([np.arange(0, S * C, 9973), [C - 1, C, S * C - 1]]))
    with Engine(seed=0xC2F0115) as eng:
        eng.sed_load(R.M2N, nb, nu, d, sig)
        eng.mh_init(C, theta0, sigma, param_blocks=np.arange(P), randomize_start=False)
        step0 = eng.step_total
        m0 = eng.mh_moments()
        eng.mh_run(nsteps)
        mom = eng.mh_moments()
        rhat = eng.rhat()
        seed = eng.seed
    src = pick // C
    vc = R.VectorChainsM2N(nu[src], d[src], sig[src], sigma, theta0.reshape(-1, P)[pick][0])
    vc.theta = theta0.reshape(-1, P)[pick].copy()
    vc.lnPr = vc.lnlike(vc.theta)
    np.testing.assert_allclose(m0["lnPr"].reshape(-1)[pick], vc.lnPr, rtol=1e-12, atol=1e-14)
    traj = []
    for t in range(step0, step0 + nsteps):
        z = np.zeros((len(pick), P))
        z[:, vc.iblock] = R.philox_normal(seed, pick, np.full(len(pick), t))
        vc.step(z, R.philox_uniform(seed, pick, np.full(len(pick), t)))
        traj.append(vc.theta.copy())
    traj = np.array(traj)
    flat = {k: v.reshape((S * C,) + v.shape[2:]) for k, v in mom.items()}
    np.testing.assert_array_equal(flat["naccept"][pick], vc.naccept)
    np.testing.assert_allclose(flat["theta"][pick], vc.theta, rtol=1e-12)
    np.testing.assert_allclose(flat["lnPr"][pick], vc.lnPr, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(flat["mean"][pick], traj.mean(axis=0), rtol=1e-12)
    # a parameter whose proposals were all rejected: exactly 0 on the device, rounding of
    # sum/n (~1e-28) in NumPy's two-pass formula, hence the absolute floor
    np.testing.assert_allclose(flat["var"][pick], traj.var(axis=0), rtol=1e-8, atol=1e-20)
    assert 0.02 < (vc.naccept.mean() - 1) / nsteps < 0.98      # chains moved and were rejected
    # Gelman-Rubin of every source (convergence.py:77-85) from the device's own moments
    W = mom["var"].mean(axis=1)
    B = mom["mean"].var(axis=1, ddof=1)
    with np.errstate(all="ignore"):
        ref = np.where(W > 0, np.sqrt((W + B) / W), np.nan)
    np.testing.assert_allclose(rhat, ref, rtol=1e-9)
    assert np.isfinite(rhat).all()


def test_c4_full_size_cl_likelihood():
    """4096 chains x 121 calibration x beam points, nb=256, npar=64: 16 chains spread over the
    batch against the oracle to 1e-12; the result does not depend on the batch (the reversed
    batch and a 100-chain slice return the same bits)."""
    from oracle import refcl as RC
    from mcmc_amd import Engine
    nb, npar, n = 256, 64, 4096
    model, d, truth = RC.synth_cl_problem(nb=nb, npar=npar, calib=0.05, beam=0.02, lmax=2400)
    rng = np.random.default_rng(404)
    q = truth[None, :] * rng.uniform(0.85, 1.15, (n, npar))
    with Engine() as eng:
        eng.cl_load([d])
        eng.cl_set_binning(model.bins, model.shapefun)
        got = eng.cl_lnlike(q)
        rev = eng.cl_lnlike(q[::-1].copy())
        part = eng.cl_lnlike(q[1000:1100].copy())
    assert np.all(np.isfinite(got))
    np.testing.assert_array_equal(rev[::-1], got)
    np.testing.assert_array_equal(part, got[1000:1100])
    idx = np.unique(np.linspace(0, n - 1, 16).astype(int))
    ref = np.array([RC.cl_lnlike(q[i], model, [d]) for i in idx])
    np.testing.assert_allclose(got[idx], ref, rtol=1e-12)


def test_c5_full_size_beam_maps():
    """1e4 maps of 256 x 256 pixels (5.2 GB of pixel data resident) x 16 chains: the maps are 8
    distinct synthetic ones repeated, so equal (map, theta) pairs anywhere in the batch must
    return the same bits; first, middle and last map against the oracle to 1e-12."""
    from oracle import refbeam as RB
    from mcmc_amd import Engine
    M, side, C = 10000, 256, 16
    x, y, d8, noise, truth8, prior = RB.synth_beam_maps(8, n=side, seed=3)
    d = np.tile(d8, (M // 8, 1))
    truth = np.tile(truth8, (M // 8, 1))
    rng = np.random.default_rng(55)
    th8 = truth8[:, None, :] * (1 + 0.02 * rng.standard_normal((8, C, 9)))
    th8[:, :, 7] = rng.uniform(0.8, 1.0, (8, C))
    th8[:, 0, :] = truth8
    th = np.tile(th8, (M // 8, 1, 1))
    with Engine() as eng:
        eng.beam_load(x, y, d, noise, center_min=prior.center_min, center_max=prior.center_max,
                      sig_min=prior.sig_min, sig_max=prior.sig_max)
        got = eng.beam_lnlike(th)
    assert got.shape == (M, C) and np.all(np.isfinite(got))
    np.testing.assert_array_equal(got.reshape(M // 8, 8, C), np.broadcast_to(got[:8], (M // 8, 8, C)))
    for m in (0, M // 2 + 3, M - 1):
        bm = RB.BeamMap(x, y, d[m], noise)
        ref = [RB.beam_lnlike(RB.beam_package(th[m, c]), bm) for c in range(C)]
        np.testing.assert_allclose(got[m], ref, rtol=1e-12)
    del truth
