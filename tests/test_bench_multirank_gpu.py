"""bench.py's multi-rank code path (config C3's chain-sharded layout: every rank holds all sources
and its own chains, R-hat all-reduced) run as 2 ranks on ONE MI355X over gloo, against the
single-process result on the union of the chains (Gelman-Rubin, MCMC/convergence.py:58-87).  The
8-GPU RCCL run is the driver's; this pins that the path computes the right thing."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_equal_the_union_of_chains(tmp_path):
    from mcmc_amd import Engine, synth
    S, C, inner, adapt, launches = 48, 8, 20, 40, 3            # warmup 1 + steps 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dump = str(tmp_path / "rhat.npy")
    env = dict(os.environ, MI355_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "C3", "--sources", str(S),
           "--chains", str(C), "--inner", str(inner), "--adapt", str(adapt), "--ess-steps", "0",
           "--incremental-steps", "0", "--corr-steps", "2000", "--no-cpu-baseline", "--dump-rhat", dump]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["workload"].startswith("C3")
    assert line["config"]["layout"].startswith("chain-sharded")
    assert line["value"] > 0 and line["rhat_allreduce_bytes"] == 3 * 6 * 8 * S
    # the correlated-proposal ESS leg runs on every rank and its figures are gathered
    corr = line["correlated_proposals"]
    assert corr["steps"] == 2000 and corr["ess_per_sec"] > 0 and corr["evals_per_s"] > 0
    got = np.load(dump)                                        # [S, P] over the 2 x C chains
    # the same chains in this process: two engines with the ranks' seeds, same sequence of launches
    nu, d, sig, _ = synth.sed_catalogue(S)
    nb = np.full(S, nu.shape[1], np.int32)
    moms = []
    for rank in range(2):
        with Engine(seed=0x5EDC0FFEE + 7919 * rank) as eng:
            eng.sed_load("M2N", nb, nu, d, sig)
            eng.mh_init(C, synth.M2N_START, synth.M2N_PROP_SIGMAS, param_blocks=np.arange(6),
                        randomize_start=True)
            for _ in range(adapt // inner):
                eng.mh_run(inner)
            eng.mh_adapt()
            for _ in range(launches):
                eng.mh_run(inner)
            moms.append(eng.mh_moments())
    mean = np.concatenate([m["mean"] for m in moms], axis=1)   # [S, 2C, P]
    var = np.concatenate([m["var"] for m in moms], axis=1)
    W = var.mean(axis=1)
    B = mean.var(axis=1, ddof=1)
    with np.errstate(invalid="ignore", divide="ignore"):
        ref = np.where(W > 0, np.sqrt((W + B) / W), np.nan)
    assert got.shape == ref.shape == (S, 6)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    np.testing.assert_allclose(got[~np.isnan(ref)], ref[~np.isnan(ref)], rtol=1e-10)
