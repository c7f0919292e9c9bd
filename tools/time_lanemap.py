"""us per Metropolis step of the fused SED sampler (submmModel2_normalized) under the two lane
mappings (mi355_mh_set_lane_map 1 / 2), product kernels through the C ABI.  GPU box:
    python tools/time_lanemap.py
"""
import sys

import numpy as np

sys.path.insert(0, ".")
from mcmc_amd import Engine          # noqa: E402
from mcmc_amd import synth           # noqa: E402
from mcmc_amd.engine import MODEL_IDS   # noqa: E402

for S, C, nbands in [(1, 1, 27), (1, 4, 27), (1, 64, 27), (16, 16, 27), (64, 32, 27), (1024, 64, 27),
                     (1, 4, 8), (16, 16, 8), (256, 16, 8), (1024, 8, 8), (1024, 16, 8), (1024, 64, 8)]:
    nu, d, sig, truth = synth.sed_catalogue(S, seed=31, nb=nbands)
    nb = np.full(S, nbands, np.int32)
    row = {}
    for lane_map in (1, 2, 0):
        with Engine(seed=1) as e:
            e.mh_set_lane_map(lane_map)
            e.sed_load(MODEL_IDS["M2N"], nb, nu, d, sig)
            e.mh_init(C, truth[:, None, :].repeat(C, axis=1), np.full(6, 0.03), param_blocks=np.arange(6),
                      randomize_start=False)
            G = e.mh_lane_group
            e.mh_run(400)
            t = []
            for _ in range(3):
                e.mh_run(400)
                t.append(e.last_kernel_ms)
            row[lane_map] = (G, float(np.min(t)) * 1e3 / 400)
    print("S=%-5d C=%-3d nb=%-2d | lane=chain %7.2f us/step | (chain,band) G=%-2d %7.2f us/step (x%.2f) | auto picks G=%d"
          % (S, C, nbands, row[1][1], row[2][0], row[2][1], row[1][1] / row[2][1], row[0][0]), flush=True)

# the extended sampler (Cholesky-correlated proposals, co-moments on) under both mappings
print("extended sampler: correlated proposals + running co-moments", flush=True)
for S, C, nbands in [(1, 4, 27), (1, 64, 27), (16, 16, 27), (1, 4, 8), (16, 16, 8), (256, 16, 8)]:
    nu, d, sig, truth = synth.sed_catalogue(S, seed=31, nb=nbands)
    nb = np.full(S, nbands, np.int32)
    rng = np.random.default_rng(1)
    A = rng.normal(size=(6, 6))
    cov = A @ A.T + 6 * np.eye(6)
    dg = np.sqrt(np.diag(cov))
    L = np.linalg.cholesky(cov / dg[:, None] / dg[None, :])
    row = {}
    for lane_map in (1, 2):
        with Engine(seed=1) as e:
            e.mh_set_lane_map(lane_map)
            e.sed_load(MODEL_IDS["M2N"], nb, nu, d, sig)
            e.mh_init_problem("sed", C, truth[:, None, :].repeat(C, axis=1), np.full(6, 0.03), param_blocks=None,
                              randomize_start=False, track_cov=True)
            e.mh_set_sqrt_matrix(L)
            e.mh_run(400)
            t = []
            for _ in range(3):
                e.mh_run(400)
                t.append(e.last_kernel_ms)
            row[lane_map] = float(np.min(t)) * 1e3 / 400
    print("S=%-5d C=%-3d nb=%-2d | lane=chain %7.2f us/step | (chain,band) %7.2f us/step (x%.2f)"
          % (S, C, nbands, row[1], row[2], row[1] / row[2]), flush=True)
