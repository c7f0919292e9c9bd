"""ms per 200 Metropolis steps of the fused SED kernel, every model, on the clamp-free fast path and
on the exact (OCML) path (the same catalogue with band 0 of every source moved to 1.4 GHz, which
takes every source out of the fast domain).  S x C chains, 8 bands.  Run on the GPU box:
    python tools/time_models.py [S] [C]
"""
import sys

import numpy as np

sys.path.insert(0, ".")
from mcmc_amd import Engine          # noqa: E402
from mcmc_amd import synth           # noqa: E402
from mcmc_amd.engine import MODEL_IDS   # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 64
nu, d, sig, truth = synth.sed_catalogue(S, seed=31)
nb = np.full(S, 8, np.int32)
nu_slow = nu.copy()
nu_slow[:, 0] = 1.4
names = {v: k for k, v in MODEL_IDS.items()}
NPARAM = {"M1N": 3, "M1N_LOGA": 3, "M2N": 6, "M2N_LOGA": 6, "M1": 2, "M2": 4, "THICK": 4, "THICK_LOGA": 4}
M = MODEL_IDS
for mid in range(8):
    P = NPARAM[names[mid]]
    th0 = truth[:, :P].copy()
    if mid in (M["M2N_LOGA"], M["M1N_LOGA"], M["THICK_LOGA"]):
        th0[:, 0] = np.log10(th0[:, 0])
    if mid == M["M2N_LOGA"]:
        th0[:, 3] = np.log10(th0[:, 3])
    if mid == M["M1"]:
        th0 = truth[:, 1:3].copy()
    if mid == M["M2"]:
        th0 = truth[:, [1, 2, 4, 5]].copy()
    if mid in (M["THICK"], M["THICK_LOGA"]):
        th0[:, 3] = 1.0
    row = []
    for cat in (nu, nu_slow):
        with Engine(seed=1) as e:
            e.sed_load(mid, nb, cat, d, sig)
            nfast = e.sed_fast_sources
            e.mh_init(C, th0[:, None, :].repeat(C, axis=1), np.full(P, 0.03), param_blocks=np.arange(P),
                      randomize_start=False)
            e.mh_run(200)
            t = []
            for _ in range(3):
                e.mh_run(200)
                t.append(e.last_kernel_ms)
            row.append((nfast, float(np.median(t))))
    print("model %d %-11s P=%d  fast %8.2f ms (%d fast sources)   exact %8.2f ms (%d)   x%.2f"
          % (mid, names.get(mid, "?"), P, row[0][1], row[0][0], row[1][1], row[1][0], row[1][1] / row[0][1]),
          flush=True)
