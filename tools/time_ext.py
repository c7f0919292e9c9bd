"""ms per 200 steps of the extended fused SED kernel (correlated proposals) with and without the
running co-moments, C2 shape.  GPU box: python tools/time_ext.py [S] [C]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from mcmc_amd import Engine, synth   # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 64
nu, d, sig, truth = synth.sed_catalogue(S, seed=31)
nb = np.full(S, 8, np.int32)
with Engine(seed=1) as e:
    e.sed_load("M2N", nb, nu, d, sig)
    e.mh_init_problem("sed", C, synth.M2N_START, synth.M2N_PROP_SIGMAS, param_blocks=np.arange(6),
                      randomize_start=True, track_cov=True)
    assert e.mh_fused_active
    e.mh_run(200)
    e.mh_run(200)
    print("first stage (block proposals, general instantiation, co-moments in memory): %.2f ms per 200 steps"
          % e.last_kernel_ms, flush=True)
    e.mh_run(600)
    e.mh_adapt_corr(no_correlations=False, do_block=False, fac=0.6)
    for cov in (True, False):
        e.mh_track_cov(cov)
        e.mh_run(200)
        t = []
        for _ in range(3):
            e.mh_run(200)
            t.append(e.last_kernel_ms)
        print("correlated proposals, co-moments %s: %.2f ms per 200 steps = %.3e evals/s"
              % ("on " if cov else "off", np.median(t), S * C * 200 / (np.median(t) * 1e-3)), flush=True)
