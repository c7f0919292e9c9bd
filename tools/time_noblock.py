"""ms per 200 Metropolis steps of the fused SED sampler without parameter blocks (every parameter moves in every
step: the general instantiation of k_mh_run), 20 000 sources x 64 chains, 8 bands.  GPU box: python tools/time_noblock.py"""
import sys

import numpy as np

sys.path.insert(0, ".")
from mcmc_amd import Engine, synth   # noqa: E402

S, C = 20000, 64
nu, d, sig, truth = synth.sed_catalogue(S, seed=31)
nb = np.full(S, 8, np.int32)
for model, P in (("M2N", 6), ("M1N", 3)):
    with Engine(seed=1) as e:
        e.sed_load(model, nb, nu, d, sig)
        e.mh_init(C, truth[:, None, :P].repeat(C, axis=1), np.full(P, 0.01), param_blocks=None, randomize_start=False)
        e.mh_run(200)
        t = []
        for _ in range(5):
            e.mh_run(200)
            t.append(e.last_kernel_ms)
        print("%s no blocks: %.2f ms per 200 steps" % (model, min(t)))
