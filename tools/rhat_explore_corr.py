"""R-hat against run length with Cholesky-correlated all-parameter proposals (step-wise engine),
after staged adaptation, 256 C2 sources x 64 chains - compare tools/rhat_explore.py."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
from mcmc_amd import Engine, synth

S, C = 256, 64
nu, d, sig, truth = synth.sed_catalogue(S)
nb = np.full(S, 8, np.int32)

def stats(eng, tag, n, t):
    r = eng.rhat()
    mom = eng.mh_moments()
    print(json.dumps(dict(tag=tag, steps=n, sec=round(t, 2), rhat_med_per_param=[round(float(x), 3) for x in np.nanmedian(r, axis=0)],
                          worst_med=round(float(np.nanmedian(np.nanmax(r, axis=1))), 3),
                          frac_lt_1p1=float((np.nanmax(r, axis=1) < 1.1).mean()),
                          acc=round(float(mom["naccept"].mean() / (eng.nsamples + 1)), 3))), flush=True)

for do_block, fac in ((False, None), (False, 0.6), (True, None)):
    with Engine(seed=11) as eng:
        eng.sed_load("M2N", nb, nu, d, sig)
        eng.mh_init_problem("sed", C, synth.M2N_START, synth.M2N_PROP_SIGMAS, param_blocks=np.arange(6),
                            randomize_start=True, track_cov=True)
        eng.mh_run(2000)
        eng.mh_adapt_corr(no_correlations=True)
        eng.mh_run(4000)
        for n in (2000, 10000, 30000):
            kw = dict(no_correlations=False, do_block=do_block, rotate=do_block)
            if fac:
                kw["fac"] = fac
            eng.mh_adapt_corr(**kw)
            t0 = time.perf_counter()
            eng.mh_run(n)
            stats(eng, "corr block=%s fac=%s" % (do_block, fac), n, time.perf_counter() - t0)
