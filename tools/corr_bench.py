"""Rate and mixing of the fused SED kernel's correlated proposals on the C2 workload."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
from mcmc_amd import Engine, synth

S, C = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, 64
nu, d, sig, truth = synth.sed_catalogue(S)
nb = np.full(S, 8, np.int32)

def stats(eng, tag, n, t):
    r = eng.rhat()
    mom = eng.mh_moments()
    print(json.dumps(dict(tag=tag, steps=n, sec=round(t, 3), evals_per_s="%.3g" % (S * C * n / t), kernel_ms=round(eng.last_kernel_ms, 2),
                          worst_med=round(float(np.nanmedian(np.nanmax(r, axis=1))), 3),
                          frac_lt_1p1=float((np.nanmax(r, axis=1) < 1.1).mean()),
                          acc=round(float(mom["naccept"].mean() / (eng.nsamples + 1)), 3))), flush=True)

with Engine(seed=11) as eng:
    eng.sed_load("M2N", nb, nu, d, sig)
    eng.mh_init_problem("sed", C, synth.M2N_START, synth.M2N_PROP_SIGMAS, param_blocks=np.arange(6),
                        randomize_start=True, track_cov=True)
    assert eng.mh_fused_active
    t0 = time.perf_counter(); eng.mh_run(2000); stats(eng, "stage0 uncorr block +cov", 2000, time.perf_counter() - t0)
    eng.mh_adapt_corr(no_correlations=True)
    t0 = time.perf_counter(); eng.mh_run(4000); stats(eng, "stage1 uncorr block +cov", 4000, time.perf_counter() - t0)
    for n, cov in ((2000, True), (10000, True), (200, False), (30000, False)):
        if cov:
            eng.mh_adapt_corr(no_correlations=False, do_block=False, fac=0.6)
        eng.mh_track_cov(cov)
        t0 = time.perf_counter(); eng.mh_run(n); stats(eng, "corr all fac=0.6 cov=%s" % cov, n, time.perf_counter() - t0)
