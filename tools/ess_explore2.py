"""Mixing with Cholesky-correlated proposals (step-wise engine) on the C2 workload (GPU)."""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
from mcmc_amd import Engine, synth

S, C = 2000, 64
nu, d, sig, _ = synth.sed_catalogue(S)
nb = np.full(S, 8, np.int32)


def report(eng, tag, measure, t):
    ess = eng.mh_ess()
    rhat = eng.rhat()
    mom = eng.mh_moments()
    em = np.nanmin(np.where(np.isfinite(ess), ess, np.nan), axis=1)
    print(json.dumps(dict(tag=tag, measure=measure, rhat_med=float(np.nanmedian(rhat)),
                          rhat_p90=float(np.nanpercentile(np.nanmax(rhat, axis=1), 90)),
                          ess_med=float(np.nanmedian(em)), acc=float(mom["naccept"].mean() / (measure + 1)),
                          sec=t)), flush=True)


def run(stages, measure=2000, pooled=False):
    eng = Engine(seed=123)
    eng.sed_load("M2N", nb, nu, d, sig)
    eng.mh_init_problem("sed", C, synth.M2N_START, synth.M2N_PROP_SIGMAS, param_blocks=np.arange(6),
                        randomize_start=True, track_cov=True)
    eng.mh_run(400)
    eng.mh_adapt_corr(no_correlations=True)
    eng.mh_run(2000)
    for n in stages:
        eng.mh_adapt_corr(no_correlations=False, do_block=False)
        eng.mh_run(n)
    eng.mh_adapt_corr(no_correlations=False, do_block=False)
    t0 = time.perf_counter()
    eng.mh_run(measure, thin=max(1, measure // 100))
    t = time.perf_counter() - t0
    report(eng, "corr stages=%s" % (stages,), measure, t)
    eng.close()


run([], 2000)
run([2000], 2000)
run([2000, 4000], 2000)
run([2000, 4000], 10000)
