"""Mixing experiments for the C2 workload: adaptation schedule vs R-hat / ESS (GPU)."""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
from mcmc_amd import Engine, synth

S, C = 4000, 64
nu, d, sig, _ = synth.sed_catalogue(S)
nb = np.full(S, 8, np.int32)


def run(schedule, measure=2000, fac=None):
    eng = Engine(seed=123)
    eng.sed_load("M2N", nb, nu, d, sig)
    eng.mh_init(C, synth.M2N_START, synth.M2N_PROP_SIGMAS, param_blocks=np.arange(6), randomize_start=True)
    for n in schedule:
        eng.mh_run(n)
        eng.mh_adapt(fac=fac)
    thin = max(1, measure // 100)
    eng.mh_run(measure, thin=thin)
    ess = eng.mh_ess()
    rhat = eng.rhat()
    mom = eng.mh_moments()
    acc = mom["naccept"].mean() / (measure + 1)
    em = np.nanmin(np.where(np.isfinite(ess), ess, np.nan), axis=1)
    out = dict(schedule=schedule, measure=measure, fac=fac, rhat_med=float(np.nanmedian(rhat)),
               rhat_p90=float(np.nanpercentile(np.nanmax(rhat, axis=1), 90)),
               ess_med=float(np.nanmedian(em)), ess_sum_per_kstep=float(np.nansum(em) / measure * 1000 / S), acc=float(acc))
    print(json.dumps(out), flush=True)
    eng.close()


for sched in ([400], [2000], [400, 2000], [400, 2000, 4000], [1000, 4000, 8000]):
    run(sched)
run([400, 2000], measure=10000)
run([400, 2000, 4000], measure=10000)
run([400, 2000, 4000], measure=10000, fac=1.5)
run([400, 2000, 4000], measure=10000, fac=0.6)
