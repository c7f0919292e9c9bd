"""Mixing per MH step of the reference's proposal modes on the C2 workload (GPU): Cholesky-correlated
all-parameter steps (Proposal.py:150-154), rotated single-direction steps (:145-148, doBlock), and the
production driver's uncorrelated single-parameter blocks, after the same staged adaptation.
usage: python tools/ess_modes.py [S] [measure_steps]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from mcmc_amd import Engine, synth   # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
C = 64
nu, d, sig, _ = synth.sed_catalogue(S)
nb = np.full(S, 8, np.int32)


def run(tag, stages, final, measure=M):
    with Engine(seed=11) as eng:
        eng.sed_load("M2N", nb, nu, d, sig)
        eng.mh_init_problem("sed", C, synth.M2N_START, synth.M2N_PROP_SIGMAS, param_blocks=np.arange(6),
                            randomize_start=True, track_cov=True)
        eng.mh_run(2000)
        eng.mh_adapt_corr(no_correlations=True)
        eng.mh_run(4000)
        for n, kw in stages:
            eng.mh_adapt_corr(**kw)
            eng.mh_run(n)
        eng.mh_adapt_corr(**final)
        eng.mh_track_cov(False)
        rows = 100
        eng.mh_run(rows, thin=1)
        t0 = time.perf_counter()
        eng.mh_run(measure, thin=measure // rows)
        t = time.perf_counter() - t0
        kms = eng.last_kernel_ms
        ess = eng.mh_ess()
        rhat = eng.rhat()
        acc = eng.mh_moments()["naccept"].mean() / (eng.nsamples + 1)
    worst = np.nanmax(rhat, axis=1)
    em = np.nanmin(np.where(np.isfinite(ess), ess, np.nan), axis=1)
    ok = worst < 1.1
    print(json.dumps(dict(tag=tag, steps=measure, kernel_ms_per_200=round(kms * 200 / measure, 3),
                          acc=round(float(acc), 3), rhat_worst_med=round(float(np.nanmedian(worst)), 3),
                          frac_lt_1p1=round(float(ok.mean()), 3), ess_med=round(float(np.nanmedian(em)), 1),
                          ess_med_converged=round(float(np.nanmedian(em[ok])), 1) if ok.any() else None,
                          ess_per_kstep_converged=round(float(np.nansum(em[ok]) / max(1, ok.sum()) / measure * 1e3), 3),
                          sec=round(t, 2))), flush=True)


corr = lambda fac: dict(no_correlations=False, do_block=False, fac=fac)
rot = lambda fac: dict(no_correlations=False, do_block=True, rotate=True, fac=fac)
unc = dict(no_correlations=True, do_block=True)
which = sys.argv[3].split(",") if len(sys.argv) > 3 else ["corr0.6", "corr1", "rot1", "rot1.7", "rot2.4", "unc"]
for w in which:
    if w.startswith("corr"):
        f = float(w[4:])
        run(w, [(2000, corr(f)), (10000, corr(f))], corr(f))
    elif w.startswith("rot"):
        f = float(w[3:])
        run(w, [(2000, rot(f)), (10000, rot(f))], rot(f))
    elif w.startswith("mix"):          # adapt with all-parameter correlated stages, measure rotated
        f = float(w[3:])
        run(w, [(2000, corr(0.6)), (10000, corr(0.6))], rot(f))
    else:
        run(w, [(2000, unc), (10000, unc)], unc)
