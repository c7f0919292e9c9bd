"""Why does R-hat stay at ~1.5 on the C2 workload?  R-hat per parameter against run length, for the
production start (randomised around (1,2,10,0.1,2,20) with the driver's widths) and for chains
started at the truth; fused SED kernel, 256 sources x 64 chains."""
import json, sys
import numpy as np
sys.path.insert(0, ".")
from mcmc_amd import Engine, synth

S, C = 256, 64
nu, d, sig, truth = synth.sed_catalogue(S)
nb = np.full(S, 8, np.int32)

def stats(eng, tag, n):
    r = eng.rhat()
    mom = eng.mh_moments()
    print(json.dumps(dict(tag=tag, steps=n, rhat_med_per_param=[round(float(x), 3) for x in np.nanmedian(r, axis=0)],
                          worst_med=round(float(np.nanmedian(np.nanmax(r, axis=1))), 3),
                          frac_lt_1p1=float((np.nanmax(r, axis=1) < 1.1).mean()),
                          acc=round(float(mom["naccept"].mean() / (eng.nsamples + 1)), 3))), flush=True)

for tag in ("production start", "start at truth"):
    with Engine(seed=11) as eng:
        eng.sed_load("M2N", nb, nu, d, sig)
        if tag == "production start":
            eng.mh_init(C, synth.M2N_START, synth.M2N_PROP_SIGMAS, param_blocks=np.arange(6), randomize_start=True)
        else:
            th0 = np.repeat(truth[:, None, :], C, axis=1)
            eng.mh_init(C, th0, np.array([0.05, 0.1, 1.0, 0.05, 0.1, 2.0]), param_blocks=np.arange(6), randomize_start=False)
        eng.mh_run(2000)
        eng.mh_adapt()
        for n in (2000, 10000, 50000, 200000):
            eng.mh_run(n)
            stats(eng, tag, n)
            eng.mh_adapt()
    # spread of chain means of one source in units of the within-chain stdev
with Engine(seed=11) as eng:
    eng.sed_load("M2N", nb, nu, d, sig)
    eng.mh_init(C, synth.M2N_START, synth.M2N_PROP_SIGMAS, param_blocks=np.arange(6), randomize_start=True)
    eng.mh_run(2000); eng.mh_adapt(); eng.mh_run(50000)
    mom = eng.mh_moments()
    s = 3
    print("truth", np.round(truth[s], 3))
    print("chain means of source 3 (first 12 chains):\n", np.round(mom["mean"][s, :12], 3))
    print("chain stdevs:\n", np.round(np.sqrt(mom["var"][s, :12]), 3))
    print("lnPr:", np.round(mom["lnPr"][s, :12], 2))
