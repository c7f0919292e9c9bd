"""Measured per-point lnL deviation of the SED fast path from the oracle (GPU): near the truth
(small chi-square: worst amplification) and at random points of the prior box."""
import sys
import numpy as np
sys.path.insert(0, ".")
from mcmc_amd import Engine, synth
from oracle import refcpu as R

S = 300
nu, d, sig, truth = synth.sed_catalogue(S, seed=5)
rng = np.random.default_rng(1)
C = 8
near = truth[:, None, :] * (1 + 0.02 * rng.standard_normal((S, C, 6)))
lo = np.array([0.01, 0, 3, 0.01, 0, 3.0]); hi = np.array([3, 3, 100, 3, 3, 100.0])
box = lo + (hi - lo) * rng.uniform(size=(S, C, 6))
with Engine() as eng:
    eng.sed_load("M2N", np.full(S, 8, np.int32), nu, d, sig)
    for tag, th in (("near truth", near), ("prior box", box)):
        th = np.ascontiguousarray(th)
        got = eng.sed_lnlike(th)
        ref = np.array([[R.sed_lnlike(R.M2N, th[s, c], R.GaussData(nu[s], d[s], sig[s])) for c in range(C)]
                        for s in range(S)])
        rel = np.abs(got - ref) / np.abs(ref)
        print("%s: max rel %.2e  median %.2e  (|lnL| min %.2g)" % (tag, rel.max(), np.median(rel), np.abs(ref).min()))
