#!/usr/bin/env python3
"""Speed of the CPU port (oracle/refcpu.Chain, what bench.py times on the GPU box as
`cpu_baseline`) relative to the reference itself (MCMC.MCMC.MC_append, imported from
/root/reference as oracle/make_golden.py does), on the same workload: submmModel2_normalized +
SEDLikelihood_normalized, 8 bands, doBlock=True, one chain, one core.  Build container only (the
reference never travels); writes tests/golden/port_over_reference.json, which bench.py reports as
cpu_baseline.port_over_reference.

    python oracle/time_port_vs_reference.py [seconds per leg]
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))

from oracle import make_golden as G          # noqa: E402
from oracle import refcpu as R               # noqa: E402
from mcmc_amd import synth                   # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
    ref = G.load_reference()
    model, like_mod, data_mod, MC = (ref["submmSED.model"], ref["submmSED.likelihood"],
                                     ref["submmSED.data"], ref["MCMC"])
    nu, d, sig, _ = synth.sed_catalogue(4)
    out = {"seconds_per_leg": seconds, "sources": []}
    rates_ref, rates_port = [], []
    for s in range(nu.shape[0]):
        # the reference: MCMC(like, startProposal, startParams, doBlock) then MC_append in chunks
        dat = data_mod.submmData(nu[s], d[s], sig[s], "src%d" % s, 0.0)
        mod = model.submmModel2_normalized
        like = like_mod.SEDLikelihood_normalized(data=dat, model=mod)
        np.random.seed(100 + s)
        start = mod.package(np.asarray(synth.M2N_START, float))
        sigmas = mod.package(np.asarray(synth.M2N_PROP_SIGMAS, float))
        m = MC.MCMC(like, startProposal=sigmas, nMC=0, startParams=start, doBlock=True)
        n_ref, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            m.MC_append(500)
            n_ref += 500
        t_ref = time.perf_counter() - t0
        # the port, same call structure
        ch = R.Chain(R.sed_problem(R.M2N, R.GaussData(nu[s], d[s], sig[s])), synth.M2N_PROP_SIGMAS,
                     synth.M2N_START, R.NumpyDraws(100 + s), do_block=True)
        n_port, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            ch.run(500)
            n_port += 500
        t_port = time.perf_counter() - t0
        rates_ref.append(n_ref / t_ref)
        rates_port.append(n_port / t_port)
        out["sources"].append({"reference_steps_per_s": n_ref / t_ref, "port_steps_per_s": n_port / t_port})
    out["reference_steps_per_s"] = float(np.mean(rates_ref))
    out["port_steps_per_s"] = float(np.mean(rates_port))
    out["port_over_reference"] = out["port_steps_per_s"] / out["reference_steps_per_s"]
    out["where"] = "build container, one core; " + os.popen("grep -m1 'model name' /proc/cpuinfo").read().split(":")[-1].strip()
    out["reference_sha256"] = G.ref_tree_sha()
    path = os.path.join(HERE, "..", "tests", "golden", "port_over_reference.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
