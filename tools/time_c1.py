"""Config C1 (the reference's own CPU-runnable case: one source - txt/Arp220.txt - one chain, the
staged sampler with the production step counts 50 000 + 500 000 + 500 000, driver.py:292-342) through
the class API on one GPU: wall time per stage and steps per second.  GPU box: python tools/time_c1.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from mcmc_amd import MCMC                          # noqa: E402
from mcmc_amd.submmSED import data, likelihood, model   # noqa: E402

g2 = np.load(os.path.join("tests", "golden", "sed_g2.npz"))
i = len(g2["nb"]) - 1                               # the Arp220 entry of the fixture
n = int(g2["nb"][i])
dat = data.submmData(g2["nu"][i, :n], g2["d"][i, :n], g2["sig"][i, :n], "Arp220", 0.018)
like = likelihood.SEDLikelihood_normalized(data=dat, model=model.submmModel2_normalized)
nMC = (50000, 500000, 500000)
t0 = time.perf_counter()
stages, ana = MCMC.sampler(like, nMC, np.array([2.0, 4.0, 4.0, 2.0, 4.0, 4.0]), np.array([1.0, 2.0, 10.0, 0.1, 2.0, 20.0]),
                           noCorrelations=True, doBlock=True, seed=3)
dt = time.perf_counter() - t0
tot = sum(len(s.samples) - 1 for s in stages)
print("C1: %d bands, 1 chain, %d stages, %d steps in %.2f s = %.3e MH steps/s (class API, traces kept on the host)"
      % (n, len(stages), tot, dt, tot / dt))
print("means", np.asarray(ana[0]).ravel()[:6])
