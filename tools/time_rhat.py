import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from mcmc_amd import Engine, synth
from mcmc_amd import distributed as D
S, C = 100000, 64
nu, d, sig, truth = synth.sed_catalogue(S, seed=31)
nb = np.full(S, 8, np.int32)
with Engine(seed=1) as e:
    e.sed_load("M2N", nb, nu, d, sig)
    e.mh_init(C, synth.M2N_START, synth.M2N_PROP_SIGMAS, param_blocks=np.arange(6), randomize_start=True)
    e.mh_run(200)
    dev = torch.device("cuda", e.device)
    for it in range(4):
        ev = []
        torch.cuda.synchronize(dev); e.sync()
        t0 = time.perf_counter()
        r = D.engine_rhat(e, Ctot=C, timing=ev)
        t1 = time.perf_counter()
        torch.cuda.synchronize(dev); e.sync()
        t2 = time.perf_counter()
        print("enqueue %.3f ms, to completion %.3f ms, events %.3f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3, ev[0][0].elapsed_time(ev[0][1])), flush=True)
    # with a launch in between, as bench.py does
    for it in range(3):
        ev = []
        e.mh_run(200)
        t0 = time.perf_counter()
        r = D.engine_rhat(e, Ctot=C, timing=ev)
        torch.cuda.synchronize(dev); e.sync()
        t2 = time.perf_counter()
        print("after mh_run: to completion %.3f ms, events %.3f ms" % ((t2 - t0) * 1e3, ev[0][0].elapsed_time(ev[0][1])), flush=True)
