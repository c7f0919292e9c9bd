import numpy as np, time, sys
sys.path.insert(0, '.')
from mcmc_amd import Engine, synth
for S, C, nbn in ((16, 64, 8), (64, 16, 27), (100, 64, 8)):
    nu, d, sig, truth = synth.sed_catalogue(S, seed=3, nb=nbn)
    nu[1, 0] = 1.4
    nb = np.full(S, nbn, np.int32)
    for lm in (1, 0):
        with Engine(seed=1) as e:
            e.mh_set_lane_map(lm)
            e.sed_load("M2N", nb, nu, d, sig)
            e.mh_init(C, truth[:, None, :].repeat(C, axis=1), np.full(6, 0.02), param_blocks=np.arange(6), randomize_start=False)
            g = e.mh_lane_group
            e.mh_run(200); e.sync()
            t = time.perf_counter(); e.mh_run(4000); e.sync(); dt = time.perf_counter() - t
            print(S, C, nbn, "lane_map", lm, "G", g, "us/step %.2f" % (dt / 4000 * 1e6), flush=True)
