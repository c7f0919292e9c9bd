"""us per MH step of beam chains, fused sampler against step-wise engine, over (maps, pixels, chains per map).
GPU box: python tools/scan_beam_mh.py"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from mcmc_amd import Engine          # noqa: E402
sys.path.insert(0, "tools")
from time_beam_mh import maps        # noqa: E402,F401


def one(M, side, C, fused, nsteps=1500):
    x, y, d, noise, truth = maps(M, side)
    with Engine(seed=1) as e:
        e.mh_set_fused("always" if fused else False)
        e.beam_load(x, y, d, noise, center_min=(-20, -20), center_max=(20, 20), sig_min=3, sig_max=8)
        sg = np.array([0.05, 0.05, 0.05, 0.05, 0.02, 0.05, 0.01, 0.0, 0.0])
        e.mh_init_problem("beam", C, truth, sg, param_blocks=np.array([0, 1, 2, 3, 4, 5, 6, 6, 6]), randomize_start=False)
        if fused and not e.mh_fused_active:
            return float("nan")
        e.mh_run(200)
        t0 = time.perf_counter()
        e.mh_run(nsteps)
        return (time.perf_counter() - t0) * 1e6 / nsteps


if __name__ == "__main__":
    for side in (16, 32, 64, 128):
        for M in (1, 16, 64, 256, 1024):
            for C in (1, 4, 8):
                if M * side * side * 8 * 3 > 2e9:
                    continue
                f, s = one(M, side, C, True), one(M, side, C, False)
                print("pixels %-6d maps %-5d chains/map %-2d fused %8.2f step-wise %8.2f  ratio %.2f"
                      % (side * side, M, C, f, s, s / f), flush=True)
