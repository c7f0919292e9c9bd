"""us per Metropolis step of beam-map chains: the fused sampler (one launch per run, csrc/beam_mh_kernels.hpp)
against the step-wise engine (chi-square and accept + propose per step, replayed as a hipGraph).  GPU box: python tools/time_beam_mh.py"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from mcmc_amd import Engine          # noqa: E402


def maps(M, side, seed=3):
    rng = np.random.default_rng(seed)
    g = np.linspace(-20, 20, side)
    x, y = [a.ravel() for a in np.meshgrid(g, g)]
    truth = np.array([0.5, -0.3, 4.0, 6.0, 0.4, 10.0, 0.1, 1.0, 0.0])
    dx, dy = x - truth[0], y - truth[1]
    c, s = np.cos(truth[4]), np.sin(truth[4])
    u, v = c * dx + s * dy, -s * dx + c * dy
    d = truth[5] * (np.exp(-0.5 * (u * u / truth[2] ** 2 + v * v / truth[3] ** 2)) + truth[6])
    return x, y, d[None] + 0.1 * rng.standard_normal((M, x.size)), 0.1, truth


if __name__ == "__main__":
  for M, side, C, fused in [(1, 64, 1, True), (1, 64, 1, False), (1, 64, 8, True), (1, 64, 8, False), (1, 32, 1, True),
                            (1, 32, 1, False), (1, 128, 1, True), (1, 128, 1, False), (1, 256, 1, True), (1, 256, 1, False),
                            (16, 64, 1, True), (16, 64, 1, False), (256, 64, 4, True), (256, 64, 4, False),
                            (1, 64, 16, True), (1, 256, 16, True), (16, 256, 16, True), (256, 256, 16, True)]:
      x, y, d, noise, truth = maps(M, side)
      with Engine(seed=1) as e:
          e.mh_set_fused(fused)
          e.beam_load(x, y, d, noise, center_min=(-20, -20), center_max=(20, 20), sig_min=3, sig_max=8)
          sg = np.array([0.05, 0.05, 0.05, 0.05, 0.02, 0.05, 0.01, 0.0, 0.0])
          e.mh_init_problem("beam", C, truth, sg, param_blocks=np.array([0, 1, 2, 3, 4, 5, 6, 6, 6]),
                            randomize_start=False)
          e.mh_run(400)
          t0 = time.perf_counter()
          e.mh_run(4000)
          dt = time.perf_counter() - t0
          print("maps %-4d pixels %-6d chains/map %-3d %-9s: %7.2f us per step = %.3e chain-steps/s"
                % (M, side * side, C, "fused" if e.mh_fused_active else "step-wise", dt * 1e6 / 4000, M * C * 4000 / dt),
                flush=True)
