#!/usr/bin/env python3
"""Timing of the Cl (config C4) and beam (config C5) likelihood kernels on one MI355X.
    python tests/bench_clbeam.py cl [nb npar calib beam chains]
    python tests/bench_clbeam.py beam [maps npix_side chains]
Prints one JSON line per workload (kernel time from HIP events on the library's stream)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
PEAK_FP64 = 78.6   # TFLOP/s, vector = matrix on MI355X


def bench_cl(nb=256, npar=64, calib=0.05, beam=0.02, chains=4096, reps=5):
    from oracle import refcl as RC       # synthetic problem generator only (inputs)
    from mcmc_amd import Engine
    import torch
    model, d, truth = RC.synth_cl_problem(nb=nb, npar=npar, calib=calib, beam=beam, lmax=2400)
    rng = np.random.default_rng(1)
    q = truth[None, :] * rng.uniform(0.9, 1.1, (chains, npar))
    with Engine() as eng:
        eng.cl_load([d])
        eng.cl_set_binning(model.bins, model.shapefun)
        qd = torch.tensor(np.ascontiguousarray(q.T), device="cuda")
        out = torch.empty(chains, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        ms, wall = [], []
        for _ in range(reps + 2):
            t0 = time.perf_counter()
            eng.cl_lnlike_dev(qd.data_ptr(), out.data_ptr(), chains)
            eng.sync()
            wall.append((time.perf_counter() - t0) * 1e3)
            ms.append(eng.last_kernel_ms)
        ms, wall = float(np.mean(ms[2:])), float(np.mean(wall[2:]))
        nm = (11 if calib > 0 else 1) * (11 if beam > 0 else 1)
        rows = chains * nm
        flop = rows * (2.0 * nb * nb + 2.0 * nb)
        print(json.dumps({"workload": "C4 Cl offset-lognormal full covariance", "chains": chains,
                          "nb": nb, "npar": npar, "marge_points": nm, "rows": rows,
                          "k_cl_form_ms": ms, "call_ms": wall, "evals_per_s": chains / (wall * 1e-3),
                          "mfma_tflops": flop / (ms * 1e-3) / 1e12,
                          "mfma_frac": flop / (ms * 1e-3) / 1e12 / PEAK_FP64,
                          "lnl_finite": bool(torch.isfinite(out).all().item())}))


def bench_beam(M=2000, side=256, C=16, reps=3):
    from oracle import refbeam as RB     # synthetic maps (inputs)
    from mcmc_amd import Engine
    import torch
    x, y, d1, noise, truth1, prior = RB.synth_beam_maps(8, n=side, seed=3)
    rng = np.random.default_rng(2)
    reps_m = (M + 7) // 8
    d = np.tile(d1, (reps_m, 1))[:M] + 0.0
    truth = np.tile(truth1, (reps_m, 1))[:M]
    th = truth[:, None, :] * (1 + 0.02 * rng.standard_normal((M, C, 9)))
    th[:, :, 7] = 1.0
    with Engine() as eng:
        eng.beam_load(x, y, d, noise, center_min=prior.center_min, center_max=prior.center_max,
                      sig_min=prior.sig_min, sig_max=prior.sig_max)
        td = torch.tensor(np.ascontiguousarray(th.reshape(M * C, 9).T), device="cuda")
        out = torch.empty(M * C, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        ms, wall = [], []
        for _ in range(reps + 1):
            t0 = time.perf_counter()
            eng.beam_lnlike_dev(td.data_ptr(), out.data_ptr(), M, C)
            eng.sync()
            wall.append((time.perf_counter() - t0) * 1e3)
            ms.append(eng.last_kernel_ms)
        ms, wall = float(np.mean(ms[1:])), float(np.mean(wall[1:]))
        npix = side * side
        flop = float(M) * C * npix * 46.0            # Npix*(32+14) per eval
        print(json.dumps({"workload": "C5 beam maps", "maps": M, "npix": npix, "chains_per_map": C,
                          "k_beam_chi2_ms": ms, "call_ms": wall, "evals_per_s": M * C / (wall * 1e-3),
                          "valu_tflops": flop / (ms * 1e-3) / 1e12,
                          "valu_frac": flop / (ms * 1e-3) / 1e12 / PEAK_FP64,
                          "hbm_GBps": M * npix * 8.0 * ((C + 7) // 8) / (ms * 1e-3) / 1e9,
                          "lnl_finite": bool(torch.isfinite(out).all().item())}))


def bench_mh(nb=64, npar=32, calib=0.0, beam=0.0, chains=4096, nsteps=300, fused=1, track_cov=0):
    """MH steps/s on a Cl problem: the fused one-launch kernel (fused=1) or the launch-per-step
    engine (fused=0, launch-bound when the likelihood is small)."""
    from oracle import refcl as RC
    from mcmc_amd import Engine
    model, d, truth = RC.synth_cl_problem(nb=nb, npar=npar, calib=calib, beam=beam, lmax=2400)
    with Engine() as eng:
        eng.cl_load([d])
        eng.cl_set_binning(model.bins, model.shapefun)
        eng.mh_set_fused(bool(fused))
        eng.mh_init_problem("cl", chains, truth, np.full(npar, 0.02), param_blocks=np.arange(npar),
                            track_cov=bool(track_cov))
        eng.mh_run(20)
        t0 = time.perf_counter()
        eng.mh_run(nsteps)
        dt = time.perf_counter() - t0
        mom = eng.mh_moments()
        print(json.dumps({"workload": "Cl MH, fused kernel" if eng.mh_fused_active else "Cl MH, launch per step",
                          "chains": chains, "nb": nb, "npar": npar, "track_cov": int(track_cov),
                          "calib": calib, "beam": beam,
                          "us_per_step": dt / nsteps * 1e6, "chain_steps_per_s": chains * nsteps / dt,
                          "accept": float(mom["naccept"].mean() / (nsteps + 21))}))


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "cl"
    args = [float(a) if "." in a else int(a) for a in sys.argv[2:]]
    {"cl": bench_cl, "beam": bench_beam, "mh": bench_mh}[what](*args)
