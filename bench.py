#!/usr/bin/env python3
"""Benchmark of the SED Metropolis-Hastings hot path (BASELINE.json metric:
"SED likelihood evals/sec"), config C2: two-component greybody, 1e5 synthetic sources x 64
chains, FP64, per GPU.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one launch of the fused propose/evaluate/accept kernel advancing every chain by
--inner MH steps (= S*C*inner likelihood evaluations) followed by the Gelman-Rubin R-hat of
every source (all-reduced over RCCL when N>1: chain-sharded layout, every GPU holds all S sources
and its own 64 chains of each, so per-GPU work is fixed: weak scaling).  Inputs are resident in
HBM before the timed region; the step loop moves no data over PCIe.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_STEP = 1342.0        # 2-comp, 8 bands, 1 transcendental = 32 FP64 FLOP
PEAK_FP64_VALU_TFLOPS = 78.6  # 256 CU x 128 FP64 FLOP/clk x 2.4 GHz (MI355X vector FP64)
# HBM bytes per chain per launch of k_mh_run<M2N>, from the PMC passes in
# profiles/r2_bench_summary.txt: (2 x FETCH_SIZE [gfx950 correction] + WRITE_SIZE) / 6.4e6 chains.
# State is read and written once per launch whatever the number of inner steps (415.7 B per chain);
# the rest is the scratch of the 50 VGPRs the three-waves-per-SIMD register allocation spills.
HBM_BYTES_PER_CHAIN_LAUNCH = 817.4


def _port_worker(job):
    """one-step-at-a-time oracle chains on sources first, first+stride, ... for `seconds`:
    (MH steps done, seconds used, ESS summed, sources touched)"""
    from oracle import refcpu as R
    from mcmc_amd import synth
    first, stride, seconds, nsources = job
    nu, d, sig, _ = synth.sed_catalogue(min(nsources, first + 64 * stride + 1))
    src, done, t_used, ess_sum, nsrc = first, 0, 0.0, 0.0, 0
    chunk = 500
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end and src < nu.shape[0]:
        data = R.GaussData(nu[src], d[src], sig[src])
        ch = R.Chain(R.sed_problem(R.M2N, data), synth.M2N_PROP_SIGMAS, synth.M2N_START,
                     R.NumpyDraws(src), do_block=True)
        t0 = time.perf_counter()
        while time.perf_counter() < t_end and len(ch.samples) < 4000:
            ch.run(chunk)
            done += chunk
        t_used += time.perf_counter() - t0
        src += stride
        nsrc += 1
        # same ESS estimator as the GPU line (BDA3 11.5), on this chain thinned to <= 128 rows
        tr = np.asarray(ch.samples)[1:]
        tr = tr[::max(1, len(tr) // 100)][:128]
        if len(tr) >= 4:
            e = np.array([R.ess_bda3(tr[:, i]) for i in range(tr.shape[1])])
            if np.any(np.isfinite(e)):
                ess_sum += float(np.nanmin(e))
    return done, t_used, ess_sum, nsrc


def cpu_baseline(nsources, seconds=6.0):
    """the oracle's one-step-at-a-time sampler (same call structure as
    MCMC.py:209-215 -> Likelihood.py:99-105 -> model.py:394-396) on sources of the same catalogue,
    on one host core and on every core of the box's CPU share (sources dealt to the processes);
    then the NumPy-vectorised variant.  Forks: must run before this process touches the GPU."""
    import multiprocessing as mp
    cores = max(1, min(16, os.cpu_count() or 1))
    done, t_used, ess_sum, nsrc = _port_worker((0, 1, seconds, nsources))
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_port_worker, [(k, cores, seconds, nsources) for k in range(cores)])
    all_rate = sum(r[0] / r[1] for r in res if r[1] > 0)
    out = {"value": all_rate, "unit": "evals/s", "cores": cores, "kind": "port",
           "per_core": all_rate / cores,
           "one_core": {"value": done / t_used, "cores": 1,
                        "ess_per_sec": ess_sum / t_used if t_used > 0 else None},
           "ess_per_sec": sum(r[2] for r in res) / max(r[1] for r in res),
           "sample": "oracle/refcpu.py (pure Python + NumPy restatement of the reference's sampler, one "
                     "MH step at a time) on %d processes, %d MH steps on %d sources of the same "
                     "catalogue, 1 chain each, %.1f s; before that %d steps on one core (%.1f s)"
                     % (cores, sum(r[0] for r in res), sum(r[3] for r in res), max(r[1] for r in res),
                        done, t_used)}
    # speed of this port relative to the reference itself, measured once in the build container
    # (oracle/time_port_vs_reference.py; the reference does not travel to the GPU box)
    try:
        with open(os.path.join(ROOT, "tests", "golden", "port_over_reference.json")) as f:
            por = json.load(f)
        out["port_over_reference"] = por["port_over_reference"]
        out["port_over_reference_note"] = ("refcpu.Chain %.3g vs reference MCMC.MC_append %.3g MH steps/s on "
                                           "one core of the build container (%s)"
                                           % (por["port_steps_per_s"], por["reference_steps_per_s"], por["where"]))
    except (OSError, KeyError, ValueError):
        pass
    out.update(cpu_vectorised(seconds=3.0))
    return out


def cpu_vectorised(seconds=3.0):
    """SURVEY 8(d)'s "best-effort CPU" lines: the same MH transition vectorised with NumPy over
    64 sources x 64 chains per process (oracle/refcpu.VectorChainsM2N), on one core and on the
    box's CPU share.  Must run before this process touches the GPU (it forks workers)."""
    import multiprocessing as mp
    from oracle import refcpu as R
    from mcmc_amd import synth
    sig, start = np.asarray(synth.M2N_PROP_SIGMAS, float), np.asarray(synth.M2N_START, float)
    done, t = R.vector_m2n_rate((0, 64, 64, seconds, sig, start))
    cores = max(1, min(16, os.cpu_count() or 1))
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(R.vector_m2n_rate, [(64 * k, 64, 64, seconds, sig, start) for k in range(cores)])
    return {"vectorised": {"value": done / t, "unit": "evals/s", "cores": 1, "kind": "port",
                           "sample": "NumPy-vectorised MH over 64 sources x 64 chains, %.1f s" % t},
            "vectorised_all_cores": {"value": sum(r[0] / r[1] for r in res), "unit": "evals/s",
                                     "cores": cores, "kind": "port",
                                     "sample": "the same in %d processes on distinct sources" % cores}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)   # x 200 = the 10 000 measured MH steps of SURVEY 8(d)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=["auto", "C2", "C3"], default="auto",
                    help="C2: 1e5 sources x 64 chains per GPU; C3: 1e6 sources x 32 chains per GPU "
                         "(= BASELINE's 1e6 x 256 on 8 GPUs, chain-sharded, R-hat all-reduced over RCCL); "
                         "auto: C2 on one GPU, C3's layout on several")
    ap.add_argument("--sources", type=int, default=0, help="override the config's sources per GPU")
    ap.add_argument("--chains", type=int, default=0, help="override the config's chains per source per GPU")
    ap.add_argument("--dump-rhat", default="", help="rank 0 writes the last R-hat [S,P] to this .npy (tests)")
    ap.add_argument("--inner", type=int, default=200, help="MH steps per chain per launch")
    ap.add_argument("--adapt", type=int, default=2000, help="untimed stage-0 steps before newMCMC (SURVEY 8(d): 2 000)")
    ap.add_argument("--ess-steps", type=int, default=10000,
                    help="MH steps of the (separately timed) ESS/s measurement; 0 = skip")
    ap.add_argument("--incremental-steps", type=int, default=3,
                    help="extra launches with the opt-in incremental sampler (0 = skip)")
    ap.add_argument("--corr-steps", type=int, default=10000,
                    help="MH steps of the correlated-proposal ESS/s measurement (one GPU only); 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=6.0)
    args = ap.parse_args()
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    config = args.config if args.config != "auto" else ("C2" if world_env == 1 else "C3")
    args.sources = args.sources or (100000 if config == "C2" else 1000000)
    args.chains = args.chains or (64 if config == "C2" else 32)

    # CPU baselines first: the multi-core line forks workers, which must happen before this
    # process initialises the GPU.  N=1 only (rank 0 of a single-process run).
    cpu = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.sources, args.cpu_seconds)

    # stdout carries exactly one JSON line: libraries that write to file descriptor 1 (RCCL prints
    # a version banner there when the first communicator is made) are sent to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    from mcmc_amd import Engine, synth
    from mcmc_amd import distributed as D

    rank, world, local = D.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run "
                         "--nproc-per-node %d" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    local = local % torch.cuda.device_count()                 # (rehearsals: more ranks than GPUs)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    S, C, inner = args.sources, args.chains, args.inner
    nu, d, sig, _ = synth.sed_catalogue(S)                    # same sources on every rank
    nb = np.full(S, nu.shape[1], np.int32)
    eng = Engine(device=local, seed=0x5EDC0FFEE + 7919 * rank)  # distinct chains per rank
    eng.sed_load("M2N", nb, nu, d, sig)
    eng.mh_init(C, synth.M2N_START, synth.M2N_PROP_SIGMAS, param_blocks=np.arange(6),
                randomize_start=True)
    # stage 0 with the production driver's proposal widths, then newMCMC(): sigma = 2.4/sqrt(6)*stdev
    # (run as launches of `inner` steps like the timed ones, so that every k_mh_run launch of a
    # profile of this command does the same work and rocprofv3's per-kernel average is the figure
    # the roofline uses)
    if args.adapt > 0:
        left = args.adapt
        while left > 0:
            eng.mh_run(min(inner, left))
            left -= min(inner, left)
        eng.mh_adapt()

    rhat_s = []                                               # wall time of the R-hat of every step

    def one_step():
        eng.mh_run(inner)                                     # synchronous: the launch has finished
        t = time.perf_counter()
        r = D.engine_rhat(eng, Ctot=C * world)                # two kernels (+ two all-reduces on several GPUs)
        rhat_s.append(time.perf_counter() - t)
        return r

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        eng.sync()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        one_step()
    kernel_ms = []
    del rhat_s[:]
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rhat = one_step()
        kernel_ms.append(eng.last_kernel_ms)
    barrier()
    elapsed = time.perf_counter() - t0
    rhat_ms = 1e3 * float(np.mean(rhat_s)) if rhat_s else 0.0
    if rank == 0:
        print("kernel ms per launch:", " ".join("%.2f" % v for v in kernel_ms), file=sys.stderr)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    # Opt-in incremental evaluation (mi355_mh_set_incremental): the same chains, a step evaluates
    # only what its parameter touches.  Reported beside the headline figure, never as `value`
    # (which stays the full evaluation of every proposed point).
    inc_out = {}
    if args.incremental_steps > 0:
        eng.mh_set_incremental(True)
        if eng.mh_incremental_active:
            eng.mh_run(inner)
            barrier()
            t2 = time.perf_counter()
            inc_ms = []
            for _ in range(args.incremental_steps):
                one_step()
                inc_ms.append(eng.last_kernel_ms)
            barrier()
            t_inc = time.perf_counter() - t2
            inc_out = {"incremental_sampler": {
                "value": float(S) * C * inner * args.incremental_steps / t_inc, "unit": "MH steps/s per GPU",
                "kernel": "k_mh_run_inc<M2N>", "kernel_ms": float(np.mean(inc_ms)),
                "note": "opt-in: same chains as the full evaluation (tests/test_sed_gpu.py), the unit "
                        "band terms of the unchanged component kept across single-parameter steps"}}
        eng.mh_set_incremental(False)

    # ESS/s (second half of BASELINE.json's metric): a separately timed run that keeps a thinned
    # trace (100 rows per chain) in HBM, then the multi-chain BDA3 estimator on the device; per
    # source the minimum over parameters, summed over sources and (chains being independent) ranks
    ess_out = {}
    if args.ess_steps > 0:
        # thinned trace rows per chain: 100, fewer when the trace [rows, P, N] would pass 32 GB
        rows = int(max(4, min(100, 32e9 // (8.0 * 6 * S * C))))
        thin = max(1, args.ess_steps // rows)
        eng.mh_run(rows, thin=1)                              # allocates the trace buffers (rows kept), untimed
        barrier()
        t1 = time.perf_counter()
        eng.mh_run(args.ess_steps, thin=thin)
        ess = eng.mh_ess()
        barrier()
        t_ess = time.perf_counter() - t1
        ess_min = np.nanmin(np.where(np.isfinite(ess), ess, np.nan), axis=1)
        tot = torch.tensor([float(np.nansum(ess_min)), t_ess], dtype=torch.float64, device=device)
        if world > 1:
            both = [torch.zeros_like(tot) for _ in range(world)]
            torch.distributed.all_gather(both, tot)
            ess_total = float(sum(b[0].item() for b in both))
            t_ess = float(max(b[1].item() for b in both))
        else:
            ess_total = float(tot[0].item())
        ess_out = {"ess_per_sec": ess_total / t_ess, "ess_steps": args.ess_steps, "ess_thin": thin,
                   "ess_median_per_source": float(np.nanmedian(ess_min)),
                   "ess_caveat": "BDA3 ESS is only meaningful for sources whose chains have converged: "
                                 "read it together with rhat_lt_1p1_frac",
                   "ess_note": "sum over sources of min-over-parameters ESS (BDA3 11.5 multi-chain "
                               "estimator on the device, %d chains x %d thinned rows per source per "
                               "GPU) / wall time of that run incl. the estimator" % (C, args.ess_steps // thin)}

    # Correlated proposals (newMCMC's Cholesky factor of the chain covariance, Proposal.py:150-154)
    # in the fused kernel's extended instantiation: mixing per second beside the production
    # configuration (driver.py:173-175 runs noCorrelations=True, which is what `value` measures)
    corr_out = {}
    if args.corr_steps > 0 and world == 1:
        eng.mh_init_problem("sed", C, synth.M2N_START, synth.M2N_PROP_SIGMAS, param_blocks=np.arange(6),
                            randomize_start=True, track_cov=True)
        if eng.mh_fused_active:
            eng.mh_run(2000)
            eng.mh_adapt_corr(no_correlations=True)
            eng.mh_run(2000)
            for n in (1000, 3000):
                eng.mh_adapt_corr(no_correlations=False, do_block=False, fac=0.6)
                eng.mh_run(n)
            eng.mh_adapt_corr(no_correlations=False, do_block=False, fac=0.6)
            eng.mh_track_cov(False)
            rows = int(max(4, min(100, 32e9 // (8.0 * 6 * S * C))))
            thin = max(1, args.corr_steps // rows)
            eng.mh_run(rows, thin=1)
            barrier()
            t3 = time.perf_counter()
            eng.mh_run(args.corr_steps, thin=thin)
            k_corr = eng.last_kernel_ms
            ess_c = eng.mh_ess()
            barrier()
            t_corr = time.perf_counter() - t3
            rh = torch.from_numpy(eng.rhat())
            em = np.nanmin(np.where(np.isfinite(ess_c), ess_c, np.nan), axis=1)
            corr_out = {"correlated_proposals": {
                "ess_per_sec": float(np.nansum(em)) / t_corr, "ess_median_per_source": float(np.nanmedian(em)),
                "steps": args.corr_steps, "evals_per_s": float(S) * C * args.corr_steps / (k_corr * 1e-3),
                "kernel": "k_mh_run<M2N, fast, extended>", "kernel_ms": k_corr,
                "rhat_median": float(torch.nanmedian(rh).item()),
                "rhat_lt_1p1_frac": float((torch.nan_to_num(rh, nan=float("inf")).amax(dim=1) < 1.1).double().mean().item()),
                "note": "after 8000 adaptation steps (two uncorrelated and two correlated stages, fac 0.6); "
                        "R-hat over the moments of the measured stage incl. its trace-allocation steps"}}

    evals_per_launch = float(S) * C * inner                   # per GPU
    total_evals = evals_per_launch * args.steps * world
    value = total_evals / elapsed
    k_ms = float(np.mean(kernel_ms))
    achieved = evals_per_launch * FLOP_PER_STEP / (k_ms * 1e-3) / 1e12
    out = {
        "metric": "SED likelihood evals/sec",
        "value": value,
        "unit": "evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "%s: submmSED two-component greybody (submmModel2_normalized), "
                               "%d synthetic sources x %d chains per GPU (%d chains per source over "
                               "the %d GPUs), 8 bands, FP64, doBlock" % (config, S, C, C * world, world),
                   "sources": S, "chains_per_source_per_gpu": C, "mh_steps_per_launch": inner,
                   "layout": "chain-sharded, R-hat all-reduce per step" if world > 1 else "single GPU",
                   "adapt_steps": args.adapt},
        "roofline": {"bound": "valu", "achieved": achieved, "peak": PEAK_FP64_VALU_TFLOPS,
                     "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_VALU_TFLOPS,
                     "traffic": HBM_BYTES_PER_CHAIN_LAUNCH * S * C,
                     "kernel": "k_mh_run<M2N>", "kernel_ms": k_ms,
                     "note": "FP64 VALU (transcendental-bound); algorithmic FLOP = 1342 per MH step "
                             "(SURVEY 8d convention, 1 transcendental = 32 FLOP) x S*C*inner steps "
                             "per launch; traffic = HBM bytes per launch from rocprofv3 PMC "
                             "(profiles/r2_bench_summary.txt; state in and out plus register-spill scratch), ~3% of the 8 TB/s roof"},
        "rhat_median": float(torch.nanmedian(rhat).item()),
        # fraction of sources whose worst parameter has R-hat < 1.1 (reference formula,
        # convergence.py:77-85) after adapt + warmup + timed steps
        "rhat_lt_1p1_frac": float((torch.nan_to_num(rhat, nan=float("inf")).amax(dim=1) < 1.1).double().mean().item()),
        "rhat_allreduce_bytes": 3 * 6 * 8 * S if world > 1 else 0,
        "rhat_ms_per_step": rhat_ms,       # R-hat of every source (kernels + collectives + their syncs), inside ms_per_step
    }
    if args.dump_rhat and rank == 0:
        np.save(args.dump_rhat, rhat.cpu().numpy())
    out.update(ess_out)
    out.update(inc_out)
    out.update(corr_out)
    if rank == 0 and cpu is not None:
        out["cpu_baseline"] = cpu
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    eng.close()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
