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
# HBM bytes per chain per launch of k_mh_run<M2N>.  NOT measured in this run: a constant taken from
# the PMC passes of tools/prof_bench.sh over this same command (TRAFFIC_SOURCE: (2 x FETCH_SIZE [gfx950
# correction] + WRITE_SIZE) / 6.4e6 chains), to be regenerated whenever the kernel's register
# allocation changes.  State is read and written once per launch whatever the number of inner steps
# (STATE_BYTES_PER_CHAIN, the algorithmic figure); the rest is the scratch of the VGPRs the
# three-waves-per-SIMD register allocation spills.
HBM_BYTES_PER_CHAIN_LAUNCH = 801.5
STATE_BYTES_PER_CHAIN = 415.7
TRAFFIC_SOURCE = "profiles/r3_bench_summary.txt"


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


_T0 = time.perf_counter()


def progress(msg):
    """one line per phase on stderr (rank 0): a long run is seen to be alive, and a stall names its phase"""
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1f s] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


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
                    help="MH steps of the (separately timed) ESS/s measurement with the production driver's "
                         "uncorrelated single-parameter blocks; 0 = skip")
    ap.add_argument("--incremental-steps", type=int, default=3,
                    help="extra launches with the opt-in incremental sampler (0 = skip)")
    ap.add_argument("--corr-steps", type=int, default=-1,
                    help="MH steps of the correlated-proposal ESS/s measurement, after 18 000 steps of staged "
                         "adaptation; 0 = skip; default: 50 000 on one GPU, 0 on several (every rank can run it "
                         "on its own chains - ask for it: it takes 70 s at config C3's 3.2e7 chains per GPU, "
                         "three times the rest of the run)")
    ap.add_argument("--trace-rows", type=int, default=100, help="thinned trace rows kept per chain in the ESS legs")
    ap.add_argument("--trace-gb", type=float, default=32.0, help="cap on the trace [rows, P, N] of an ESS leg, GB")
    ap.add_argument("--c3-slice-steps", type=int, default=3,
                    help="one GPU only: launches of one GPU's slice of config C3 (1e6 sources x 32 chains) "
                         "timed after the C2 legs - the same-shape anchor of a multi-GPU run; 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=6.0)
    args = ap.parse_args()
    if os.environ.get("MI355_BENCH_WATCHDOG"):               # seconds: dump every thread's Python stack, repeatedly
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["MI355_BENCH_WATCHDOG"]), repeat=True, file=sys.stderr)
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    config = args.config if args.config != "auto" else ("C2" if world_env == 1 else "C3")
    if args.corr_steps < 0:
        args.corr_steps = 50000 if world_env == 1 else 0
    args.sources = args.sources or (100000 if config == "C2" else 1000000)
    args.chains = args.chains or (64 if config == "C2" else 32)

    # CPU baselines first: the multi-core line forks workers, which must happen before this
    # process initialises the GPU.  N=1 only (rank 0 of a single-process run).
    cpu = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1 and not args.no_cpu_baseline:
        progress("CPU baselines (oracle port on the host cores)")
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
    progress("rank %d of %d on cuda:%d: %s, %d sources x %d chains" % (rank, world, local, config, S, C))
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
    progress("catalogue resident, chains started; %d adaptation steps" % args.adapt)
    if args.adapt > 0:
        left = args.adapt
        while left > 0:
            eng.mh_run(min(inner, left))
            left -= min(inner, left)
        eng.mh_adapt()

    rhat_ev = []                                              # (start, end) events of the R-hat of every step

    def one_step():
        eng.mh_run(inner)                                     # synchronous: the launch has finished
        # four small kernels (+ two all-reduces on several GPUs), enqueued without a host wait; the next
        # launch is enqueued at once and starts behind the last of them
        return D.engine_rhat(eng, Ctot=C * world, timing=rhat_ev)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        eng.sync()
        torch.cuda.synchronize(device)

    progress("warmup: %d launches of %d MH steps + R-hat" % (args.warmup, inner))
    for _ in range(args.warmup):
        one_step()
    kernel_ms = []
    barrier()
    del rhat_ev[:]
    progress("timed region: %d launches" % args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rhat = one_step()
        kernel_ms.append(eng.last_kernel_ms)
    barrier()
    elapsed = time.perf_counter() - t0
    progress("timed region done: %.2f ms per step" % (1e3 * elapsed / args.steps))
    rhat_ms = float(np.mean([a.elapsed_time(b) for a, b in rhat_ev])) if rhat_ev else 0.0
    if rank == 0:
        print("kernel ms per launch:", " ".join("%.2f" % v for v in kernel_ms), file=sys.stderr)
        print("rhat ms per step:", " ".join("%.2f" % a.elapsed_time(b) for a, b in rhat_ev), file=sys.stderr)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    # Opt-in incremental evaluation (mi355_mh_set_incremental): the same chains, a step evaluates
    # only what its parameter touches.  Reported beside the headline figure, never as `value`
    # (which stays the full evaluation of every proposed point).
    inc_out = {}
    if args.incremental_steps > 0:
        progress("incremental sampler leg")
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

    # ESS/s (second half of BASELINE.json's metric): separately timed runs that keep a thinned trace
    # (100 rows per chain) in HBM, then the multi-chain BDA3 estimator on the device; per source the
    # minimum over parameters.  BDA3's estimate means something only where the chains of a source agree,
    # so the rate counts the sources whose worst parameter has R-hat < 1.1 over the measured stage
    # (reference formula, convergence.py:77-85) and says which fraction of the catalogue that is; the
    # time is that of the whole run (every source), estimator included.
    def ess_summary(ess, rh, seconds):
        em = np.nanmin(np.where(np.isfinite(ess), ess, np.nan), axis=1)
        worst = np.where(np.isfinite(rh), rh, np.inf).max(axis=1)
        ok = worst < 1.1
        conv = float(np.nansum(em[ok]))
        return {"ess_per_sec": conv / seconds, "rhat_lt_1p1_frac": float(ok.mean()),
                "rhat_median": float(np.nanmedian(rh)),
                "ess_median_per_converged_source": float(np.nanmedian(em[ok])) if ok.any() else None,
                "ess_per_sec_all_sources": float(np.nansum(em)) / seconds,
                "seconds": seconds}, conv

    def trace_rows():      # thinned trace rows per chain: --trace-rows, fewer when [rows, P, N] would pass --trace-gb
        return int(max(4, min(args.trace_rows, args.trace_gb * 1e9 // (8.0 * 6 * S * C))))

    prod_out = {}
    if args.ess_steps > 0:
        # the production driver's proposal (driver.py:173-175: noCorrelations=True, doBlock=True), the
        # configuration `value` is measured on: it mixes slowly on this workload (two barely constrained
        # greybodies per source), few sources converge in 10 000 steps
        rows = trace_rows()
        thin = max(1, args.ess_steps // rows)
        progress("ESS leg, production proposal: %d steps, %d trace rows" % (args.ess_steps, rows))
        eng.mh_adapt()                                        # a stage of its own: R-hat over these steps only
        eng.mh_run(rows, thin=1)                              # allocates the trace buffers (rows kept), untimed
        barrier()
        t1 = time.perf_counter()
        eng.mh_run(args.ess_steps, thin=thin)
        progress("ESS estimator (BDA3 on the device)")
        ess = eng.mh_ess()
        barrier()
        t_ess = time.perf_counter() - t1
        summ, conv = ess_summary(ess, eng.rhat(), t_ess)
        if world > 1:
            tot = torch.tensor([conv, t_ess], dtype=torch.float64, device=device)
            both = [torch.zeros_like(tot) for _ in range(world)]
            torch.distributed.all_gather(both, tot)
            t_ess = float(max(b[1].item() for b in both))
            summ["ess_per_sec"] = float(sum(b[0].item() for b in both)) / t_ess
            summ["seconds"] = t_ess
        summ.update({"steps": args.ess_steps, "thin": thin,
                     "proposal": "uncorrelated single-parameter blocks, sigma = 2.4/sqrt(6) stdev (MCMC.py:379-385)",
                     "note": "per GPU: sum over converged sources of min-over-parameters ESS (BDA3 11.5 on the "
                             "device, %d chains x %d thinned rows per source) / wall time of the run incl. the "
                             "estimator; R-hat per rank over its own chains" % (C, args.ess_steps // thin)})
        prod_out = {"production_proposal_ess": summ}

    # Correlated proposals (newMCMC's Cholesky factor of the chain covariance, MCMC.py:389-403,
    # Proposal.py:150-154) in the fused kernel's instantiation for them, adapted in stages until the
    # chains of most sources agree (tests/test_mixing_gpu.py has the same schedule): the ESS/s figure of
    # the line.  driver.py:173-175 itself runs noCorrelations=True, which is what `value` measures.
    corr_out = {}
    if args.corr_steps > 0:
        eng.mh_init_problem("sed", C, synth.M2N_START, synth.M2N_PROP_SIGMAS, param_blocks=np.arange(6),
                            randomize_start=True, track_cov=True)
        if eng.mh_fused_active:
            progress("ESS leg, correlated proposals: 18 000 adaptation steps, then %d measured" % args.corr_steps)
            barrier()
            t_ad = time.perf_counter()
            eng.mh_run(2000)                                  # stage 0: the driver's widths, doBlock
            eng.mh_adapt_corr(no_correlations=True)
            eng.mh_run(4000)
            progress("  correlated stages: 2000 + 10 000 steps")
            for n in (2000, 10000):                           # correlated stages, all parameters at once
                eng.mh_adapt_corr(no_correlations=False, do_block=False, fac=0.6)
                eng.mh_run(n)
            eng.mh_adapt_corr(no_correlations=False, do_block=False, fac=0.6)
            eng.mh_track_cov(False)
            rows = trace_rows()
            thin = max(1, args.corr_steps // rows)
            eng.mh_run(rows, thin=1)
            barrier()
            progress("  measured stage: %d steps, %d trace rows" % (args.corr_steps, rows))
            t3 = time.perf_counter()
            eng.mh_run(args.corr_steps, thin=thin)
            k_corr = eng.last_kernel_ms
            ess_c = eng.mh_ess()
            barrier()
            t_corr = time.perf_counter() - t3
            summ, conv_c = ess_summary(ess_c, eng.rhat(), t_corr)
            if world > 1:                                     # same sources on every rank, other chains: the ESS add up
                tot = torch.tensor([conv_c, t_corr, k_corr], dtype=torch.float64, device=device)
                both = [torch.zeros_like(tot) for _ in range(world)]
                torch.distributed.all_gather(both, tot)
                t_corr = float(max(b[1].item() for b in both))
                k_corr = float(max(b[2].item() for b in both))
                summ["ess_per_sec"] = float(sum(b[0].item() for b in both)) / t_corr
                summ["seconds"] = t_corr
            summ.update({"steps": args.corr_steps, "thin": thin, "adapt_steps": 18000, "adapt_seconds": t3 - t_ad,
                         "ess_per_sec_incl_adaptation": summ["ess_per_sec"] * t_corr / (t_corr + t3 - t_ad),
                         "evals_per_s": float(S) * C * world * args.corr_steps / (k_corr * 1e-3),
                         "kernel": "k_mh_run<M2N, fast, extended, Cholesky-correlated>", "kernel_ms": k_corr,
                         "kernel_ms_per_200_steps": k_corr * 200.0 / args.corr_steps,
                         "proposal": "Cholesky-correlated, all parameters per step, sigma = 0.6 stdev",
                         "note": "after 18 000 adaptation steps (two uncorrelated and two correlated stages); "
                                 "R-hat over the moments of the measured stage incl. its trace-allocation steps"
                                 + ("; on several GPUs every rank adapts and converges its own %d chains per source "
                                    "(R-hat per rank, fractions and medians are rank 0's), ESS and evals summed over "
                                    "the ranks" % C if world > 1 else "")})
            corr_out = {"correlated_proposals": summ, "ess_per_sec": summ["ess_per_sec"],
                        "ess_rhat_lt_1p1_frac": summ["rhat_lt_1p1_frac"],
                        "ess_note": "ESS/s of the correlated-proposal leg (see correlated_proposals), counted over "
                                    "the sources with R-hat < 1.1 only"}
    if not corr_out and prod_out:
        corr_out = {"ess_per_sec": prod_out["production_proposal_ess"]["ess_per_sec"],
                    "ess_rhat_lt_1p1_frac": prod_out["production_proposal_ess"]["rhat_lt_1p1_frac"],
                    "ess_note": "ESS/s of the production proposal (see production_proposal_ess), counted over "
                                "the sources with R-hat < 1.1 only"}

    # One GPU's slice of config C3 (1e6 sources x 32 chains: what every rank of the multi-GPU run holds)
    # on this one GPU: the same-shape anchor for a scaling curve whose N = 1 point is C2
    c3_out = {}
    if args.c3_slice_steps > 0 and world == 1 and config == "C2":
        S3, C3 = 1000000, 32
        progress("C3 slice leg: %d sources x %d chains" % (S3, C3))
        nu3, d3, sig3, _ = synth.sed_catalogue(S3)
        eng.sed_load("M2N", np.full(S3, nu3.shape[1], np.int32), nu3, d3, sig3)
        eng.mh_init(C3, synth.M2N_START, synth.M2N_PROP_SIGMAS, param_blocks=np.arange(6), randomize_start=True)
        eng.mh_run(inner)
        eng.mh_adapt()
        eng.mh_run(inner)
        ev3, k3 = [], []
        D.engine_rhat(eng, Ctot=C3)
        barrier()
        t4 = time.perf_counter()
        for _ in range(args.c3_slice_steps):
            eng.mh_run(inner)
            k3.append(eng.last_kernel_ms)
            D.engine_rhat(eng, Ctot=C3, timing=ev3)
        barrier()
        t_c3 = time.perf_counter() - t4
        c3_out = {"c3_slice": {"sources": S3, "chains_per_source": C3, "launches": args.c3_slice_steps,
                               "evals_per_s": float(S3) * C3 * inner * args.c3_slice_steps / t_c3,
                               "ms_per_step": 1e3 * t_c3 / args.c3_slice_steps, "kernel_ms": float(np.mean(k3)),
                               "rhat_ms_per_step": float(np.mean([a.elapsed_time(b) for a, b in ev3])),
                               "rhat_exposed_ms_per_step": 1e3 * t_c3 / args.c3_slice_steps - float(np.mean(k3)),
                               "note": "one GPU's share of C3 (1e6 x 256 over 8 GPUs, chain-sharded) run alone: "
                                       "divide a multi-GPU value by N times this evals_per_s, not by the C2 value"}}

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
                     "traffic_source": "constant (not measured in this run): rocprofv3 PMC passes of this command, "
                                       + TRAFFIC_SOURCE,
                     "traffic_algorithmic": STATE_BYTES_PER_CHAIN * S * C,
                     "kernel": "k_mh_run<M2N>", "kernel_ms": k_ms,
                     "note": "FP64 VALU (transcendental-bound); algorithmic FLOP = 1342 per MH step "
                             "(SURVEY 8d convention, 1 transcendental = 32 FLOP) x S*C*inner steps "
                             "per launch; traffic = HBM bytes per launch (chain state in and out plus "
                             "register-spill scratch), ~3% of the 8 TB/s roof"},
        "rhat_median": float(torch.nanmedian(rhat).item()),
        # fraction of sources whose worst parameter has R-hat < 1.1 (reference formula,
        # convergence.py:77-85) after adapt + warmup + timed steps
        "rhat_lt_1p1_frac": float((torch.nan_to_num(rhat, nan=float("inf")).amax(dim=1) < 1.1).double().mean().item()),
        "rhat_allreduce_bytes": 3 * 6 * 8 * S if world > 1 else 0,
        # R-hat of every source (kernels + collectives): from its first kernel to its result, by events; the next
        # sampler launch starts behind it.  What a step pays for it in all (host gaps included) is
        # rhat_exposed_ms_per_step = ms_per_step - the sampler kernel's own time.
        "rhat_ms_per_step": rhat_ms,
        "rhat_exposed_ms_per_step": 1e3 * elapsed / args.steps - k_ms,
    }
    if args.dump_rhat and rank == 0:
        np.save(args.dump_rhat, rhat.cpu().numpy())
    out.update(corr_out)
    out.update(prod_out)
    out.update(inc_out)
    out.update(c3_out)
    if rank == 0 and cpu is not None:
        out["cpu_baseline"] = cpu
    progress("done")
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    eng.close()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
