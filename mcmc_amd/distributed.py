"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

(source, chain) pairs are independent Markov chains, so the step loop needs no communication.
Two layouts:
  * source-sharded: rank r owns sources shard_range(S, r, W) and all their chains - no collective.
  * chain-sharded : every rank holds all S sources and its own C chains of each; Gelman-Rubin
    R-hat (MCMC/convergence.py:58-87) then needs per-source sums over all W*C chains:
    two all-reduces of [P,S] and [2,P,S] float64, off the step loop.
"""
import os

import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous balanced shard [lo, hi) of n units for `rank` of `world`."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_from_env(backend=None):
    """Join the job torchrun / torch.distributed.run started (RANK, WORLD_SIZE, MASTER_*).
    Returns (rank, world, local_rank).  World size 1 needs no process group."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # MI355_DIST_FORCE=1: create the process group for a single rank too, so that a one-GPU box
    # runs the collectives of the multi-rank path through RCCL itself
    force = os.environ.get("MI355_DIST_FORCE", "") not in ("", "0")
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            # MI355_DIST_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs
            # than ranks (gloo stages CUDA tensors through the host)
            backend = os.environ.get("MI355_DIST_BACKEND") or \
                ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def _world(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def allreduce_sum_(t, group=None):
    """In-place sum over the ranks of `group` (a no-op without a process group).  RCCL ("nccl") reduces
    the device tensor in place.  gloo - the rehearsal backend for boxes with fewer GPUs than ranks - has
    no device path of its own: torch stages such tensors through pinned host memory on pool streams, and
    with two processes on one MI355X that staging stalled at config C3's sizes (48 / 96 MB per call), so
    the copy to the host and back is made here, synchronously."""
    if dist.is_available() and dist.is_initialized():
        if t.is_cuda and "nccl" not in str(dist.get_backend(group)):
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def gelman_rubin_allreduce(partial, finish, C_local, P, S, device, group=None, Ctot=None):
    """Chain-sharded Gelman-Rubin.

    partial(pass_, gmean, out): fill `out` with this rank's sums over its C_local chains:
         pass 0: out[P,S] = sum_c mean_c ;  pass 1: out[2,P,S] = (sum_c (mean_c-gmean)^2, sum_c var_c)
    finish(sums, Ctot, out): out[P,S] = sqrt((W+B)/W)  (NaN where W<=0)
    Both operate on torch float64 tensors resident on `device`; on a GPU they are the HIP
    kernels behind mi355_rhat_partial_async / mi355_rhat_finish_async.  Centred second pass:
    sum(mean^2) - (sum mean)^2/C would cancel catastrophically in FP64.
    Ctot: the number of chains per source over all ranks if the caller already knows it (it is
    all-reduced here otherwise: one more collective and a host round trip per call).
    Returns (rhat [S,P] tensor, Ctot)."""
    if Ctot is None:
        t = torch.tensor([float(C_local)], dtype=torch.float64, device=device)
        allreduce_sum_(t, group)
        Ctot = int(round(float(t.item())))
    s0 = torch.empty((P, S), dtype=torch.float64, device=device)
    partial(0, None, s0)
    allreduce_sum_(s0, group)
    gmean = s0 / float(Ctot)
    s1 = torch.empty((2, P, S), dtype=torch.float64, device=device)
    partial(1, gmean, s1)
    allreduce_sum_(s1, group)
    out = torch.empty((P, S), dtype=torch.float64, device=device)
    finish(s1, Ctot, out)
    return out.t().contiguous(), Ctot


def engine_rhat(engine, group=None, Ctot=None, timing=None):
    """R-hat of every loaded source over the chains of all ranks in `group` (chain-sharded
    layout); with one rank this equals Engine.rhat().  Ctot: chains per source over all ranks
    when every rank knows it (saves a collective and a host round trip per call; all ranks must
    pass it or none).  Returns a [S,P] CUDA tensor (a view of the [P,S] result).

    Nothing here waits on the host (RCCL; the gloo rehearsal path copies through the host, see
    allreduce_sum_): the engine's HIP stream (mi355_stream) and torch's current stream - the one the
    collectives run on - wait for each other through events.  On the device
    the pieces run one after the other, and the last one (sqrt((W+B)/W), convergence.py:77-85:
    mi355_rhat_finish_async) is on the engine's stream again, so the next mi355_mh_run - which the
    host may enqueue at once - starts behind it.  Measured on one MI355X at config C3's slice
    (1e6 sources x 32 chains, one-rank RCCL group): letting the next sampler launch run beside the
    second all-reduce and the last step, as this function did until late in round 3, cost the
    sampler more (130.7 -> 139.5 ms: it shares its CUs with them) than the 1.2 ms the R-hat takes
    by itself (132.5 ms per step in sequence, 140.8 ms side by side).
    The caller synchronises torch's stream (or uses the tensor on it) before reading the result on
    the host.
    timing: a list that receives (start, end) torch events around the whole computation."""
    device = torch.device("cuda", engine.device)
    cur = torch.cuda.current_stream(device)
    eng_stream = torch.cuda.ExternalStream(int(engine.stream), device=device)
    if timing is not None:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record(eng_stream)
    out = torch.empty((engine.P, engine.S), dtype=torch.float64, device=device)
    multi = (dist.is_available() and dist.is_initialized()) or not (Ctot is None or Ctot == engine.C)
    if not multi:
        # no process group, one rank holds every chain: the library's own four kernels on the engine's stream
        eng_stream.wait_stream(cur)                           # `out` comes from torch's stream
        engine.rhat_allreduce(None, engine.C, out.data_ptr())
    else:
        if Ctot is None:
            t = torch.tensor([float(engine.C)], dtype=torch.float64, device=device)
            allreduce_sum_(t, group)
            Ctot = int(round(float(t.item())))
        sums = torch.empty((3, engine.P, engine.S), dtype=torch.float64, device=device)
        s0, s1 = sums[0], sums[1:]
        eng_stream.wait_stream(cur)                           # the buffers come from torch's stream
        engine.rhat_partial(0, s0.data_ptr())                 # sum over this rank's chains of the chain means
        cur.wait_stream(eng_stream)
        allreduce_sum_(s0, group)
        s0.mul_(1.0 / float(Ctot))                            # the mean over all chains, as the library's k_scale
        eng_stream.wait_stream(cur)
        engine.rhat_partial(1, s1.data_ptr(), s0.data_ptr())  # (sum (mean_c - gmean)^2, sum var_c), centred
        cur.wait_stream(eng_stream)
        allreduce_sum_(s1, group)
        eng_stream.wait_stream(cur)
        engine.rhat_finish(s1.data_ptr(), Ctot, out.data_ptr())
    # torch's stream waits for the engine's: whatever reuses `sums` (freed here, an allocation of torch's
    # stream) or reads `out` on torch's stream comes after the engine's last kernel
    cur.wait_stream(eng_stream)
    if timing is not None:
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record(cur)
        timing.append((ev0, ev1))
    return out.t()
