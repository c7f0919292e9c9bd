"""World-size-2 (gloo, CPU) test of the multi-GPU layer: sharding helpers and the chain-sharded
Gelman-Rubin all-reduce protocol of mcmc_amd/distributed.py.  On a GPU the partial / finish
callables are the HIP kernels (mi355_rhat_partial_async / _finish_async); here the oracle plays
their role so that the orchestration (what is reduced, in what order, with which chain count)
is what gets tested against the reference definition (convergence.py:58-87)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mcmc_amd import distributed as D
from oracle import refcpu as R


def test_shard_range_is_a_balanced_partition():
    for n in (0, 1, 7, 8, 100000, 1000003):
        for w in (1, 2, 3, 8):
            parts = [D.shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, S, P, C_per_rank, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    rng = np.random.default_rng(1000)                      # same data on every rank ...
    C_all = sum(C_per_rank)
    means = rng.normal(5.0, 0.3, (S, C_all, P)) + 1e6      # large offset: needs the centred pass
    var = rng.uniform(0.5, 2.0, (S, C_all, P))
    var[:, :, 1] = 0.0                                     # a fixed parameter: W=0 -> NaN
    lo = sum(C_per_rank[:rank])
    hi = lo + C_per_rank[rank]                             # ... each rank owns chains [lo,hi)
    m_loc, v_loc = means[:, lo:hi], var[:, lo:hi]

    def partial(pass_, gmean, out):
        if pass_ == 0:
            out.copy_(torch.from_numpy(m_loc.sum(axis=1).T.copy()))
        else:
            g = gmean.numpy().T[:, None, :]
            out[0].copy_(torch.from_numpy(((m_loc - g) ** 2).sum(axis=1).T.copy()))
            out[1].copy_(torch.from_numpy(v_loc.sum(axis=1).T.copy()))

    def finish(sums, Ctot, out):
        within = sums[1] / Ctot
        between = sums[0] / (Ctot - 1)
        val = torch.sqrt((within + between) / within)
        out.copy_(torch.where(within > 0, val, torch.full_like(val, float("nan"))))

    rhat, Ctot = D.gelman_rubin_allreduce(partial, finish, C_per_rank[rank], P, S,
                                          torch.device("cpu"))
    assert Ctot == C_all
    np.save(os.path.join(out_dir, "rhat_%d.npy" % rank), rhat.numpy())
    if rank == 0:
        np.savez(os.path.join(out_dir, "inputs.npz"), means=means, var=var)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("C_per_rank", [(4, 4), (3, 5)])
def test_chain_sharded_rhat_allreduce_world2(tmp_path, C_per_rank):
    S, P, world = 6, 4, 2
    mp.spawn(_worker, args=(world, _free_port(), S, P, C_per_rank, str(tmp_path)), nprocs=world,
             join=True)
    r0 = np.load(tmp_path / "rhat_0.npy")
    r1 = np.load(tmp_path / "rhat_1.npy")
    np.testing.assert_array_equal(r0, r1)                  # every rank ends with the same R-hat
    inp = np.load(tmp_path / "inputs.npz")
    for s in range(S):
        ref, idx = R.gelman_rubin(inp["means"][s], inp["var"][s])
        assert list(idx) == [True, False, True, True]
        assert np.isnan(r0[s, 1])
        np.testing.assert_allclose(r0[s, idx], ref, rtol=1e-9)
