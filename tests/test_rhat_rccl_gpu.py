"""mi355_rhat_allreduce_async: the multi-GPU R-hat behind the C ABI alone (RCCL called by the library
on the caller's ncclComm_t, no torch.distributed).  One GPU here: a one-rank communicator made with
ncclCommInitAll through ctypes - the all-reduces run through RCCL and leave the sums as they are, so
the result must equal mi355_rhat's bit for bit.  (The multi-rank arithmetic - partial sums of chain
shards added up - is what tests/test_distributed.py checks with gloo and tests/
test_bench_multirank_gpu.py through torch.distributed.)"""
import ctypes
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def load_rccl():
    import torch
    cand = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")) + \
        ["/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"]
    for path in cand:
        try:
            return ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            continue
    pytest.skip("no RCCL library found")


def test_rhat_allreduce_on_a_one_rank_communicator_equals_rhat():
    import torch
    from mcmc_amd import Engine, synth
    rccl = load_rccl()
    S, C = 300, 16
    nu, d, sig, truth = synth.sed_catalogue(S, seed=5)
    nb = np.full(S, nu.shape[1], np.int32)
    with Engine(seed=3) as eng:
        eng.sed_load("M2N", nb, nu, d, sig)
        eng.mh_init(C, synth.M2N_START, synth.M2N_PROP_SIGMAS, param_blocks=np.arange(6))
        eng.mh_run(300)
        want = eng.rhat()                                             # [S, P]
        comm = ctypes.c_void_p()
        devs = (ctypes.c_int * 1)(eng.device)
        rccl.ncclCommInitAll.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        assert rccl.ncclCommInitAll(ctypes.byref(comm), 1, devs) == 0
        try:
            out = torch.empty((eng.P, S), dtype=torch.float64, device="cuda:%d" % eng.device)
            eng.rhat_allreduce(comm.value, C, out.data_ptr())
            eng.sync()
            assert np.array_equal(out.cpu().numpy().T, want, equal_nan=True)
            with pytest.raises(Exception, match="Ctot"):
                eng.rhat_allreduce(comm.value, C - 1, out.data_ptr())
            with pytest.raises(Exception, match="null communicator"):
                eng.rhat_allreduce(None, 2 * C, out.data_ptr())   # chains elsewhere, and no way to reach them
            out.zero_()
            eng.rhat_allreduce(None, C, out.data_ptr())            # every chain is here: no collective needed
            eng.sync()
            assert np.array_equal(out.cpu().numpy().T, want, equal_nan=True)
        finally:
            rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
            rccl.ncclCommDestroy(comm)
