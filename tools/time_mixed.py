"""Rate of a mixed catalogue (2e4 sources x 64 chains, ONE of them with a radio point outside the fast
kernels' domain) against the same catalogue without that point: the chain blocks of the slow source run
through the exact path on a second, high-priority stream beside the fast launch (1.02-1.06x;
serialised, the two launches would take 1.5x).  A measurement, not a parity test: the bit-identity of the
fast sources' chains with and without slow neighbours is tests/test_sed_gpu.py::
test_mixed_catalogue_splits_source_by_source.  GPU box: python tools/time_mixed.py"""
import sys

import numpy as np

sys.path.insert(0, ".")
from mcmc_amd import Engine           # noqa: E402
from oracle import refcpu as R        # noqa: E402  (catalogue generator only)

S, C, P = 20000, 64, 6
nu, d, sig, truth = R.synth_sed_catalogue(S, seed=31)
nb = np.full(S, 8, np.int32)
nu2 = nu.copy()
nu2[S // 2, 0] = 1.4
for attempt in range(3):
    ms = {}
    for tag, cat in (("clean", nu), ("mixed", nu2)):
        with Engine(seed=1) as e:
            e.sed_load(R.M2N, nb, cat, d, sig)
            e.mh_init(C, truth[:, None, :P].repeat(C, axis=1), np.full(P, 0.03),
                      param_blocks=np.arange(P), randomize_start=False)
            e.mh_run(200)
            t = []
            for _ in range(5):
                e.mh_run(200)
                t.append(e.last_kernel_ms)
            ms[tag] = float(np.min(t))
    print("200 steps of %d x %d chains: all fast %.2f ms, one slow source %.2f ms (%.3f)"
          % (S, C, ms["clean"], ms["mixed"], ms["mixed"] / ms["clean"]), flush=True)
