"""Likelihood of a normalised beam model, interface of the reference's NormalizedBeamLikelihood
(MCMC/BeamFit/NormalizedBeamLikelihood.py:4-19): lnLike = -chi2/2 with the per-pixel residual
reduction on the GPU (csrc/beam_kernels.hpp)."""
import numpy as np

from .. import Likelihood
from ..engine import Engine


class NormalizedBeamLikelihood(Likelihood.Likelihood):
    problem_kind = "beam"

    def __init__(self, data=None, model=None, device=0):
        self._device = device
        self._eng = None
        self._key = None
        super().__init__(data, model)

    def setdata(self, data):
        super().setdata(data)
        self._eng = None

    def load_into(self, eng):
        """all data sets of this likelihood as maps of one engine (they share one theta)"""
        ds = self.data
        npix = ds[0].n
        if any(d.n != npix for d in ds):
            raise NotImplementedError("data sets of one beam likelihood must have equal sizes")
        x = np.array([np.ravel(d.x) for d in ds])
        y = np.array([np.ravel(d.y) for d in ds])
        dd = np.array([np.ravel(d.d) for d in ds])
        if all(d.nsig == 1 for d in ds):
            sig = np.array([float(np.ravel(d.sig)[0]) for d in ds])
            sig = sig[0] if len(ds) == 1 else np.broadcast_to(sig[:, None], dd.shape).copy()
        else:
            sig = np.array([np.broadcast_to(np.ravel(d.sig), (npix,)) for d in ds])
        eng.beam_load(x, y, dd, sig, **self.model.prior_state())

    @property
    def engine(self):
        key = (len(self.data), tuple(getattr(d, "version", 0) for d in self.data),
               tuple(sorted((k, str(v)) for k, v in self.model.prior_state().items())))
        if self._eng is None or self._key != key:
            eng = Engine(device=self._device)
            self.load_into(eng)
            self._eng, self._key = eng, key
        return self._eng

    def lnLike1(self, model_vals, data):                             # :9-16
        return -0.5 * data.chi2(model_vals)

    def lnNorm1(self, data=None):                                    # :18-19
        return -0.5 * data.lnDetN

    def lnLike(self, params):
        """params = model.package(flat) (the reference's structured tuple) or a flat [9] /
        [n, 9] array; sum over the data sets (Likelihood.py:99-105)."""
        flat = self._flat(params)
        eng = self.engine
        nd = len(self.data)
        if flat.ndim == 1:
            th = np.broadcast_to(flat, (nd, 1, 9))
            return float(sum(eng.beam_lnlike(np.ascontiguousarray(th))[:, 0]))
        th = np.broadcast_to(flat[None], (nd,) + flat.shape)
        return eng.beam_lnlike(np.ascontiguousarray(th)).sum(axis=0)

    def _flat(self, params):
        if isinstance(params, tuple) and len(params) == 5 and np.ndim(params[0]) == 1:
            xy, s12, ang, amp, plane = params
            return np.array([xy[0], xy[1], s12[0], s12[1], ang, amp, plane[0], plane[1], plane[2]],
                            dtype=np.float64)
        return np.asarray(params, dtype=np.float64)
