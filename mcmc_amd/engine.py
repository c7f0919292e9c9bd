"""Thin numpy-facing wrapper of the C ABI: one Engine = one mi355_ctx = one GPU + one stream.

The reference-shaped classes (mcmc_amd.MCMC, .Likelihood, .submmSED ...) sit on top of this."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import BeamPrior, ClDataset, Mi355Error, SedModel, c_double_p, c_int32_p, c_int64_p, c_uint8_p, check

MODEL_IDS = {"M1N": 0, "M1N_LOGA": 1, "M2N": 2, "M2N_LOGA": 3, "M1": 4, "M2": 5, "THICK": 6,
             "THICK_LOGA": 7}


def _dp(a):
    return a.ctypes.data_as(c_double_p) if a is not None else None


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError("expected array of shape %s, got %s" % (tuple(shape), a.shape))
    return a


class Engine:
    def __init__(self, device=0, seed=0x5EDC0FFEE):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.mi355_init(int(device), C.c_uint64(int(seed) & (2 ** 64 - 1)), C.byref(h)))
        self._h = h
        self.device = int(device)
        self.seed = int(seed)
        self.P = self.S = self.C = self.nblock = 0
        self.nb_max = 0

    def close(self):
        if getattr(self, "_h", None):
            self.lib.mi355_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def N(self):
        return self.S * self.C

    @property
    def stream(self):
        return self.lib.mi355_stream(self._h)

    @property
    def last_kernel_ms(self):
        """HIP-event duration of the main kernel of the last mh_run / sed_lnlike call."""
        return float(self.lib.mi355_last_kernel_ms(self._h))

    def sync(self):
        check(self.lib.mi355_sync(self._h))

    # ---- SED catalogue -----------------------------------------------------------------
    def default_model(self, model):
        m = SedModel()
        check(self.lib.mi355_sed_model_default(int(MODEL_IDS.get(model, model)), C.byref(m)))
        return m

    def sed_load(self, model, nb, nu, d, sig, consts=None):
        """nb [S] int32; nu, d, sig [S, nb_max]."""
        m = consts if isinstance(consts, SedModel) else self.default_model(model)
        if isinstance(consts, dict):
            for k, v in consts.items():
                setattr(m, k, v)
        nu = _f64(nu)
        if nu.ndim != 2:
            raise ValueError("nu must be [S, nb_max]")
        S, nbm = nu.shape
        d = _f64(d, (S, nbm))
        sig = _f64(sig, (S, nbm))
        nb = np.ascontiguousarray(nb, dtype=np.int32)
        if nb.shape != (S,):
            raise ValueError("nb must be [S]")
        check(self.lib.mi355_sed_load(self._h, C.byref(m), S, nbm, nb.ctypes.data_as(c_int32_p),
                                      _dp(nu), _dp(d), _dp(sig)))
        self.model = m
        self.P, self.S, self.nb_max = int(m.nparam), int(S), int(nbm)
        self.S_sed = int(S)
        self.C = 0

    def sed_lnlike(self, theta, s0=0, derived=False):
        """theta [ns, C, P] -> lnL [ns, C] (and derived [ns, C, 2])."""
        theta = _f64(theta)
        if theta.ndim != 3 or theta.shape[2] != self.P:
            raise ValueError("theta must be [ns, C, %d]" % self.P)
        ns, Cc, _ = theta.shape
        out = np.empty((ns, Cc))
        der = np.empty((ns, Cc, 2)) if derived else None
        check(self.lib.mi355_sed_lnlike(self._h, int(s0), ns, Cc, _dp(theta), _dp(out), _dp(der)))
        return (out, der) if derived else out

    def mh_set_lane_map(self, mode):
        """0 automatic, 1 lane = chain, 2 lanes = (chain, band) in the fused SED sampler
        (mi355_mh_set_lane_map); same chains either way"""
        check(self.lib.mi355_mh_set_lane_map(self._h, int(mode)))

    @property
    def mh_lane_group(self):
        """lanes per chain the next mh_run would use (1, 8 or 32; -1: no fused SED chains)"""
        return int(self.lib.mi355_mh_lane_group(self._h))

    @property
    def sed_fast_sources(self):
        """number of loaded sources inside the domain of the clamp-free kernels
        (mi355_sed_fast_sources); the others evaluate through the exact path, source by source"""
        return int(self.lib.mi355_sed_fast_sources(self._h))

    def sed_lnlike_dev(self, theta_ptr, lnl_ptr, ns, Cc, s0=0, derived_ptr=None):
        check(self.lib.mi355_sed_lnlike_dev_async(self._h, int(s0), int(ns), int(Cc),
                                                  C.c_void_p(theta_ptr), C.c_void_p(lnl_ptr),
                                                  C.c_void_p(derived_ptr) if derived_ptr else None))

    def sed_flux(self, theta, nu):
        """theta [n,P] (or [P]), nu [nnu] -> flux [n, nnu] ([n, nnu, 2] for submmModel2)."""
        theta = _f64(theta)
        single = theta.ndim == 1
        th = theta.reshape(-1, self.P)
        nu = _f64(nu).reshape(-1)
        nc = 2 if int(self.model.model) == MODEL_IDS["M2"] else 1
        out = np.empty((th.shape[0], nu.size, nc))
        check(self.lib.mi355_sed_flux(self._h, th.shape[0], _dp(th), nu.size, _dp(nu), _dp(out)))
        if nc == 1:
            out = out[..., 0]
        return out[0] if single else out

    def sed_prior_package(self, theta):
        theta = _f64(theta)
        flat = theta.reshape(-1, self.P)
        pr = np.empty(flat.shape[0])
        pk = np.empty_like(flat)
        check(self.lib.mi355_sed_prior_package(self._h, flat.shape[0], _dp(flat), _dp(pr), _dp(pk)))
        return pr.reshape(theta.shape[:-1]), pk.reshape(theta.shape)

    # ---- Metropolis-Hastings -------------------------------------------------------------
    def mh_init(self, C_chains, theta0, sigma, param_blocks=None, randomize_start=True):
        theta0 = _f64(theta0)
        sigma = _f64(sigma)
        Cc = int(C_chains)
        self.P, self.S = int(self.model.nparam), self.S_sed     # (a step-wise sampler may have changed them)
        per_t = int(theta0.ndim > 1)
        per_s = int(sigma.ndim > 1)
        if per_t:
            theta0 = _f64(theta0.reshape(self.S, Cc, self.P))
        elif theta0.shape != (self.P,):
            raise ValueError("theta0 must be [P] or [S,C,P]")
        if per_s:
            sigma = _f64(sigma.reshape(self.S, Cc, self.P))
        elif sigma.shape != (self.P,):
            raise ValueError("sigma must be [P] or [S,C,P]")
        if param_blocks is None:
            pb, nblock = None, 0
        else:
            pb = np.ascontiguousarray(param_blocks, dtype=np.int32)
            if pb.shape != (self.P,):
                raise ValueError("param_blocks must be [P]")
            nblock = int(pb.max()) + 1
        check(self.lib.mi355_mh_init(self._h, Cc, _dp(theta0), per_t, _dp(sigma), per_s,
                                     pb.ctypes.data_as(c_int32_p) if pb is not None else None,
                                     nblock, int(bool(randomize_start))))
        self.C, self.nblock = Cc, nblock

    PROBLEMS = {"sed": 0, "cl": 1, "beam": 2}

    def mh_init_problem(self, problem, C_chains, theta0, sigma, param_blocks=None,
                        randomize_start=True, track_cov=False):
        """Step-wise sampler over the loaded problem ("sed", "cl" or "beam"): any proposal of
        Proposal.py (uncorrelated, Cholesky-correlated, rotated).  Chain layout [units, C, P] with
        units = sources (sed), 1 (cl) or maps (beam)."""
        kind = self.PROBLEMS[problem] if isinstance(problem, str) else int(problem)
        if kind == 0:
            P, S = int(self.model.nparam), self.S_sed
        elif kind == 1:
            P, S = self.cl_npar, 1
        else:
            P, S = 9, self.beam_M
        Cc = int(C_chains)
        theta0, sigma = _f64(theta0), _f64(sigma)
        per_t, per_s = int(theta0.ndim > 1), int(sigma.ndim > 1)
        if per_t:
            theta0 = _f64(theta0.reshape(S, Cc, P))
        elif theta0.shape != (P,):
            raise ValueError("theta0 must be [P] or [units,C,P]")
        if per_s:
            sigma = _f64(sigma.reshape(S, Cc, P))
        elif sigma.shape != (P,):
            raise ValueError("sigma must be [P] or [units,C,P]")
        if param_blocks is None:
            pb, nblock = None, 0
        else:
            pb = np.ascontiguousarray(param_blocks, dtype=np.int32)
            if pb.shape[0] < P:                       # parameters beyond the table are in no block
                pb = np.concatenate([pb, np.full(P - pb.shape[0], -1, np.int32)])
            pb = np.ascontiguousarray(pb[:P])
            nblock = int(pb.max()) + 1
        check(self.lib.mi355_mh_init_problem(
            self._h, kind, Cc, _dp(theta0), per_t, _dp(sigma), per_s,
            pb.ctypes.data_as(c_int32_p) if pb is not None else None, nblock,
            int(bool(randomize_start)), int(bool(track_cov))))
        self.P, self.S, self.C, self.nblock = P, S, Cc, nblock

    def mh_set_sqrt_matrix(self, L, rotate=False):
        if L is None:
            check(self.lib.mi355_mh_set_sqrt_matrix(self._h, None, 0, 0))
            return
        L = _f64(L)
        per = int(L.ndim > 2)
        L = _f64(L.reshape(self.S, self.C, self.P, self.P)) if per else _f64(L, (self.P, self.P))
        check(self.lib.mi355_mh_set_sqrt_matrix(self._h, _dp(L), per, int(bool(rotate))))

    def mh_adapt_corr(self, fac=None, random_restart=False, no_correlations=False, rotate=False,
                      do_block=None):
        """newMCMC() for every chain, with the correlated / rotated proposals of the reference."""
        db = -1 if do_block is None else int(bool(do_block))
        check(self.lib.mi355_mh_adapt_corr(self._h, float(fac) if fac else 0.0,
                                           int(bool(random_restart)), int(bool(no_correlations)),
                                           int(bool(rotate)), db))
        if do_block is not None and not do_block:
            self.nblock = 0

    def mh_covar(self, normalized=True):
        out = np.empty((self.S, self.C, self.P, self.P))
        check(self.lib.mi355_mh_covar(self._h, _dp(out), int(bool(normalized))))
        return out

    def mh_sqrt_matrix(self):
        out = np.empty((self.S, self.C, self.P, self.P))
        check(self.lib.mi355_mh_get_sqrt_matrix(self._h, _dp(out)))
        return out

    @property
    def proposal_mode(self):
        return int(self.lib.mi355_mh_proposal_mode(self._h))

    def mh_run(self, nsteps, thin=0, inj_normals=None, inj_uniforms=None):
        if inj_normals is not None:
            inj_normals = _f64(inj_normals, (nsteps, self.N, self.P))
            inj_uniforms = _f64(inj_uniforms, (nsteps, self.N))
        check(self.lib.mi355_mh_run(self._h, int(nsteps), int(thin), _dp(inj_normals),
                                    _dp(inj_uniforms)))

    def mh_moments(self):
        S, Cc, P = self.S, self.C, self.P
        out = dict(theta=np.empty((S, Cc, P)), lnPr=np.empty((S, Cc)),
                   naccept=np.empty((S, Cc), np.int64),
                   accept=np.empty((S, Cc, max(self.nblock, 0)), np.int64))
        have = self.nsamples > 0
        if have:
            out["mean"] = np.empty((S, Cc, P))
            out["var"] = np.empty((S, Cc, P))
        check(self.lib.mi355_mh_moments(
            self._h, _dp(out.get("mean")), _dp(out.get("var")),
            out["naccept"].ctypes.data_as(c_int64_p),
            out["accept"].ctypes.data_as(c_int64_p) if self.nblock else None,
            _dp(out["lnPr"]), _dp(out["theta"])))
        return out

    def mh_sigmas(self):
        out = np.empty((self.S, self.C, self.P))
        check(self.lib.mi355_mh_sigmas(self._h, _dp(out)))
        return out

    @property
    def nsamples(self):
        return int(self.lib.mi355_mh_nsamples(self._h))

    @property
    def step_total(self):
        return int(self.lib.mi355_mh_stream_pos(self._h, 0))

    @property
    def stage(self):
        return int(self.lib.mi355_mh_stream_pos(self._h, 1))

    def mh_set_incremental(self, on=True):
        """incremental evaluation in the fused SED sampler (mi355_mh_set_incremental): a step
        re-evaluates only what its parameter touches; same chain, about twice the step rate for
        the two-component models in single-parameter block mode"""
        check(self.lib.mi355_mh_set_incremental(self._h, 1 if on else 0))

    def mh_track_cov(self, on=True):
        """pause / resume the running co-moments of a sampler started with track_cov=True"""
        check(self.lib.mi355_mh_track_cov(self._h, 1 if on else 0))

    def mh_set_fused(self, on=True):
        """one-launch runs of step-wise problems where a fused kernel exists (binnedCl); on by
        default, off forces the launch-per-step engine (mi355_mh_set_fused)"""
        check(self.lib.mi355_mh_set_fused(self._h, 2 if on == "always" else (1 if on else 0)))

    @property
    def mh_fused_active(self):
        return bool(self.lib.mi355_mh_fused_active(self._h))

    @property
    def mh_incremental_active(self):
        return bool(self.lib.mi355_mh_incremental_active(self._h))

    def set_stream_pos(self, step, stage):
        check(self.lib.mi355_mh_set_stream_pos(self._h, int(step), int(stage)))

    def mh_trace(self):
        nk = int(self.lib.mi355_mh_ntrace(self._h))
        S, Cc, P = self.S, self.C, self.P
        samples = np.empty((nk, S, Cc, P))
        lnPr = np.empty((nk, S, Cc))
        acc = np.empty((nk, S, Cc), np.uint8)
        check(self.lib.mi355_mh_trace(self._h, _dp(samples), _dp(lnPr), acc.ctypes.data_as(c_uint8_p)))
        return samples, lnPr, acc.astype(bool)

    def mh_adapt(self, fac=None, random_restart=False):
        check(self.lib.mi355_mh_adapt(self._h, float(fac) if fac else 0.0, int(bool(random_restart))))

    def rng_dump(self, step0, nsteps, N, npn):
        z = np.empty((nsteps, N, npn))
        u = np.empty((nsteps, N))
        check(self.lib.mi355_rng_dump(self._h, int(step0), int(nsteps), int(N), int(npn), _dp(z), _dp(u)))
        return z, u

    def gauss_quadform(self, A, sig2, B=None):
        """rows of A [m,n] (or one vector [n]) -> A^T N^-1 B per row, N = diag(sig2)."""
        A = _f64(A)
        single = A.ndim == 1
        A2 = A.reshape(1, -1) if single else A
        m, n = A2.shape
        sig2 = _f64(sig2).reshape(-1)
        b_rows = 0
        if B is not None:
            B = _f64(B)
            B = B.reshape(-1, n)
            b_rows = B.shape[0]
        out = np.empty(m)
        check(self.lib.mi355_gauss_quadform(self._h, m, n, _dp(A2), _dp(B), b_rows, _dp(sig2),
                                            sig2.size, _dp(out)))
        return float(out[0]) if single else out

    def math_probe(self, fn, a, b=None, c=None):
        a = _f64(a)
        b = _f64(b, a.shape) if b is not None else None
        c = _f64(c, a.shape) if c is not None else None
        out = np.empty_like(a)
        check(self.lib.mi355_math_probe(self._h, int(fn), a.size, _dp(a), _dp(b), _dp(c), _dp(out)))
        return out

    # ---- binnedCl / ClData -------------------------------------------------------------------
    def cl_load(self, datasets):
        """datasets: sequence of objects with the ClData_CosmoMC attributes (num_points, obs,
        N_inv | var, xfactors, has_xfactor, window, win_min, win_max, calib_uncertainty,
        beam_uncertain, beam_err, has_pol, inc_pol)."""
        arr = (ClDataset * len(datasets))()
        keep = []

        def f64(a):
            a = np.ascontiguousarray(a, dtype=np.float64)
            keep.append(a)
            return a.ctypes.data_as(c_double_p)

        def u8(a):
            a = np.ascontiguousarray(a, dtype=np.uint8)
            keep.append(a)
            return a.ctypes.data_as(c_uint8_p)

        def i32(a):
            a = np.ascontiguousarray(a, dtype=np.int32)
            keep.append(a)
            return a.ctypes.data_as(c_int32_p)

        self.cl_nb = []
        for k, d in enumerate(datasets):
            nb = int(d.num_points)
            window = np.asarray(d.window, dtype=np.float64)
            if window.ndim == 3 and window.shape[:2] == (nb, 1):      # TT-only file data (ncls = 1)
                window = np.concatenate((window, np.zeros((nb, 2, window.shape[2]))), axis=1)
            if window.ndim != 3 or window.shape[0] != nb or window.shape[1] != 3:
                raise ValueError("window must be [num_points, 3, lmax+1] (or [num_points, 1, lmax+1])")
            corr = bool(getattr(d, "has_corr_errors", getattr(d, "N_inv", None) is not None))
            e = arr[k]
            e.num_points, e.win_len = nb, window.shape[2]
            e.obs = f64(np.asarray(d.obs).reshape(nb))
            e.N_inv = f64(np.asarray(d.N_inv).reshape(nb, nb)) if corr else None
            e.var = None if corr else f64(np.asarray(d.var).reshape(nb))
            hasx = bool(getattr(d, "has_xfactors", False))
            e.xfactors = f64(np.asarray(d.xfactors).reshape(nb)) if hasx else None
            e.has_xfactor = u8(np.asarray(d.has_xfactor).reshape(nb)) if hasx else None
            e.window = f64(window)
            e.win_min, e.win_max = i32(d.win_min), i32(d.win_max)
            e.calib_uncertainty = float(getattr(d, "calib_uncertainty", 0.0))
            beam = bool(getattr(d, "beam_uncertain", False))
            e.beam_err = f64(np.broadcast_to(np.asarray(d.beam_err, np.float64), (nb,))) if beam else None
            pol = bool(getattr(d, "has_pol", False))
            e.has_pol = int(pol)
            e.inc_pol = u8(np.asarray(d.inc_pol).reshape(nb)) if pol else None
            self.cl_nb.append(nb)
        check(self.lib.mi355_cl_load(self._h, len(datasets), arr))
        self.cl_npar = 0

    def cl_set_binning(self, bins, shapefun):
        """bins: per spectrum [nbin, 2] inclusive ranges; shapefun [nCl, lmax+1]."""
        bins = [np.asarray(b, dtype=np.int32).reshape(-1, 2) for b in bins]
        nbins = np.array([len(b) for b in bins], dtype=np.int32)
        flat = np.ascontiguousarray(np.concatenate(bins), dtype=np.int32)
        lmax = int(flat.max())
        shapefun = _f64(np.atleast_2d(shapefun))
        if shapefun.shape != (len(bins), lmax + 1):
            raise ValueError("shapefun must be [nCl=%d, lmax+1=%d] (truncate it to the largest binned "
                             "multipole), got %s" % (len(bins), lmax + 1, shapefun.shape))
        check(self.lib.mi355_cl_set_binning(self._h, len(bins), nbins.ctypes.data_as(c_int32_p),
                                            flat.ctypes.data_as(c_int32_p), _dp(shapefun), lmax))
        self.cl_npar = int(self.lib.mi355_cl_nparam(self._h))
        self.cl_lmax = lmax

    def cl_lnlike(self, q):
        q = _f64(q)
        single = q.ndim == 1
        q2 = q.reshape(-1, self.cl_npar)
        out = np.empty(q2.shape[0])
        check(self.lib.mi355_cl_lnlike(self._h, q2.shape[0], _dp(q2), _dp(out)))
        return float(out[0]) if single else out

    def cl_lnlike_dev(self, q_ptr, lnl_ptr, n):
        check(self.lib.mi355_cl_lnlike_dev_async(self._h, int(n), C.c_void_p(q_ptr), C.c_void_p(lnl_ptr)))

    def cl_prior(self, q):
        q = _f64(q)
        single = q.ndim == 1
        q2 = q.reshape(-1, self.cl_npar)
        out = np.empty(q2.shape[0])
        check(self.lib.mi355_cl_prior(self._h, q2.shape[0], _dp(q2), _dp(out)))
        return out[0] if single else out

    def cl_model(self, q):
        q = _f64(q)
        single = q.ndim == 1
        q2 = q.reshape(-1, self.cl_npar)
        out = np.empty((q2.shape[0], 3, self.cl_lmax + 1))
        check(self.lib.mi355_cl_model(self._h, q2.shape[0], _dp(q2), _dp(out)))
        return out[0] if single else out

    def cl_bandpowers(self, idata, q=None, Cl=None):
        nb = self.cl_nb[idata]
        if q is not None:
            a = _f64(q).reshape(-1, self.cl_npar)
            what, rows, L = 0, 0, 0
        else:
            a = _f64(Cl)
            if a.ndim == 1:
                a = a[None, None, :]
            elif a.ndim == 2:
                a = a[None]
            what, rows, L = 1, a.shape[1], a.shape[2]
        out = np.empty((a.shape[0], nb))
        check(self.lib.mi355_cl_bandpowers(self._h, int(idata), what, a.shape[0], _dp(a), rows, L, _dp(out)))
        return out

    def cl_data_op(self, idata, op, vec):
        nb = self.cl_nb[idata]
        v = _f64(vec)
        single = v.ndim == 1
        v2 = v.reshape(-1, nb)
        out = np.empty(v2.shape if op in (0, 1) else v2.shape[0])
        check(self.lib.mi355_cl_data_op(self._h, int(idata), int(op), v2.shape[0], _dp(v2), _dp(out)))
        return out[0] if single else out

    def cl_calc_lnlike(self, idata, Cl):
        a = _f64(Cl)
        single = a.ndim <= 2
        if a.ndim == 1:
            a = a[None, None, :]
        elif a.ndim == 2:
            a = a[None]
        out = np.empty(a.shape[0])
        check(self.lib.mi355_cl_calc_lnlike(self._h, int(idata), a.shape[0], _dp(a), a.shape[1],
                                            a.shape[2], _dp(out)))
        return float(out[0]) if single else out

    # ---- BeamFit -----------------------------------------------------------------------------
    def beam_load(self, x, y, d, sig, center_min=None, center_max=None, sig_min=None, sig_max=None):
        """d [M, npix] (or [npix]); x, y [npix] shared or [M, npix]; sig scalar, [M] or [M, npix]."""
        d = _f64(d)
        if d.ndim == 1:
            d = d[None]
        M, npix = d.shape
        x, y = _f64(x), _f64(y)
        per_map = int(x.ndim == 2)
        if x.shape != y.shape or x.shape[-1] != npix or (per_map and x.shape[0] != M):
            raise ValueError("DataSizeError: x, y must be [npix] or [M, npix] matching d")
        sig = _f64(np.atleast_1d(sig))
        if sig.size == 1:
            kind = 0
        elif sig.shape == (M,) and M != npix:
            kind = 1
        elif sig.size == M * npix:
            kind, sig = 2, _f64(sig.reshape(M, npix))
        else:
            raise ValueError("DataSizeError: %d data points but %d sigmas" % (npix, sig.size))
        pr = BeamPrior()
        pr.has_center = int(center_min is not None)
        if center_min is not None:
            pr.center_min[0], pr.center_min[1] = float(center_min[0]), float(center_min[1])
            pr.center_max[0], pr.center_max[1] = float(center_max[0]), float(center_max[1])
        pr.has_sig_min = int(sig_min is not None)
        pr.sig_min = float(sig_min) if sig_min is not None else 0.0
        pr.has_sig_max = int(sig_max is not None)
        pr.sig_max = float(sig_max) if sig_max is not None else 0.0
        check(self.lib.mi355_beam_load(self._h, M, npix, _dp(x), _dp(y), per_map, _dp(d), _dp(sig),
                                       kind, C.byref(pr)))
        self.beam_M, self.beam_npix = M, npix

    def beam_lnlike(self, theta, m0=0):
        """theta [nm, C, 9] -> lnL [nm, C]; [9] -> float (map m0)."""
        theta = _f64(theta)
        single = theta.ndim == 1
        th = theta.reshape(1, 1, 9) if single else (theta[None] if theta.ndim == 2 else theta)
        nm, Cc, _ = th.shape
        out = np.empty((nm, Cc))
        check(self.lib.mi355_beam_lnlike(self._h, int(m0), nm, Cc, _dp(_f64(th)), _dp(out)))
        if single:
            return float(out[0, 0])
        return out[0] if theta.ndim == 2 else out

    def beam_lnlike_dev(self, theta_ptr, lnl_ptr, nm, Cc, m0=0):
        check(self.lib.mi355_beam_lnlike_dev_async(self._h, int(m0), int(nm), int(Cc),
                                                   C.c_void_p(theta_ptr), C.c_void_p(lnl_ptr)))

    def beam_prior_package(self, theta):
        theta = _f64(theta)
        flat = _f64(theta.reshape(-1, 9))
        pr = np.empty(flat.shape[0])
        pk = np.empty_like(flat)
        check(self.lib.mi355_beam_prior_package(self._h, flat.shape[0], _dp(flat), _dp(pr), _dp(pk)))
        return pr.reshape(theta.shape[:-1]), pk.reshape(theta.shape)

    def beam_model(self, theta, x, y):
        theta = _f64(theta)
        single = theta.ndim == 1
        flat = _f64(theta.reshape(-1, 9))
        x, y = _f64(x).reshape(-1), _f64(y).reshape(-1)
        out = np.empty((flat.shape[0], x.size))
        check(self.lib.mi355_beam_model(self._h, flat.shape[0], _dp(flat), x.size, _dp(x), _dp(y), _dp(out)))
        return out[0] if single else out

    def beam_gauss2d(self, x, y, x0, y0, cxx, cxy, cyy):
        """gauss2d of BeamModel.py:288-290 at the points (x, y) for a packed inverse covariance"""
        x = _f64(x)
        shape = x.shape
        x = np.ascontiguousarray(x.reshape(-1))
        y = _f64(np.broadcast_to(np.asarray(y, np.float64), shape)).reshape(-1)
        out = np.empty(x.size)
        check(self.lib.mi355_beam_gauss2d(self._h, x.size, _dp(x), _dp(y), float(x0), float(y0),
                                          float(cxx), float(cxy), float(cyy), _dp(out)))
        return out.reshape(shape)

    def beam_stats(self, m=0, sigcut=None, do_abs=False):
        out = np.empty(4)
        check(self.lib.mi355_beam_stats(self._h, int(m), int(sigcut is not None),
                                        float(sigcut) if sigcut is not None else 0.0, int(do_abs),
                                        _dp(out)))
        return tuple(out)

    # ---- Gelman-Rubin ----------------------------------------------------------------------
    def rhat(self):
        out = np.empty((self.S, self.P))
        check(self.lib.mi355_rhat(self._h, _dp(out)))
        return out

    def rhat_noalloc(self, rhat_ptr):
        """mi355_rhat with no allocation anywhere: result [P, S] into device memory at rhat_ptr,
        asynchronous on the engine's stream (sync() before reading)."""
        check(self.lib.mi355_rhat_noalloc_async(self._h, C.c_void_p(rhat_ptr)))

    def rhat_allreduce(self, rccl_comm, Ctot, rhat_ptr):
        """R-hat over the chains of all ranks of an RCCL communicator (its ncclComm_t as an integer /
        c_void_p; None: this rank holds every chain): result [P, S] into device memory at rhat_ptr,
        asynchronous on the engine's stream.
        The path for callers without torch.distributed; engine_rhat (mcmc_amd.distributed) is the other."""
        check(self.lib.mi355_rhat_allreduce_async(self._h, C.c_void_p(rccl_comm or None), int(Ctot), C.c_void_p(rhat_ptr)))

    def mh_ess(self):
        """Effective sample size [S, P] from the trace of the last mh_run (BDA3 11.5)."""
        out = np.empty((self.S, self.P))
        check(self.lib.mi355_mh_ess(self._h, _dp(out)))
        return out

    def gelman_rubin(self, means, variances):
        means = _f64(means)
        variances = _f64(variances, means.shape)
        out = np.empty(means.shape[1])
        check(self.lib.mi355_gelman_rubin(self._h, means.shape[0], means.shape[1], _dp(means),
                                          _dp(variances), _dp(out)))
        return out

    def rhat_partial(self, pass_, sums_ptr, gmean_ptr=None):
        check(self.lib.mi355_rhat_partial_async(self._h, int(pass_),
                                                C.c_void_p(gmean_ptr) if gmean_ptr else None,
                                                C.c_void_p(sums_ptr)))

    def rhat_finish(self, sums_ptr, Ctot, rhat_ptr):
        check(self.lib.mi355_rhat_finish_async(self._h, C.c_void_p(sums_ptr), int(Ctot),
                                               C.c_void_p(rhat_ptr)))


_DEFAULT = {}


def default_engine(device=0):
    """Process-wide engine for stateless calls (GaussianData.quadform, probes)."""
    if device not in _DEFAULT:
        _DEFAULT[device] = Engine(device=device)
    return _DEFAULT[device]


__all__ = ["Engine", "Mi355Error", "MODEL_IDS", "default_engine"]
