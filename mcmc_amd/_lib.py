"""ctypes binding of libmi355mcmc.so (include/mi355mcmc.h).

There is no CPU fallback: if the shared library is missing, or no MI355X is visible when a
context is created, the caller gets an exception - never a silently different code path.
"""
import ctypes as C
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355MCMC_LIB") or os.path.join(_HERE, "libmi355mcmc.so")
ABI_VERSION = 3

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)
c_uint8_p = C.POINTER(C.c_uint8)


class SedModel(C.Structure):
    """mirror of mi355_sed_model"""
    _fields_ = [("model", C.c_int32), ("nparam", C.c_int32), ("h_over_k", C.c_double),
                ("nu_b", C.c_double), ("prefac", C.c_double), ("Tmin", C.c_double),
                ("Tmax", C.c_double), ("bmin", C.c_double), ("bmax", C.c_double),
                ("nu_rescale", C.c_double)]


class ClDataset(C.Structure):
    """mirror of mi355_cl_dataset"""
    _fields_ = [("num_points", C.c_int32), ("win_len", C.c_int32), ("obs", c_double_p),
                ("N_inv", c_double_p), ("var", c_double_p), ("xfactors", c_double_p),
                ("has_xfactor", c_uint8_p), ("window", c_double_p), ("win_min", c_int32_p),
                ("win_max", c_int32_p), ("calib_uncertainty", C.c_double), ("beam_err", c_double_p),
                ("has_pol", C.c_int32), ("inc_pol", c_uint8_p)]


class BeamPrior(C.Structure):
    """mirror of mi355_beam_prior"""
    _fields_ = [("center_min", C.c_double * 2), ("center_max", C.c_double * 2),
                ("sig_min", C.c_double), ("sig_max", C.c_double), ("has_center", C.c_int32),
                ("has_sig_min", C.c_int32), ("has_sig_max", C.c_int32)]


# name -> (restype, argtypes); every symbol include/mi355mcmc.h declares
SIGNATURES = {
    "mi355_init": (C.c_int, [C.c_int, C.c_uint64, C.POINTER(C.c_void_p)]),
    "mi355_free": (None, [C.c_void_p]),
    "mi355_last_error": (C.c_char_p, []),
    "mi355_abi_version": (C.c_int, []),
    "mi355_last_kernel_ms": (C.c_double, [C.c_void_p]),
    "mi355_stream": (C.c_void_p, [C.c_void_p]),
    "mi355_sync": (C.c_int, [C.c_void_p]),
    "mi355_sed_model_default": (C.c_int, [C.c_int32, C.POINTER(SedModel)]),
    "mi355_sed_nparam": (C.c_int, [C.c_int32]),
    "mi355_sed_load": (C.c_int, [C.c_void_p, C.POINTER(SedModel), C.c_int64, C.c_int32, c_int32_p,
                                 c_double_p, c_double_p, c_double_p]),
    "mi355_sed_lnlike": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, c_double_p,
                                   c_double_p, c_double_p]),
    "mi355_sed_lnlike_dev_async": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p,
                                             C.c_void_p, C.c_void_p]),
    "mi355_sed_flux": (C.c_int, [C.c_void_p, C.c_int64, c_double_p, C.c_int64, c_double_p, c_double_p]),
    "mi355_sed_prior_package": (C.c_int, [C.c_void_p, C.c_int64, c_double_p, c_double_p, c_double_p]),
    "mi355_mh_init": (C.c_int, [C.c_void_p, C.c_int32, c_double_p, C.c_int32, c_double_p, C.c_int32,
                                c_int32_p, C.c_int32, C.c_int32]),
    "mi355_mh_run": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, c_double_p, c_double_p]),
    "mi355_mh_moments": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_int64_p, c_int64_p,
                                   c_double_p, c_double_p]),
    "mi355_mh_sigmas": (C.c_int, [C.c_void_p, c_double_p]),
    "mi355_sed_fast_sources": (C.c_int64, [C.c_void_p]),
    "mi355_mh_set_lane_map": (C.c_int, [C.c_void_p, C.c_int32]),
    "mi355_mh_lane_group": (C.c_int, [C.c_void_p]),
    "mi355_mh_set_incremental": (C.c_int, [C.c_void_p, C.c_int32]),
    "mi355_mh_incremental_active": (C.c_int, [C.c_void_p]),
    "mi355_mh_stream_pos": (C.c_int64, [C.c_void_p, C.c_int32]),
    "mi355_mh_set_stream_pos": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64]),
    "mi355_mh_nsamples": (C.c_int64, [C.c_void_p]),
    "mi355_mh_ntrace": (C.c_int64, [C.c_void_p]),
    "mi355_mh_trace": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_uint8_p]),
    "mi355_mh_adapt": (C.c_int, [C.c_void_p, C.c_double, C.c_int32]),
    "mi355_rng_dump": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, c_double_p,
                                 c_double_p]),
    "mi355_gauss_quadform": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, c_double_p, c_double_p,
                                       C.c_int64, c_double_p, C.c_int64, c_double_p]),
    "mi355_math_probe": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, c_double_p, c_double_p, c_double_p,
                                   c_double_p]),
    "mi355_cl_load": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(ClDataset)]),
    "mi355_cl_set_binning": (C.c_int, [C.c_void_p, C.c_int32, c_int32_p, c_int32_p, c_double_p, C.c_int32]),
    "mi355_cl_nparam": (C.c_int32, [C.c_void_p]),
    "mi355_cl_lnlike": (C.c_int, [C.c_void_p, C.c_int64, c_double_p, c_double_p]),
    "mi355_cl_lnlike_dev_async": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "mi355_cl_prior": (C.c_int, [C.c_void_p, C.c_int64, c_double_p, c_double_p]),
    "mi355_cl_model": (C.c_int, [C.c_void_p, C.c_int64, c_double_p, c_double_p]),
    "mi355_cl_bandpowers": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, c_double_p,
                                      C.c_int32, C.c_int32, c_double_p]),
    "mi355_cl_data_op": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, c_double_p, c_double_p]),
    "mi355_cl_calc_lnlike": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, c_double_p, C.c_int32,
                                       C.c_int32, c_double_p]),
    "mi355_beam_load": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, c_double_p, c_double_p, C.c_int32,
                                  c_double_p, c_double_p, C.c_int32, C.POINTER(BeamPrior)]),
    "mi355_beam_lnlike": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, c_double_p, c_double_p]),
    "mi355_beam_lnlike_dev_async": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p,
                                              C.c_void_p]),
    "mi355_beam_prior_package": (C.c_int, [C.c_void_p, C.c_int64, c_double_p, c_double_p, c_double_p]),
    "mi355_beam_model": (C.c_int, [C.c_void_p, C.c_int64, c_double_p, C.c_int64, c_double_p, c_double_p,
                                   c_double_p]),
    "mi355_beam_gauss2d": (C.c_int, [C.c_void_p, C.c_int64, c_double_p, c_double_p, C.c_double, C.c_double,
                                    C.c_double, C.c_double, C.c_double, c_double_p]),
    "mi355_beam_stats": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_int32, c_double_p]),
    "mi355_mh_init_problem": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, c_double_p, C.c_int32,
                                        c_double_p, C.c_int32, c_int32_p, C.c_int32, C.c_int32,
                                        C.c_int32]),
    "mi355_mh_track_cov": (C.c_int, [C.c_void_p, C.c_int32]),
    "mi355_mh_set_fused": (C.c_int, [C.c_void_p, C.c_int32]),
    "mi355_mh_fused_active": (C.c_int, [C.c_void_p]),
    "mi355_mh_set_sqrt_matrix": (C.c_int, [C.c_void_p, c_double_p, C.c_int32, C.c_int32]),
    "mi355_mh_adapt_corr": (C.c_int, [C.c_void_p, C.c_double, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_int32]),
    "mi355_mh_covar": (C.c_int, [C.c_void_p, c_double_p, C.c_int32]),
    "mi355_mh_get_sqrt_matrix": (C.c_int, [C.c_void_p, c_double_p]),
    "mi355_mh_proposal_mode": (C.c_int32, [C.c_void_p]),
    "mi355_rhat_partial_async": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "mi355_rhat_finish_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "mi355_gelman_rubin": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, c_double_p, c_double_p, c_double_p]),
    "mi355_rhat": (C.c_int, [C.c_void_p, c_double_p]),
    "mi355_rhat_noalloc_async": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mi355_rhat_allreduce_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "mi355_mh_ess": (C.c_int, [C.c_void_p, c_double_p]),
}

_lib = None


class Mi355Error(RuntimeError):
    pass


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.  A process must run on ONE HIP runtime:
    if this library pulls in /opt/rocm's copy first, a later `import torch` binds to it and finds
    no GPU.  So when torch is installed but not imported yet, its runtime is loaded first (the
    order every `import torch; import mcmc_amd` program has anyway)."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load the shared library (once) and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    _share_torch_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise Mi355Error(
            "libmi355mcmc.so not found at %s: build it with `make -C mcmc_amd/csrc` "
            "(or __graft_entry__.build()); mcmc_amd has no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.mi355_abi_version() != ABI_VERSION:
        raise Mi355Error("libmi355mcmc.so ABI %d != binding ABI %d" %
                         (lib.mi355_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise Mi355Error(load().mi355_last_error().decode("utf-8", "replace"))
