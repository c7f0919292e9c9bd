"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol
include/mi355mcmc.h declares, and refuses to work without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "mi355mcmc.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from mcmc_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build first: make -C mcmc_amd/csrc"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "library does not export %s" % s
    # and the ctypes binding table covers exactly the header
    assert sorted(_lib.SIGNATURES) == syms


def test_binding_loads_and_abi_version_matches():
    from mcmc_amd import _lib
    lib = _lib.load()
    assert lib.mi355_abi_version() == _lib.ABI_VERSION
    assert lib.mi355_sed_nparam(2) == 6 and lib.mi355_sed_nparam(99) == -1
    m = _lib.SedModel()
    assert lib.mi355_sed_model_default(2, ctypes.byref(m)) == 0
    assert (m.nparam, m.h_over_k, m.nu_b, m.Tmin, m.Tmax, m.bmax) == (6, 0.04799237, 1000.0, 3.0, 100.0, 3.0)


def test_no_cpu_fallback_without_gpu():
    """Without a visible GPU the product path must fail loudly, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    from mcmc_amd import Engine, Mi355Error
    with pytest.raises(Mi355Error, match="no HIP device|hip"):
        Engine(device=0)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mcmc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
