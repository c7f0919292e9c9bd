#!/usr/bin/env python3
"""Bound the golden vectors' own arithmetic spread (test infrastructure; SURVEY.md section 8c,
"third-party arithmetic"): the reference computes with NumPy ufuncs whose expm1 / power / exp /
log dispatch to different SIMD code paths on different hosts.  This script regenerates the
sed / cl / beam groups of oracle/make_golden.py - i.e. runs the REFERENCE again - with the wide
SIMD paths switched off (NPY_DISABLE_CPU_FEATURES) into a scratch directory, compares every
floating-point array with the committed fixture of the same name and writes the largest relative
difference per file and key to tests/golden/simd_spread.json.

Run in the build container only:   python oracle/simd_spread.py
The parity tolerance of the tests (1e-12 relative on lnL) has to sit above this spread, and
tests/test_golden_spread.py checks that it does."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "..", "tests", "golden")
FILES = ["sed_g1.npz", "sed_g2.npz", "sed_g3.npz", "sed_g4.npz", "cl_g5.npz", "beam_g6.npz"]
DISABLE = "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR AVX2 FMA3"


def rel_spread(a, b):
    """largest |a-b| / max(|a|, tiny) over entries finite in both; non-finite entries must agree
    in kind (nan with nan, +-inf with the same inf)"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    fin = np.isfinite(a) & np.isfinite(b)
    with np.errstate(invalid="ignore"):
        same_kind = bool(np.all((a[~fin] == b[~fin]) | (np.isnan(a[~fin]) & np.isnan(b[~fin]))))
    if not fin.any():
        return 0.0, same_kind
    den = np.maximum(np.abs(a[fin]), np.finfo(float).tiny)
    return float(np.max(np.abs(a[fin] - b[fin]) / den)), same_kind


def main():
    out = {"disabled": DISABLE, "numpy": np.__version__, "files": {}}
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, NPY_DISABLE_CPU_FEATURES=DISABLE, MCMC_GOLDEN_OUT=tmp)
        subprocess.run([sys.executable, os.path.join(HERE, "make_golden.py"), "sed", "cl", "beam"],
                       env=env, check=True, stdout=subprocess.DEVNULL)
        for name in FILES:
            ref, alt = np.load(os.path.join(GOLD, name)), np.load(os.path.join(tmp, name))
            rows, worst = {}, 0.0
            assert sorted(ref.files) == sorted(alt.files), name
            for key in ref.files:
                if ref[key].dtype.kind != "f":
                    assert np.array_equal(ref[key], alt[key]), (name, key)      # integers, flags, text
                    continue
                assert ref[key].shape == alt[key].shape, (name, key)
                spread, same_kind = rel_spread(ref[key], alt[key])
                assert same_kind, (name, key)
                if spread > 0.0:
                    rows[key] = spread
                worst = max(worst, spread)
            out["files"][name] = {"max_rel": worst, "keys": dict(sorted(rows.items(), key=lambda kv: -kv[1])[:8])}
            print("%-14s max rel spread %.3e" % (name, worst))
    with open(os.path.join(GOLD, "simd_spread.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
