"""The golden vectors' own arithmetic spread (oracle/simd_spread.py: the reference run twice, with
NumPy's wide SIMD code paths on and off) has to sit below the parity tolerance the GPU tests use,
or those tests would be measuring the host's libm rather than the kernels."""
import json
import os

import pytest

RTOL_LNL = 1e-12                                           # north-star tolerance on lnL
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def spread():
    with open(os.path.join(HERE, "golden", "simd_spread.json")) as f:
        return json.load(f)


def test_every_group_was_compared(spread):
    assert sorted(spread["files"]) == ["beam_g6.npz", "cl_g5.npz", "sed_g1.npz", "sed_g2.npz",
                                       "sed_g3.npz", "sed_g4.npz"]
    assert "AVX512F" in spread["disabled"] and "AVX2" in spread["disabled"]


def test_likelihood_values_spread_below_the_parity_tolerance(spread):
    worst = 0.0
    for name, rec in spread["files"].items():
        for key, val in rec["keys"].items():
            if "lnl" in key or "lnPr" in key or "chi2" in key or key.startswith(("grey", "bb_")):
                worst = max(worst, val)
    assert 0.0 < worst < 0.2 * RTOL_LNL, worst            # measured: 1.6e-13 (SEDLikelihood2)


def test_only_the_ill_conditioned_amplitudes_exceed_it(spread):
    """SEDLikelihood2's ML amplitudes come from a nearly singular 2x2 solve: they move by 1e-9
    between code paths, which is why tests/test_sed_gpu.py holds that model to 1e-8 and checks
    the derived amplitudes of SEDLikelihood1 only."""
    over = {(name, key) for name, rec in spread["files"].items()
            for key, val in rec["keys"].items() if val >= RTOL_LNL}
    assert over <= {("sed_g2.npz", "derived_5")}, over
