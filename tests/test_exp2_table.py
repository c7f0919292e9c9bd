"""The 2048-entry table behind the device's 2^y kernel (mcmc_amd/csrc/exp2_table.inc, written by
tools/gen_exp2_table.py): every entry is 2^(i/2048) correctly rounded to double."""
import os
import re

from mpmath import mp, mpf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_table_entries_are_correctly_rounded():
    text = open(os.path.join(ROOT, "mcmc_amd", "csrc", "exp2_table.inc")).read()
    vals = [float.fromhex(t) for t in re.findall(r"0x1\.[0-9a-f]+p[+-]\d+", text)]
    assert len(vals) == 2048 and vals[0] == 1.0
    mp.prec = 200
    for i, v in enumerate(vals):
        exact = mpf(2) ** (mpf(i) / 2048)
        assert v == float(exact), i          # mpmath -> float rounds to nearest
        assert 1.0 <= v < 2.0
    assert all(b > a for a, b in zip(vals, vals[1:]))


def test_header_uses_the_same_size():
    src = open(os.path.join(ROOT, "mcmc_amd", "csrc", "fastmath.hpp")).read()
    assert "constexpr int kExp2Bits = 11;" in src
    assert '#include "exp2_table.inc"' in src
