"""Seeded synthetic inputs of SURVEY.md section 8(d) for the benchmark configurations.

Input generation only (host numpy, run once before any timed region); it is not a likelihood
code path.  tests/ cross-check these arrays against the oracle's own generator."""
import numpy as np

SPEED_OF_LIGHT = 299792.458   # micron GHz (submmSED/data.py:28)
LAMBDA_UM = np.array([60.0, 100.0, 160.0, 250.0, 350.0, 500.0, 850.0, 1200.0])


def sed_catalogue(S, seed=20240521, nb=8, frac_err=0.07, h_over_k=0.04799237, nu_b=1000.0):
    """Two-component greybody sources (configs C2/C3): nu, d, sig [S, nb] and truth [S, 6] =
    (A1, beta1, T1, A2, beta2, T2) inside the reference's prior box (model.py:59-61)."""
    rng = np.random.default_rng(seed)
    lam = LAMBDA_UM[:nb] if nb <= 8 else np.geomspace(60.0, 1200.0, nb)
    z = rng.uniform(0.0, 0.3, S)
    nu = SPEED_OF_LIGHT / lam[None, :] * (1.0 + z[:, None])
    A1 = 10.0 ** rng.uniform(-0.5, 0.5, S)
    b1 = rng.uniform(1.0, 2.5, S)
    T1 = rng.uniform(15.0, 30.0, S)
    A2 = rng.uniform(0.05, 0.5, S) * A1
    b2 = rng.uniform(1.0, 2.5, S)
    T2 = rng.uniform(35.0, 60.0, S)
    truth = np.stack([A1, b1, T1, A2, b2, T2], axis=1)

    def grey(b, T):
        x = h_over_k * nu / T[:, None]
        xn = h_over_k * nu_b / T[:, None]
        return (nu / nu_b) ** (3 + b[:, None]) * np.expm1(xn) / np.expm1(x)

    m = A1[:, None] * grey(b1, T1) + A2[:, None] * grey(b2, T2)
    sig = frac_err * m
    d = m + sig * rng.standard_normal(m.shape)
    return nu, d, sig, truth


# production driver settings for the two-component run (submmSED/driver.py:301,329-332)
M2N_START = np.array([1.0, 2.0, 10.0, 0.1, 2.0, 20.0])
M2N_PROP_SIGMAS = np.array([2.0, 4.0, 4.0, 2.0, 4.0, 4.0])
