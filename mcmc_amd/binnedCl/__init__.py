"""Binned C_l model and likelihood (reference: MCMC/binnedCl/)."""
from . import binnedClLikelihood, binnedClModel  # noqa: F401
