"""Sub-mm SED fitting: models, likelihoods and data containers (reference: MCMC/submmSED/)."""
from . import data, likelihood, model  # noqa: F401
