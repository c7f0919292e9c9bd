"""CMB band-power data containers (reference: MCMC/ClData/)."""
from . import ClData_CosmoMC  # noqa: F401
