"""mcmc_amd: MI355X-native Metropolis-Hastings likelihood hot path with the class API of
defjaf/MCMC (Data / Model / Likelihood / Proposal / MCMC).  All arithmetic runs in
libmi355mcmc.so (hand-written HIP for gfx950); there is no CPU fallback."""
from ._lib import LIB_PATH, Mi355Error  # noqa: F401
from .engine import Engine, MODEL_IDS  # noqa: F401

__version__ = "0.1.0"
