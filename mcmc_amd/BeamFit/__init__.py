"""2-D Gaussian beam fitting (reference: MCMC/BeamFit/)."""
from . import BeamData, BeamModel, NormalizedBeamLikelihood, OffsetNormalizedBeamModel  # noqa: F401
