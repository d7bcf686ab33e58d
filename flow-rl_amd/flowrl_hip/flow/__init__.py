from .ddpm import DDPM, SUPPORTED_SOLVERS  # noqa: F401
from .continuous_ddpm import ContinuousDDPM  # noqa: F401
from .cnf import ContinuousNormalizingFlow  # noqa: F401
