from .base import ContrastiveLoss, SparsityLoss, TemporalSmoothnessLoss  # noqa: F401
from .mgfn import MGFNLoss  # noqa: F401
