from .configuration_mgfn import MGFNConfig  # noqa: F401
from .modeling_mgfn import (  # noqa: F401
    MGFNForVideoAnomalyDetection,
    MGFNModel,
    MGFNModelOutput,
    MGFNVideoAnomalyDetectionOutput,
    mgfn_param_shapes,
)
