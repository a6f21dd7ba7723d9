from anomaly_detection_on_video_amd.models.mgfn.modeling_mgfn import *  # noqa: F401,F403
from anomaly_detection_on_video_amd.models.mgfn.modeling_mgfn import MGFNForVideoAnomalyDetection, MGFNModel  # noqa: F401
