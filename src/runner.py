from anomaly_detection_on_video_amd.runner import *  # noqa: F401,F403
from anomaly_detection_on_video_amd.runner import Trainer, VideoAnomalyDetectionRunner  # noqa: F401
