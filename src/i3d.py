from anomaly_detection_on_video_amd.i3d import *  # noqa: F401,F403
from anomaly_detection_on_video_amd.i3d import Bottleneck, I3Res50, build_i3d_feature_extractor, print_model_size  # noqa: F401
