from anomaly_detection_on_video_amd.loss.base import ContrastiveLoss, SparsityLoss, TemporalSmoothnessLoss  # noqa: F401
