from anomaly_detection_on_video_amd.loss import ContrastiveLoss, MGFNLoss, SparsityLoss, TemporalSmoothnessLoss  # noqa: F401
