from anomaly_detection_on_video_amd.loss.mgfn import MGFNLoss  # noqa: F401
