from anomaly_detection_on_video_amd.models.mgfn import MGFNConfig, MGFNForVideoAnomalyDetection, MGFNModel  # noqa: F401
