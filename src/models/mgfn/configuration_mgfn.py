from anomaly_detection_on_video_amd.models.mgfn.configuration_mgfn import MGFNConfig  # noqa: F401
