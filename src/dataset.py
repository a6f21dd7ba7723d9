from anomaly_detection_on_video_amd.dataset import FeatureDataset, build_feature_dataset, write_synthetic_feature_zips  # noqa: F401
