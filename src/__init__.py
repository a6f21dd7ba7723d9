"""Import-path compatibility with the reference repo: `src.i3d`, `src.models.mgfn...`, `src.loss`,
`src.runner`, `src.dataset` resolve to the MI355X-native implementation in
`anomaly_detection_on_video_amd`, so `configs/*.yaml` `_target_` strings and user imports written
against jinmang2/anomaly_detection_on_video keep working."""
