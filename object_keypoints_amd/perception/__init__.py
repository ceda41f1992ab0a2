"""Host-side mirror of the reference's `perception` package for the keypoint-inference hot path.

Same class names, constructor/call signatures, tensor layouts and `state_dict` keys as
ethz-asl/object_keypoints (`perception.models`, `perception.pipeline`,
`perception.utils.camera_utils`), with the device work done by libokp_hip.so.
"""
