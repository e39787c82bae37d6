"""stablekeypoints_amd -- MI355X-native implementation of the StableKeypoints token-optimisation
hot path (hooked SD-UNet cross-attention maps -> [T,R,R] reduction -> sharpening/equivariance
losses -> gradient of the learned text embedding), behind the reference's own API names.

Sub-modules mirror the reference's `unsupervised_keypoints` package for this path only:
`ptp_utils`, `optimize`, `optimize_token`, `invertable_transform`, `eval`.
"""
__version__ = "0.1.0"
