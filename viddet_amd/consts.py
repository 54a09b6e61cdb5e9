"""Constants of the yolo3_darknet53 definition shared by the network object and the (NumPy-only) host data pipeline.

models/definitions/yolo/wrappers.py:80-84 (anchors, strides); models/definitions/layers.py:68-69 (BatchNorm / LeakyReLU)."""
ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]
STRIDES = [8, 16, 32]
BN_EPS, BN_MOMENTUM, LEAKY_SLOPE = 1e-5, 0.9, 0.1
