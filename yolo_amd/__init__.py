"""MI355X-native inference hot path of the YOLOv5-v2.0 fork: conv forward -> Detect decode -> NMS.

Layout: csrc/ holds the HIP kernels and the C-ABI (liby5hip.so); _lib.py loads it through ctypes;
engine.py lowers a live models.yolo.Model into a launch plan; nms.py drives the NMS kernels;
models/ and utils/ mirror the reference's Python interface; dist.py shards batches over GPUs.
"""
__version__ = '0.1.0'
