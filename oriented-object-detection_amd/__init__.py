"""oriented-object-detection_amd -- MI355X (gfx950) drop-in for the Detect_OBB.py sliding-window OBB inference path.

Host side mirrors the reference's call surface (same names / argument meaning / error behaviour):
    compute_polygon_iou, merge_detections, cross_scale_consensus_filter, detect_symbols, process_image, YOLO
and routes everything through libobbhip.so (hand-written HIP kernels behind the C-ABI in include/obbhip.h).
"""
__version__ = "0.1.0"
