"""TEST INFRASTRUCTURE ONLY.

CPU oracle for the Detect_OBB.py hot path: a plain-C restatement of the geometry half
(`obb_oracle.c`), a PyTorch-CPU fp32 restatement of the YOLO11-OBB forward / decode / ProbIoU
fast-NMS (`yolo11_obb.py`, `postproc.py`) and a loop-for-loop pure-Python restatement of the
script's control flow (`pyref.py`).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product package (oriented-object-detection_amd/) never does.
"""
