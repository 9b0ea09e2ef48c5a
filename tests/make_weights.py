#!/usr/bin/env python3
"""Writes the synthetic YOLO11-OBB checkpoints ("OBBW" blobs) used by bench.py and the smoke test.

No trained weights exist offline (the reference's best*.pt are Google-Drive links), so the benchmark
runs the published architecture with seeded, variance-calibrated random weights.  The generator lives with the oracle
model definition (test infrastructure); the blob it writes is plain data consumed by the product through
obb_model_load -- the product never imports the oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


# Class-logit bias of the synthetic head.  -8.0 gives about a dozen candidates per 416-px noise tile at conf 0.25 (a few survive
# NMS), i.e. the detection density of the reference's own sample outputs (Output/Test1.xlsx: 34 detections over 9 + 90 tiles);
# the unit tests use the oracle's denser default (-4.2) to stress the post-processing kernels.
BENCH_CLS_BIAS = -8.0


def path_for(scale="n", nc=12, ch=3, seed=0, cls_bias=BENCH_CLS_BIAS):
    return os.path.join(ROOT, "weights", f"yolo11{scale}-obb_nc{nc}_ch{ch}_seed{seed}_clsb{abs(cls_bias):g}.obbw")


def ensure(scale="n", nc=12, ch=3, seed=0, cls_bias=BENCH_CLS_BIAS):
    p = path_for(scale, nc, ch, seed, cls_bias)
    if not os.path.exists(p):
        from oracle.yolo11_obb import Yolo11OBB
        os.makedirs(os.path.dirname(p), exist_ok=True)
        blob = Yolo11OBB(scale, nc=nc, ch=ch, seed=seed, cls_bias=cls_bias).to_blob()
        with open(p + ".tmp", "wb") as f:
            f.write(blob)
        os.replace(p + ".tmp", p)
    return p


if __name__ == "__main__":
    for args in (("n", 12, 3, 0),):
        print(ensure(*args))
