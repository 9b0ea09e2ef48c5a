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


def path_for(scale="n", nc=12, ch=3, seed=0):
    return os.path.join(ROOT, "weights", f"yolo11{scale}-obb_nc{nc}_ch{ch}_seed{seed}.obbw")


def ensure(scale="n", nc=12, ch=3, seed=0):
    p = path_for(scale, nc, ch, seed)
    if not os.path.exists(p):
        from oracle.yolo11_obb import Yolo11OBB
        os.makedirs(os.path.dirname(p), exist_ok=True)
        blob = Yolo11OBB(scale, nc=nc, ch=ch, seed=seed).to_blob()
        with open(p + ".tmp", "wb") as f:
            f.write(blob)
        os.replace(p + ".tmp", p)
    return p


if __name__ == "__main__":
    for args in (("n", 12, 3, 0), ("n", 12, 3, 1), ("n", 12, 4, 0)):
        print(ensure(*args))
