"""Post-processing rooflines (SURVEY.md section 8(d) "Post-proc benches"): time per call (HIP events on the launch stream), algorithmic
bytes per SURVEY 8(d), achieved GB/s and the fraction of the 8 TB/s HBM peak, for
    decode + ProbIoU Fast-NMS (obb_decode_nms, 1024 tiles of 416 px, bench weights), conf 0.25 and 0.001
    k_iou_pairs (obb_poly_iou_pairs)           136 B per pair
    obb_merge_detections at N in {1024, 4096, 16384, 65536}   N * 72 B in + one 8-B edge per suppressing pair (+ N B keep)
    obb_sort_desc_stable at the same N          N * 12 B
    obb_consensus at the dual-scale sizes       (N1 + N2) * 72 B
Writes a text table (stdout): python tools/postproc_bench.py > profiles/r02_postproc.txt"""
import sys

import numpy as np
import torch

sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import make_weights
import synth
import oriented_object_detection_amd  # noqa: F401
from oriented_object_detection_amd import ops
from oriented_object_detection_amd.model import YOLO

PEAK = 8000.0  # GB/s, MI355X_MICROARCH.md chip table


def t_us(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def row(name, us, nbytes, extra=""):
    gbs = nbytes / (us * 1e-6) / 1e9
    print(f"{name:58s} {us:10.1f} us  {nbytes / 1e6:10.2f} MB  {gbs:8.1f} GB/s  {gbs / PEAK * 100:6.2f} % of 8 TB/s  {extra}")


print("# post-processing kernels, MI355X, HIP-event time per call; bytes = ALGORITHMIC bytes of SURVEY 8(d) (not PMC traffic)")
m = YOLO(make_weights.ensure("n", 12, 3, 0), imgsz=416)
B, A = 1024, 3549
tiles = torch.as_tensor(np.random.default_rng(0).integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)).cuda()
cmax = torch.empty((B, A), dtype=torch.float32, device="cuda")
head = ops.forward(tiles, cmax=cmax)  # cmax: the largest class logit per anchor, written by the head's fused class tails (obb_forward_gate)
torch.cuda.synchronize()
pred = ops.decode(head[:64].contiguous(), 416, 416)
conf = pred[..., 4:16].amax(-1)
print("# candidates per tile at conf 0.25: mean %.1f max %d" % (float((conf > 0.25).sum(1).float().mean()), int((conf > 0.25).sum(1).max())))
for c in (0.25, 0.001):
    us = t_us(lambda: ops.decode_nms(head, 416, 416, c, 0.7, 300), reps=10)
    det, cnt = ops.decode_nms(head, 416, 416, c, 0.7, 300)
    ncand = float(cnt.float().mean())
    # decode reads 77 logits and writes 17 values per anchor; NMS reads 28 B per candidate; the candidate-first kernel reads
    # the class logits of every anchor (48 B) and whole rows of the survivors only
    row(f"obb_decode_nms  1024 tiles x 3549 anchors  conf {c}", us, B * A * (77 + 17) * 4, f"kept/tile {ncand:.1f}; bytes the candidate-first kernel needs: {B * A * 48 / 1e6:.0f} MB")
    us = t_us(lambda: ops.decode_nms(head, 416, 416, c, 0.7, 300, cmax=cmax), reps=10)
    row(f"obb_decode_nms_gate (dense class-maximum gate)  conf {c}", us, B * A * (77 + 17) * 4, f"bytes this form needs: {B * A * 4 / 1e6:.0f} MB of maxima + the rows of the candidates")
us = t_us(lambda: ops.decode(head, 416, 416), reps=10)
row("obb_decode (full decode, parity tap)  1024 tiles", us, B * A * (77 + 17) * 4)
del head, tiles, pred, cmax
m.close()

npair = 1 << 22
b, _, _, _ = synth.make_dets(2, 1 << 16, extent=1000.0)
Aa = torch.tensor(b).cuda().repeat(npair // len(b), 1)
Bb = Aa.roll(1, 0).contiguous()
us = t_us(lambda: ops.poly_iou_pairs(Aa, Bb), reps=10)
out = ops.poly_iou_pairs(Aa, Bb)
row(f"obb_poly_iou_pairs  {npair} pairs", us, npair * 136, "overlapping pairs %.1f %%" % (100 * float((out > 0).double().mean())))
b2, _, _, _ = synth.make_dets(3, 1 << 16, extent=60000.0)
A2 = torch.tensor(b2).cuda().repeat(npair // len(b2), 1)
B2 = A2.roll(1, 0).contiguous()
us = t_us(lambda: ops.poly_iou_pairs(A2, B2), reps=10)
row(f"obb_poly_iou_pairs  {npair} pairs, disjoint envelopes", us, npair * 136, "(envelope reject only: the streaming rate)")
del Aa, Bb, A2, B2

for n in (1024, 4096, 16384, 65536):
    bb, cc, ss, _ = synth.make_dets(1, n, extent=4096.0)
    Bt, Ct, St = torch.tensor(bb).cuda(), torch.tensor(cc).cuda(), torch.tensor(ss).cuda()
    us = t_us(lambda: ops.merge_detections(Bt, Ct, St, 0.4), reps=10)
    o, k, nk = ops.merge_detections(Bt, Ct, St, 0.4)
    row(f"obb_merge_detections  N = {n} (4096^2 px, 35 % near-duplicates)", us, n * 72 + n, "kept %d" % int(nk.item()))
    us = t_us(lambda: ops.sort_desc_stable(St), reps=10)
    row(f"obb_sort_desc_stable  N = {n}", us, n * 12)

# survivors of the dual-scale bench step: ~51 k at the 128-px scale, ~12 k at the 416-px scale, ~12 per 416-px tile area
for n1, n2, ext in ((51000, 12000, 18000.0), (12000, 3000, 9000.0), (3000, 800, 4500.0)):
    rng = np.random.default_rng(5)
    base_b, base_c, _, base_x = synth.make_dets(7, n1, extent=ext, dup_frac=0.1)
    pick = rng.permutation(n1)[:n2]
    x2 = base_x[pick].copy()
    x2[:, :2] += rng.normal(0, 4.0, (n2, 2)).astype(np.float32)
    loc = synth.xywhr_to_corners_f32(x2).astype(np.float64)
    off = base_b[pick][:, :2] - synth.xywhr_to_corners_f32(base_x[pick]).astype(np.float64)[:, :2]
    loc[:, 0::2] += np.round(off[:, 0:1]); loc[:, 1::2] += np.round(off[:, 1:2])
    allb = torch.tensor(np.concatenate([base_b, loc])).cuda()
    allc = torch.tensor(np.concatenate([base_c, base_c[pick]]).astype(np.int32)).cuda()
    alls = torch.tensor(rng.uniform(0.25, 1.0, n1 + n2).astype(np.float32).astype(np.float64)).cuda()
    us = t_us(lambda: ops.consensus(allb, allc, alls, [0, n1, n1 + n2]), reps=5)
    idx, nout = ops.consensus(allb, allc, alls, [0, n1, n1 + n2])
    row(f"obb_consensus  N1 = {n1}, N2 = {n2}", us, (n1 + n2) * 72, "kept %d" % int(nout.item()))
