#!/usr/bin/env python3
"""Headline benchmark: 416-px tiles/sec of the Detect_OBB.py hot path on MI355X (BASELINE.json).

A "step" is one pass of the hot path over one batch (default 1024) of synthetic 416x416x3 tiles that are already resident in HBM
(config[1]: YOLOv11n-OBB 3-ch 416x416 tiled inference, single scale):
    uint8 tiles -> fused preprocess + YOLO11n-OBB forward (MFMA implicit-GEMM convs) -> decode -> ProbIoU Fast-NMS
    -> result construction -> border filter -> per-tile polygon-IoU merge -> [N>1: RCCL all-gather of survivor records]
    -> final whole-batch polygon-IoU merge (the fusion step of process_image).
Nothing is cached or skipped between steps; consecutive steps are software-pipelined on two HIP streams (the forward of step
k+1 overlaps the post-processing of step k; --no-pipeline disables it).  With --gpus N every rank processes its own batch (weak
scaling) and the survivors of all ranks are exchanged every step exactly as the multi-GPU tiler does.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (MFMA, for the conv kernel family: algorithmic FLOPs of
the forward / HIP-event time of the forward) and "cpu_baseline" (the CPU restatement of the same path, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FLOP_PER_TILE = 2 * 1392703312  # YOLO11n-OBB nc=12, 3x416x416
FLOP_PER_TILE_4CH = 2797866656  # same table, 4-channel input (BASELINE configs[3])
PEAK_TFLOPS = 2500.0            # dense fp16/bf16 MFMA, MI355X_MICROARCH.md chip table


def synthetic_rects(B, tile=416, step=316, cols=16):
    """tile rectangles of a virtual map scanned with the reference's stride (416 - 100)"""
    r = np.zeros((B, 4), np.int32)
    for t in range(B):
        x, y = (t % cols) * step, (t // cols) * step
        r[t] = (x, y, x + tile, y + tile)
    return r


def cpu_baseline(budget_s=15.0):
    """The CPU restatement of the same path (oracle/: torch-CPU fp32 forward, one tile per call like the reference,
    + oracle post-processing + C geometry), timed on this box's host cores on a bounded sample."""
    from oracle import pipeline as opl
    from oracle.yolo11_obb import Yolo11OBB
    torch.set_num_threads(min(os.cpu_count() or 1, 32))  # more threads only add synchronisation cost on these small convolutions
    net = Yolo11OBB("n", nc=12, ch=3, seed=0)
    om = opl.OracleModel(net, 416, "fp32")
    rng = np.random.default_rng(0)
    n, t0, dets = 0, time.time(), []
    while True:
        crop = rng.integers(0, 256, (416, 416, 3), dtype=np.uint8)
        img_dets = opl.detect_symbols(crop, om, 416, 100)  # a 416x416 "image" = exactly one full tile
        dets.extend(img_dets)
        n += 1
        if (time.time() - t0 > budget_s and n >= 8) or n >= 512:
            break
    from oracle import geom as og
    og.merge_detections(dets, 0.4)
    dt = time.time() - t0
    return {"value": n / dt, "unit": "tiles/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} synthetic 416x416x3 tiles, one model call per tile (reference loop), torch-CPU fp32 + C geometry, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("OBB_BENCH_BATCH", 1024)), help="tiles per GPU per step (SURVEY 8(d): B in {1, 16, 64, 256, 1024})")
    ap.add_argument("--no-pipeline", dest="pipeline", action="store_false", help="run the steps strictly one after the other (no forward / post-processing overlap)")
    ap.add_argument("--precision", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--channels", type=int, default=3, choices=[3, 4], help="4 = BASELINE configs[3]: every step also builds the RGB + DT-edge "
                    "input of a 4-channel checkpoint from the BGR tiles (build_multich) before the forward; not the headline configuration")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible and there is no CPU fallback")
    # one process per GPU; OBB_FORCE_DEVICE / OBB_DIST_BACKEND exist only to rehearse the N > 1 code path on a 1-GPU box (gloo)
    local = int(os.environ.get("OBB_FORCE_DEVICE", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    backend = os.environ.get("OBB_DIST_BACKEND", "nccl")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import make_weights
    import oriented_object_detection_amd  # noqa: F401
    from oriented_object_detection_amd import detect as D
    from oriented_object_detection_amd import dist as DD
    from oriented_object_detection_amd import ops
    from oriented_object_detection_amd.model import YOLO

    if rank == 0:
        wpath = make_weights.ensure("n", 12, args.channels, 0)
    if world > 1:
        dist.barrier()
    wpath = make_weights.path_for("n", 12, args.channels, 0)
    model = YOLO(wpath, imgsz=416, precision=args.precision)
    cfg = D.Config(tile_sizes=(416,), overlaps=(100,))
    B = args.batch
    tiles = torch.as_tensor(np.random.default_rng(rank).integers(0, 256, (B, 416, 416, 3), dtype=np.uint8)).to(dev)
    rects = synthetic_rects(B * world)
    rects_dev = torch.as_tensor(rects).to(dev)
    tile_ids = torch.arange(rank * B, (rank + 1) * B, dtype=torch.int32, device=dev)
    fwd_ev = []

    # Two HIP streams, software-pipelined over steps: the forward of step k+1 (whole chip, HBM/MFMA bound) is enqueued before the
    # host starts the post-processing of step k (per-tile NMS / merge kernels: latency bound, a few workgroups, host-visible counts).
    # Every step still performs all of its work inside the timed region; nothing is reused between steps (heads are double-buffered).
    s_fwd, s_post = torch.cuda.Stream(), torch.cuda.Stream()
    head_bufs = [torch.zeros((B, 3549, 80), dtype=torch.float32, device=dev) for _ in range(2)]  # stable addresses -> hipGraph replay
    # 4-channel mode: the input builder runs on the forward's stream in front of it (measured: on a stream of its own, under the previous
    # step's forward, it costs more than it hides -- 70.3 k vs 72.4 k tiles/s: the forward's two chains and the post-processing stream
    # already occupy the hardware queues)
    tiles4 = torch.zeros((B, 416, 416, 4), dtype=torch.uint8, device=dev) if args.channels == 4 else None
    md = cfg.max_det

    def launch_forward(k, timed):
        with torch.cuda.stream(s_fwd):
            model._ensure_active()
            e0, e1 = torch.cuda.Event(enable_timing=timed), torch.cuda.Event(enable_timing=timed)
            if tiles4 is not None:
                ops.build_multich(tiles, out=tiles4)  # inside the step, outside the forward's event pair
            e0.record()
            head = ops.forward(tiles if tiles4 is None else tiles4, out=head_bufs[k % 2])
            e1.record()
        if timed:
            fwd_ev.append((e0, e1))
        return head, e1

    def postprocess(head, ready):
        with torch.cuda.stream(s_post):
            s_post.wait_event(ready)
            det, cnt = ops.decode_nms(head, 416, 416, cfg.conf_predict, cfg.iou_nms, md)
            valid = (torch.arange(md, device=dev)[None, :] < cnt[:, None]).reshape(-1)
            rows = torch.nonzero(valid).squeeze(1)
            if rows.numel():
                d = det.reshape(-1, 7)[rows].contiguous()
                slot = (rows // md).long()
                _, pts = ops.results(d, None)
                rec = D._tile_records(pts, d[:, 5].int().contiguous(), d[:, 4].contiguous(), tile_ids[slot].contiguous(), rects_dev, cfg, 416)
            else:
                rec = D.TileRecords.empty(dev)
            if world > 1:
                rec = DD.all_gather_records(rec)
            ds = D.records_to_detset(rec, rects_dev, cfg, 416)
            merged, _ = D.merge_detections_device(ds, cfg.iou_threshold)
            done = torch.cuda.Event()
            done.record()
        return len(rec), len(merged), done

    def run_steps(n, timed):
        cur = torch.cuda.current_stream()
        s_fwd.wait_stream(cur)
        s_post.wait_stream(cur)
        res, pending, dones = (0, 0), None, [None, None]
        for k in range(n):
            if dones[k % 2] is not None:
                s_fwd.wait_event(dones[k % 2])  # head buffer k%2 is free once step k-2 has been post-processed
            f = launch_forward(k, timed)
            if pending is not None:
                r = postprocess(*pending[1])
                res, dones[pending[0] % 2] = r[:2], r[2]
                if not args.pipeline:
                    s_fwd.wait_event(r[2])
            pending = (k, f)
            if not args.pipeline:  # strictly sequential variant: finish step k before anything of step k+1 is enqueued
                r = postprocess(*pending[1])
                res, dones[k % 2] = r[:2], r[2]
                s_fwd.wait_event(r[2])
                pending = None
        if pending is not None:
            r = postprocess(*pending[1])
            res = r[:2]
        cur.wait_stream(s_post)
        cur.wait_stream(s_fwd)
        return res

    nrec, nmerged = run_steps(args.warmup, False) if args.warmup else (0, 0)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nrec, nmerged = run_steps(args.steps, True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    fwd_ms = float(np.mean([a.elapsed_time(b) for a, b in fwd_ev])) if fwd_ev else float("nan")

    def profiled_traffic():
        """HBM bytes of one forward from the committed rocprofv3 PMC passes of this build (FETCH_SIZE x2 + WRITE_SIZE, collected at 256
        tiles, see tools/profile_all.sh), scaled to this step's tiles; None if the profile is not there."""
        import re
        try:
            head = open(os.path.join(ROOT, "profiles", "r01_forward_hbm_traffic_b256.txt")).read(600)
            m = re.search(r"-> ([0-9.]+) MB / tile", head)
            return float(m.group(1)) * 1e6 * B if m else None
        except OSError:
            return None

    if rank == 0:
        tiles_per_s = world * B * args.steps / dt
        achieved = B * (FLOP_PER_TILE if args.channels == 3 else FLOP_PER_TILE_4CH) / (fwd_ms * 1e-3) / 1e12
        out = {
            "metric": "416px tiles/sec (whole node), YOLOv11n-OBB %dch" % args.channels,
            "value": tiles_per_s, "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if args.precision == "f16" else "bf16", "data": "synthetic",
            "config": {"workload": ("YOLOv11n-OBB 3ch 416x416 tiled inference, single-scale (BASELINE configs[1]): forward + decode + "
                                    "ProbIoU NMS + border filter + per-tile merge + final polygon-IoU merge") if args.channels == 3 else
                                   ("YOLOv11n-OBB 4ch (RGB + DT-edge) 416x416 tiled inference, single-scale (BASELINE configs[3] per GPU): build_multich + "
                                    "forward + decode + ProbIoU NMS + border filter + per-tile merge + final polygon-IoU merge"),
                       "tiles_per_gpu_per_step": B, "step_pipelining": bool(args.pipeline), "nc": 12, "weights": "synthetic seeded, cls bias -8 (no checkpoint offline)",
                       "survivor_records_per_step": nrec, "final_detections": nmerged},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_TFLOPS,
                         "traffic": profiled_traffic(), "traffic_source": "profiles/r01_forward_hbm_traffic_b256.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                         "passes of this build, bytes per tile x tiles per step)", "kernel": "k_conv_igemm family (whole forward)", "forward_ms": fwd_ms},
        }
        if args.channels == 4:
            out["roofline"]["traffic"] = None  # the committed PMC passes are of the 3-channel forward
        if not args.no_cpu_baseline and world == 1 and args.channels == 3:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
