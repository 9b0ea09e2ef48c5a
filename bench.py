#!/usr/bin/env python3
"""Headline benchmark: 416-px tiles/sec of the Detect_OBB.py hot path on MI355X (BASELINE.json).

A "step" is one pass of the hot path over one batch (default 1024) of synthetic 416x416x3 tiles that are already resident in HBM
(configs[1]: YOLOv11n-OBB 3-ch 416x416 tiled inference, single scale):
    uint8 tiles -> fused preprocess + YOLO11n-OBB forward (MFMA implicit-GEMM convs) -> decode -> ProbIoU Fast-NMS
    -> result construction -> border filter -> per-tile polygon-IoU merge -> [N>1: RCCL all-gather of survivor records]
    -> final whole-batch polygon-IoU merge (the fusion step of process_image).
Nothing is cached or skipped between steps; consecutive steps are software-pipelined on two HIP streams (the forward of step
k+1 overlaps the post-processing of step k; --no-pipeline disables it).  With --gpus N every rank processes its own batch (weak
scaling) and the survivors of all ranks are exchanged every step exactly as the multi-GPU tiler does.

`python bench.py --gpus N` launches its own N ranks (torch.distributed.run, one process per GPU) when it is not already running under
a launcher; under `python -m torch.distributed.run ... bench.py --gpus N` it uses the ranks it is given.

Prints ONE JSON line (rank 0).  Objects beside the contract fields:
  "roofline"      MFMA roofline of the forward (algorithmic FLOPs / HIP-event time) against the peak of the arithmetic the line ran in
  "also"          the other BASELINE configs timed the same way in the headline's arithmetic: "dual_scale" (configs[2]) and "ch4"
                  (configs[3], per GPU); and "f16": the opt-in 16-bit fast mode (fp16 storage, fp32 accumulate) on the same three
                  workloads, each with its own roofline against the 2.5 PFLOP/s fp16 MFMA peak
The headline runs in fp32 arithmetic (obb_set_option precision 32: exact-f32 MFMA, peak 157.3 TFLOP/s) because that is what the
reference computes (Detect_OBB.py:79-83, half=False); --precision f16 / bf16 makes the 16-bit mode the headline instead.
  "cpu_baseline"  the CPU restatement of the same path (bounded sample; baseline only)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FLOP_PER_TILE = 2 * 1392703312  # YOLO11n-OBB nc=12, 3x416x416
FLOP_PER_TILE_4CH = 2797866656  # same table, 4-channel input (BASELINE configs[3])
FLOP_PER_TILE_128 = 262767104   # 3x128x128
PEAK_TFLOPS = {"f16": 2500.0, "bf16": 2500.0, "f32": 157.3}  # dense MFMA peaks, MI355X_MICROARCH.md chip table


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("OBB_BENCH_BATCH", 1024)), help="tiles per GPU per step (SURVEY 8(d): B in {1, 16, 64, 256, 1024})")
    ap.add_argument("--no-pipeline", dest="pipeline", action="store_false", help="run the steps strictly one after the other (no forward / post-processing overlap)")
    ap.add_argument("--precision", default="f32", choices=["f16", "bf16", "f32"], help="arithmetic of the HEADLINE line (default fp32 = the reference's; the f16 figures are reported under also.f16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the dual-scale / 4-channel / f16 measurements beside the headline")
    ap.add_argument("--channels", type=int, default=3, choices=[3, 4], help="4 = BASELINE configs[3] as the headline: every step also builds the RGB + DT-edge "
                    "input of a 4-channel checkpoint from the BGR tiles (build_multich) before the forward")
    return ap.parse_args()


def synthetic_rects(B, tile=416, step=316, cols=16):
    """tile rectangles of a virtual map scanned with the reference's stride (tile - overlap)"""
    import numpy as np
    r = np.zeros((B, 4), np.int32)
    for t in range(B):
        x, y = (t % cols) * step, (t // cols) * step
        r[t] = (x, y, x + tile, y + tile)
    return r


def cpu_baseline(budget_s=15.0):
    """The CPU restatement of the same path (oracle/: torch-CPU fp32 forward, one tile per call like the reference,
    + oracle post-processing + C geometry), timed on this box's host cores on a bounded sample."""
    import numpy as np
    import torch
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
    return {"value": n / dt, "unit": "tiles/s", "cores": torch.get_num_threads(), "host_logical_cpus": os.cpu_count(), "kind": "port",
            "sample": f"{n} synthetic 416x416x3 tiles, one model call per tile (reference loop), torch-CPU fp32 + C geometry, {dt:.1f} s"}


def spawn_ranks(n):
    """`python bench.py --gpus N` outside a launcher: start the N ranks ourselves.  The parent never touches the GPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to record a scaling point with a wrong n_gpus")

    import numpy as np
    import torch
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

    def weights(ch, seed):
        if rank == 0:
            make_weights.ensure("n", 12, ch, seed)
        if world > 1:
            dist.barrier()
        return make_weights.path_for("n", 12, ch, seed)

    B = args.batch
    s_fwd, s_post, s_pre = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()

    class Scale:
        """one tile size of a workload: the model, its resident synthetic tiles, double-buffered heads, tile rectangles"""

        def __init__(self, model, px, n, seed, rects, tile_ids, channels=3):
            self.model, self.px, self.n = model, px, n
            self.tiles = torch.as_tensor(np.random.default_rng(seed).integers(0, 256, (n, px, px, 3), dtype=np.uint8)).to(dev)
            # 4-channel mode: the input builder of step k + 1 runs on a stream of its own under the forward of step k (double-buffered
            # 4-channel tiles); it is part of the step, outside the forward's event pair
            self.tiles4 = [torch.zeros((n, px, px, 4), dtype=torch.uint8, device=dev) for _ in range(2)] if channels == 4 else None
            self.ev_pre, self.ev_used = [None, None], [None, None]
            A = ops.model_info(px, px)["anchors"]
            self.heads = [torch.zeros((n, A, 80), dtype=torch.float32, device=dev) for _ in range(2)]  # stable addresses -> hipGraph replay
            self.cmaxs = [torch.zeros((n, A), dtype=torch.float32, device=dev) for _ in range(2)]     # largest class logit per anchor (obb_forward_gate)
            self.gate = model.precision in ("f32", "fp32")  # (the fp32 plan writes the maxima in its fused class tails; 16-bit plans would need an extra pass)
            self.rects_dev = torch.as_tensor(rects).to(dev)
            self.tile_ids = tile_ids
            self.cfg = D.Config(tile_sizes=(px,), overlaps=(100 if px > 128 else 30,))
            self.ev = []

        def prepare(self, k):  # on s_pre: build_multich of step k (no-op for 3-channel models)
            if self.tiles4 is None or self.ev_pre[k % 2] is not None:
                return
            with torch.cuda.stream(s_pre):
                if self.ev_used[k % 2] is not None:
                    s_pre.wait_event(self.ev_used[k % 2])  # the forward of step k - 2 has consumed this buffer
                ops.build_multich(self.tiles, out=self.tiles4[k % 2])
                self.ev_pre[k % 2] = torch.cuda.Event()
                self.ev_pre[k % 2].record()

        def forward(self, k, timed):  # on s_fwd
            self.model._ensure_active()
            e0, e1 = torch.cuda.Event(enable_timing=timed), torch.cuda.Event(enable_timing=timed)
            src = self.tiles
            if self.tiles4 is not None:
                self.prepare(k)
                s_fwd.wait_event(self.ev_pre[k % 2])
                self.ev_pre[k % 2] = None
                src = self.tiles4[k % 2]
            e0.record()
            head = ops.forward(src, out=self.heads[k % 2], cmax=self.cmaxs[k % 2] if self.gate else None)
            e1.record()
            if self.tiles4 is not None:
                self.ev_used[k % 2] = e1
                self.prepare(k + 1)  # overlaps this forward
            if timed:
                self.ev.append((e0, e1))
            return head

        def records(self, head):  # on s_post: decode -> Fast-NMS -> results -> border filter -> per-tile merge -> exchange records
            self.model._ensure_active()
            md = self.cfg.max_det
            cmax = None if not self.gate else (self.cmaxs[0] if head.data_ptr() == self.heads[0].data_ptr() else self.cmaxs[1])
            det, cnt = ops.decode_nms(head, self.px, self.px, self.cfg.conf_predict, self.cfg.iou_nms, md, zero=False, cmax=cmax)
            margin = self.cfg.margin_for(self.px) if self.cfg.APPLY_BORDER_FILTER else 0
            rec, _, n = ops.tile_survivors(det, cnt, None, self.tile_ids, self.rects_dev, margin, self.cfg.iou_threshold, self.cfg.strike_cls)
            return D.TileRecords.from_packed(rec, n)  # packed rows + their count, both still on the device

    def measure(scales, steps, warmup, pipeline=True):
        """-> dict(dt, nrec, nmerged, fwd_ms per scale).  Two HIP streams, software-pipelined over steps: the forwards of step k+1 (whole
        chip) are enqueued before the host starts the post-processing of step k (per-tile NMS / merge kernels: latency-bound, host-visible
        counts).  Every step performs all of its work inside the timed region; nothing is reused between steps (heads double-buffered)."""
        for sc in scales:
            sc.ev = []

        def launch_forward(k, timed):
            with torch.cuda.stream(s_fwd):
                heads = [sc.forward(k, timed) for sc in scales]
                ready = torch.cuda.Event()
                ready.record()
            return heads, ready

        def postprocess(heads, ready):
            with torch.cuda.stream(s_post):
                s_post.wait_event(ready)
                sets, nrec = {}, 0
                for sc, head in zip(scales, heads):
                    rec = sc.records(head)
                    if world > 1:
                        rec = DD.all_gather_records(rec)
                    nrec += len(rec)  # the one host read of this scale's step: the survivor count sizes the fusion launches
                    sets[sc.px] = D.records_to_detset(rec, sc.rects_dev, sc.cfg, sc.px)
                fused = D.cross_scale_consensus_filter_device(sets) if len(scales) > 1 else sets[scales[0].px]  # Detect_OBB.py:290
                merged, _ = D.merge_detections_device(fused, scales[0].cfg.iou_threshold)                    # :291 (kept rows + device count)
                done = torch.cuda.Event()
                done.record()
            return nrec, merged, done

        def run_steps(n, timed):
            cur = torch.cuda.current_stream()
            s_fwd.wait_stream(cur)
            s_post.wait_stream(cur)
            s_pre.wait_stream(cur)
            res, pending, dones = (0, 0), None, [None, None]
            for k in range(n):
                if dones[k % 2] is not None:
                    s_fwd.wait_event(dones[k % 2])  # head buffer k%2 is free once step k-2 has been post-processed
                f = launch_forward(k, timed)
                if pending is not None:
                    r = postprocess(*pending[1])
                    res, dones[pending[0] % 2] = r[:2], r[2]
                pending = (k, f)
                if not pipeline:  # strictly sequential variant: finish step k before anything of step k+1 is enqueued
                    r = postprocess(*pending[1])
                    res, dones[k % 2] = r[:2], r[2]
                    s_fwd.wait_event(r[2])
                    pending = None
            if pending is not None:
                r = postprocess(*pending[1])
                res = r[:2]
            cur.wait_stream(s_post)
            cur.wait_stream(s_fwd)
            cur.wait_stream(s_pre)
            for sc in scales:  # (a prepared but unused buffer of the step after the last one: drop it, the next run starts clean)
                sc.ev_pre, sc.ev_used = [None, None], [None, None]
            return res

        if warmup:
            run_steps(warmup, False)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nrec, merged = run_steps(steps, True)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        nmerged = len(merged)  # the final count is read once, outside the timed region
        return {"dt": dt, "nrec": nrec, "nmerged": nmerged, "fwd_ms": [float(np.mean([a.elapsed_time(b) for a, b in sc.ev])) for sc in scales]}

    def roofline(precision, flop_per_step, fwd_ms):
        ach = flop_per_step / (fwd_ms * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": ach, "peak": PEAK_TFLOPS[precision], "unit": "TFLOP/s", "frac": ach / PEAK_TFLOPS[precision], "forward_ms": fwd_ms}

    def profiled_traffic(precision, nt):
        """HBM bytes of one forward from the committed rocprofv3 PMC passes of this build (FETCH_SIZE x2 + WRITE_SIZE, collected at 256
        tiles, see tools/profile_all.sh), scaled to `nt` tiles; (None, None) if no profile of that arithmetic is there."""
        import re
        stem = "forward_f32_hbm_traffic_b256.txt" if precision == "f32" else "forward_hbm_traffic_b256.txt"
        for rnd in ("r04_", "r03_", "r02_", "r01_"):
            try:
                m = re.search(r"-> ([0-9.]+) MB / tile", open(os.path.join(ROOT, "profiles", rnd + stem)).read(600))
                if m:
                    return float(m.group(1)) * 1e6 * nt, "profiles/" + rnd + stem
            except OSError:
                pass
        return None, None

    def single_scale(precision, channels):
        model = YOLO(weights(channels, 0), imgsz=416, precision=precision)
        rects = synthetic_rects(B * world)
        tile_ids = torch.arange(rank * B, (rank + 1) * B, dtype=torch.int32, device=dev)
        return model, [Scale(model, 416, B, rank, rects, tile_ids, channels)]

    KERNEL = {"f32": "k_conv_f32 family (whole forward, v_mfma_f32_16x16x4_f32)", "f16": "k_conv_igemm family (whole forward, v_mfma_f32_16x16x32_f16)",
              "bf16": "k_conv_igemm family (whole forward, v_mfma_f32_16x16x32_bf16)"}

    def run_single(precision, channels, steps, warmup):
        """configs[1] (3 channels) / configs[3] per GPU (4 channels) in the given arithmetic -> (throughput dict, roofline dict)"""
        model, scales = single_scale(precision, channels)
        h = measure(scales, steps, warmup, args.pipeline)
        flop = FLOP_PER_TILE if channels == 3 else FLOP_PER_TILE_4CH
        rf = roofline(precision, B * flop, h["fwd_ms"][0])
        traffic, src = profiled_traffic(precision, B) if channels == 3 else (None, None)
        rf.update({"traffic": traffic, "traffic_source": (src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, bytes per tile x tiles per step)") if src else None,
                   "kernel": KERNEL[precision]})
        del scales
        model.close()
        return h, rf

    def run_dual(precision, steps):
        """BASELINE configs[2]: dual-scale (128 + 416) late fusion.  The virtual map of the B 416-px tiles (16 columns, stride 316) is scanned
        at the 128-px scale as well (stride 98): both forwards, both per-tile paths, cross_scale_consensus_filter and the final merge inside
        every step (Detect_OBB.py:277-291)"""
        cols, rows_ = 16, (B + 15) // 16
        Wm, Hm = (cols - 1) * 316 + 416, (rows_ - 1) * 316 + 416
        c128, r128 = len(range(0, Wm - 127, 98)), len(range(0, Hm - 127, 98))  # full 128-px tiles only
        n128 = c128 * r128
        rects128 = np.zeros((n128, 4), np.int32)
        for t in range(n128):
            x, y = (t % c128) * 98, (t // c128) * 98
            rects128[t] = (x, y, x + 128, y + 128)
        m128 = YOLO(weights(3, 1), imgsz=128, precision=precision)
        m416 = YOLO(weights(3, 0), imgsz=416, precision=precision)
        sc = [Scale(m128, 128, n128, 7, rects128, torch.arange(n128, dtype=torch.int32, device=dev)),
              Scale(m416, 416, B, 0, synthetic_rects(B), torch.arange(B, dtype=torch.int32, device=dev))]
        r = measure(sc, steps, 2, args.pipeline)
        flopd = n128 * FLOP_PER_TILE_128 + B * FLOP_PER_TILE
        rf = roofline(precision, flopd, r["fwd_ms"][0] + r["fwd_ms"][1])
        rf["by_scale"] = {"128": roofline(precision, n128 * FLOP_PER_TILE_128, r["fwd_ms"][0]), "416": roofline(precision, B * FLOP_PER_TILE, r["fwd_ms"][1])}
        d = {"workload": "BASELINE configs[2]: dual-scale (128 + 416) 3ch late fusion: per step %d 128-px tiles + %d 416-px tiles of one %dx%d virtual map -> "
                         "both forwards + decode + NMS + per-tile merges + cross_scale_consensus_filter + final merge" % (n128, B, Wm, Hm),
             "value": B * steps / r["dt"], "unit": "416px-tile map areas/s (each with its %.1f 128-px tiles)" % (n128 / B), "dtype": precision,
             "tiles_per_s_both_scales": (B + n128) * steps / r["dt"], "steps": steps, "ms_per_step": r["dt"] / steps * 1e3,
             "forward_ms": {"128": r["fwd_ms"][0], "416": r["fwd_ms"][1]}, "roofline": rf,
             "survivor_records_per_step": r["nrec"], "final_detections": r["nmerged"]}
        del sc
        m128.close()
        m416.close()
        return d

    def run_ch4(precision, steps):
        h4, rf4 = run_single(precision, 4, steps, 2)
        return {"workload": "BASELINE configs[3] per GPU: build_multich (DT-edge channel) + 4-channel forward + the same post-processing",
                "value": B * steps / h4["dt"], "unit": "tiles/s", "dtype": precision, "steps": steps, "ms_per_step": h4["dt"] / steps * 1e3, "roofline": rf4}

    # ------------------------------------------------------------------ headline: configs[1] (or configs[3] with --channels 4)
    h, rf = run_single(args.precision, args.channels, args.steps, args.warmup)
    out = None
    if rank == 0:
        out = {
            "metric": "416px tiles/sec (whole node), YOLOv11n-OBB %dch" % args.channels,
            "value": world * B * args.steps / h["dt"], "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": h["dt"] / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": ("YOLOv11n-OBB 3ch 416x416 tiled inference, single-scale (BASELINE configs[1]): forward + decode + "
                                    "ProbIoU NMS + border filter + per-tile merge + final polygon-IoU merge") if args.channels == 3 else
                                   ("YOLOv11n-OBB 4ch (RGB + DT-edge) 416x416 tiled inference, single-scale (BASELINE configs[3] per GPU): build_multich + "
                                    "forward + decode + ProbIoU NMS + border filter + per-tile merge + final polygon-IoU merge"),
                       "tiles_per_gpu_per_step": B, "step_pipelining": bool(args.pipeline), "nc": 12, "weights": "synthetic seeded, cls bias -8 (no checkpoint offline)",
                       "arithmetic": {"f32": "fp32 weights / activations / accumulation end to end (what Detect_OBB.py computes with half=False)",
                                      "f16": "fp16 storage, fp32 accumulate (opt-in fast mode)", "bf16": "bf16 storage, fp32 accumulate (opt-in fast mode)"}[args.precision],
                       "survivor_records_per_step": h["nrec"], "final_detections": h["nmerged"]},
            "roofline": rf,
        }

    extras = not args.no_extras and world == 1
    if extras:
        stepsd = max(3, args.steps // 2)
        also = {"dual_scale": run_dual(args.precision, stepsd)}
        if args.channels == 3:
            also["ch4"] = run_ch4(args.precision, stepsd)
        if args.precision == "f32":
            # ------------------------------------------------------------ the opt-in 16-bit fast mode on the same workloads (never the headline)
            h16, rf16 = run_single("f16", args.channels, args.steps, 2)
            f16 = {"value": B * args.steps / h16["dt"], "unit": "tiles/s", "dtype": "f16", "steps": args.steps, "warmup": 2, "ms_per_step": h16["dt"] / args.steps * 1e3,
                   "roofline": rf16, "survivor_records_per_step": h16["nrec"], "final_detections": h16["nmerged"],
                   "note": "fp16 storage / fp32 accumulate (YOLO(..., precision='f16')): detections agree with the fp32 pipeline within the tolerances of "
                           "tests/test_gpu_fp32.py, not bit for bit",
                   "dual_scale": run_dual("f16", stepsd)}
            if args.channels == 3:
                f16["ch4"] = run_ch4("f16", stepsd)
            also["f16"] = f16
        if rank == 0:
            out["also"] = also

    if rank == 0:
        if not args.no_cpu_baseline and world == 1 and args.channels == 3:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
