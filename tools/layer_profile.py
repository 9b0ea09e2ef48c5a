"""Per-layer timing of one forward: run under `rocprofv3 --kernel-trace`, then `python tools_layer_profile.py report <dir>`."""
import csv, glob, sys, numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "report":
    plan = [l for l in open(sys.argv[2] + "/plan.txt").read().strip().split("\n")]
    B = int(open(sys.argv[2] + "/B.txt").read())
    ops = [l for l in plan if not l.startswith("total_macs")]
    f = glob.glob(sys.argv[2] + "/*/*_kernel_trace.csv")[0]
    rows = [r for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ker = [r for r in rows if "obb::k_conv" in r["Kernel_Name"] or "k_dwconv3" in r["Kernel_Name"] or "k_maxpool5" in r["Kernel_Name"] or "k_upsample2" in r["Kernel_Name"] or "k_attention" in r["Kernel_Name"] or "k_fused_chain" in r["Kernel_Name"] or "k_stem_conv" in r["Kernel_Name"] or "k_front" in r["Kernel_Name"] or "k_sppf_pools" in r["Kernel_Name"] or "k_bneck_stripe" in r["Kernel_Name"] or "k_c3k_image" in r["Kernel_Name"] or "k_dwpw_stripe" in r["Kernel_Name"] or "_f32" in r["Kernel_Name"]]
    nf = len(ker) // len(ops)
    ker = ker[-len(ops):]  # last forward
    tot = 0
    out = []
    for o, r in zip(ops, ker):
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        macs = float(o.split("macs")[-1])
        tf = 2 * macs * B / (us * 1e-6) / 1e12 if us > 0 else 0
        out.append((us, tf, o, r["VGPR_Count"] if "VGPR_Count" in r else "", r.get("LDS_Block_Size", "")))
        tot += us
    for us, tf, o, v, l in out:
        print("%8.1f us %7.1f TF  vgpr %s lds %s | %s" % (us, tf, v, l, o[:150]))
    print("total %.1f us, forwards seen %d" % (tot, nf))
else:
    import torch
    sys.path.insert(0, "tests"); sys.path.insert(0, ".")
    import make_weights, os
    import oriented_object_detection_amd
    from oriented_object_detection_amd import ops
    from oriented_object_detection_amd.model import YOLO
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/layers"
    os.makedirs(out, exist_ok=True)
    S = int(os.environ.get("OBB_SIZE", "416"))  # tile side (the 128-px scale of the dual-scale config: OBB_SIZE=128, weights of seed 1)
    m = YOLO(make_weights.ensure("n", 12, 3, 0 if S == 416 else 1), imgsz=S, precision=os.environ.get("OBB_PREC", "f16"),
             engine_options={k: bool(int(v)) for k, v in (kv.split("=") for kv in os.environ.get("OBB_OPTS", "").split(",") if kv)})  # e.g. OBB_OPTS=xtile=0
    open(out + "/plan.txt", "w").write("\n".join(ops.debug_plan(S, S)))
    open(out + "/B.txt", "w").write(str(B))
    tiles = torch.as_tensor(np.random.default_rng(0).integers(0, 256, (B, S, S, 3), dtype=np.uint8)).cuda()
    for _ in range(3):
        ops.forward(tiles)
    torch.cuda.synchronize()
