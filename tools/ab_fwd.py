"""Same-box A/B of the forward: each configuration runs in its own process (so that OBB_LIB / engine options can differ), the
configurations alternate `rounds` times, and the table shows every run and the medians -- box-to-box differences of +-3 % hide +-1 % effects
in separate gpurun calls.  Usage:  python tools/ab_fwd.py [--b 1024] [--rounds 3] [--prec f32] [--size 416] name=ENV1=v1,ENV2=v2 ...
   e.g.  base=OBB_LIB=tools/scratch/libobbhip_base.so  new=  plain=OBB_OPTS=blk32=0
Worker mode (internal): python tools/ab_fwd.py --worker B prec size"""
import os, subprocess, sys

if len(sys.argv) > 1 and sys.argv[1] == "--worker":
    import numpy as np, torch
    sys.path.insert(0, "tests"); sys.path.insert(0, ".")
    import make_weights
    import oriented_object_detection_amd  # noqa: F401
    from oriented_object_detection_amd import ops
    from oriented_object_detection_amd.model import YOLO
    B, prec, S = int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    opts = {k: bool(int(v)) for k, v in (kv.split("=") for kv in os.environ.get("OBB_OPTS", "").split(",") if kv)}
    m = YOLO(make_weights.ensure("n", 12, 3, 0 if S == 416 else 1), imgsz=S, precision=prec, engine_options=opts)
    tiles = torch.as_tensor(np.random.default_rng(0).integers(0, 256, (B, S, S, 3), dtype=np.uint8)).cuda()
    for _ in range(4):
        ops.forward(tiles)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = []
    for _ in range(3):
        e0.record()
        for _ in range(6):
            ops.forward(tiles)
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 6)
    print("MS %.4f" % sorted(best)[1])
    sys.exit(0)

args = sys.argv[1:]
B, rounds, prec, size = 1024, 3, "f32", 416
cfgs = []
i = 0
while i < len(args):
    if args[i] == "--b": B = int(args[i + 1]); i += 2
    elif args[i] == "--rounds": rounds = int(args[i + 1]); i += 2
    elif args[i] == "--prec": prec = args[i + 1]; i += 2
    elif args[i] == "--size": size = int(args[i + 1]); i += 2
    else:
        name, _, envs = args[i].partition("=")
        env = {}
        for kv in [e for e in envs.split(",") if e]:
            k, _, v = kv.partition("=")
            if k == "OBB_OPTS" and "OBB_OPTS" in env: env[k] += "," + v
            else: env[k] = v
        # OBB_OPTS=a=0 -> key OBB_OPTS, value a=0 (several engine options: OBB_OPTS=a=0,OBB_OPTS=b=0)
        cfgs.append((name, env)); i += 1
res = {n: [] for n, _ in cfgs}
for r in range(rounds):
    for n, env in cfgs:
        e = dict(os.environ, **env)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(B), prec, str(size)], env=e, capture_output=True, text=True)
        ms = [float(l.split()[1]) for l in out.stdout.splitlines() if l.startswith("MS ")]
        if not ms:
            print(n, "FAILED", out.stderr[-600:], flush=True); continue
        res[n].append(ms[0])
        print(f"round {r} {n:12s} {ms[0]:8.3f} ms per {B} tiles", flush=True)
print("# medians (ms per %d tiles of %d px, %s)" % (B, size, prec))
for n, v in res.items():
    if v:
        print(f"{n:12s} {sorted(v)[len(v) // 2]:8.3f}   runs {' '.join('%.3f' % x for x in v)}")
