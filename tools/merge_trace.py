"""kernel trace target: obb_merge_detections on the dense synthetic sets of tools/postproc_bench.py (run under rocprofv3 --kernel-trace --stats);
prints the HIP-event time per call as well.  Usage: python tools/merge_trace.py [N ...]"""
import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import synth
import oriented_object_detection_amd
from oriented_object_detection_amd import ops

for n in [int(a) for a in sys.argv[1:]] or [16384]:
    bb, cc, ss, _ = synth.make_dets(1, n, extent=4096.0)
    Bt, Ct, St = torch.tensor(bb).cuda(), torch.tensor(cc).cuda(), torch.tensor(ss).cuda()
    for _ in range(3):
        ops.merge_detections(Bt, Ct, St, 0.4)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        o, k, nk = ops.merge_detections(Bt, Ct, St, 0.4)
    e1.record()
    torch.cuda.synchronize()
    print(f"N = {n}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per call, kept {int(nk.item())}", flush=True)
