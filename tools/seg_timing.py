import sys, time, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import synth
import oriented_object_detection_amd
from oriented_object_detection_amd import ops
def run(nseg, per):
    n = nseg * per
    b, c, s, _ = synth.make_dets(1, n, extent=4096.0)
    B, Cc, S = torch.tensor(b).cuda(), torch.tensor(c).cuda(), torch.tensor(s).cuda()
    seg = torch.arange(0, n + 1, per, dtype=torch.int32).cuda()
    for _ in range(3): ops.merge_segments(B, Cc, S, seg, 0.4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.merge_segments(B, Cc, S, seg, 0.4)
    e1.record(); torch.cuda.synchronize()
    print(f"merge_segments nseg={nseg} per={per}: {e0.elapsed_time(e1)/20*1e3:.1f} us")
for nseg, per in ((1, 12), (256, 12), (256, 100), (256, 300), (1, 300), (2048, 12)):
    run(nseg, per)
