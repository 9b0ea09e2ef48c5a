"""obb_consensus on 51 k + 12 k synthetic rows; run under rocprofv3 --kernel-trace --stats"""
import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import synth
import oriented_object_detection_amd
from oriented_object_detection_amd import ops
n1, n2 = 51000, 12000
ext = 4096.0 * (n1 / 16384) ** 0.5
b1, c1, s1, _ = synth.make_dets(11, n1, extent=ext)
b2 = b1[:n2] + np.random.default_rng(0).normal(0, 2.0, (n2, 1))
c2, s2 = c1[:n2], s1[:n2] * 0.9
bb = torch.tensor(np.concatenate([b1, b2])).cuda(); cc = torch.tensor(np.concatenate([c1, c2])).cuda(); ss = torch.tensor(np.concatenate([s1, s2])).cuda()
for _ in range(6):
    kc = ops.consensus(bb, cc, ss, [0, n1, n1 + n2])
torch.cuda.synchronize()
print("kept", int(kc.sum()))
