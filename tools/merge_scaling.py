"""Time of the global (replicated) final merge vs number of records: the N-GPU bench merges N x (records of 1024 tiles) on every rank."""
import sys, time, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import synth
import oriented_object_detection_amd
from oriented_object_detection_amd import detect as D, ops
dev = torch.device("cuda")
for n in (12000, 24000, 48000, 96000):
    # boxes like the bench's survivors: ~12 per 416-px tile on a 16-column virtual map, stride 316
    tiles = n // 12
    rng = np.random.default_rng(0)
    t = np.repeat(np.arange(tiles), 12)[:n]
    ox, oy = (t % 16) * 316.0, (t // 16) * 316.0
    c = np.stack([ox + rng.uniform(20, 396, n), oy + rng.uniform(20, 396, n)], 1)
    w, h, a = rng.uniform(10, 60, n), rng.uniform(10, 60, n), rng.uniform(0, np.pi / 2, n)
    v1 = np.stack([np.cos(a), np.sin(a)], 1) * (w / 2)[:, None]; v2 = np.stack([-np.sin(a), np.cos(a)], 1) * (h / 2)[:, None]
    boxes = np.concatenate([c + v1 + v2, c + v1 - v2, c - v1 - v2, c - v1 + v2], 1)
    b = torch.as_tensor(boxes).to(dev); cls = torch.as_tensor(rng.integers(0, 12, n).astype(np.int32)).to(dev)
    conf = torch.as_tensor(rng.uniform(0.25, 1, n).astype(np.float32).astype(np.float64)).to(dev)
    for _ in range(2):
        order, keep, _ = ops.merge_detections(b, cls, conf, 0.4)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        order, keep, _ = ops.merge_detections(b, cls, conf, 0.4)
    torch.cuda.synchronize()
    print(n, "records: merge %.2f ms, kept %d" % ((time.perf_counter() - t0) / 3 * 1e3, int(keep.sum())))
