"""Multi-GPU form of the tile loop: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

Tiles are independent through forward / decode / Fast-NMS / border filter / per-tile merge (Detect_OBB.py:216-264 has
no cross-tile state), so the tile list is sharded across ranks with no data-path collective.  Only the fusion steps
(:290-291) need every detection of an image: survivors are exchanged once per image as fixed 48-byte records in ONE
fixed-capacity all-gather (row 0 of every rank's buffer carries its count, written on the device; (capacity + 1) x 48 B per rank:
0.2 MB at the starting capacity, 0.8 MB per rank at the bench's 16 384 -- 6.3 MB gathered at 8 ranks; a single step), compacted by one kernel (obb_gather_compact) whose small count tensor is the step's one host
read, after which every rank holds the identical, tile-ordered record list and runs the fusion replicated.
RCCL itself has not been executed yet (no multi-GPU box was available to the builder): the exchange is covered by 2-rank gloo
tests on CPU and a 2-rank gloo rehearsal on one MI355X.
"""
import torch
import torch.distributed as dist

from .detect import (DEFAULT, DetSet, TileRecords, cross_scale_consensus_filter_device, detect_symbols_records,
                     merge_detections_device, records_to_detset)


def shard_bounds(n_items, rank, world):
    """Contiguous balanced split: rank r owns [lo, hi).  Concatenating shards in rank order restores item order."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_records(rec, group=None, capacity=None):
    """Variable-length all-gather of TileRecords in ONE collective: every rank contributes a fixed-capacity int32 [capacity + 1, 12]
    buffer whose row 0 carries its record count (written on the device: the local count is never read by the host).  After the
    collective one kernel (obb_gather_compact) packs the valid rows of all ranks densely, in rank order == tile order for contiguous
    shards, and leaves the per-rank counts + their total in one small tensor: that tensor is the single host read of the step.
    The capacity (records per rank) must be the same on every rank: by default it is sticky per process group (keyed by the group's
    global-rank tuple, so a later group with a recycled Python id never inherits it) -- it starts at 4096 and, when some rank's count
    exceeds it (every rank sees the same counts, so every rank takes this branch), grows to the next power of two for a second exchange
    and for all later steps.  An explicit `capacity` (tests) is checked for equality across the ranks with one small all-reduce before it
    is used.  Unused rows are zeros, never uninitialised memory.  Nothing is ever truncated."""
    from . import ops
    world = dist.get_world_size(group)
    if world == 1:
        return rec
    dev = torch.device(rec.device)
    nccl = dist.get_backend(group) == "nccl"
    packed = rec.packed if rec.packed is not None else rec.pack()          # [rows >= n, 12] on the device
    n_dev = rec.count if rec.count is not None else torch.tensor([len(rec)], dtype=torch.int32, device=dev)
    key = tuple(dist.get_process_group_ranks(group if group is not None else dist.group.WORLD))
    if capacity:
        cap = int(capacity)
        lohi = torch.tensor([cap, -cap], dtype=torch.int64, device=dev if nccl else "cpu")
        dist.all_reduce(lohi, op=dist.ReduceOp.MAX, group=group)
        if int(lohi[0]) != cap or int(lohi[1]) != -cap:
            raise ValueError(f"all_gather_records: capacity differs between ranks (this rank {cap}, max {int(lohi[0])}, min {-int(lohi[1])})")
    else:
        cap = _CAPACITY.get(key, _CAPACITY_START)

    def exchange(cap):
        send = torch.zeros((cap + 1, 12), dtype=torch.int32, device=dev)
        send[0, 0:1] = n_dev                                               # device-to-device: no host read of the local count
        m = min(cap, packed.shape[0])
        send[1:1 + m] = packed[:m]                                         # rows past the count are never looked at by the receiver
        if nccl:  # RCCL exchanges the device buffers directly over xGMI
            recv = torch.empty((world, cap + 1, 12), dtype=torch.int32, device=dev)
            dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group)
        else:     # gloo rehearsal (CPU tests, single-GPU boxes): staged through host memory
            recv_h = torch.empty(world * (cap + 1) * 12, dtype=torch.int32)
            dist.all_gather_into_tensor(recv_h, send.view(-1).cpu(), group=group)
            recv = recv_h.view(world, cap + 1, 12).to(dev)
        if dev.type == "cpu":  # host tensors (the 2-rank gloo test on CPU): the same row bookkeeping by plain indexing, no arithmetic involved
            cs = recv[:, 0, 0].tolist()
            rows = torch.cat([recv[r, 1:1 + min(c, cap)] for r, c in enumerate(cs)], 0)
            return rows, cs + [int(rows.shape[0])]
        rows, counts = ops.gather_compact(recv)
        return rows, counts.tolist()                                       # the one host read of this step: world counts + the total

    rows, counts = exchange(cap)
    if max(counts[:world]) > cap:  # some rank did not fit
        cap = 1 << (max(counts[:world]) - 1).bit_length()
        rows, counts = exchange(cap)
    _CAPACITY[key] = max(_CAPACITY.get(key, 0), cap)
    total = counts[world]
    if total == 0:
        return TileRecords.empty(dev)
    out = TileRecords.from_packed(rows, None)
    out._n = total
    return out


_CAPACITY = {}  # global ranks of the process group -> sticky capacity (records per rank)
_CAPACITY_START = 4096


def detect_symbols_distributed(image, model, tile_size, overlap, cfg=DEFAULT, conf=None, batch=256, group=None):
    """Every rank holds the image, processes its contiguous shard of the tile list and receives all survivors."""
    from . import ops
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    H, W, _ = image.shape
    ntiles = len(ops.tile_grid(H, W, tile_size, overlap))
    lo, hi = shard_bounds(ntiles, rank, world)
    rec, rects_dev = detect_symbols_records(image, model, tile_size, overlap, cfg, conf, batch, tile_subset=list(range(lo, hi)))
    rec = all_gather_records(rec, group)
    return records_to_detset(rec, rects_dev, cfg, tile_size)


def process_image_distributed(image, models, cfg=DEFAULT, group=None):
    sets = {}
    for tile_size, overlap, model in zip(cfg.tile_sizes, cfg.overlaps, models):
        sets[tile_size] = detect_symbols_distributed(image, model, tile_size, overlap, cfg, group=group)
    consensus = cross_scale_consensus_filter_device(sets)
    merged, _ = merge_detections_device(consensus, cfg.iou_threshold)
    return merged
