"""Multi-GPU form of the tile loop: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

Tiles are independent through forward / decode / Fast-NMS / border filter / per-tile merge (Detect_OBB.py:216-264 has
no cross-tile state), so the tile list is sharded across ranks with no data-path collective.  Only the fusion steps
(:290-291) need every detection of an image: survivors are exchanged once per image as fixed 48-byte records in ONE
fixed-capacity all-gather (row 0 of every rank's buffer carries its count; KB- to MB-scale, latency-bound, a single
step, one host read), after which every rank holds the identical, tile-ordered record list and runs the fusion replicated.
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
    buffer whose row 0 carries its record count; the `world` counts are then read with one host transfer and the valid rows sliced.
    Result is ordered by (rank, local order) == tile order for contiguous shards.  `capacity` (records per rank) defaults to the next
    power of two above this rank's count, agreed through the same collective: a rank whose count exceeds the agreed capacity triggers
    one second, larger exchange (rare: survivors are a few per tile), so nothing is ever truncated."""
    world = dist.get_world_size(group)
    if world == 1:
        return rec
    dev = rec.tile.device
    # RCCL ("nccl") exchanges device buffers directly over xGMI; the gloo rehearsal path (CPU tests, single-GPU boxes) stages
    # through host memory.
    xdev = dev if dist.get_backend(group) == "nccl" else torch.device("cpu")
    buf = rec.pack().to(xdev)
    n = buf.shape[0]

    def exchange(cap):
        send = torch.zeros((cap + 1, 12), dtype=torch.int32, device=xdev)
        send[0, 0] = n
        m = min(n, cap)
        send[1:1 + m] = buf[:m]
        recv = torch.empty(world * (cap + 1) * 12, dtype=torch.int32, device=xdev)
        dist.all_gather_into_tensor(recv, send.view(-1), group=group)
        recv = recv.view(world, cap + 1, 12)
        return recv, recv[:, 0, 0].tolist()  # the one host read of this step

    cap = int(capacity) if capacity else _DEFAULT_CAPACITY.get("cap", 1024)
    recv, counts = exchange(cap)
    if max(counts) > cap:  # some rank did not fit: every rank sees the same counts, so every rank takes this branch
        cap = 1 << (max(counts) - 1).bit_length()
        _DEFAULT_CAPACITY["cap"] = max(_DEFAULT_CAPACITY.get("cap", 1024), cap)  # sticky: the next steps start large enough
        recv, counts = exchange(cap)
    if sum(counts) == 0:
        return TileRecords.empty(dev)
    rows = torch.cat([recv[r, 1:1 + c] for r, c in enumerate(counts) if c], 0)
    return TileRecords.unpack(rows.contiguous().to(dev))


_DEFAULT_CAPACITY = {}


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
