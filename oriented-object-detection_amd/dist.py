"""Multi-GPU form of the tile loop: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

Tiles are independent through forward / decode / Fast-NMS / border filter / per-tile merge (Detect_OBB.py:216-264 has
no cross-tile state), so the tile list is sharded across ranks with no data-path collective.  Only the fusion steps
(:290-291) need every detection of an image: survivors are exchanged once per image as fixed 48-byte records
(all-gather of counts, then one all-gather of padded record buffers -- KB-scale, latency-bound, a single step), after
which every rank holds the identical, tile-ordered record list and runs the fusion replicated.
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


def all_gather_records(rec, group=None):
    """Variable-length all-gather of TileRecords; result is ordered by (rank, local order) == tile order for
    contiguous shards."""
    world = dist.get_world_size(group)
    if world == 1:
        return rec
    dev = rec.tile.device
    # RCCL ("nccl") exchanges device buffers directly over xGMI; the gloo rehearsal path (CPU tests, single-GPU boxes) stages
    # through host memory.
    xdev = dev if dist.get_backend(group) == "nccl" else torch.device("cpu")
    buf = rec.pack().to(xdev)
    n = torch.tensor([buf.shape[0]], dtype=torch.int64, device=xdev)
    counts = [torch.zeros(1, dtype=torch.int64, device=xdev) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    mx = max(counts)
    if mx == 0:
        return rec
    padded = torch.zeros((mx, 12), dtype=torch.int32, device=xdev)
    padded[: buf.shape[0]] = buf
    outs = [torch.zeros((mx, 12), dtype=torch.int32, device=xdev) for _ in range(world)]
    dist.all_gather(outs, padded, group=group)
    return TileRecords.unpack(torch.cat([o[:c] for o, c in zip(outs, counts)], 0).contiguous().to(dev))


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
