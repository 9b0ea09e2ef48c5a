"""Writes the "OBBW" weight blob that `YOLO(...)` of this package loads, from a trained Ultralytics OBB checkpoint.

Run this ONCE where ultralytics is installed (it is not needed, and not available, where the blob is used):

    python tools/export_obbw.py best416.pt best416.obbw

The blob holds every convolution of the FUSED model (`model.fuse()`: BatchNorm folded into the conv, `Detect_OBB.py:26` loads the same
checkpoint and Ultralytics fuses it before predicting): name (the module path, without the `.conv` of a `Conv` wrapper), input / output
channels, kernel, stride, groups, activation flag (SiLU or none), fp32 OIHW weights and bias -- the layout `oracle/yolo11_obb.py::to_blob`
writes for the synthetic checkpoints and `csrc/engine.hip::obb_model_load` parses.  `blob_from_module` works on any module tree with
Ultralytics' structure (`Conv` wrappers = modules with a `.conv` Conv2d and an `.act`; the head's last layers = bare Conv2d), which is how
tests/test_host_logic_cpu.py exercises it without ultralytics."""
import struct
import sys

REG_MAX = 16
SCALES = {"n": (0.50, 0.25, 1024), "s": (0.50, 0.50, 1024), "m": (0.50, 1.00, 512), "l": (1.00, 1.00, 512), "x": (1.00, 1.50, 512)}  # depth, width, max_channels (yolo11-obb.yaml)


def conv_records(root):
    """-> list of (name, c1, c2, k, s, g, act, weight fp32 [c2, c1/g, k, k], bias fp32 [c2]) for every Conv2d under `root` (a fused model)."""
    import torch
    from torch import nn
    wrapped, recs = {}, []
    for name, mod in root.named_modules():
        conv = getattr(mod, "conv", None)
        if isinstance(conv, nn.Conv2d) and not isinstance(mod, nn.Conv2d):
            if getattr(mod, "bn", None) is not None and isinstance(getattr(mod, "bn"), nn.modules.batchnorm._BatchNorm):
                raise ValueError(f"{name}: BatchNorm not folded -- call model.fuse() first")
            wrapped[(name + ".conv") if name else "conv"] = (name, isinstance(getattr(mod, "act", None), nn.SiLU))
    for name, mod in root.named_modules():
        if not isinstance(mod, nn.Conv2d):
            continue
        rec_name, act = wrapped.get(name, (name, False))
        k, s, g = mod.kernel_size[0], mod.stride[0], mod.groups
        if mod.kernel_size[0] != mod.kernel_size[1] or mod.stride[0] != mod.stride[1] or mod.padding[0] != k // 2 or mod.dilation[0] != 1:
            raise ValueError(f"{name}: only square, 'same'-padded, undilated convolutions exist in YOLO11-OBB")
        w = mod.weight.detach().to(torch.float32).contiguous().cpu()
        b = (mod.bias.detach() if mod.bias is not None else torch.zeros(mod.out_channels)).to(torch.float32).contiguous().cpu()
        recs.append((rec_name, mod.in_channels, mod.out_channels, k, s, g, act, w, b))
    return recs


def blob_from_records(recs, nc, ch, scale):
    depth, width, max_ch = SCALES[scale]
    hdr = struct.pack("<4sIIiiffii", b"OBBW", 1, len(recs), nc, ch, width, depth, max_ch, REG_MAX) + struct.pack("<8s", scale.encode())
    rec_size = 64 + 6 * 4 + 2 * 8
    data_off = (len(hdr) + rec_size * len(recs) + 63) // 64 * 64
    table, chunks, off = b"", [], data_off
    for name, c1, c2, k, s, g, act, w, b in recs:
        if len(name.encode()) > 63:
            raise ValueError(f"record name too long: {name}")
        wb, bb = w.numpy().astype("<f4").tobytes(), b.numpy().astype("<f4").tobytes()
        table += struct.pack("<64siiiiiiQQ", name.encode(), c1, c2, k, s, g, int(act), off, off + len(wb))
        chunks.append(wb + bb)
        off += len(wb) + len(bb)
    return hdr + table + b"\0" * (data_off - len(hdr) - len(table)) + b"".join(chunks)


def blob_from_module(root, nc, scale):
    recs = conv_records(root)
    first = next(r for r in recs if r[0] == "model.0")
    return blob_from_records(recs, nc, first[1], scale)


def main(argv):
    if len(argv) != 3:
        raise SystemExit(__doc__)
    from ultralytics import YOLO
    y = YOLO(argv[1])
    det = y.model
    det.float().eval()
    det.fuse()
    scale = str(det.yaml.get("scale") or "n")
    nc = int(det.yaml.get("nc", len(det.names)))
    blob = blob_from_module(det, nc, scale)
    with open(argv[2], "wb") as f:
        f.write(blob)
    print(f"{argv[2]}: {len(blob)} bytes, scale {scale}, nc {nc}")


if __name__ == "__main__":
    main(sys.argv)
