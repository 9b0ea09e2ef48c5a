"""TEST INFRASTRUCTURE ONLY -- PyTorch-CPU restatement of the YOLO11-OBB forward the reference reaches through
`model(net_input, conf=...)` (Detect_OBB.py:81-83) in ultralytics==8.3.196 (requirements.txt:3).

ultralytics is NOT under /root/reference and not installable here, so this restates its published architecture
(yolo11-obb.yaml + nn/modules: Conv, C3k2, C3k, Bottleneck, SPPF, C2PSA/PSABlock/Attention, OBB head) -- SURVEY.md
Appendix A3.  Pinned by the published parameter table (n: 2.66 M params with nc=80) in tests/test_oracle_model.py;
otherwise **parity unpinned** (no real checkpoint or Ultralytics install is available offline).

Precisions:
  * "fp32"          -- what the reference computes on CPU (half=False).
  * "fp64"          -- the same fp32 weights and inputs evaluated in double precision: the yardstick that tells how far any fp32
                       evaluation (torch's own included) sits from the exact result of the same network.
  * "f16" / "bf16"  -- same graph with the HIP path's rounding points (16-bit storage of every activation tensor and
                       of the weights, fp32 accumulate/bias/SiLU/residual) so that kernel bugs are not hidden behind
                       a loose tolerance.

Weights are synthetic (no checkpoint exists offline): seeded, variance-preserving, BN already folded.
"""
import math
import struct

import numpy as np
import torch
import torch.nn.functional as F

SCALES = {  # depth, width, max_channels  (yolo11-obb.yaml `scales`)
    "n": (0.50, 0.25, 1024),
    "s": (0.50, 0.50, 1024),
    "m": (0.50, 1.00, 512),
    "l": (1.00, 1.00, 512),
    "x": (1.00, 1.50, 512),
}
REG_MAX = 16


def make_divisible(x, d=8):
    return int(math.ceil(x / d) * d)


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def half_round(t, mode):
    """round-to-nearest-even to the 16-bit storage type of the HIP path ("f16" or "bf16")"""
    return t.to(torch.float16 if mode == "f16" else torch.bfloat16).to(torch.float32)


class ConvRec:
    __slots__ = ("name", "c1", "c2", "k", "s", "g", "act", "w", "b")

    def __init__(self, name, c1, c2, k, s, g, act):
        self.name, self.c1, self.c2, self.k, self.s, self.g, self.act = name, c1, c2, k, s, g, act
        self.w = None
        self.b = None


class Yolo11OBB:
    """Functional YOLO11-OBB with BN-folded convs.  `convs` is an ordered dict name -> ConvRec (names follow the
    Ultralytics state-dict paths, e.g. "model.2.m.0.cv1")."""

    def __init__(self, scale="n", nc=12, ch=3, seed=0, cls_bias=-4.2):
        self.scale, self.nc, self.ch = scale, nc, ch
        self.depth, self.width, self.max_ch = SCALES[scale]
        self.convs = {}
        self._define()
        self._init_weights(seed, cls_bias)

    # ------------------------------------------------------------------ graph definition (channels only)
    def _c(self, c):
        return make_divisible(min(c, self.max_ch) * self.width, 8)

    def _n(self, n):
        return max(round(n * self.depth), 1) if n > 1 else n

    def _conv(self, name, c1, c2, k=1, s=1, g=1, act=True):
        self.convs[name] = ConvRec(name, c1, c2, k, s, g, act)

    def _bottleneck(self, name, c1, c2, e):
        c_ = int(c2 * e)
        self._conv(name + ".cv1", c1, c_, 3)
        self._conv(name + ".cv2", c_, c2, 3)

    def _c3k(self, name, c1, c2, n=2):
        c_ = int(c2 * 0.5)
        self._conv(name + ".cv1", c1, c_, 1)
        self._conv(name + ".cv2", c1, c_, 1)
        self._conv(name + ".cv3", 2 * c_, c2, 1)
        for i in range(n):
            self._bottleneck(f"{name}.m.{i}", c_, c_, 1.0)

    def _c3k2(self, name, c1, c2, n, c3k, e):
        c = int(c2 * e)
        self._conv(name + ".cv1", c1, 2 * c, 1)
        self._conv(name + ".cv2", (2 + n) * c, c2, 1)
        for i in range(n):
            if c3k:
                self._c3k(f"{name}.m.{i}", c, c, 2)
            else:
                self._bottleneck(f"{name}.m.{i}", c, c, 0.5)
        return c

    def _define(self):
        c = self._c
        n2 = self._n(2)
        big = self.scale in "mlx"
        self.cfg = {}
        self._conv("model.0", self.ch, c(64), 3, 2)
        self._conv("model.1", c(64), c(128), 3, 2)
        self.cfg[2] = (c(128), c(256), n2, big or False, 0.25)
        self._conv("model.3", c(256), c(256), 3, 2)
        self.cfg[4] = (c(256), c(512), n2, big or False, 0.25)
        self._conv("model.5", c(512), c(512), 3, 2)
        self.cfg[6] = (c(512), c(512), n2, True, 0.5)
        self._conv("model.7", c(512), c(1024), 3, 2)
        self.cfg[8] = (c(1024), c(1024), n2, True, 0.5)
        self.cfg[13] = (c(1024) + c(512), c(512), n2, big or False, 0.5)
        self.cfg[16] = (c(512) + c(512), c(256), n2, big or False, 0.5)
        self.cfg[19] = (c(256) + c(512), c(512), n2, big or False, 0.5)
        self.cfg[22] = (c(512) + c(1024), c(1024), n2, True, 0.5)
        for i in (2, 4, 6, 8):
            self._c3k2(f"model.{i}", *self.cfg[i])
        # SPPF
        c9 = c(1024)
        self._conv("model.9.cv1", c9, c9 // 2, 1)
        self._conv("model.9.cv2", c9 // 2 * 4, c9, 1)
        # C2PSA
        self.psa_n = n2
        cp = int(c9 * 0.5)
        self.psa_c = cp
        self.psa_heads = cp // 64
        self._conv("model.10.cv1", c9, 2 * cp, 1)
        self._conv("model.10.cv2", 2 * cp, c9, 1)
        for i in range(n2):
            nh = self.psa_heads
            hd = cp // nh
            kd = int(hd * 0.5)
            self._conv(f"model.10.m.{i}.attn.qkv", cp, cp + nh * kd * 2, 1, act=False)
            self._conv(f"model.10.m.{i}.attn.proj", cp, cp, 1, act=False)
            self._conv(f"model.10.m.{i}.attn.pe", cp, cp, 3, 1, g=cp, act=False)
            self._conv(f"model.10.m.{i}.ffn.0", cp, cp * 2, 1)
            self._conv(f"model.10.m.{i}.ffn.1", cp * 2, cp, 1, act=False)
        for i in (13, 16):
            self._c3k2(f"model.{i}", *self.cfg[i])
        self._conv("model.17", c(256), c(256), 3, 2)
        self._c3k2("model.19", *self.cfg[19])
        self._conv("model.20", c(512), c(512), 3, 2)
        self._c3k2("model.22", *self.cfg[22])
        # OBB head
        chs = (c(256), c(512), c(1024))
        self.head_ch = chs
        c2 = max(16, chs[0] // 4, REG_MAX * 4)
        c3 = max(chs[0], min(self.nc, 100))
        c4 = max(chs[0] // 4, 1)
        self.head_c = (c2, c3, c4)
        for i, x in enumerate(chs):
            self._conv(f"model.23.cv2.{i}.0", x, c2, 3)
            self._conv(f"model.23.cv2.{i}.1", c2, c2, 3)
            self._conv(f"model.23.cv2.{i}.2", c2, 4 * REG_MAX, 1, act=False)
        for i, x in enumerate(chs):
            self._conv(f"model.23.cv3.{i}.0.0", x, x, 3, g=x)
            self._conv(f"model.23.cv3.{i}.0.1", x, c3, 1)
            self._conv(f"model.23.cv3.{i}.1.0", c3, c3, 3, g=c3)
            self._conv(f"model.23.cv3.{i}.1.1", c3, c3, 1)
            self._conv(f"model.23.cv3.{i}.2", c3, self.nc, 1, act=False)
        for i, x in enumerate(chs):
            self._conv(f"model.23.cv4.{i}.0", x, c4, 3)
            self._conv(f"model.23.cv4.{i}.1", c4, c4, 3)
            self._conv(f"model.23.cv4.{i}.2", c4, 1, 1, act=False)

    # ------------------------------------------------------------------ synthetic weights
    def _init_weights(self, seed, cls_bias):
        g = torch.Generator().manual_seed(seed)
        for name, r in self.convs.items():
            fan_in = (r.c1 // r.g) * r.k * r.k
            gain = 1.676  # keeps pre-activation variance ~1 when the input is SiLU(N(0,1)) (E[silu^2] = 0.356)
            final = name.startswith("model.23.") and name.endswith(".2")
            if final:
                gain = 1.0
            r.w = (torch.randn((r.c2, r.c1 // r.g, r.k, r.k), generator=g) * (gain / math.sqrt(fan_in))).float()
            r.b = (torch.randn(r.c2, generator=g) * 0.05).float()
            if final and ".cv2." in name:
                r.b = r.b + 1.0  # Ultralytics bias_init: box branch 1.0
            if final and ".cv3." in name:
                r.b = r.b + cls_bias  # rare-but-non-empty conf > 0.25
        # data-dependent rescale on seeded noise tiles so that activations stay O(1) through all 23 layers
        cal = np.random.default_rng(seed + 12345).integers(0, 256, (8, 416, 416, self.ch), dtype=np.uint8)
        self.calib = True
        self.forward_raw(cal, "fp32")
        self.calib = False

    def n_params(self):
        return sum(r.w.numel() + r.b.numel() for r in self.convs.values())

    def macs(self, h, w):
        """MACs of one forward (convs + attention), SURVEY.md Appendix B accounting."""
        tot = 0
        sizes = self._trace_sizes(h, w)
        for name, r in self.convs.items():
            ho, wo = sizes[name]
            tot += ho * wo * r.c2 * (r.c1 // r.g) * r.k * r.k
        nh, cp = self.psa_heads, self.psa_c
        hd = cp // nh
        kd = hd // 2
        N = (h // 32) * (w // 32)
        tot += self.psa_n * nh * (N * N * kd + N * N * hd)
        return tot

    def _trace_sizes(self, h, w):
        s = {}
        lvl = {0: 2, 1: 4, 2: 4, 3: 8, 4: 8, 5: 16, 6: 16, 7: 32, 8: 32, 9: 32, 10: 32, 13: 16, 16: 8, 17: 16, 19: 16, 20: 32, 22: 32}
        for name in self.convs:
            parts = name.split(".")
            li = int(parts[1])
            if li == 23:
                st = (8, 16, 32)[int(parts[3])]
            else:
                st = lvl[li]
            s[name] = (h // st, w // st)
        return s

    # ------------------------------------------------------------------ forward
    def _q(self, t):
        return half_round(t, self.mode) if self.bf16 else t

    def _apply_conv(self, name, x, residual=None, out_f32=False):
        r = self.convs[name]
        if self.calib:  # data-dependent init: per-channel unit variance / zero mean pre-activations (what a folded BN gives)
            y0 = F.conv2d(x, r.w, None, stride=r.s, padding=r.k // 2, groups=r.g)
            mu = y0.mean((0, 2, 3))
            sd = y0.std((0, 2, 3)).clamp_min(1e-3)
            final = name.startswith("model.23.") and name.endswith(".2")
            r.w = (r.w / sd.view(-1, 1, 1, 1)).contiguous()
            r.b = r.b - mu / sd if not final else r.b - mu / sd
        w = half_round(r.w, self.mode) if self.bf16 else r.w
        b = r.b
        if self.mode == "fp64":
            w, b = w.double(), b.double()
        y = F.conv2d(x, w, b, stride=r.s, padding=r.k // 2, groups=r.g)
        if r.act:
            y = y / (1.0 + torch.exp(-y))  # SiLU
        if residual is not None:
            y = residual + y
        y = y if out_f32 else self._q(y)
        if self.taps is not None:
            self.taps[name] = y
        return y

    def _bneck(self, name, x):
        return self._apply_conv(name + ".cv2", self._apply_conv(name + ".cv1", x), residual=x)

    def _c3k_f(self, name, x, n=2):
        a = self._apply_conv(name + ".cv1", x)
        for i in range(n):
            a = self._bneck(f"{name}.m.{i}", a)
        b = self._apply_conv(name + ".cv2", x)
        return self._apply_conv(name + ".cv3", torch.cat([a, b], 1))

    def _c3k2_f(self, li, x):
        c1, c2, n, c3k, e = self.cfg[li]
        name = f"model.{li}"
        y = list(self._apply_conv(name + ".cv1", x).chunk(2, 1))
        for i in range(n):
            y.append(self._c3k_f(f"{name}.m.{i}", y[-1]) if c3k else self._bneck(f"{name}.m.{i}", y[-1]))
        return self._apply_conv(name + ".cv2", torch.cat(y, 1))

    def _attention(self, name, x):
        B, C, H, W = x.shape
        N = H * W
        nh = self.psa_heads
        hd = C // nh
        kd = hd // 2
        qkv = self._apply_conv(name + ".qkv", x)
        q, k, v = qkv.view(B, nh, kd * 2 + hd, N).split([kd, kd, hd], dim=2)
        attn = (q.transpose(-2, -1) @ k) * (kd ** -0.5)
        attn = attn.softmax(dim=-1)
        o = self._q((v @ attn.transpose(-2, -1)).reshape(B, C, H, W))
        if self.taps is not None:
            self.taps[name] = o
        o = self._apply_conv(name + ".pe", v.reshape(B, C, H, W), residual=o)  # x = attn_out + pe(v)
        return o

    def _psa(self, name, x):
        a = self._attention(name + ".attn", x)
        x = self._apply_conv(name + ".attn.proj", a, residual=x)
        f = self._apply_conv(name + ".ffn.0", x)
        return self._apply_conv(name + ".ffn.1", f, residual=x)

    @torch.no_grad()
    def forward_raw(self, tiles_u8_nhwc, precision="fp32", taps=None):
        """tiles uint8 [B,H,W,ch] (BGR for ch==3, exactly what the reference hands to model(...)).
        -> raw head [B, A, 64+nc+1] fp32: per anchor 4x16 DFL logits, nc class logits, 1 angle logit."""
        self.mode = precision
        self.bf16 = precision in ("bf16", "f16")  # 16-bit storage emulation on
        self.taps = taps
        self.calib = getattr(self, "calib", False)
        x = torch.as_tensor(np.ascontiguousarray(tiles_u8_nhwc))
        if self.ch == 3:
            x = x.flip(-1)  # BGR -> RGB only for 3-channel input (Appendix A2)
        x = self._q(x.permute(0, 3, 1, 2).float() / 255.0)
        if precision == "fp64":
            x = x.double()
        x0 = self._apply_conv("model.0", x)
        x1 = self._apply_conv("model.1", x0)
        x2 = self._c3k2_f(2, x1)
        x3 = self._apply_conv("model.3", x2)
        x4 = self._c3k2_f(4, x3)
        x5 = self._apply_conv("model.5", x4)
        x6 = self._c3k2_f(6, x5)
        x7 = self._apply_conv("model.7", x6)
        x8 = self._c3k2_f(8, x7)
        # SPPF
        y = [self._apply_conv("model.9.cv1", x8)]
        for _ in range(3):
            y.append(F.max_pool2d(y[-1], 5, 1, 2))
        x9 = self._apply_conv("model.9.cv2", torch.cat(y, 1))
        # C2PSA
        a, b = self._apply_conv("model.10.cv1", x9).split((self.psa_c, self.psa_c), 1)
        for i in range(self.psa_n):
            b = self._psa(f"model.10.m.{i}", b)
        x10 = self._apply_conv("model.10.cv2", torch.cat([a, b], 1))
        up = lambda t: t.repeat_interleave(2, 2).repeat_interleave(2, 3)  # nearest x2
        x13 = self._c3k2_f(13, torch.cat([up(x10), x6], 1))
        x16 = self._c3k2_f(16, torch.cat([up(x13), x4], 1))
        x17 = self._apply_conv("model.17", x16)
        x19 = self._c3k2_f(19, torch.cat([x17, x13], 1))
        x20 = self._apply_conv("model.20", x19)
        x22 = self._c3k2_f(22, torch.cat([x20, x10], 1))
        outs = []
        for i, f in enumerate((x16, x19, x22)):
            p = f"model.23.cv2.{i}"
            box = self._apply_conv(p + ".2", self._apply_conv(p + ".1", self._apply_conv(p + ".0", f)), out_f32=True)
            p = f"model.23.cv3.{i}"
            t = self._apply_conv(p + ".0.1", self._apply_conv(p + ".0.0", f))
            t = self._apply_conv(p + ".1.1", self._apply_conv(p + ".1.0", t))
            cls = self._apply_conv(p + ".2", t, out_f32=True)
            p = f"model.23.cv4.{i}"
            ang = self._apply_conv(p + ".2", self._apply_conv(p + ".1", self._apply_conv(p + ".0", f)), out_f32=True)
            o = torch.cat([box, cls, ang], 1)  # [B, 64+nc+1, h, w]
            outs.append(o.flatten(2))
        if taps is not None:
            taps.update({"x0": x0, "x1": x1, "x2": x2, "x4": x4, "x6": x6, "x8": x8, "x9": x9, "x10": x10, "x13": x13,
                         "x16": x16, "x19": x19, "x22": x22})
        self.taps = None
        return torch.cat(outs, 2).transpose(1, 2).contiguous()

    # ------------------------------------------------------------------ weight blob ("OBBW" v1) for obb_model_load
    def to_blob(self):
        recs = list(self.convs.values())
        hdr = struct.pack("<4sIIiiffii", b"OBBW", 1, len(recs), self.nc, self.ch, self.width, self.depth, self.max_ch, REG_MAX)
        hdr += struct.pack("<8s", self.scale.encode())
        rec_size = 64 + 6 * 4 + 2 * 8
        data_off = len(hdr) + rec_size * len(recs)
        data_off = (data_off + 63) // 64 * 64
        table, chunks, off = b"", [], data_off
        for r in recs:
            wb = r.w.contiguous().numpy().astype("<f4").tobytes()
            bb = r.b.contiguous().numpy().astype("<f4").tobytes()
            table += struct.pack("<64siiiiiiQQ", r.name.encode(), r.c1, r.c2, r.k, r.s, r.g, int(r.act), off, off + len(wb))
            chunks.append(wb + bb)
            off += len(wb) + len(bb)
        pad = b"\0" * (data_off - len(hdr) - len(table))
        return hdr + table + pad + b"".join(chunks)


def make_anchors(h, w):
    """anchor points (x, y) and strides for P3,P4,P5, row-major per level (Appendix A4)."""
    pts, st = [], []
    for s in (8, 16, 32):
        hh, ww = h // s, w // s
        sx = torch.arange(ww, dtype=torch.float32) + 0.5
        sy = torch.arange(hh, dtype=torch.float32) + 0.5
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        pts.append(torch.stack((xx, yy), -1).view(-1, 2))
        st.append(torch.full((hh * ww,), float(s)))
    return torch.cat(pts), torch.cat(st)
