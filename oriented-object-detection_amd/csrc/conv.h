// Implicit-GEMM convolution on MFMA (bf16 x bf16 -> fp32) for gfx950: launch descriptor + host-side weight packing.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "halfx.h"

namespace obb {

typedef half_bits_t bf16_t;  // raw 16-bit storage (fp16 or bf16, see halfx.h)

// One fused conv launch: out[b, oy, ox, co] = epilogue( sum_{ky,kx,ci} in[b, oy*s+ky-p, ox*s+kx-p, ci] * W[co][ci][ky][kx] )
// epilogue: + bias -> SiLU (optional) -> + residual (optional) -> bf16 (or fp32) store into a channel slice of `out`.
// Tensors are NHWC slices: element (b, y, x, c) of X lives at X.p + b*X.bs + (y*W + x)*X.cs + X.co + c.
// Channel-blocked variant (cpb > 0): the buffer is [C / blk][image][pixel][blk] with blk = 8 * cpb channels per block, so that a
// consumer which walks the channels in stages (or reads a channel slice) touches dense memory instead of 16..32-byte pieces of wide
// pixel rows: element (b, y, x, c) lives at X.p + (c / blk) * X.ps + b * X.bs + (y*W + x) * blk + c % blk   (co must be a multiple of blk).
struct TensorRef {
    void *p = nullptr;
    int64_t bs = 0;  // batch stride (elements)
    int cs = 0;      // pixel stride = channels of the underlying buffer (elements); blk for a blocked buffer
    int co = 0;      // channel offset of the slice
    int cpb = 0;     // 16-byte chunks (8 channels) per block; 0 = plain NHWC
    int64_t ps = 0;  // block stride (elements), blocked buffers only
};

struct ConvLaunch {
    TensorRef in, out, res;  // res.p == nullptr -> no residual
    const bf16_t *wpk = nullptr;  // packed weights (pack_conv_weights)
    const float *bias = nullptr;  // padded to a multiple of 64 floats
    const bf16_t *lut = nullptr;  // u8 -> bf16(v/255) table for the network input (in_u8)
    int B = 0, Hin = 0, Win = 0, Hout = 0, Wout = 0;
    int cin = 0;    // logical input channels (u8 input: 3 or 4, staged as 8)
    int cout = 0;
    int ks = 1, stride = 1;
    int act = 1;
    int in_u8 = 0, out_f32 = 0, flip_bgr = 0;
    int f16 = 1;  // storage type: 1 = fp16, 0 = bf16
    // 1x1 over a virtual concat (up_c > 0): input channels [0, up_c) = nearest-x2 upsample of `in` (a tensor of half the resolution),
    // channels [up_c, cin) = `in2` (full resolution).  1-D launches only (B = 1, Hin = 1, Win = batch * up_HW pixels).
    TensorRef in2;
    int up_c = 0, up_W = 0, up_HW = 0;  // full-resolution width and pixels per image
    // optional fused trailing 1x1 conv without activation (the last layer of an OBB-head branch): its fp32 output rows go to tail_out,
    // the 16-bit output of THIS layer is then never written (conv_tail_supported lists the shapes that have a kernel)
    const bf16_t *tail_wpk = nullptr;  // pack_conv_weights(w2, tail_cout, cout, 1, {.., NF = tail_cout <= 16 ? 1 : 4, CK = cout})
    const float *tail_bias = nullptr;  // padded to a multiple of 64 floats
    int tail_cout = 0;
    bool tail_act16 = false;           // the tail has SiLU and writes ITS 16-bit result to `out` (cv1 of a C3k2 block behind a stride-2 conv)
    TensorRef tail_out;
    int tail_out_hw = 0;
    int out_hw = 0;  // > 0: 1-D launch whose OUTPUT is split per image: pixel P -> (b = P / out_hw, P % out_hw) with out.bs
    // tiling (chosen by plan_conv)
    int TH = 1, TW = 64, MF = 1, NF = 4, CK = 32;
    int tiles_y = 1, tiles_x = 1;
    // NI > 1: one tile = NI whole images of a small map (TH x TW = Hout x Wout, NI * TH * TW <= 64 * MF): the 4 x 4 level of the 128-px scale
    // would otherwise run as 8 x 8 tiles, three quarters of them padding.  3x3 layers with 64-cout groups and a 16-bit output only
    // (conv_ni_supported).
    int NI = 1;
};

struct ConvTiling { int TH, TW, MF, NF, CK; };

// chooses tile shape / fragment blocking for a layer
ConvTiling plan_conv(int ks, int stride, int cin, int cout, int Hout, int Wout, bool pair = true);  // pair: 64 -> 64-cout 3x3 layers on k_conv3_pair

// Host: repack fp32 OIHW weights into MFMA A-operand fragment order (bf16, RNE):
//   [cout_block][stage][kstep][nf][lane 0..63][8]
// with the cout permutation that makes every lane own 4*NF contiguous output channels (see conv.hip).
std::vector<bf16_t> pack_conv_weights(const float *w_oihw, int cout, int cin, int ks, const ConvTiling &t,
                                      const int *cout_perm /* optional: logical cout -> source row */, int in_u8, bool f16);

int conv_ksteps(int ks, int CK);
size_t conv_lds_bytes(const ConvLaunch &L);
bool conv_tail_supported(int ks, int MF, int NF, int cout1, int cout2, bool act16 = false, int TH = 0);
// NI (images per tile) for a launch planned on the generic small-map tile, or 1: the shapes the multi-image form has a kernel path for
int conv_ni_supported(const ConvLaunch &L);
hipError_t launch_conv(const ConvLaunch &L, hipStream_t st);

}  // namespace obb
