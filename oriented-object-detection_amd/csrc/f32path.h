// fp32-arithmetic form of the OBBModel forward (obb_set_option "precision" = 32): what the reference computes with Ultralytics'
// default half=False (Detect_OBB.py:79-83).  Activations and weights stay fp32 end to end; the convolutions run on the exact-f32
// matrix instruction v_mfma_f32_16x16x4_f32 (bit-for-bit a k-ordered fmaf chain), everything else on the fp32 VALU.  One kernel per
// layer (no fusion): every activation is observable, which is what the tight parity tests want.  See f32path.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "conv.h"

namespace obb {

// out[b, oy, ox, co] = epilogue( sum in[b, oy*s+ky-p, ox*s+kx-p, ci] * W[co][ci][ky][kx] ); TensorRef offsets count fp32 elements
// (in: bytes of the uint8 tile when in_u8).  Plain NHWC slices only.
struct Conv32Launch {
    TensorRef in, out, res;
    const float *wpk = nullptr;   // pack_conv32_weights
    const float *bias = nullptr;  // cout floats, padded to a multiple of 64
    const float *lut = nullptr;   // 256 floats: (float)v / 255.0f (IEEE division, what `im.float() / 255` gives)
    int B = 0, Hin = 0, Win = 0, Hout = 0, Wout = 0;
    int cin = 0, cout = 0, ks = 1, stride = 1, act = 1, in_u8 = 0, flip_bgr = 0;
    int out_hw = 0;  // > 0: 1-D launch (B = 1, H = 1, W = batch * pixels) whose output rows are split per image: P -> (P / out_hw, P % out_hw)
    // tiling (plan_conv32): output tile TH x TW (<= 208 pixels = 13 fragments of 16), CK input channels per LDS stage, WC waves along cout
    int TH = 1, TW = 208, CK = 16, WC = 4;
    int tiles_y = 1, tiles_x = 1;
};

struct Conv32Tiling { int TH, TW, CK, WC; };
Conv32Tiling plan_conv32(int ks, int stride, int cin, int cout, int Hout, int Wout, bool in_u8);
// fp32 OIHW -> A-operand order [cout fragment of 16][stage][k16 step][lane][4]: lane (r = lane & 15, g = lane >> 4) holds the weights of
// cout r for the four k values of its 4-channel chunk q = 4 * step + g (tap = q / (CK/4), channels 4 * (q % (CK/4)) ..+3); element s of
// the vector feeds MFMA step s.  cout_perm (optional): logical cout -> source row.
std::vector<float> pack_conv32_weights(const float *w_oihw, int cout, int cin, int ks, const Conv32Tiling &t, const int *cout_perm, bool in_u8);
size_t conv32_lds_bytes(const Conv32Launch &L);
hipError_t launch_conv32(const Conv32Launch &L, hipStream_t st);

hipError_t launch_dwconv3_f32(const TensorRef &in, const TensorRef &out, const TensorRef &res, const float *w9c, const float *bias, int B, int H, int W,
                              int C, int act, hipStream_t st);
hipError_t launch_maxpool5_f32(const TensorRef &in, const TensorRef &out, int B, int H, int W, int C, hipStream_t st);
hipError_t launch_upsample2_f32(const TensorRef &in, const TensorRef &out, int B, int H, int W, int C, hipStream_t st);
hipError_t launch_attention_f32(const TensorRef &qkv, const TensorRef &out, int B, int N, int nh, int kd, int hd, hipStream_t st);

}  // namespace obb
