// fp32-arithmetic form of the OBBModel forward (obb_set_option "precision" = 32): what the reference computes with Ultralytics'
// default half=False (Detect_OBB.py:79-83).  Activations and weights stay fp32 end to end; the convolutions run on the exact-f32
// matrix instruction v_mfma_f32_16x16x4_f32 (bit-for-bit a k-ordered fmaf chain), everything else on the fp32 VALU.  Fused forms (a
// trailing 1x1 behind its producer, Upsample + Concat read in place, merged sibling convs) keep the engine's switches: with "tail" = 0
// the plan is one kernel per layer and every activation is observable, which is what the tight parity tests want.  See f32path.hip.
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
    // 1x1 over a virtual concat (up_c > 0): input channels [0, up_c) = nearest-x2 upsample of `in` (a tensor of half the resolution),
    // channels [up_c, cin) = `in2` (full resolution); neither the upsampled tensor nor the concat exists.  1-D launches only.
    TensorRef in2;
    int up_c = 0, up_W = 0, up_HW = 0;  // full-resolution width and pixels per image
    // optional fused trailing 1x1 conv (tail_cout > 0): consumes THIS layer's activated output tile straight from LDS (this layer's
    // own output is then never written); tail_act = SiLU on the tail; its rows go to tail_out (an activation slice or the head tensor)
    const float *tail_w = nullptr;  // pack_conv32_weights(w2, tail_cout, cout, 1, {1, *, CK = cout, *})
    const float *tail_b = nullptr;  // padded like `bias`
    int tail_cout = 0, tail_act = 0;
    TensorRef tail_out;
    int tail_out_hw = 0;
    // (TAIL writing the class logits of the head) per-anchor maximum of the tail's outputs, dense float[image][cmax_bs] + cmax_off: the
    // candidate gate of obb_decode_nms_gate reads 4 bytes per anchor instead of a 48-byte piece of every 320-byte head row
    float *cmax = nullptr;
    int64_t cmax_bs = 0;
    // tiling (plan_conv32): output tile TH x TW (<= 16 * (8 / WC) * MFM pixels), CK input channels per LDS stage, WC of the workgroup's 8
    // waves along cout (16 couts each), the other 8 / WC along the pixel fragments, MFM fragments of 16 pixels per wave
    int TH = 1, TW = 208, CK = 16, WC = 4, MFM = 7;
    int dw = 0, dw_act = 1;  // dw: a depthwise 3x3 (+ bias, SiLU if dw_act) runs in front of this 1x1 inside the launch (wpk from pack_dwpw32_weights)
    int NI = 1;  // > 1: one tile = NI whole images of a small map (TH x TW = the map; the 8 x 8 / 4 x 4 levels of the 128-px scale)
    int tiles_y = 1, tiles_x = 1;
    int NC = 1;  // cout fragments per wave: 2 = WC waves x 32 couts, MFM <= 4 pixel fragments (plain / VCAT forms; wpk packed for it)
    int xtile = 1;  // resident workgroups that walk several tiles, the next tile's first stage fetched under this tile's last k loop (plain / VCAT forms)
};

struct Conv32Tiling { int TH, TW, CK, WC, MFM, NI, NC; };  // (NC 0 / 1: one cout fragment per wave)
// vcat: the input is a virtual [upsample | skip] concat; nc2: the layer may take the two-fragments-per-wave form (no tail, no depthwise prologue)
Conv32Tiling plan_conv32(int ks, int stride, int cin, int cout, int Hout, int Wout, bool in_u8, bool vcat = false, bool nc2 = false);
// true if the 1x1 conv (cout1 -> cout2) can run as the fused tail of a layer tiled as `t`
bool conv32_tail_supported(const Conv32Tiling &t, int cout1, int cout2);
// DWConv 3x3 -> Conv 1x1 as one launch (k_conv_f32 DW): tiling (TH = 0: no kernel for the shapes) and the per-stage weight blocks
// [pw fragments of the stage][9 depthwise taps + bias of the stage's channels]
Conv32Tiling plan_dwpw32(int cin, int cout, int H, int W);
std::vector<float> pack_dwpw32_weights(const float *pw_oihw, int cout, int cin, const float *dw_c9, const float *dw_bias, const Conv32Tiling &t);
// fp32 OIHW -> A-operand order [cout fragment of 16][stage][k16 step][lane][4]: lane (r = lane & 15, g = lane >> 4) holds the weights of
// cout r for the four k values of its 4-channel chunk q = 4 * step + g (tap = q / (CK/4), channels 4 * (q % (CK/4)) ..+3); element s of
// the vector feeds MFMA step s.  cout_perm (optional): logical cout -> source row.
std::vector<float> pack_conv32_weights(const float *w_oihw, int cout, int cin, int ks, const Conv32Tiling &t, const int *cout_perm, bool in_u8);
size_t conv32_lds_bytes(const Conv32Launch &L);
hipError_t launch_conv32(const Conv32Launch &L, hipStream_t st);

// model.0 (3x3 s2 on the uint8 tile) as full-width row stripes: see k_stem_f32
struct Stem32Launch {
    const uint8_t *in = nullptr;  // [B][Hin][Win][cin] bytes
    TensorRef out;                // fp32 NHWC slice, Hin/2 x Win/2 x cout
    const float *wpk = nullptr;   // pack_stem32_weights
    const float *bias = nullptr, *lut = nullptr;
    int B = 0, Hin = 0, Win = 0, cin = 3, cout = 16, act = 1;
};
bool stem32_supported(int cin, int cout, int ks, int stride, int Hin, int Win);
std::vector<float> pack_stem32_weights(const float *w_oihw, int cout, int cin, bool flip_bgr);
hipError_t launch_stem32(const Stem32Launch &L, hipStream_t st);

hipError_t launch_dwconv3_f32(const TensorRef &in, const TensorRef &out, const TensorRef &res, const float *w9c, const float *bias, int B, int H, int W,
                              int C, int act, hipStream_t st);
hipError_t launch_maxpool5_f32(const TensorRef &in, const TensorRef &out, int B, int H, int W, int C, hipStream_t st);
hipError_t launch_upsample2_f32(const TensorRef &in, const TensorRef &out, int B, int H, int W, int C, hipStream_t st);
hipError_t launch_sppf_pools_f32(const TensorRef &cat, int B, int H, int W, int C, hipStream_t st);  // cat = [x | m1 | m2 | m3], C channels each
hipError_t launch_attention_f32(const TensorRef &qkv, const TensorRef &out, int B, int N, int nh, int kd, int hd, bool use_mfma, hipStream_t st);

}  // namespace obb
