// Implicit-GEMM convolution on MFMA (bf16 x bf16 -> fp32) for gfx950: launch descriptor + host-side weight packing.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

namespace obb {

typedef uint16_t bf16_t;  // raw bits

// One fused conv launch: out[b, oy, ox, co] = epilogue( sum_{ky,kx,ci} in[b, oy*s+ky-p, ox*s+kx-p, ci] * W[co][ci][ky][kx] )
// epilogue: + bias -> SiLU (optional) -> + residual (optional) -> bf16 (or fp32) store into a channel slice of `out`.
// Tensors are NHWC slices: element (b, y, x, c) of X lives at X.p + b*X.bs + (y*W + x)*X.cs + X.co + c.
struct TensorRef {
    void *p = nullptr;
    int64_t bs = 0;  // batch stride (elements)
    int cs = 0;      // pixel stride = channels of the underlying buffer (elements)
    int co = 0;      // channel offset of the slice
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
    // tiling (chosen by plan_conv)
    int TH = 1, TW = 64, MF = 1, NF = 4, CK = 32;
    int tiles_y = 1, tiles_x = 1;
};

struct ConvTiling { int TH, TW, MF, NF, CK; };

// chooses tile shape / fragment blocking for a layer
ConvTiling plan_conv(int ks, int stride, int cin, int cout, int Hout, int Wout);

// Host: repack fp32 OIHW weights into MFMA A-operand fragment order (bf16, RNE):
//   [cout_block][stage][kstep][nf][lane 0..63][8]
// with the cout permutation that makes every lane own 4*NF contiguous output channels (see conv.hip).
std::vector<bf16_t> pack_conv_weights(const float *w_oihw, int cout, int cin, int ks, const ConvTiling &t,
                                      const int *cout_perm /* optional: logical cout -> source row */, int in_u8);

int conv_ksteps(int ks, int CK);
size_t conv_lds_bytes(const ConvLaunch &L);
hipError_t launch_conv(const ConvLaunch &L, hipStream_t st);

inline bf16_t f32_to_bf16(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN stays NaN
    return (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
inline float bf16_to_f32(bf16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

}  // namespace obb
