// Network input layer (Conv 3x3 s2 on the uint8 tile, preprocess fused) as full-width row stripes: see stem.hip.
#pragma once
#include "conv.h"

namespace obb {

struct StemLaunch {
    const uint8_t *in = nullptr;  // uint8 NHWC tiles [B, Hin, Win, cin]
    TensorRef out;                // 16-bit NHWC slice [B, Hin/2, Win/2, cout]
    const bf16_t *wpk = nullptr;  // pack_stem_weights
    const float *bias = nullptr;  // cout floats (padded to a multiple of 64)
    int B = 0, Hin = 0, Win = 0, cin = 3, cout = 16, act = 1, f16 = 1;
};

bool stem_supported(int cin, int cout, int ks, int stride, int Hin, int Win);
bool stem_scale_is_exact(bool f16);  // v * (1/255) and v / 255 round to the same 16-bit value for every byte v
std::vector<bf16_t> pack_stem_weights(const float *w_oihw, int cout, int cin, bool flip_bgr, bool f16);
hipError_t launch_stem(const StemLaunch &L, hipStream_t st);

}  // namespace obb
