// Network front in one launch: model.0 (Conv 3x3 s2 on the uint8 tile, preprocess fused) -> model.1 (Conv 3x3 s2) -> model.2.cv1 (Conv 1x1),
// all with SiLU; the two intermediate tensors never exist in memory.  See front.hip.
#pragma once
#include "conv.h"

namespace obb {

struct FrontLaunch {
    const uint8_t *in = nullptr;  // uint8 NHWC tiles [B, Hin, Win, cin]
    TensorRef out;                // 16-bit slice [B, Hin/4, Win/4, 32] (plain NHWC or channel-blocked)
    const bf16_t *w0 = nullptr;   // pack_stem_weights(model.0, 16 couts)
    const bf16_t *w1 = nullptr;   // pack_conv_weights(model.1, 32, 16, 3, {13, 13, 3, NF 2, CK 16})
    const bf16_t *w2 = nullptr;   // pack_conv_weights(model.2.cv1, 32, 32, 1, {1, 1, 1, NF 2, CK 32})
    const float *b0 = nullptr, *b1 = nullptr, *b2 = nullptr;  // biases (padded to 64 floats)
    int B = 0, Hin = 0, Win = 0, cin = 3, f16 = 1;
};

// shapes the kernel exists for: 3 or 4 input channels -> 16 -> 32 -> 32, tile sides multiples of 52 (13 x 13 output tiles at 1/4 resolution)
bool front_supported(int cin, int c0, int c1, int c2, int Hin, int Win);
hipError_t launch_front(const FrontLaunch &L, hipStream_t st);

}  // namespace obb
