// Fused Bottleneck (3x3 -> 3x3 + shortcut) over full-width row stripes: see bneck.hip.
#pragma once
#include "conv.h"

namespace obb {

struct BneckLaunch {
    TensorRef y1, y2;  // input / output members of a channel-blocked concat buffer (block size = C): dense [image][pixel][C] planes
    const bf16_t *w1pk = nullptr, *w2pk = nullptr;  // pack_conv_weights(.., ks 3, {NF = 1, CK = C}) and (.., ks 3, {NF = C/16, CK = C/2})
    const float *bias1 = nullptr, *bias2 = nullptr;  // padded to a multiple of 64 floats
    int B = 0, H = 0, W = 0, C = 0, f16 = 1;
};

bool bneck_supported(int C, int H, int W);
hipError_t launch_bneck(const BneckLaunch &L, hipStream_t st);

}  // namespace obb
