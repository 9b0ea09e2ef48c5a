// Fused Bottleneck (3x3 -> 3x3 + shortcut) over full-width row stripes: see bneck.hip.
#pragma once
#include <vector>

#include "conv.h"

namespace obb {

struct BneckLaunch {
    TensorRef y1, y2;  // input / output members of a channel-blocked concat buffer (block size = C): dense [image][pixel][C] planes
    const bf16_t *w1pk = nullptr, *w2pk = nullptr;  // pack_conv_weights(.., ks 3, {NF = 1, CK = C}) and (.., ks 3, {NF = C/16, CK = C/2})
    const float *bias1 = nullptr, *bias2 = nullptr;  // padded to a multiple of 64 floats
    int B = 0, H = 0, W = 0, C = 0, f16 = 1;
    // optional closing 1x1 of the C3k2 block (cv2 over [y0 | y1 | y2] -> CO channels, SiLU) fused behind conv 2: y2 is then never written
    int CO = 0;
    TensorRef y0, out;                    // block 0 of the same concat buffer; the block's output tensor (plain NHWC or channel-blocked)
    const bf16_t *wc32pk = nullptr;       // pack_conv_weights(w, CO, 3C, 1, {NF = CO/16, CK = C == 32 ? 96 : 32}): the 32-wide k steps
    const bf16_t *wc16pk = nullptr;       // C = 16 only: pack_bneck_k16(w, CO, 48, 32): the 16-wide step over y2
    const float *biasc = nullptr;         // CO floats
};

bool bneck_supported(int C, int H, int W);
bool bneck_cv2_supported(int C, int CO);
std::vector<bf16_t> pack_bneck_k16(const float *w, int cout, int cin, int c0, bool f16);
hipError_t launch_bneck(const BneckLaunch &L, hipStream_t st);

}  // namespace obb
