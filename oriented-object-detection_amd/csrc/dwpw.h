// Depthwise 3x3 (+SiLU) -> pointwise 1x1 (+SiLU) [-> plain 1x1 to the head tensor] over row stripes: see dwpw.hip.
#pragma once
#include <vector>

#include "conv.h"

namespace obb {

struct DwPwLaunch {
    TensorRef in, out;   // plain NHWC 16-bit tensors (out: 64 channels; unused when tail_cout > 0)
    TensorRef tail_out;  // fp32 rows of the head tensor (tail_cout > 0)
    void *sink = nullptr;  // >= 1 KiB of device scratch: lanes of the tail without a cout store there (stores stay unconditional)
    const bf16_t *dw_w = nullptr;   // [9][cin] in the storage type
    const float *dw_b = nullptr;    // [cin]
    const bf16_t *pw_w = nullptr;   // pack_conv_weights(w, 64, cin, 1, {NF = 4, CK = cin})
    const float *pw_b = nullptr;    // [64]
    const bf16_t *tail_w = nullptr; // pack_dwpw_tail(w2, tail_cout, f16)
    const float *tail_b = nullptr;  // [16]
    int tail_cout = 0;
    int B = 0, H = 0, W = 0, cin = 0, f16 = 1;
};

bool dwpw_supported(int cin, int cout, int H, int W, int tail_cout);
// A-operand fragments of the trailing 1x1 (64 -> cout2 <= 16): [k step 0 / 1][lane][8]; a lane of the B operand holds 16 consecutive
// channels of its pixel, so k slot 8 (lane >> 4) + e of step s carries channel 16 (lane >> 4) + 8 s + e
std::vector<bf16_t> pack_dwpw_tail(const float *w, int cout2, bool f16);
hipError_t launch_dwpw(const DwPwLaunch &L, hipStream_t st);

}  // namespace obb
