// fp32 1x1 convolution with the activations read straight from global memory into the MFMA B operand: see pw32.hip
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "conv.h"

namespace obb {

struct Pw32Launch {
    TensorRef in;    // plain NHWC fp32, the whole (sub-)batch as one dense pixel row: in.p + pixel * in.cs + in.co
    TensorRef out;   // plain NHWC or 8-channel blocks per image (TensorRef::cpb = 2)
    TensorRef res;   // optional residual (plain NHWC), added behind the activation
    const float *wpk = nullptr;   // pack_pw32_weights
    const float *bias = nullptr;  // cout floats (natural channel order), padded to a multiple of 64
    int64_t npix = 0;             // B * H * W
    int hw = 0;                   // pixels per image (the blocked output needs (image, pixel))
    int cin = 0, cout = 0, act = 1;
};

bool pw32_supported(int cin, int cout);
// [cout block of 64][16-channel piece][cout fragment (4)][lane][4]: lane (r = lane & 15, g = lane >> 4) of fragment f holds the weights of output
// channel 64 cb + 16 (r >> 2) + 4 f + (r & 3) for input channels 16 piece + 4 g .. + 3 -- a lane's 16 accumulator rows are 16 consecutive
// output channels of its pixel.  cout_perm (optional): logical cout -> source row.
std::vector<float> pack_pw32_weights(const float *w_oi, int cout, int cin, const int *cout_perm);
hipError_t launch_pw32(const Pw32Launch &L, hipStream_t st);

}  // namespace obb
