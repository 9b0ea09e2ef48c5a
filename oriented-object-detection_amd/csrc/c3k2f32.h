// fp32 C3k2 block with one Bottleneck (the 104 x 104 block of the n / s scales: model.2) as ONE launch: see c3k2f32.hip
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "conv.h"

namespace obb {

struct C3k2F32Launch {
    TensorRef cat;  // the block's cv1 output [y0 | y1]: plain NHWC fp32, 2 C channels from cat.co on
    TensorRef out;  // the block's output (CO channels): plain NHWC or 8-channel blocks (TensorRef::cpb = 2)
    const float *w1 = nullptr, *w2 = nullptr, *wc = nullptr;  // pack_conv32_weights forms: Bottleneck cv1 (3x3, C -> C/2, one 16-channel stage),
                                                              // cv2 (3x3, C/2 -> C, one 8-channel stage), closing 1x1 (3 C -> CO, one 48-channel
                                                              // stage, rows permuted by c3k2f32_cout_perm)
    const float *b1 = nullptr, *b2 = nullptr, *bc = nullptr;  // biases; bc in the permuted row order, padded to CO
    int B = 0, H = 0, W = 0, C = 16, CO = 64;
};

bool c3k2f32_supported(int C, int CO, int H, int W);
// accumulator row r of cout fragment f holds output channel perm[16 f + r] = 16 (r >> 2) + 4 f + (r & 3): a lane's 16 results are 16
// consecutive channels (64 contiguous bytes, two whole 8-channel blocks)
std::vector<int> c3k2f32_cout_perm(int CO);
void c3k2f32_tile(int H, int W, int &TH, int &TW);
hipError_t launch_c3k2f32(const C3k2F32Launch &L, hipStream_t st);

}  // namespace obb
