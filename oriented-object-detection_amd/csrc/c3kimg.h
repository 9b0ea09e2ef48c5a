// Inner C3k block of the stride-32 level as one persistent workgroup per image: see c3kimg.hip.
#pragma once
#include "conv.h"

namespace obb {

struct C3kImgLaunch {
    TensorRef in, out;            // plain NHWC slices: 128 input channels, 128 output channels, 13 x 13 pixels
    const bf16_t *wts = nullptr;  // weight stream: c3kimg_pieces() 16-byte pieces = the layers' MFMA fragments (pack_conv_weights order), layer after layer:
                                  //   [cv1|cv2] (cout 128, K 128), m.0.cv1, m.0.cv2, m.1.cv1, m.1.cv2 (3x3 64 -> 64), cv3 (cout 128, K 128)
    const float *bias = nullptr;  // [6][128] floats, same layer order
    int B = 0, f16 = 1;
};

bool c3kimg_supported(int H, int W, int c_in, int c_hidden, int c_out, int n);
int c3kimg_pieces();  // 16-byte pieces of the weight stream
hipError_t launch_c3kimg(const C3kImgLaunch &L, hipStream_t st);

}  // namespace obb
