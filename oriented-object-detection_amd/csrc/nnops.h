// Launchers of the non-GEMM forward layers (nnops.hip).
#pragma once
#include "conv.h"

namespace obb {
hipError_t launch_dwconv3(const TensorRef &in, const TensorRef &out, const TensorRef &res, const bf16_t *w16, const float *bias, int B,
                          int H, int W, int C, int act, bool f16, hipStream_t st);
hipError_t launch_maxpool5(const TensorRef &in, const TensorRef &out, int B, int H, int W, int C, bool f16, hipStream_t st);
hipError_t launch_upsample2(const TensorRef &in, const TensorRef &out, int B, int H, int W, int C, hipStream_t st);
hipError_t launch_attention(const TensorRef &qkv, const TensorRef &out, int B, int N, int nh, int kd, int hd, bool f16, hipStream_t st);
}  // namespace obb
