// Launchers of the non-GEMM forward layers (nnops.hip).
#pragma once
#include "conv.h"

namespace obb {
hipError_t launch_dwconv3(const TensorRef &in, const TensorRef &out, const TensorRef &res, const bf16_t *w16, const float *bias, int B,
                          int H, int W, int C, int act, bool f16, hipStream_t st);
hipError_t launch_maxpool5(const TensorRef &in, const TensorRef &out, int B, int H, int W, int C, bool f16, hipStream_t st);
// the three chained 5x5 max pools of SPPF over member 0 of the concat buffer `cat` (C channels per member), results into members 1..3
hipError_t launch_sppf_pools(const TensorRef &cat, int B, int H, int W, int C, bool f16, hipStream_t st);
hipError_t launch_upsample2(const TensorRef &in, const TensorRef &out, int B, int H, int W, int C, hipStream_t st);
hipError_t launch_attention(const TensorRef &qkv, const TensorRef &out, int B, int N, int nh, int kd, int hd, bool f16, bool use_mfma, hipStream_t st);
}  // namespace obb
