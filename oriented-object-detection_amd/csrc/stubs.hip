// Temporary: entry points declared in include/obbhip.h whose kernels are not written yet.  Each returns an error
// (never a silent fallback).  This file shrinks to nothing as the kernels land.
#include "ctx.h"
using namespace obb;
#define NOT_YET(ctx, name) return set_error(ctx, OBB_ERR_STATE, name ": not implemented in this build")
extern "C" {
int obb_gather_tiles(obb_ctx *ctx, const uint8_t *, int32_t, int32_t, int32_t, const int32_t *, int32_t, int32_t, uint8_t *, obb_stream_t) { NOT_YET(ctx, "obb_gather_tiles"); }
int obb_letterbox(obb_ctx *ctx, const uint8_t *, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, uint8_t *, int32_t, int32_t, obb_stream_t) { NOT_YET(ctx, "obb_letterbox"); }
int obb_decode_nms(obb_ctx *ctx, const float *, int32_t, int32_t, int32_t, float, float, int32_t, float *, int32_t *, obb_stream_t) { NOT_YET(ctx, "obb_decode_nms"); }
int obb_decode(obb_ctx *ctx, const float *, int32_t, int32_t, int32_t, float *, obb_stream_t) { NOT_YET(ctx, "obb_decode"); }
int obb_probiou_nms(obb_ctx *ctx, const float *, const float *, int64_t, float, int32_t *, uint8_t *, obb_stream_t) { NOT_YET(ctx, "obb_probiou_nms"); }
int obb_results(obb_ctx *ctx, const float *, const float *, int64_t, float *, float *, obb_stream_t) { NOT_YET(ctx, "obb_results"); }
}
