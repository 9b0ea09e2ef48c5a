// Per-row device functions of the post-processing chain, shared by the stand-alone kernels (k_results, k_tile_post) and by the fused
// per-tile kernel of obb_tile_survivors: ONE definition of each arithmetic step, so that the fused path is bit-identical by construction.
#pragma once
#include <hip/hip_runtime.h>

namespace obb {

static constexpr float kPiF = 3.14159274101257324f;      // (float)math.pi
static constexpr float kHalfPiF = 1.57079637050628662f;  // (float)(math.pi / 2)

__device__ __forceinline__ float py_remainder(float a, float b) {  // torch.remainder / Python % for b > 0
    float r = fmodf(a, b);
    if (r != 0.0f && ((r < 0.0f) != (b < 0.0f))) r += b;
    return r;
}

// Ultralytics construct_result for one NMS row d = (x, y, w, h, conf, cls, angle): regularize_rboxes, scale_boxes(xywh=True) with the
// letterbox (gain, pad_x, pad_y) in lb3 (nullptr: identity), xywhr2xyxyxyxy.  xywhr5 may be nullptr.
__device__ __forceinline__ void results_row(const float *__restrict__ d, const float *__restrict__ lb3, float *__restrict__ xywhr5, float *__restrict__ p) {
    float x = d[0], y = d[1], w = d[2], h = d[3], t = d[6];
    bool swap = py_remainder(t, kPiF) >= kHalfPiF;  // regularize_rboxes
    float w_ = swap ? h : w, h_ = swap ? w : h;
    t = py_remainder(t, kHalfPiF);
    if (lb3) {  // scale_boxes(xywh=True): subtract the letterbox pad, divide by the gain
        float gain = lb3[0], px = lb3[1], py = lb3[2];
        x -= px; y -= py;
        x /= gain; y /= gain; w_ /= gain; h_ /= gain;
    }
    if (xywhr5) { xywhr5[0] = x; xywhr5[1] = y; xywhr5[2] = w_; xywhr5[3] = h_; xywhr5[4] = t; }
    float c = cosf(t), s = sinf(t);  // xywhr2xyxyxyxy
    float v1x = w_ / 2.0f * c, v1y = w_ / 2.0f * s;
    float v2x = -h_ / 2.0f * s, v2y = h_ / 2.0f * c;
    p[0] = x + v1x + v2x; p[1] = y + v1y + v2y;
    p[2] = x + v1x - v2x; p[3] = y + v1y - v2y;
    p[4] = x - v1x - v2x; p[5] = y - v1y - v2y;
    p[6] = x - v1x + v2x; p[7] = y - v1y + v2y;
}

// Per-detection body of detect_symbols (Detect_OBB.py:229-262) for local corners lp[8] of a detection in the tile (x, y, x2, y2):
// global corners (float64), border filter, strike angle.
__device__ __forceinline__ void tile_post_row(const float *__restrict__ lp, int cls, int x, int y, int x2, int y2, int margin, int strike_cls,
                                              double *__restrict__ g, double &angle, bool &inside) {
    double p[8];
    for (int k = 0; k < 8; ++k) p[k] = (double)lp[k];  // float(v) widening, Detect_OBB.py:229
    for (int k = 0; k < 4; ++k) { g[2 * k] = p[2 * k] + (double)x; g[2 * k + 1] = p[2 * k + 1] + (double)y; }  // :233-234
    double cx = (g[0] + g[2] + g[4] + g[6]) / 4.0, cy = (g[1] + g[3] + g[5] + g[7]) / 4.0;  // :163-164
    double cxr = cx - (double)x, cyr = cy - (double)y, m = (double)margin;
    double cw = (double)(x2 - x), ch = (double)(y2 - y);
    bool in = true;
    if (margin > 0) in = (m <= cxr && cxr <= (cw - m)) && (m <= cyr && cyr <= (ch - m));  // :174, :242
    inside = in;
    double a = 0.0;
    if (cls == strike_cls) {  // :251-254 uses the LOCAL points
        a = atan2(p[6] - p[0], p[7] - p[1]) * (180.0 / 3.141592653589793);
        a = (a > 0) ? 180 - a : fabs(a);
    }
    angle = a;
}

}  // namespace obb
