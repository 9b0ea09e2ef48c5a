// LDS-resident layer chains: several consecutive layers of the OBBModel forward executed by ONE kernel per spatial tile, with
// every intermediate activation kept in LDS (never written to HBM).  Used for the blocks whose unfused form is bound by HBM
// traffic and launch latency rather than by the matrix pipe (the 104x104 C3k2 block, the class / angle branches of the OBB head).
#pragma once
#include "conv.h"

namespace obb {

enum FusedStepType { FS_CONV = 0, FS_DW = 1, FS_STORE = 2 };
constexpr int kFusedMaxSteps = 8;
constexpr int kFusedThreads = 512;

// One layer of a chain.  All regions are LDS arrays [pixel][channel] of 16-bit values with `pst` bytes per pixel and `w` pixels
// per row; a step produces the `oh x ow` pixel block whose origin is (tile origin - halo) in the level's pixel grid.
struct FusedStep {
    int type = FS_CONV;
    int oh = 0, ow = 0, halo = 0;
    float inv_ow = 0.f;
    // input region: LDS byte offset, pixel pitch, row length, position of tap (0,0) of output pixel (0,0), channel byte offset
    int in_off = 0, in_pst = 0, in_w = 0, in_y0 = 0, in_x0 = 0, in_cb = 0;
    int cin = 0, ks = 1, sh = 0, stride = 1;  // sh = log2(cin / 8)
    // output region (FS_CONV / FS_DW), same description; to_global = 1: fp32 rows of the caller's head tensor instead
    int out_off = 0, out_pst = 0, out_w = 0, out_y0 = 0, out_x0 = 0, out_cb = 0;
    int cout = 0, act = 1, mask = 0, to_global = 0;
    // residual region (res_off < 0: none)
    int res_off = -1, res_pst = 0, res_w = 0, res_y0 = 0, res_x0 = 0, res_cb = 0;
    // weights: LDS byte offset (FS_CONV: MFMA A fragments [cb][kstep][nf][lane][8]; FS_DW: halves [9][C]); bias fp32 in global memory
    int w_off = 0, NF = 1, ncb = 1, kst = 1;
    const float *bias = nullptr;
};

struct FusedLaunch {
    TensorRef in, out;        // chain input (16-bit NHWC slice); final output (16-bit slice for FS_STORE, fp32 head rows for to_global)
    int B = 0, H = 0, W = 0;  // level dims (all layers of a chain are stride 1 on one pyramid level)
    int TH = 13, TW = 13;     // output tile
    int in_halo = 0, in_C = 0, in_off = 0, in_pst = 0;  // prefetched load of the chain input: (TH + 2 halo) x (TW + 2 halo) pixels
    int nsteps = 0;
    FusedStep steps[kFusedMaxSteps];
    const bf16_t *wts = nullptr;  // all steps' weights, copied to LDS once per workgroup
    int w_bytes = 0, w_lds_off = 0;
    int lds_bytes = 0;
    int f16 = 1;
};

hipError_t launch_fused(const FusedLaunch &L, hipStream_t st);

}  // namespace obb
