// Context, error plumbing and workspace pool shared by every C-ABI entry point of libobbhip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/obbhip.h"

namespace obb {
struct Model;  // engine.hip
// obb_set_option switches of the forward engine.  Every fused form keeps a switch that restores its separate launches: the A/B parity
// tests compare the two on identical inputs.  graph / fwd_split / microbatch steer how a forward is issued (their defaults may be
// preset from the environment for profiling runs: OBB_GRAPH, OBB_FWD_SPLIT, OBB_MICROBATCH -- read once per context).
struct EngineOpts {
    bool tail = true, tail16 = true, bneck = true, bneck_cv2 = true, c3kimg = true, dwpw = true, upfold = true, stem = true, front = true, pair = true, hmerge = true,
         sppf_fuse = true, attn_mfma = true, xtile = true, nitile = true, nc2 = true, blk32 = true, c3k2f = true, pw32 = true, graph = true;
    int fwd_split = 0 /* auto = 2 chains */, microbatch = 1024;
};
}

struct obb_ctx {
    int device = 0;
    std::string err;
    // grow-only device scratch, one slot per purpose so that concurrent users inside one call never alias
    struct Slot { void *p = nullptr; size_t bytes = 0; };
    std::vector<Slot> ws;
    std::shared_ptr<obb::Model> model;                   // model of the active slot
    std::map<int, std::shared_ptr<obb::Model>> slots;    // parked models (obb_set_option "model_slot")
    int slot = 0;
    bool opt_f16 = true;
    bool opt_f32 = false;  // fp32 arithmetic end to end ("precision" = 32)
    obb::EngineOpts opt;   // obb_set_option switches
    void *workspace(int slot, size_t bytes);
    ~obb_ctx();
};

namespace obb {
int set_error(obb_ctx *ctx, int code, const char *fmt, ...);
extern thread_local std::string g_tls_error;

#define OBB_HIP(ctx, call)                                                                                  \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return obb::set_error(ctx, OBB_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                                  __FILE__, __LINE__);                                                      \
    } while (0)

#define OBB_REQUIRE(ctx, cond, ...)                                      \
    do {                                                                 \
        if (!(cond)) return obb::set_error(ctx, OBB_ERR_INVALID, __VA_ARGS__); \
    } while (0)

#define OBB_LAUNCH_CHECK(ctx) OBB_HIP(ctx, hipGetLastError())

// workspace slots
enum { WS_GEOM_A = 0, WS_GEOM_B, WS_GEOM_C, WS_GEOM_D, WS_GEOM_E, WS_NMS_A, WS_NMS_B, WS_NMS_C, WS_NMS_D, WS_SURV_A, WS_SURV_B, WS_SEL, WS_DT, WS_TRAIN_A, WS_TRAIN_B, WS_TRAIN_C, WS_COUNT };  // (WS_TRAIN_*: the training kernels' own scratch -- never shared with the inference path's slots)

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
}  // namespace obb
