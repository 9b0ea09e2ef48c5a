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
struct Model;  // engine.h
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
    bool opt_tail = true;
    bool opt_fuse = false;  // LDS-resident layer chains (fused.hip): parity-tested, but slower than layer-by-layer on MI355X so far
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
enum { WS_GEOM_A = 0, WS_GEOM_B, WS_GEOM_C, WS_GEOM_D, WS_GEOM_E, WS_NMS_A, WS_NMS_B, WS_NMS_C, WS_NMS_D, WS_COUNT };

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
}  // namespace obb
