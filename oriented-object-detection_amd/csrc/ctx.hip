// Context lifetime + error plumbing of libobbhip.so (C-ABI in include/obbhip.h).
#include "ctx.h"

#include <unistd.h>

#include <algorithm>
#include <cstdlib>

namespace obb {
thread_local std::string g_tls_error;

int set_error(obb_ctx *ctx, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_tls_error = buf;
    if (ctx) ctx->err = buf;
    return code;
}
}  // namespace obb

void *obb_ctx::workspace(int slot, size_t bytes) {
    if ((int)ws.size() <= slot) ws.resize(slot + 1);
    Slot &s = ws[slot];
    if (s.bytes < bytes) {
        if (s.p) (void)hipFree(s.p);
        size_t want = bytes + bytes / 4 + 256;
        if (hipMalloc(&s.p, want) != hipSuccess) { s.p = nullptr; s.bytes = 0; return nullptr; }
        s.bytes = want;
    }
    return s.p;
}

obb_ctx::~obb_ctx() {
    for (auto &s : ws)
        if (s.p) (void)hipFree(s.p);
}

extern "C" {

int obb_version(void) { return 100; }

int obb_ctx_create(int device, obb_ctx **out) {
    if (!out) return obb::set_error(nullptr, OBB_ERR_INVALID, "obb_ctx_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    for (int attempt = 0; attempt < 10 && (e != hipSuccess || ndev <= 0); ++attempt) {  // a GPU waking from its low-power state can miss the first probe
        (void)hipGetLastError();
        usleep(300 * 1000);
        e = hipGetDeviceCount(&ndev);
    }
    if (e != hipSuccess || ndev <= 0)
        return obb::set_error(nullptr, OBB_ERR_HIP, "obb_ctx_create: no HIP device visible (%s)", hipGetErrorString(e));
    if (device < 0 || device >= ndev)
        return obb::set_error(nullptr, OBB_ERR_INVALID, "obb_ctx_create: device %d out of range [0,%d)", device, ndev);
    OBB_HIP(nullptr, hipSetDevice(device));
    hipDeviceProp_t prop;
    OBB_HIP(nullptr, hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
        return obb::set_error(nullptr, OBB_ERR_STATE, "obb_ctx_create: device %d is %s; this library is built for gfx950 only",
                              device, prop.gcnArchName);
    obb_ctx *c = new obb_ctx();
    c->device = device;
    if (const char *v = getenv("OBB_GRAPH")) c->opt.graph = atoi(v) != 0;  // profiling presets (per-layer kernel traces want eager, one-chain runs)
    if (const char *v = getenv("OBB_FWD_SPLIT")) c->opt.fwd_split = std::max(0, std::min(4, atoi(v)));
    if (const char *v = getenv("OBB_MICROBATCH")) c->opt.microbatch = std::max(1, std::min(1024, atoi(v)));
    *out = c;
    return OBB_OK;
}

int obb_ctx_destroy(obb_ctx *ctx) {
    if (!ctx) return OBB_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    delete ctx;
    return OBB_OK;
}

const char *obb_last_error(const obb_ctx *ctx) { return ctx ? ctx->err.c_str() : obb::g_tls_error.c_str(); }

}  // extern "C"
