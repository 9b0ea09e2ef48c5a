// Training-step slice: the optimiser update behind `model.train(...)` (Train_OBB.py:796-841).  Ultralytics'
// trainer builds torch.optim.SGD(momentum, nesterov=True) or torch.optim.AdamW(betas=(momentum, 0.999)) ("auto": by the iteration count,
// restated in train.py) over three parameter groups (weights with decay / norm weights / biases); here a group is ONE flat fp32 buffer
// (parameters, gradients and optimiser state side by side) and the update of the whole group is one streaming launch -- 16 (SGD) or 20
// (AdamW) bytes read and 8 / 12 written per parameter, float4 per lane, against the one launch per tensor and per elementary operation of
// the unfused form.  The arithmetic follows torch.optim's single-tensor reference order operation by operation (tests compare with it).
#include <cmath>

#include "ctx.h"

namespace obb {

// torch.optim.SGD (dampening 0, maximize False): d = g + wd p;  buf = first ? d : mu buf + d;  d = nesterov ? d + mu buf : buf;  p -= lr d
__global__ __launch_bounds__(256) void k_sgd_step(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ buf, int64_t n, float lr, float mu,
                                                 float wd, int nesterov, int first) {
    const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= n) return;
    if (i4 + 4 <= n) {
        float4 pv = *reinterpret_cast<const float4 *>(p + i4);
        const float4 gv = *reinterpret_cast<const float4 *>(g + i4);
        // (momentum 0: `buf` may be NULL -- include/obbhip.h: "no buffer touched")
        float4 bv = (first || mu == 0.f) ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4 *>(buf + i4);
        float pe[4] = {pv.x, pv.y, pv.z, pv.w}, ge[4] = {gv.x, gv.y, gv.z, gv.w}, be[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float d = wd != 0.f ? ge[j] + wd * pe[j] : ge[j];
            if (mu != 0.f) {
                be[j] = first ? d : mu * be[j] + d;
                d = nesterov ? d + mu * be[j] : be[j];
            }
            pe[j] = pe[j] - lr * d;
        }
        *reinterpret_cast<float4 *>(p + i4) = make_float4(pe[0], pe[1], pe[2], pe[3]);
        if (mu != 0.f) *reinterpret_cast<float4 *>(buf + i4) = make_float4(be[0], be[1], be[2], be[3]);
        return;
    }
    for (int64_t i = i4; i < n; ++i) {
        float d = wd != 0.f ? g[i] + wd * p[i] : g[i];
        if (mu != 0.f) {
            const float b = first ? d : mu * buf[i] + d;
            buf[i] = b;
            d = nesterov ? d + mu * b : b;
        }
        p[i] = p[i] - lr * d;
    }
}

// torch.optim.AdamW (amsgrad False): p *= 1 - lr wd;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g g;
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)            (bias corrections computed on the host in double, as torch does)
__device__ __forceinline__ void adamw_one(float &p, float g, float &m, float &v, float lr, float b1, float b2, float eps, float wd, float step_size, float sqrt_bc2) {
    p = p * (1.0f - lr * wd);
    m = m + (g - m) * (1.0f - b1);         // torch: exp_avg.lerp_(grad, 1 - beta1)
    v = v * b2 + (1.0f - b2) * g * g;      // torch: exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(v) / sqrt_bc2 + eps;  // torch: (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    p = p - step_size * (m / denom);
}

__global__ __launch_bounds__(256) void k_adamw_step(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, int64_t n,
                                                   float lr, float b1, float b2, float eps, float wd, float step_size, float sqrt_bc2) {
    const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= n) return;
    if (i4 + 4 <= n) {
        float4 pv = *reinterpret_cast<const float4 *>(p + i4), mv = *reinterpret_cast<const float4 *>(m + i4), vv = *reinterpret_cast<const float4 *>(v + i4);
        const float4 gv = *reinterpret_cast<const float4 *>(g + i4);
        adamw_one(pv.x, gv.x, mv.x, vv.x, lr, b1, b2, eps, wd, step_size, sqrt_bc2);
        adamw_one(pv.y, gv.y, mv.y, vv.y, lr, b1, b2, eps, wd, step_size, sqrt_bc2);
        adamw_one(pv.z, gv.z, mv.z, vv.z, lr, b1, b2, eps, wd, step_size, sqrt_bc2);
        adamw_one(pv.w, gv.w, mv.w, vv.w, lr, b1, b2, eps, wd, step_size, sqrt_bc2);
        *reinterpret_cast<float4 *>(p + i4) = pv;
        *reinterpret_cast<float4 *>(m + i4) = mv;
        *reinterpret_cast<float4 *>(v + i4) = vv;
        return;
    }
    for (int64_t i = i4; i < n; ++i) adamw_one(p[i], g[i], m[i], v[i], lr, b1, b2, eps, wd, step_size, sqrt_bc2);
}

}  // namespace obb

using namespace obb;

extern "C" {

int obb_sgd_step(obb_ctx *ctx, float *param, const float *grad, float *momentum_buf, int64_t n, float lr, float momentum, float weight_decay,
                 int32_t nesterov, int32_t first_step, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && n >= 0, "obb_sgd_step: bad arguments");
    if (n == 0) return OBB_OK;
    OBB_REQUIRE(ctx, param && grad && (momentum == 0.f || momentum_buf), "obb_sgd_step: NULL buffer");
    OBB_REQUIRE(ctx, !(nesterov && momentum <= 0.f), "obb_sgd_step: nesterov needs a momentum (as torch.optim.SGD)");
    OBB_REQUIRE(ctx, ((uintptr_t)param | (uintptr_t)grad | (uintptr_t)momentum_buf) % 16 == 0, "obb_sgd_step: buffers must be 16-byte aligned");
    const int64_t nb = (n + 1023) / 1024;
    OBB_REQUIRE(ctx, nb < (1ll << 31), "obb_sgd_step: n too large");
    hipLaunchKernelGGL(k_sgd_step, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)s, param, grad, momentum_buf, n, lr, momentum, weight_decay, nesterov, first_step);
    OBB_HIP(ctx, hipGetLastError());
    return OBB_OK;
}

int obb_adamw_step(obb_ctx *ctx, float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, int64_t step, float lr, float beta1,
                   float beta2, float eps, float weight_decay, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && n >= 0 && step >= 1, "obb_adamw_step: bad arguments (step counts from 1)");
    if (n == 0) return OBB_OK;
    OBB_REQUIRE(ctx, param && grad && exp_avg && exp_avg_sq, "obb_adamw_step: NULL buffer");
    OBB_REQUIRE(ctx, ((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0, "obb_adamw_step: buffers must be 16-byte aligned");
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step), bc2 = 1.0 - std::pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1), sqrt_bc2 = (float)std::sqrt(bc2);
    const int64_t nb = (n + 1023) / 1024;
    OBB_REQUIRE(ctx, nb < (1ll << 31), "obb_adamw_step: n too large");
    hipLaunchKernelGGL(k_adamw_step, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)s, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step_size,
                       sqrt_bc2);
    OBB_HIP(ctx, hipGetLastError());
    return OBB_OK;
}

}  // extern "C"
