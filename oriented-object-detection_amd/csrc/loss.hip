// First slice of the training step (`model.train(...)`, Train_OBB.py:796-841 -> ultralytics==8.3.196
// v8OBBLoss -> RotatedBboxLoss): the ProbIoU rotated-box loss, forward AND backward in one kernel.
//
//   loss = sum_i (1 - probiou(pred_i, target_i)) * weight_i / target_scores_sum          
//   probiou: Gaussian (Bhattacharyya / Hellinger) overlap of the two boxes' covariance ellipses (Appendix A4, arXiv:2106.06072)
//
// One thread per matched (prediction, target) pair: the forward value and the closed-form gradient with respect to the prediction's
// (x, y, w, h, theta) come out of the same registers, so the pair's 40 + 4 bytes are read once and 4 + 20 bytes written -- an
// HBM-streaming kernel (68 B per pair).  The per-pair losses are summed by a second, tree-ordered launch (deterministic; no float
// atomics).  Gradients match torch.autograd on the same formula (tests/test_gpu_loss.py); clamps pass gradient inside their range and
// block it outside, like torch.clamp.
#include "ctx.h"

namespace obb {

__global__ __launch_bounds__(256) void k_probiou_loss(const float *__restrict__ pred, const float *__restrict__ target, const float *__restrict__ weight,
                                                     int64_t n, float inv_tss, float *__restrict__ loss_elem, float *__restrict__ grad) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float eps = 1e-7f;
    const float x1 = pred[i * 5], y1 = pred[i * 5 + 1], w1 = pred[i * 5 + 2], h1 = pred[i * 5 + 3], t1a = pred[i * 5 + 4];
    const float x2 = target[i * 5], y2 = target[i * 5 + 1], w2 = target[i * 5 + 2], h2 = target[i * 5 + 3], t2a = target[i * 5 + 4];
    const float wt = weight ? weight[i] : 1.0f;
    // covariance terms (_get_covariance_matrix)
    const float a1 = w1 * w1 / 12.0f, b1 = h1 * h1 / 12.0f, c1 = cosf(t1a), s1 = sinf(t1a);
    const float a2 = w2 * w2 / 12.0f, b2 = h2 * h2 / 12.0f, c2 = cosf(t2a), s2 = sinf(t2a);
    const float A1 = a1 * c1 * c1 + b1 * s1 * s1, B1 = a1 * s1 * s1 + b1 * c1 * c1, C1 = (a1 - b1) * c1 * s1;
    const float A2 = a2 * c2 * c2 + b2 * s2 * s2, B2 = a2 * s2 * s2 + b2 * c2 * c2, C2 = (a2 - b2) * c2 * s2;
    const float A = A1 + A2, B = B1 + B2, C = C1 + C2;
    const float dx = x1 - x2, dy = y1 - y2;
    const float D = A * B - C * C, den = D + eps;
    const float N1 = A * dy * dy + B * dx * dx, N2 = C * dx * dy;
    const float T1 = 0.25f * N1 / den, T2 = -0.5f * N2 / den;
    const float d1raw = A1 * B1 - C1 * C1, d2raw = A2 * B2 - C2 * C2;
    const float d1 = fmaxf(d1raw, 0.0f), d2 = fmaxf(d2raw, 0.0f);
    const float r = sqrtf(d1 * d2), g = 4.0f * r + eps;
    const float u = D / g;
    const float T3 = 0.5f * logf(u + eps);
    const float braw = T1 + T2 + T3;
    const float bd = fminf(fmaxf(braw, eps), 100.0f);
    const float ex = expf(-bd);
    const float hd = sqrtf(1.0f - ex + eps);
    loss_elem[i] = hd * wt;  // (1 - iou) * weight with iou = 1 - hd
    // ---- backward: dL/dbd, then the chain down to (x, y, w, h, theta) of the prediction
    float gb = (braw >= eps && braw <= 100.0f) ? wt * inv_tss * ex / (2.0f * hd) : 0.0f;
    const float iden = 1.0f / den, iden2 = iden * iden;
    // d(bd) / d(A, B, C, dx, dy)
    float gA = 0.25f * (dy * dy * iden - N1 * B * iden2) + 0.5f * N2 * B * iden2;
    float gB = 0.25f * (dx * dx * iden - N1 * A * iden2) + 0.5f * N2 * A * iden2;
    float gC = 0.5f * N1 * C * iden2 - 0.5f * dx * dy * iden - N2 * C * iden2;
    const float gdx = 0.5f * B * dx * iden - 0.5f * C * dy * iden;
    const float gdy = 0.5f * A * dy * iden - 0.5f * C * dx * iden;
    const float k3 = 0.5f / (u + eps);
    const float gD = k3 / g, gg = -k3 * D / (g * g);
    gA += gD * B; gB += gD * A; gC += gD * (-2.0f * C);
    // g = 4 sqrt(d1 d2) + eps: only det1 belongs to the prediction
    const float gdet1 = (d1raw >= 0.0f && r > 0.0f) ? gg * 2.0f * d2 / r : 0.0f;
    const float gA1 = gA + gdet1 * B1, gB1 = gB + gdet1 * A1, gC1 = gC + gdet1 * (-2.0f * C1);
    const float cs = c1 * s1;
    const float ga = gA1 * c1 * c1 + gB1 * s1 * s1 + gC1 * cs;
    const float gbb = gA1 * s1 * s1 + gB1 * c1 * c1 - gC1 * cs;
    const float gt = gA1 * (-2.0f * C1) + gB1 * (2.0f * C1) + gC1 * (a1 - b1) * (c1 * c1 - s1 * s1);
    float *go = grad + i * 5;
    go[0] = gb * gdx; go[1] = gb * gdy; go[2] = gb * ga * (w1 / 6.0f); go[3] = gb * gbb * (h1 / 6.0f); go[4] = gb * gt;
}

// deterministic sum: each block reduces 4096 consecutive values in a fixed tree, a second pass (same kernel, one block) reduces the partials
__global__ __launch_bounds__(256) void k_sum_f32(const float *__restrict__ v, int64_t n, float scale, float *__restrict__ out) {
    __shared__ double sh[256];
    const int64_t base = (int64_t)blockIdx.x * 4096;
    double acc = 0.0;
    for (int k = threadIdx.x; k < 4096; k += 256)
        if (base + k < n) acc += (double)v[base + k];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = (float)(sh[0] * (double)scale);
}


// ---- DFL (distribution focal loss) of the box branch: per (box, side) a 16-way distribution over integer distances; the target distance
// t in [0, reg_max - 1) is shared out between its two neighbouring bins:  l = CE(logits, floor t) * (ceil' - t) + CE(logits, floor t + 1) *
// (t - floor t); per box the mean over the 4 sides, times the box weight, over target_scores_sum (ultralytics DFLoss + RotatedBboxLoss).
// One thread per (box, side): 16 logits in, 16 gradients out: d l / d logit_k = softmax_k - (wl [k = tl] + wr [k = tr]).
template <int R>
__global__ __launch_bounds__(256) void k_dfl_loss(const float *__restrict__ logits, const float *__restrict__ target, const float *__restrict__ weight,
                                                 int64_t n4, float inv_tss, float *__restrict__ loss_elem, float *__restrict__ grad) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (box, side)
    if (i >= n4) return;
    float x[R];
#pragma unroll
    for (int k = 0; k < R; k += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(logits + i * R + k);
        x[k] = v.x; x[k + 1] = v.y; x[k + 2] = v.z; x[k + 3] = v.w;
    }
    float m = x[0];
#pragma unroll
    for (int k = 1; k < R; ++k) m = fmaxf(m, x[k]);
    float e[R], se = 0.f;
#pragma unroll
    for (int k = 0; k < R; ++k) { e[k] = expf(x[k] - m); se += e[k]; }
    const float lse = m + logf(se);
    const float t = fminf(fmaxf(target[i], 0.0f), (float)(R - 1) - 0.01f);
    const int tl = (int)t, tr = tl + 1;
    const float wl = (float)tr - t, wr = 1.0f - wl;
    float xl = 0.f, xr = 0.f;
#pragma unroll
    for (int k = 0; k < R; ++k) { xl = k == tl ? x[k] : xl; xr = k == tr ? x[k] : xr; }
    const float wt = (weight ? weight[i >> 2] : 1.0f) * 0.25f;
    loss_elem[i] = ((lse - xl) * wl + (lse - xr) * wr) * wt;
    const float gs = wt * inv_tss, inv_se = 1.0f / se;
#pragma unroll
    for (int k = 0; k < R; k += 4) {
        float g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = (e[k + j] * inv_se - (k + j == tl ? wl : 0.f) - (k + j == tr ? wr : 0.f)) * gs;
        *reinterpret_cast<float4 *>(grad + i * R + k) = make_float4(g[0], g[1], g[2], g[3]);
    }
}

// ---- classification term: BCEWithLogitsLoss(reduction="none")(logits, targets).sum() / target_scores_sum (v8OBBLoss), element-wise:
// l = max(x, 0) - x t + log(1 + exp(-|x|)),  d l / d x = sigmoid(x) - t
__global__ __launch_bounds__(256) void k_bce_loss(const float *__restrict__ logits, const float *__restrict__ target, int64_t n, float inv_tss,
                                                 float *__restrict__ loss_elem, float *__restrict__ grad) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = logits[i], t = target[i];
    const float ea = expf(-fabsf(x));
    loss_elem[i] = fmaxf(x, 0.0f) - x * t + log1pf(ea);
    const float sig = x >= 0.0f ? 1.0f / (1.0f + ea) : ea / (1.0f + ea);
    grad[i] = (sig - t) * inv_tss;
}

}  // namespace obb

using namespace obb;

// deterministic sum of elem[0..n) * scale -> *loss (tree of k_sum_f32 launches)
static int sum_to_scalar(obb_ctx *ctx, const float *elem, int64_t n, float scale, float *loss, hipStream_t st, const char *who) {
    const int64_t nb1 = cdiv(n, 4096), nb2 = cdiv(nb1, 4096);
    OBB_REQUIRE(ctx, nb2 <= 4096, "%s: n too large", who);
    float *part = (float *)ctx->workspace(WS_GEOM_B, sizeof(float) * (size_t)(nb1 + nb2 + 8));
    if (!part) return set_error(ctx, OBB_ERR_HIP, "%s: workspace allocation failed", who);
    if (nb1 == 1) {
        hipLaunchKernelGGL(k_sum_f32, dim3(1), dim3(256), 0, st, elem, n, scale, loss);
    } else {
        hipLaunchKernelGGL(k_sum_f32, dim3((unsigned)nb1), dim3(256), 0, st, elem, n, 1.0f, part);
        if (nb2 == 1) hipLaunchKernelGGL(k_sum_f32, dim3(1), dim3(256), 0, st, (const float *)part, nb1, scale, loss);
        else {
            hipLaunchKernelGGL(k_sum_f32, dim3((unsigned)nb2), dim3(256), 0, st, (const float *)part, nb1, 1.0f, part + nb1);
            hipLaunchKernelGGL(k_sum_f32, dim3(1), dim3(256), 0, st, (const float *)(part + nb1), nb2, scale, loss);
        }
    }
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

extern "C" int obb_dfl_loss(obb_ctx *ctx, const float *pred_dist, const float *target_ltrb, const float *weight, int64_t n, int32_t reg_max,
                            float target_scores_sum, float *loss, float *grad_pred, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && n >= 0 && loss, "obb_dfl_loss: bad arguments");
    hipStream_t st = (hipStream_t)s;
    if (n == 0) { OBB_HIP(ctx, hipMemsetAsync(loss, 0, sizeof(float), st)); return OBB_OK; }
    OBB_REQUIRE(ctx, pred_dist && target_ltrb && grad_pred, "obb_dfl_loss: NULL buffer");
    OBB_REQUIRE(ctx, reg_max == 16, "obb_dfl_loss: reg_max %d unsupported (the OBB head uses 16 bins)", reg_max);
    OBB_REQUIRE(ctx, target_scores_sum > 0.0f, "obb_dfl_loss: target_scores_sum must be positive");
    OBB_REQUIRE(ctx, (((uintptr_t)pred_dist | (uintptr_t)grad_pred) & 15) == 0, "obb_dfl_loss: logits and gradient rows must be 16-byte aligned");
    const int64_t n4 = n * 4;
    float *elem = (float *)ctx->workspace(WS_GEOM_A, sizeof(float) * (size_t)n4);
    if (!elem) return set_error(ctx, OBB_ERR_HIP, "obb_dfl_loss: workspace allocation failed");
    hipLaunchKernelGGL(k_dfl_loss<16>, dim3((unsigned)cdiv(n4, 256)), dim3(256), 0, st, pred_dist, target_ltrb, weight, n4, 1.0f / target_scores_sum, elem, grad_pred);
    OBB_LAUNCH_CHECK(ctx);
    return sum_to_scalar(ctx, elem, n4, 1.0f / target_scores_sum, loss, st, "obb_dfl_loss");
}

extern "C" int obb_bce_loss(obb_ctx *ctx, const float *logits, const float *targets, int64_t n, float target_scores_sum, float *loss, float *grad_logits,
                            obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && n >= 0 && loss, "obb_bce_loss: bad arguments");
    hipStream_t st = (hipStream_t)s;
    if (n == 0) { OBB_HIP(ctx, hipMemsetAsync(loss, 0, sizeof(float), st)); return OBB_OK; }
    OBB_REQUIRE(ctx, logits && targets && grad_logits, "obb_bce_loss: NULL buffer");
    OBB_REQUIRE(ctx, target_scores_sum > 0.0f, "obb_bce_loss: target_scores_sum must be positive");
    float *elem = (float *)ctx->workspace(WS_GEOM_A, sizeof(float) * (size_t)n);
    if (!elem) return set_error(ctx, OBB_ERR_HIP, "obb_bce_loss: workspace allocation failed");
    hipLaunchKernelGGL(k_bce_loss, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, logits, targets, n, 1.0f / target_scores_sum, elem, grad_logits);
    OBB_LAUNCH_CHECK(ctx);
    return sum_to_scalar(ctx, elem, n, 1.0f / target_scores_sum, loss, st, "obb_bce_loss");
}

extern "C" int obb_probiou_loss(obb_ctx *ctx, const float *pred, const float *target, const float *weight, int64_t n, float target_scores_sum,
                                float *loss, float *grad_pred, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && n >= 0 && loss, "obb_probiou_loss: bad arguments");
    hipStream_t st = (hipStream_t)s;
    if (n == 0) { OBB_HIP(ctx, hipMemsetAsync(loss, 0, sizeof(float), st)); return OBB_OK; }
    OBB_REQUIRE(ctx, pred && target && grad_pred, "obb_probiou_loss: NULL buffer");
    OBB_REQUIRE(ctx, target_scores_sum > 0.0f, "obb_probiou_loss: target_scores_sum must be positive");
    const int64_t nb1 = cdiv(n, 4096), nb2 = cdiv(nb1, 4096);
    OBB_REQUIRE(ctx, nb2 <= 4096, "obb_probiou_loss: n too large");
    float *elem = (float *)ctx->workspace(WS_GEOM_A, sizeof(float) * (size_t)n);
    float *part = (float *)ctx->workspace(WS_GEOM_B, sizeof(float) * (size_t)(nb1 + nb2 + 8));
    if (!elem || !part) return set_error(ctx, OBB_ERR_HIP, "obb_probiou_loss: workspace allocation failed");
    hipLaunchKernelGGL(k_probiou_loss, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, pred, target, weight, n, 1.0f / target_scores_sum, elem, grad_pred);
    OBB_LAUNCH_CHECK(ctx);
    const float inv = 1.0f / target_scores_sum;
    if (nb1 == 1) {
        hipLaunchKernelGGL(k_sum_f32, dim3(1), dim3(256), 0, st, (const float *)elem, n, inv, loss);
    } else {
        hipLaunchKernelGGL(k_sum_f32, dim3((unsigned)nb1), dim3(256), 0, st, (const float *)elem, n, 1.0f, part);
        if (nb2 == 1) hipLaunchKernelGGL(k_sum_f32, dim3(1), dim3(256), 0, st, (const float *)part, nb1, inv, loss);
        else {
            hipLaunchKernelGGL(k_sum_f32, dim3((unsigned)nb2), dim3(256), 0, st, (const float *)part, nb1, 1.0f, part + nb1);
            hipLaunchKernelGGL(k_sum_f32, dim3(1), dim3(256), 0, st, (const float *)(part + nb1), nb2, inv, loss);
        }
    }
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}
