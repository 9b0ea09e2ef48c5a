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

}  // namespace obb

using namespace obb;

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
