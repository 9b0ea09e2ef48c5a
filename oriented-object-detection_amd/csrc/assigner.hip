// Second slice of the training step (`model.train(...)`, Train_OBB.py:796-841 -> ultralytics==8.3.196
// v8OBBLoss.__call__ -> RotatedTaskAlignedAssigner(topk = 10, alpha = 0.5, beta = 6.0)): which anchors learn which ground-truth box.
//
//   for every image b and ground-truth box g (valid where mask_gt):
//     in_gts[a]   anchor centre inside the rotated box (projections on two adjacent sides, inclusive)
//     overlap[a]  clamp(probiou(gt_g, pred_a), 0) where in_gts, else 0
//     metric[a]   score[b, a, label_g] ^ alpha * overlap[a] ^ beta
//     positives   the topk anchors by metric that are also in_gts
//   an anchor claimed by several boxes goes to the one with the largest overlap (first on ties, like torch.argmax);
//   target_scores[b, a, label] = metric[a] * max_a'(overlap of g's positives) / (max_a'(metric of g's positives) + eps)
//
// Kernels (wavefront primitives, no host-visible intermediate):
//   k_assign_rows    one 256-thread workgroup per (image, box): in_gts + ProbIoU + metric for every anchor (HBM streaming: 20 + 4 B in,
//                    4 + 4 + 1 B out per pair), then ten rounds of a workgroup-wide arg-max (lane-local maxima, DPP-free shuffles across
//                    the wave, 4 partials in LDS) mark the topk row entries
//   k_assign_resolve one thread per (image, anchor): claims summed over the boxes, arg-max of the overlaps where there are several,
//                    the anchor's target box / label / gt index, the final positive mask
//   k_assign_rowmax  one workgroup per (image, box): maxima of metric and overlap over the box's final positives
//   k_assign_scores  one thread per (image, anchor): the normalised target score row
// Ties in the topk can only occur at metric 0 (score > 0 always; overlap^6 underflows below ~1e-7), where the positive carries zero
// weight in every loss term: they are broken towards the lower anchor index.
#include "ctx.h"

namespace obb {

__device__ __forceinline__ void cov3(float w, float h, float t, float &A, float &B, float &C) {  // _get_covariance_matrix
    const float a = w * w / 12.0f, b = h * h / 12.0f, c = cosf(t), s = sinf(t);
    A = a * c * c + b * s * s; B = a * s * s + b * c * c; C = (a - b) * c * s;
}

// ultralytics.utils.metrics.probiou (CIoU = False), operation order of the torch expression
__device__ __forceinline__ float probiou_f(float x1, float y1, float A1, float B1, float C1, float x2, float y2, float A2, float B2, float C2) {
    const float eps = 1e-7f;
    const float A = A1 + A2, B = B1 + B2, C = C1 + C2;
    const float den = A * B - C * C + eps;
    const float t1 = ((A * (y1 - y2) * (y1 - y2) + B * (x1 - x2) * (x1 - x2)) / den) * 0.25f;
    const float t2 = ((C * (x2 - x1) * (y1 - y2)) / den) * 0.5f;
    const float d1 = fmaxf(A1 * B1 - C1 * C1, 0.0f), d2 = fmaxf(A2 * B2 - C2 * C2, 0.0f);
    const float t3 = logf((A * B - C * C) / (4.0f * sqrtf(d1 * d2) + eps) + eps) * 0.5f;
    const float bd = fminf(fmaxf(t1 + t2 + t3, eps), 100.0f);
    const float hd = sqrtf(1.0f - expf(-bd) + eps);
    return 1.0f - hd;
}

__global__ __launch_bounds__(256) void k_assign_rows(const float *__restrict__ pd_scores, const float *__restrict__ pd_bboxes, const float *__restrict__ anc,
                                                    const int32_t *__restrict__ gt_labels, const float *__restrict__ gt_bboxes, const uint8_t *__restrict__ mask_gt,
                                                    int na, int nc, int nmax, int topk, float alpha, float beta, float *__restrict__ overlaps,
                                                    float *__restrict__ metric, uint8_t *__restrict__ mask_pos) {
    __shared__ float swv[4];
    __shared__ int swi[4];
    __shared__ int win_s;
    const int row = blockIdx.x;  // b * nmax + g
    const int b = row / nmax;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *ov = overlaps + (int64_t)row * na, *me = metric + (int64_t)row * na;
    uint8_t *mp = mask_pos + (int64_t)row * na;
    const bool valid = mask_gt[row] != 0;
    const float *gb = gt_bboxes + (int64_t)row * 5;
    const float gx = gb[0], gy = gb[1], gw = gb[2], gh = gb[3], gt = gb[4];
    // xywhr2xyxyxyxy corners a, b, d (torch: vec1 = (w/2 cos, w/2 sin), vec2 = (-h/2 sin, h/2 cos); pt1 = ctr + vec1 + vec2, pt2 = ctr + vec1 - vec2, pt4 = ctr - vec1 + vec2)
    const float cs = cosf(gt), sn = sinf(gt);
    const float v1x = gw / 2.0f * cs, v1y = gw / 2.0f * sn, v2x = -gh / 2.0f * sn, v2y = gh / 2.0f * cs;
    const float ax = gx + v1x + v2x, ay = gy + v1y + v2y, bx = gx + v1x - v2x, by = gy + v1y - v2y, dx_ = gx - v1x + v2x, dy_ = gy - v1y + v2y;
    const float abx = bx - ax, aby = by - ay, adx = dx_ - ax, ady = dy_ - ay;
    const float nab = abx * abx + aby * aby, nad = adx * adx + ady * ady;
    float A1, B1, C1;
    cov3(gw, gh, gt, A1, B1, C1);
    int label = gt_labels[row];
    label = label < 0 ? 0 : (label >= nc ? nc - 1 : label);
    for (int a = tid; a < na; a += 256) {
        const float px = anc[a * 2], py = anc[a * 2 + 1];
        const float apx = px - ax, apy = py - ay;
        const float dab = apx * abx + apy * aby, dad = apx * adx + apy * ady;
        const bool in = valid && dab >= 0.0f && dab <= nab && dad >= 0.0f && dad <= nad;
        float o = 0.0f, m = 0.0f;
        if (in) {
            const float *pb = pd_bboxes + ((int64_t)b * na + a) * 5;
            float A2, B2, C2;
            cov3(pb[2], pb[3], pb[4], A2, B2, C2);
            o = fmaxf(probiou_f(gx, gy, A1, B1, C1, pb[0], pb[1], A2, B2, C2), 0.0f);
            const float s = pd_scores[((int64_t)b * na + a) * nc + label];
            m = powf(s, alpha) * powf(o, beta);
        }
        ov[a] = o; me[a] = m;
        mp[a] = in ? 2 : 0;  // bit 1: in_gts (& valid); bit 0 is set by the topk rounds below
    }
    constexpr int PER = 40;  // anchors per thread held in registers for the topk rounds (na <= 256 * PER); a thread re-reads its OWN stores
    float mv[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) { const int a = tid + k * 256; mv[k] = a < na ? me[a] : -1.0f; }
    // topk: `topk` rounds of a workgroup-wide arg-max over the not yet taken entries (largest value, lowest anchor index on ties)
    unsigned long long taken = 0ull;
    for (int r = 0; r < topk; ++r) {
        float best = -1.0f;
        int bi = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int a = tid + k * 256;
            if (a < na && !((taken >> k) & 1ull) && (mv[k] > best)) { best = mv[k]; bi = a; }  // ascending a: the first maximum stays
        }
        for (int d = 32; d >= 1; d >>= 1) {
            const float ob = __shfl_xor(best, d);
            const int oi = __shfl_xor(bi, d);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) { swv[wave] = best; swi[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float bb = swv[0]; int ii = swi[0];
            for (int w = 1; w < 4; ++w)
                if (swv[w] > bb || (swv[w] == bb && swi[w] < ii)) { bb = swv[w]; ii = swi[w]; }
            win_s = ii;
        }
        __syncthreads();
        const int win = win_s;
        if (win != 0x7fffffff && (win & 255) == tid) {
            taken |= 1ull << (win >> 8);
            if (valid) mp[win] |= 1;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_assign_resolve(const float *__restrict__ overlaps, uint8_t *__restrict__ mask_pos, const int32_t *__restrict__ gt_labels,
                                                       const float *__restrict__ gt_bboxes, int bs, int na, int nmax, int32_t *__restrict__ target_labels,
                                                       float *__restrict__ target_bboxes, uint8_t *__restrict__ fg_mask, int32_t *__restrict__ target_gt_idx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)bs * na) return;
    const int b = (int)(i / na), a = (int)(i % na);
    int claims = 0, first = -1, amax = 0;
    float omax = -1.0f;
    for (int g = 0; g < nmax; ++g) {
        const int64_t e = ((int64_t)b * nmax + g) * na + a;
        const bool pos = mask_pos[e] == 3;  // topk & in_gts (& valid)
        if (pos) { ++claims; if (first < 0) first = g; }
        const float o = overlaps[e];
        if (o > omax) { omax = o; amax = g; }  // first maximal value, like torch.argmax
    }
    const int tg = claims > 1 ? amax : (claims == 1 ? first : 0);
    for (int g = 0; g < nmax; ++g) {  // final positive mask as bit 2
        const int64_t e = ((int64_t)b * nmax + g) * na + a;
        const bool pos = claims > 1 ? (g == amax) : (mask_pos[e] == 3);
        mask_pos[e] = (uint8_t)((mask_pos[e] & 3) | (pos ? 4 : 0));
    }
    fg_mask[i] = (uint8_t)(claims > 0);
    target_gt_idx[i] = tg;
    int lab = gt_labels[b * nmax + tg];
    target_labels[i] = lab < 0 ? 0 : lab;
    for (int k = 0; k < 5; ++k) target_bboxes[i * 5 + k] = gt_bboxes[((int64_t)b * nmax + tg) * 5 + k];
}

__global__ __launch_bounds__(256) void k_assign_rowmax(const float *__restrict__ overlaps, const float *__restrict__ metric, const uint8_t *__restrict__ mask_pos, int na,
                                                      float *__restrict__ pos_metric, float *__restrict__ pos_overlap) {
    __shared__ float sm[4], so[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float m = 0.0f, o = 0.0f;  // (align_metric * mask_pos).amax: zeros where not positive
    for (int a = tid; a < na; a += 256) {
        const int64_t e = (int64_t)row * na + a;
        if (mask_pos[e] & 4) { m = fmaxf(m, metric[e]); o = fmaxf(o, overlaps[e]); }
    }
    for (int d = 32; d >= 1; d >>= 1) { m = fmaxf(m, __shfl_xor(m, d)); o = fmaxf(o, __shfl_xor(o, d)); }
    if (lane == 0) { sm[wave] = m; so[wave] = o; }
    __syncthreads();
    if (tid == 0) {
        pos_metric[row] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
        pos_overlap[row] = fmaxf(fmaxf(so[0], so[1]), fmaxf(so[2], so[3]));
    }
}

__global__ __launch_bounds__(256) void k_assign_scores(const float *__restrict__ metric, const float *__restrict__ pos_metric, const float *__restrict__ pos_overlap,
                                                      const uint8_t *__restrict__ fg_mask, const int32_t *__restrict__ target_gt_idx, const int32_t *__restrict__ target_labels,
                                                      int bs, int na, int nc, int nmax, float eps, float *__restrict__ target_scores) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)bs * na) return;
    const int b = (int)(i / na), a = (int)(i % na);
    float norm = 0.0f;
    int lab = -1;
    if (fg_mask[i]) {
        const int row = b * nmax + target_gt_idx[i];
        norm = metric[(int64_t)row * na + a] * pos_overlap[row] / (pos_metric[row] + eps);
        lab = target_labels[i];
    }
    for (int c = 0; c < nc; ++c) target_scores[i * nc + c] = (c == lab) ? norm : 0.0f;
}

}  // namespace obb

using namespace obb;

extern "C" int obb_rotated_tal_assign(obb_ctx *ctx, const float *pd_scores, const float *pd_bboxes, const float *anc_points, const int32_t *gt_labels,
                                      const float *gt_bboxes, const uint8_t *mask_gt, int32_t bs, int32_t na, int32_t nc, int32_t n_max, int32_t topk, float alpha,
                                      float beta, int32_t *target_labels, float *target_bboxes, float *target_scores, uint8_t *fg_mask, int32_t *target_gt_idx,
                                      obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && bs >= 0 && na >= 0 && nc >= 1 && n_max >= 0 && topk >= 1, "obb_rotated_tal_assign: bad arguments");
    OBB_REQUIRE(ctx, na <= 256 * 40, "obb_rotated_tal_assign: more than 10240 anchors per image");
    if (bs == 0 || na == 0) return OBB_OK;
    OBB_REQUIRE(ctx, target_labels && target_bboxes && target_scores && fg_mask && target_gt_idx, "obb_rotated_tal_assign: NULL output");
    hipStream_t st = (hipStream_t)s;
    const int64_t nba = (int64_t)bs * na;
    if (n_max == 0) {  // no ground truth in the batch: everything background (TaskAlignedAssigner.forward's early return)
        OBB_HIP(ctx, hipMemsetAsync(target_labels, 0, nba * 4, st));
        OBB_HIP(ctx, hipMemsetAsync(target_bboxes, 0, nba * 20, st));
        OBB_HIP(ctx, hipMemsetAsync(target_scores, 0, nba * nc * 4, st));
        OBB_HIP(ctx, hipMemsetAsync(fg_mask, 0, nba, st));
        OBB_HIP(ctx, hipMemsetAsync(target_gt_idx, 0, nba * 4, st));
        return OBB_OK;
    }
    OBB_REQUIRE(ctx, pd_scores && pd_bboxes && anc_points && gt_labels && gt_bboxes && mask_gt, "obb_rotated_tal_assign: NULL input");
    const int64_t pairs = (int64_t)bs * n_max * na;
    OBB_REQUIRE(ctx, (int64_t)bs * n_max < (1ll << 31) && cdiv(nba, 256) < (1ll << 31), "obb_rotated_tal_assign: problem too large");
    float *ov = (float *)ctx->workspace(WS_GEOM_A, pairs * 4), *me = (float *)ctx->workspace(WS_GEOM_B, pairs * 4);
    uint8_t *mp = (uint8_t *)ctx->workspace(WS_GEOM_C, pairs);
    float *rowm = (float *)ctx->workspace(WS_GEOM_D, (size_t)bs * n_max * 8);
    if (!ov || !me || !mp || !rowm) return set_error(ctx, OBB_ERR_HIP, "obb_rotated_tal_assign: workspace allocation failed");
    float *rowo = rowm + (size_t)bs * n_max;
    hipLaunchKernelGGL(k_assign_rows, dim3((unsigned)(bs * n_max)), dim3(256), 0, st, pd_scores, pd_bboxes, anc_points, gt_labels, gt_bboxes, mask_gt, (int)na, (int)nc,
                       (int)n_max, (int)topk, alpha, beta, ov, me, mp);
    hipLaunchKernelGGL(k_assign_resolve, dim3((unsigned)cdiv(nba, 256)), dim3(256), 0, st, ov, mp, gt_labels, gt_bboxes, (int)bs, (int)na, (int)n_max, target_labels,
                       target_bboxes, fg_mask, target_gt_idx);
    hipLaunchKernelGGL(k_assign_rowmax, dim3((unsigned)(bs * n_max)), dim3(256), 0, st, ov, me, mp, (int)na, rowm, rowo);
    hipLaunchKernelGGL(k_assign_scores, dim3((unsigned)cdiv(nba, 256)), dim3(256), 0, st, me, rowm, rowo, fg_mask, target_gt_idx, target_labels, (int)bs, (int)na, (int)nc,
                       (int)n_max, 1e-9f, target_scores);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}
