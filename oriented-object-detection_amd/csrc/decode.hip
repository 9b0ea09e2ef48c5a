// Everything the Ultralytics OBB predictor does around the network (call site Detect_OBB.py:81-83, results consumed at
// :228-231), restated as wave64 HIP kernels in fp32 with torch's operation order (built with -ffp-contract=off):
//   k_decode        OBB head _inference: DFL softmax-expectation, dist2rbox, angle = (sigmoid-0.25)*pi, class sigmoid
//   k_nms_tile      non_max_suppression(rotated=True): conf filter, (conf, cls) = max over classes, class offset 7680,
//                   stable score sort, ProbIoU Fast-NMS (keep j iff no higher-scored i with probiou >= thr), max_det
//   k_results       construct_result: regularize_rboxes, scale_boxes(xywh=True), xywhr2xyxyxyxy
//   k_gather_tiles / k_letterbox   the tiler's crop (Detect_OBB.py:218-220) and LetterBox(auto=True) preprocess
// SURVEY.md Appendix A2/A4/A6.  These are HBM/latency-bound kernels: one workgroup per tile, candidates kept in
// anchor order by ballot/prefix-sum compaction so that ties sort exactly like a stable argsort.
#include <algorithm>
#include <mutex>
#include <cmath>
#include <cstdlib>

#include "ctx.h"
#include "post_device.h"

namespace obb {

static constexpr int kRegMaxD = 16;
static constexpr float kMaxWh = 7680.0f;

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---------------------------------------------------------------------------------------------- decode
// box (x, y, w, h in letterboxed-input pixels) and angle of anchor `a` from its head row: DFL softmax-expectation, dist2rbox, angle
// (one definition for the full decode and for the candidate-first path: identical arithmetic by construction)
__device__ __forceinline__ void decode_anchor(const float *__restrict__ hp, int a, int nc, int h, int w, float &ox, float &oy, float &ow, float &oh, float &oang) {
    // anchor of this row: levels P3, P4, P5 concatenated, row-major inside a level, centres at +0.5
    int n8 = (h / 8) * (w / 8), n16 = (h / 16) * (w / 16);
    int stride, lw, la;
    if (a < n8) { stride = 8; lw = w / 8; la = a; }
    else if (a < n8 + n16) { stride = 16; lw = w / 16; la = a - n8; }
    else { stride = 32; lw = w / 32; la = a - n8 - n16; }
    float ax = (float)(la % lw) + 0.5f, ay = (float)(la / lw) + 0.5f;
    float d[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float x[kRegMaxD];
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < kRegMaxD; ++k) { x[k] = hp[s * kRegMaxD + k]; m = fmaxf(m, x[k]); }
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < kRegMaxD; ++k) { x[k] = expf(x[k] - m); sum += x[k]; }
        float e = 0.f;
#pragma unroll
        for (int k = 0; k < kRegMaxD; ++k) e += (x[k] / sum) * (float)k;
        d[s] = e;
    }
    float ang = (sigmoid_f(hp[4 * kRegMaxD + nc]) - 0.25f) * kPiF;
    float c = cosf(ang), s = sinf(ang);
    float xf = (d[2] - d[0]) / 2.0f, yf = (d[3] - d[1]) / 2.0f;
    float x = xf * c - yf * s, y = xf * s + yf * c;
    float fs = (float)stride;
    ox = (x + ax) * fs;
    oy = (y + ay) * fs;
    ow = (d[0] + d[2]) * fs;
    oh = (d[1] + d[3]) * fs;
    oang = ang;
}

// full decode of every anchor (parity tap obb_decode, and the fall-back of tiles with more candidates than the fast path holds:
// `only_flagged` != nullptr restricts the work to tiles whose count is -1)
__global__ __launch_bounds__(256) void k_decode(const float *__restrict__ head, int B, int A, int nc, int h, int w,
                                               float *__restrict__ pred, const int32_t *__restrict__ only_flagged) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * A) return;
    if (only_flagged && only_flagged[i / A] != -1) return;
    int a = (int)(i % A);
    const int no = (4 * kRegMaxD + nc + 1 + 3) / 4 * 4, np = 4 + nc + 1;  // head rows are padded to a multiple of 4 floats
    const float *hp = head + i * no;
    float *pp = pred + i * np;
    float ang;
    decode_anchor(hp, a, nc, h, w, pp[0], pp[1], pp[2], pp[3], ang);
    for (int k = 0; k < nc; ++k) pp[4 + k] = sigmoid_f(hp[4 * kRegMaxD + k]);
    pp[4 + nc] = ang;
}

// ---------------------------------------------------------------------------------------------- ProbIoU
struct RBox { float x, y, A, B, C, det, tr; };  // offset centre + covariance terms (batch_probiou / _get_covariance_matrix); tr = A + B

__device__ __forceinline__ RBox make_rbox(float x, float y, float w, float h, float t) {
    RBox r;
    float a = w * w / 12.0f, b = h * h / 12.0f;
    float c = cosf(t), s = sinf(t);
    float c2 = c * c, s2 = s * s;
    r.x = x; r.y = y;
    r.A = a * c2 + b * s2;
    r.B = a * s2 + b * c2;
    r.C = (a - b) * c * s;
    float dd = r.A * r.B - r.C * r.C;
    r.det = dd > 0.0f ? dd : 0.0f;
    r.tr = r.A + r.B;
    return r;
}

__device__ __forceinline__ float probiou(const RBox &p, const RBox &q) {
    const float eps = 1e-7f;
    float sa = p.A + q.A, sb = p.B + q.B, sc = p.C + q.C;
    float den = sa * sb - sc * sc;
    float dy = p.y - q.y, dx = p.x - q.x;
    float t1 = ((sa * (dy * dy) + sb * (dx * dx)) / (den + eps)) * 0.25f;
    float t2 = ((sc * (q.x - p.x) * dy) / (den + eps)) * 0.5f;
    float t3 = logf(den / (4.0f * sqrtf(p.det * q.det) + eps) + eps) * 0.5f;
    float bd = t1 + t2 + t3;
    bd = bd < eps ? eps : (bd > 100.0f ? 100.0f : bd);
    float hd = sqrtf(1.0f - expf(-bd) + eps);
    return 1.0f - hd;
}

// Cheap decision for pairs that are nowhere near the threshold: the same Bhattacharyya distance evaluated with the hardware
// reciprocal / log / sqrt (relative error ~1e-6).  probiou >= thr  <=>  bd <= BDmax (monotone), so bd_fast outside a guard band
// around BDmax decides the pair; anything inside the band (|bd_fast - BDmax| <= 1e-4 + 1e-3 * BDmax) is re-evaluated exactly.
// returns +1 hit, -1 miss, 0 undecided
__device__ __forceinline__ int probiou_fast_decision(const RBox &p, const RBox &q, float bdmax) {
    const float eps = 1e-7f;
    float sa = p.A + q.A, sb = p.B + q.B, sc = p.C + q.C;
    float den = sa * sb - sc * sc;
    float dy = p.y - q.y, dx = p.x - q.x;
    float inv = __frcp_rn(den + eps);
    float t1 = (sa * (dy * dy) + sb * (dx * dx)) * inv * 0.25f;
    float t2 = (sc * (q.x - p.x) * dy) * inv * 0.5f;
    float t3 = __logf(den * __frcp_rn(4.0f * __fsqrt_rn(p.det * q.det) + eps) + eps) * 0.5f;
    float bd = t1 + t2 + t3;
    float band = 1e-4f + 1e-3f * bdmax;
    if (!(bd == bd)) return 0;  // NaN: let the exact path decide
    if (bd > bdmax + band) return -1;
    if (bd < bdmax - band) return 1;
    return 0;
}

static float probiou_bdmax(float thr) {
    double s = 1.0 + 1e-7 - (1.0 - (double)thr) * (1.0 - (double)thr);
    if (!(thr > 1e-3f) || s <= 0.0 || s >= 1.0) return -1.0f;  // disables the fast path
    return (float)(-log(s));
}

// Conservative far-apart test.  probiou >= thr  <=>  bd <= BDmax = -log(1 + eps - (1 - thr)^2), and
// bd >= t1 + t2 = d^T (S1 + S2)^-1 d / 4 >= |d|^2 / (4 trace(S1 + S2))  (t3 >= 0 by AM-GM on the determinants).
// So |d|^2 > kq * (tr1 + tr2) with kq = 4 * BDmax * 1.05 proves probiou < thr without evaluating it; pairs anywhere near the
// threshold always take the exact path, so decisions are unchanged.
__device__ __forceinline__ bool far_apart(const RBox &p, const RBox &q, float kq) {
    float dx = p.x - q.x, dy = p.y - q.y;
    return dx * dx + dy * dy > kq * (p.tr + q.tr);
}

static float far_apart_factor(float thr) {
    double s = 1.0 + 1e-7 - (1.0 - (double)thr) * (1.0 - (double)thr);
    if (!(thr > 1e-3f) || s <= 0.0 || s >= 1.0) return INFINITY;  // never reject
    return (float)(4.0 * -log(s) * 1.05);
}

// block-wide ordered compaction helper: returns this thread's output slot (or -1) and advances *base by the chunk total
__device__ __forceinline__ int ordered_slot(bool flag, int *wave_tot /* LDS[16] */, int &base) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6;
    unsigned long long bal = __ballot(flag);
    int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(bal);
    __syncthreads();
    int off = 0, tot = 0;
    for (int k = 0; k < nw; ++k) { if (k < wave) off += wave_tot[k]; tot += wave_tot[k]; }
    int slot = flag ? base + off + before : -1;
    base += tot;
    __syncthreads();
    return slot;
}

struct NmsScratch {  // per tile, capacity A rows each
    int32_t *cand;   // anchor index of candidate k (anchor order)
    float *cscore;   // its confidence
    int32_t *ccls;
    int32_t *order;  // sorted position -> candidate k
    RBox *rb;        // sorted order
    uint8_t *keep;   // sorted order
};

// One workgroup (1024 threads) per tile.  LDS_RESIDENT: every per-candidate array lives in LDS (39 B per anchor,
// 138 KB for the 3549 anchors of a 416-px tile) so the O(n^2) rank sort and pair loop never leave the CU; tiles with
// more anchors than fit use the same code on global scratch.
template <bool LDS_RESIDENT>
__global__ __launch_bounds__(1024) void k_nms_tile(const float *__restrict__ pred, int A, int nc, float conf_thres, float iou_thres,
                                                  int max_det, int max_nms, float kq, float bdmax, NmsScratch S, float *__restrict__ out,
                                                  int32_t *__restrict__ count, int only_flagged) {
    extern __shared__ __attribute__((aligned(16))) char nms_smem[];
    __shared__ int wave_tot[16];
    const int b = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
    if (only_flagged && count[b] != -1) return;  // fall-back launch: only the tiles the candidate-first kernel handed over
    const int np = 4 + nc + 1;
    const float *pb = pred + (int64_t)b * A * np;
    RBox *rb; float *cscore; int32_t *cand, *order; uint8_t *ccls, *scls, *keep;
    if constexpr (LDS_RESIDENT) {
        char *p = nms_smem;
        cscore = (float *)p; p += ((size_t)A * 4 + 15) / 16 * 16;
        rb = (RBox *)p; p += (size_t)A * sizeof(RBox);
        cand = (int32_t *)p; p += (size_t)A * 4;
        order = (int32_t *)p; p += (size_t)A * 4;
        ccls = (uint8_t *)p; p += A;
        scls = (uint8_t *)p; p += A;
        keep = (uint8_t *)p;
    } else {
        const int64_t AS = (A + 3) & ~3;  // scratch stride per tile (keeps the float4 score reads aligned)
        rb = S.rb + (int64_t)b * AS; cscore = S.cscore + (int64_t)b * AS; cand = S.cand + (int64_t)b * AS; order = S.order + (int64_t)b * AS;
        ccls = S.keep + (int64_t)(3 * (int64_t)b) * AS; scls = ccls + AS; keep = scls + AS;
    }

    // 1. candidates in anchor order: conf = max over classes (first maximum), conf > conf_thres
    int n = 0;
    for (int a0 = 0; a0 < A; a0 += NT) {
        int a = a0 + tid;
        float best = -INFINITY;
        int bj = 0;
        if (a < A) {
            const float *cp = pb + (int64_t)a * np + 4;
            for (int j = 0; j < nc; ++j) { float v = cp[j]; if (v > best) { best = v; bj = j; } }
        }
        bool flag = (a < A) && (best > conf_thres);
        int slot = ordered_slot(flag, wave_tot, n);
        if (flag) { cand[slot] = a; cscore[slot] = best; ccls[slot] = (uint8_t)bj; }
    }
    __syncthreads();
    if (n == 0) { if (tid == 0) count[b] = 0; return; }

    // 2. stable descending rank sort by confidence (torch.argsort(descending=True, stable) restated); n > max_nms keeps the top max_nms
    for (int i = tid; i < n; i += NT) {
        float si = cscore[i];
        int rank = 0;
        int n4 = n & ~3;
        for (int j = 0; j < n4; j += 4) {
            float4 sj = *reinterpret_cast<const float4 *>(cscore + j);
            rank += (sj.x > si) | ((sj.x == si) & (j < i));
            rank += (sj.y > si) | ((sj.y == si) & (j + 1 < i));
            rank += (sj.z > si) | ((sj.z == si) & (j + 2 < i));
            rank += (sj.w > si) | ((sj.w == si) & (j + 3 < i));
        }
        for (int j = n4; j < n; ++j) { float sj = cscore[j]; rank += (sj > si) | ((sj == si) & (j < i)); }
        order[rank] = i;
    }
    __syncthreads();
    if (n > max_nms) n = max_nms;
    for (int r = tid; r < n; r += NT) {
        int k = order[r];
        const float *pp = pb + (int64_t)cand[k] * np;
        float c = (float)ccls[k] * kMaxWh;  // class offset: boxes of different classes never overlap
        rb[r] = make_rbox(pp[0] + c, pp[1] + c, pp[2], pp[3], pp[4 + nc]);
        scls[r] = ccls[k];
    }
    __syncthreads();

    // 3. Fast-NMS: keep r iff no i < r with probiou(i, r) >= thr  (suppressed boxes still suppress).
    //    One wave per row r, lanes sweep the earlier boxes 64 at a time (coalesced LDS reads, balanced work, early exit per
    //    row); the class test and the far-apart bound reject almost every pair before the exact ProbIoU.
    const bool skip_other_cls = iou_thres > 1e-3f;  // offset boxes of another class have probiou ~ 0
    {
        const int lane = tid & 63, wave = tid >> 6, nwave = NT >> 6;
        for (int r = wave; r < n; r += nwave) {
            RBox q = rb[r];
            uint8_t cq = scls[r];
            bool hit = false;
            for (int i0 = 0; i0 < r; i0 += 64) {
                int i = i0 + lane;
                bool h = false;
                if (i < r && !(skip_other_cls && scls[i] != cq)) {
                    RBox p = rb[i];
                    if (!far_apart(p, q, kq)) {
                        int dec = bdmax > 0.0f ? probiou_fast_decision(p, q, bdmax) : 0;
                        h = dec > 0 || (dec == 0 && probiou(p, q) >= iou_thres);
                    }
                }
                if (__ballot(h)) { hit = true; break; }
            }
            if (lane == 0) keep[r] = (uint8_t)!hit;
        }
    }
    __syncthreads();

    // 4. first max_det survivors in score order -> rows (x, y, w, h, conf, cls, theta)
    int m = 0;
    float *ob = out + (int64_t)b * max_det * 7;
    for (int r0 = 0; r0 < n && m < max_det; r0 += NT) {
        int r = r0 + tid;
        bool flag = (r < n) && keep[r];
        int slot = ordered_slot(flag, wave_tot, m);
        if (flag && slot < max_det) {
            int k = order[r];
            const float *pp = pb + (int64_t)cand[k] * np;
            float *o = ob + (int64_t)slot * 7;
            o[0] = pp[0]; o[1] = pp[1]; o[2] = pp[2]; o[3] = pp[3];
            o[4] = cscore[k]; o[5] = (float)ccls[k]; o[6] = pp[4 + nc];
        }
    }
    if (tid == 0) count[b] = m < max_det ? m : max_det;
}

// Candidate-first form of decode + non_max_suppression(rotated=True) (obb_decode_nms).  A 416-px tile has 3549 anchors and, at conf
// 0.25, about a dozen candidates: decoding every anchor and sizing the NMS for all of them spends its time on rows that are dropped.
//
// k_cand_nms: one 256-thread workgroup per tile, 18 KB of LDS (several per CU, next to the convolution kernels of the next forward):
//   1. candidate test on the class logits only (48 of the 320 bytes of a head row, three 16-B loads): an anchor whose largest logit
//      is below logit(conf) - guard cannot pass; the others get the exact sigmoid of every class (first maximum, conf > thr) exactly
//      like the full decode; survivors are compacted in anchor order (stable ties) into LDS and into the tile's global scratch rows;
//   2. DFL / dist2rbox / angle for the survivors only (decode_anchor: the code k_decode runs);
//   3. stable rank sort, covariance terms, Fast-NMS and the max_det cut on LDS arrays of kCandCap rows.
// A tile with more than kCandCap candidates (a saturated tile, or metrics mode at conf 0.001) is appended to a device-side list and
// finished by kernels that spread ONE tile over many workgroups, so that the slowest tile no longer sets the time of the whole call:
// k_heavy_prep + k_heavy_rows (sort-free, every candidate of the tile in LDS: predict-mode thresholds, tiles up to ~4300 anchors),
// k_heavy_sort / k_heavy_nms / k_heavy_out (larger inputs) or the round form (metrics mode).  Results are identical to the full
// decode + k_nms_tile path (kept as the parity reference in tests): same candidates, same arithmetic, same order.
static constexpr int kCandCap = 256;
static constexpr int kHeavySlots = 64;   // flagged tiles are walked by this many block columns (grid-stride)

struct HeavyScratch {  // per tile, stride AS rows (AS = A rounded up to 4)
    int32_t *cand;     // anchor of candidate k (anchor order)
    float *cscore;     // its confidence
    uint8_t *ccls;     // its class
    RBox *rb;          // sorted order: covariance terms (class offset applied)
    float *sbox;       // sorted order: [5] x, y, w, h, theta
    float *sscore;     // sorted order
    uint8_t *scls, *keep;  // sorted order
    int32_t *ncand;    // [B] candidates of a flagged tile
    int32_t *flist;    // [B] flagged tiles
    int32_t *nflag;    // [1]
    int32_t *sidx;     // sorted order: position in the candidate list
    int32_t *nkept;    // [B] kept rows of the rounds processed so far
    int32_t *done;     // [B] snapshot of nkept >= max_det taken between rounds
    int32_t *wbase;    // [B + 1] k_heavy_rows: first work item of flagged tile f (prefix sums of the tiles' workgroup shares)
};

// class scores of one anchor: three 16-byte loads instead of nc scalar ones (rows are padded to a multiple of 4 floats, columns past nc
// are masked); returns the exact (best sigmoid, first maximum) pair, or best = -inf when the largest logit is below the gate
__device__ __forceinline__ void class_best(const float *__restrict__ cp, int nc, float logit_gate, float &best, int &bj) {
    float lg[16];
    float mx = -INFINITY;
    const int n4 = (nc + 3) >> 2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q < n4) {
            const float4 v = *reinterpret_cast<const float4 *>(cp + 4 * q);
            lg[4 * q] = v.x; lg[4 * q + 1] = v.y; lg[4 * q + 2] = v.z; lg[4 * q + 3] = v.w;
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (j < nc) mx = fmaxf(mx, lg[j]);
    best = -INFINITY; bj = 0;
    if (mx > logit_gate) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < nc) { const float v = sigmoid_f(lg[j]); if (v > best) { best = v; bj = j; } }
    }
}

template <bool WIDE /* nc <= 16: vector loads */>
__global__ __launch_bounds__(256) void k_cand_nms(const float *__restrict__ head, const float *__restrict__ cmax, int A, int nc, int h, int w, float conf_thres, float logit_gate,
                                                 float iou_thres, int max_det, float kq, float bdmax, HeavyScratch S, float *__restrict__ out,
                                                 int32_t *__restrict__ count) {
    __shared__ int wave_tot[16];
    __shared__ __attribute__((aligned(16))) RBox rb[kCandCap];
    __shared__ float cbox[kCandCap][5];
    __shared__ float cscore[kCandCap];
    __shared__ int32_t cand[kCandCap], order[kCandCap];
    __shared__ uint8_t ccls[kCandCap], scls[kCandCap], keep[kCandCap];
    constexpr int NT = 256;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int no = (4 * kRegMaxD + nc + 1 + 3) / 4 * 4;
    const float *hb = head + (int64_t)b * A * no;
    const int64_t AS = (A + 3) & ~3;
    int32_t *gcand = S.cand + (int64_t)b * AS;
    float *gscore = S.cscore + (int64_t)b * AS;
    uint8_t *gcls = S.ccls + (int64_t)b * AS;

    // 1. candidates in anchor order
    int n = 0;
    if constexpr (WIDE) {
        // Two passes per 4096 anchors, so that the scan pays two memory latencies instead of one per 256 anchors: (a) the gate -- cmax, the
        // forward's per-anchor maximum of the class logits (dense: 4 bytes per anchor instead of a 48-byte piece of every 320-byte row), all
        // 16 values of a thread loaded up front; without cmax every anchor passes -- compacted in anchor order into an LDS list; (b) the exact
        // class scores of the listed anchors (class_best: the same values and decisions as before), compacted again into the candidates.
        __shared__ uint16_t gl[4096];
        for (int a00 = 0; a00 < A; a00 += 4096) {
            float cm[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int a = a00 + k * NT + tid;
                cm[k] = (cmax && a < A) ? cmax[(int64_t)b * A + a] : INFINITY;
            }
            int ng = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (a00 + k * NT >= A) break;  // (uniform)
                const int a = a00 + k * NT + tid;
                const bool flag = (a < A) && (cm[k] > logit_gate);
                const int slot = ordered_slot(flag, wave_tot, ng);
                if (flag) gl[slot] = (uint16_t)(a - a00);
            }
            __syncthreads();
            for (int k0 = 0; k0 < ng; k0 += NT) {
                const int k = k0 + tid;
                float best = -INFINITY;
                int bj = 0, a = 0;
                if (k < ng) {
                    a = a00 + gl[k];
                    class_best(hb + (int64_t)a * no + 4 * kRegMaxD, nc, logit_gate, best, bj);
                }
                const bool flag = (k < ng) && (best > conf_thres);
                const int slot = ordered_slot(flag, wave_tot, n);
                if (flag) {
                    gcand[slot] = a; gscore[slot] = best; gcls[slot] = (uint8_t)bj;  // (the copy a tile above kCandCap is finished from)
                    if (slot < kCandCap) { cand[slot] = a; cscore[slot] = best; ccls[slot] = (uint8_t)bj; }
                }
            }
            __syncthreads();  // gl is rewritten by the next 4096 anchors
        }
    } else {
        for (int a0 = 0; a0 < A; a0 += NT) {
            const int a = a0 + tid;
            float best = -INFINITY;
            int bj = 0;
            if (a < A) {
                const float *cp = hb + (int64_t)a * no + 4 * kRegMaxD;
                float mx = -INFINITY;
                for (int j = 0; j < nc; ++j) mx = fmaxf(mx, cp[j]);
                if (mx > logit_gate)
                    for (int j = 0; j < nc; ++j) { const float v = sigmoid_f(cp[j]); if (v > best) { best = v; bj = j; } }
            }
            const bool flag = (a < A) && (best > conf_thres);
            const int slot = ordered_slot(flag, wave_tot, n);
            if (flag) {
                gcand[slot] = a; gscore[slot] = best; gcls[slot] = (uint8_t)bj;
                if (slot < kCandCap) { cand[slot] = a; cscore[slot] = best; ccls[slot] = (uint8_t)bj; }
            }
        }
    }
    __syncthreads();
    if (n == 0) { if (tid == 0) count[b] = 0; return; }
    if (n > kCandCap) {  // handed to the row-parallel kernels
        if (tid == 0) { S.ncand[b] = n; S.nkept[b] = 0; S.flist[atomicAdd(S.nflag, 1)] = b; count[b] = 0; }  // (nkept: k_heavy_rows' arrival counter)
        return;
    }

    // 2. decode the candidates
    for (int k = tid; k < n; k += NT) {
        const int a = cand[k];
        decode_anchor(hb + (int64_t)a * no, a, nc, h, w, cbox[k][0], cbox[k][1], cbox[k][2], cbox[k][3], cbox[k][4]);
    }
    // 3a. stable descending rank sort by confidence
    for (int i = tid; i < n; i += NT) {
        const float si = cscore[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) { const float sj = cscore[j]; rank += (sj > si) | ((sj == si) & (j < i)); }
        order[rank] = i;
    }
    __syncthreads();
    for (int r = tid; r < n; r += NT) {
        const int k = order[r];
        const float c = (float)ccls[k] * kMaxWh;  // class offset: boxes of different classes never overlap
        rb[r] = make_rbox(cbox[k][0] + c, cbox[k][1] + c, cbox[k][2], cbox[k][3], cbox[k][4]);
        scls[r] = ccls[k];
    }
    __syncthreads();
    // 3b. Fast-NMS: keep r iff no i < r with probiou(i, r) >= thr (suppressed boxes still suppress); one wave per row
    const bool skip_other_cls = iou_thres > 1e-3f;
    {
        const int lane = tid & 63, wave = tid >> 6, nwave = NT >> 6;
        for (int r = wave; r < n; r += nwave) {
            const RBox q = rb[r];
            const uint8_t cq = scls[r];
            bool hit = false;
            for (int i0 = 0; i0 < r; i0 += 64) {
                const int i = i0 + lane;
                bool hh = false;
                if (i < r && !(skip_other_cls && scls[i] != cq)) {
                    const RBox p = rb[i];
                    if (!far_apart(p, q, kq)) {
                        const int dec = bdmax > 0.0f ? probiou_fast_decision(p, q, bdmax) : 0;
                        hh = dec > 0 || (dec == 0 && probiou(p, q) >= iou_thres);
                    }
                }
                if (__ballot(hh)) { hit = true; break; }
            }
            if (lane == 0) keep[r] = (uint8_t)!hit;
        }
    }
    __syncthreads();
    // 4. first max_det survivors in score order -> rows (x, y, w, h, conf, cls, theta)
    int m = 0;
    float *ob = out + (int64_t)b * max_det * 7;
    for (int r0 = 0; r0 < n && m < max_det; r0 += NT) {
        const int r = r0 + tid;
        const bool flag = (r < n) && keep[r];
        const int slot = ordered_slot(flag, wave_tot, m);
        if (flag && slot < max_det) {
            const int k = order[r];
            float *o = ob + (int64_t)slot * 7;
            o[0] = cbox[k][0]; o[1] = cbox[k][1]; o[2] = cbox[k][2]; o[3] = cbox[k][3];
            o[4] = cscore[k]; o[5] = (float)ccls[k]; o[6] = cbox[k][4];
        }
    }
    if (tid == 0) count[b] = m < max_det ? m : max_det;
}

// ---- tiles above kCandCap candidates, one tile spread over many workgroups.  grid (kHeavySlots, chunks): block column s walks the flagged
//      tiles s, s + kHeavySlots, ...
// rank of candidate i among the tile's n scores (stable, descending) + its decoded box -> sorted arrays
__global__ __launch_bounds__(256) void k_heavy_sort(const float *__restrict__ head, int A, int nc, int h, int w, int max_nms, HeavyScratch S) {
    __shared__ __attribute__((aligned(16))) float sc[1024];
    const int tid = threadIdx.x;
    const int no = (4 * kRegMaxD + nc + 1 + 3) / 4 * 4;
    const int64_t AS = (A + 3) & ~3;
    const int nflag = *S.nflag;
    for (int f = blockIdx.x; f < nflag; f += gridDim.x) {
        const int b = S.flist[f];
        const int n = S.ncand[b];
        const float *gscore = S.cscore + (int64_t)b * AS;
        for (int i0 = blockIdx.y * 256; i0 < n; i0 += gridDim.y * 256) {  // (uniform per block: the barriers below are safe)
            const int i = i0 + tid;
            const float si = i < n ? gscore[i] : 0.f;
            int rank = 0;
            for (int j0 = 0; j0 < n; j0 += 1024) {  // 1024 scores per LDS chunk (padding -inf never outranks anything), four compares per 16-B read
                __syncthreads();
                for (int q = tid; q < 1024; q += 256) sc[q] = j0 + q < n ? gscore[j0 + q] : -INFINITY;
                __syncthreads();
                const int lim = min(1024, (n - j0 + 3) & ~3);
                for (int j = 0; j < lim; j += 4) {
                    const float4 sj = *reinterpret_cast<const float4 *>(sc + j);
                    const int jj = j0 + j;
                    rank += (sj.x > si) | ((sj.x == si) & (jj < i));
                    rank += (sj.y > si) | ((sj.y == si) & (jj + 1 < i));
                    rank += (sj.z > si) | ((sj.z == si) & (jj + 2 < i));
                    rank += (sj.w > si) | ((sj.w == si) & (jj + 3 < i));
                }
            }
            if (i < n && rank < max_nms) {  // n > max_nms keeps the top max_nms
                const int a = S.cand[(int64_t)b * AS + i];
                const int cls = S.ccls[(int64_t)b * AS + i];
                float x, y, ww, hh, t;
                decode_anchor(head + ((int64_t)b * A + a) * no, a, nc, h, w, x, y, ww, hh, t);
                const int64_t r = (int64_t)b * AS + rank;
                const float c = (float)cls * kMaxWh;
                S.rb[r] = make_rbox(x + c, y + c, ww, hh, t);
                float *sb = S.sbox + r * 5;
                sb[0] = x; sb[1] = y; sb[2] = ww; sb[3] = hh; sb[4] = t;
                S.sscore[r] = si;
                S.scls[r] = (uint8_t)cls;
            }
        }
    }
}

// Fast-NMS rows of a flagged tile: one wave per row, rows interleaved over gridDim.y * 4 waves
__global__ __launch_bounds__(256) void k_heavy_nms(int A, float iou_thres, int max_nms, float kq, float bdmax, HeavyScratch S) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int gw = blockIdx.y * 4 + (tid >> 6), nw = gridDim.y * 4;
    const int64_t AS = (A + 3) & ~3;
    const bool skip_other_cls = iou_thres > 1e-3f;
    const int nflag = *S.nflag;
    for (int f = blockIdx.x; f < nflag; f += gridDim.x) {
        const int b = S.flist[f];
        const int n = min(S.ncand[b], max_nms);
        const RBox *rb = S.rb + (int64_t)b * AS;
        const uint8_t *scls = S.scls + (int64_t)b * AS;
        uint8_t *keep = S.keep + (int64_t)b * AS;
        for (int r = gw; r < n; r += nw) {
            const RBox q = rb[r];
            const uint8_t cq = scls[r];
            bool hit = false;
            // 256 predecessors per step: a lane's four rows are loaded together (four independent global loads in flight -- the loop used to pay
            // one memory latency per 64 predecessors, ~30 of them in sequence for the last rows of a 2000-candidate tile)
            for (int i0 = 0; i0 < r; i0 += 256) {
                RBox p[4];
                uint8_t cp[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = min(i0 + u * 64 + lane, r - 1);
                    p[u] = rb[i]; cp[u] = scls[i];
                }
                bool hh = false;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * 64 + lane;
                    if (i < r && !(skip_other_cls && cp[u] != cq) && !far_apart(p[u], q, kq)) {
                        const int dec = bdmax > 0.0f ? probiou_fast_decision(p[u], q, bdmax) : 0;
                        hh = hh || dec > 0 || (dec == 0 && probiou(p[u], q) >= iou_thres);
                    }
                }
                if (__ballot(hh)) { hit = true; break; }
            }
            if (lane == 0) keep[r] = (uint8_t)!hit;
        }
    }
}

// ---- the round form of the heavy path (metrics mode: thousands of candidates per tile, of which only the first max_det survivors in
//      score order are ever output).  A Fast-NMS row is kept iff NO higher-scored row overlaps it -- kept or not -- so the keep flag of
//      row r needs rows [0, r) only: the rows are processed in rounds of kRoundRows in score order and a tile stops at the first round
//      boundary where it has max_det survivors.  Nothing behind that boundary is decoded or compared; the rows in front of it get exactly
//      the flags of the all-rows form above (k_heavy_sort / k_heavy_nms stay as its parity reference).
static constexpr int kRoundRows = 512;

// stable descending sort of a flagged tile's candidate scores: one workgroup per tile, bitonic network on 64-bit keys in LDS
// key = score bits (positive floats order like their bit patterns) << 32 | ~position (ties: the earlier candidate = lower anchor first)
__global__ __launch_bounds__(1024) void k_heavy_bitonic(int A, int max_det, HeavyScratch S) {
    extern __shared__ unsigned long long skey[];
    const int tid = threadIdx.x;
    const int64_t AS = (A + 3) & ~3;
    const int nflag = *S.nflag;
    for (int f = blockIdx.x; f < nflag; f += gridDim.x) {
        const int b = S.flist[f];
        const int n = S.ncand[b];
        int NP = 1024;
        while (NP < n) NP <<= 1;
        const float *gscore = S.cscore + (int64_t)b * AS;
        __syncthreads();
        for (int i = tid; i < NP; i += 1024)
            skey[i] = i < n ? ((unsigned long long)__float_as_uint(gscore[i]) << 32) | (unsigned long long)(0xffffffffu - (unsigned)i) : 0ull;
        __syncthreads();
        for (int k = 2; k <= NP; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < NP; i += 1024) {
                    const int l = i ^ j;
                    if (l > i) {
                        const unsigned long long a = skey[i], c = skey[l];
                        const bool desc = (i & k) == 0;  // descending blocks first: the final order is descending
                        if (desc ? a < c : a > c) { skey[i] = c; skey[l] = a; }
                    }
                }
                __syncthreads();
            }
        int32_t *sidx = S.sidx + (int64_t)b * AS;
        for (int i = tid; i < n; i += 1024) sidx[i] = (int32_t)(0xffffffffu - (unsigned)(skey[i] & 0xffffffffull));
        if (tid == 0) { S.nkept[b] = 0; S.done[b] = 0; }
    }
}

// rows [r0, r1) of every unfinished flagged tile: decode + covariance terms into the sorted arrays; also takes the `done` snapshot
__global__ __launch_bounds__(256) void k_heavy_decode_rows(const float *__restrict__ head, int A, int nc, int h, int w, int r0, int r1, int max_det, HeavyScratch S) {
    const int no = (4 * kRegMaxD + nc + 1 + 3) / 4 * 4;
    const int64_t AS = (A + 3) & ~3;
    const int nflag = *S.nflag;
    for (int f = blockIdx.x; f < nflag; f += gridDim.x) {
        const int b = S.flist[f];
        const bool fin = S.nkept[b] >= max_det;  // (complete: the previous round's kernel has ended)
        if (blockIdx.y == 0 && threadIdx.x == 0) S.done[b] = fin;
        if (fin) continue;
        const int n = S.ncand[b];
        const int rank = r0 + blockIdx.y * 256 + threadIdx.x;
        if (rank >= r1 || rank >= n) continue;
        const int i = S.sidx[(int64_t)b * AS + rank];
        const int a = S.cand[(int64_t)b * AS + i];
        const int cls = S.ccls[(int64_t)b * AS + i];
        float x, y, ww, hh, t;
        decode_anchor(head + ((int64_t)b * A + a) * no, a, nc, h, w, x, y, ww, hh, t);
        const int64_t r = (int64_t)b * AS + rank;
        const float c = (float)cls * kMaxWh;
        S.rb[r] = make_rbox(x + c, y + c, ww, hh, t);
        float *sb = S.sbox + r * 5;
        sb[0] = x; sb[1] = y; sb[2] = ww; sb[3] = hh; sb[4] = t;
        S.sscore[r] = S.cscore[(int64_t)b * AS + i];
        S.scls[r] = (uint8_t)cls;
    }
}

// Fast-NMS flags of rows [r0, r1): one wave per row against every earlier row; the round's survivors are counted into nkept
__global__ __launch_bounds__(256) void k_heavy_nms_rows(int A, float iou_thres, int r0, int r1, float kq, float bdmax, HeavyScratch S) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int gw = blockIdx.y * 4 + (tid >> 6), nw = gridDim.y * 4;
    const int64_t AS = (A + 3) & ~3;
    const bool skip_other_cls = iou_thres > 1e-3f;
    const int nflag = *S.nflag;
    for (int f = blockIdx.x; f < nflag; f += gridDim.x) {
        const int b = S.flist[f];
        if (S.done[b]) continue;
        const int n = min(S.ncand[b], r1);
        const RBox *rb = S.rb + (int64_t)b * AS;
        const uint8_t *scls = S.scls + (int64_t)b * AS;
        uint8_t *keep = S.keep + (int64_t)b * AS;
        int kept = 0;
        for (int r = r0 + gw; r < n; r += nw) {
            const RBox q = rb[r];
            const uint8_t cq = scls[r];
            bool hit = false;
            for (int i0 = 0; i0 < r; i0 += 64) {
                const int i = i0 + lane;
                bool hh = false;
                if (i < r && !(skip_other_cls && scls[i] != cq)) {
                    const RBox p = rb[i];
                    if (!far_apart(p, q, kq)) {
                        const int dec = bdmax > 0.0f ? probiou_fast_decision(p, q, bdmax) : 0;
                        hh = dec > 0 || (dec == 0 && probiou(p, q) >= iou_thres);
                    }
                }
                if (__ballot(hh)) { hit = true; break; }
            }
            if (lane == 0) keep[r] = (uint8_t)!hit;
            kept += !hit;
        }
        if (lane == 0 && kept) atomicAdd(&S.nkept[b], kept);
    }
}

// first max_det survivors of a flagged tile in score order -> output rows
__global__ __launch_bounds__(256) void k_heavy_out(int A, int max_det, int max_nms, HeavyScratch S, float *__restrict__ out, int32_t *__restrict__ count) {
    __shared__ int wave_tot[16];
    const int tid = threadIdx.x;
    const int64_t AS = (A + 3) & ~3;
    const int nflag = *S.nflag;
    for (int f = blockIdx.x; f < nflag; f += gridDim.x) {
        const int b = S.flist[f];
        const int n = min(S.ncand[b], max_nms);
        const uint8_t *keep = S.keep + (int64_t)b * AS;
        float *ob = out + (int64_t)b * max_det * 7;
        int m = 0;
        for (int r0 = 0; r0 < n && m < max_det; r0 += 256) {
            const int r = r0 + tid;
            const bool flag = (r < n) && keep[r];
            const int slot = ordered_slot(flag, wave_tot, m);
            if (flag && slot < max_det) {
                const int64_t g = (int64_t)b * AS + r;
                const float *sb = S.sbox + g * 5;
                float *o = ob + (int64_t)slot * 7;
                o[0] = sb[0]; o[1] = sb[1]; o[2] = sb[2]; o[3] = sb[3];
                o[4] = S.sscore[g]; o[5] = (float)S.scls[g]; o[6] = sb[4];
            }
        }
        if (tid == 0) count[b] = m < max_det ? m : max_det;
        __syncthreads();
    }
}

// ---- the LDS-resident, SORT-FREE form of the all-rows heavy path (k_heavy_prep + k_heavy_rows instead of k_heavy_sort + k_heavy_nms +
//      k_heavy_out).  Those three are each the latency of their slowest tile -- a 2000-candidate tile: an O(n^2) rank count (83 us), then ~8 rows
//      per wave with up to 8 dependent global loads each (97 us), then the compaction -- 185 us per 1024 tiles of which 36 are flagged.
// Fast-NMS keeps row r iff NO candidate that precedes it in (score descending, position ascending) order overlaps it: that is a property of
// the SET of predecessors, so the rows need not be sorted to be judged -- only the survivors (a few dozen) need their rank for the output.
//   k_heavy_prep   one thread per candidate of a flagged tile, in CANDIDATE order: decode_anchor (the code of every other path) -> covariance
//                  terms (class offset applied) + decoded box, to global scratch;
//   k_heavy_rows   a flagged tile of n candidates is shared by G ~ n^2 / 2^17 <= gridDim.y workgroups of 512 threads.  Each loads the tile's
//                  terms, scores and classes into LDS (one array per term: a wave's 64 partners are 64 consecutive words; 37 bytes per
//                  candidate, 131 KB for the 3549 anchors of a 416-px tile, + 16 KB of score histogram) and takes rows r = part, part + G, ...: a wave scans ALL n candidates,
//                  256 per step, starting at r's own block (neighbouring anchors are the likeliest suppressors, and any hit ends the row),
//                  testing "precedes r" on the scores; every operand is an LDS read.  The LAST workgroup to finish a tile (device-scope
//                  counter) compacts the survivors in candidate order, ranks them among themselves and writes the first max_det rows.
// Same candidates, same pair arithmetic, same rows as the three-kernel form (kept for inputs whose candidates do not fit the LDS, above
// ~4300 anchors) and as obb_decode_nms_full (tests).
__device__ __forceinline__ int heavy_share(int n, int gmax) { return max(1, min(gmax, (int)(((int64_t)n * n + (1 << 17) - 1) >> 17))); }
static constexpr int kHeavyShareMax = 32;

__global__ __launch_bounds__(256) void k_heavy_prep(const float *__restrict__ head, int A, int nc, int h, int w, HeavyScratch S) {
    __shared__ int wave_tot[16];
    const int no = (4 * kRegMaxD + nc + 1 + 3) / 4 * 4;
    const int64_t AS = (A + 3) & ~3;
    const int nflag = *S.nflag;
    if (blockIdx.x == 0 && blockIdx.y == 0) {  // the work list of k_heavy_rows: tile f owns items [wbase[f], wbase[f + 1])
        int base = 0;
        for (int f0 = 0; f0 < nflag; f0 += 256) {
            const int f = f0 + threadIdx.x;
            const int g = f < nflag ? heavy_share(S.ncand[S.flist[f]], kHeavyShareMax) : 0;
            // exclusive prefix over the block (the ordered_slot pattern, with counts instead of flags)
            int v = g;
            const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o); if (lane >= o) v += t; }
            if (lane == 63) wave_tot[wv] = v;
            __syncthreads();
            int off = 0, tot = 0;
            for (int k = 0; k < 4; ++k) { if (k < wv) off += wave_tot[k]; tot += wave_tot[k]; }
            if (f < nflag) S.wbase[f] = base + off + v - g;
            base += tot;
            __syncthreads();
        }
        if (threadIdx.x == 0) S.wbase[nflag] = base;
    }
    for (int f = blockIdx.x; f < nflag; f += gridDim.x) {
        const int b = S.flist[f];
        const int n = S.ncand[b];
        const int64_t tb = (int64_t)b * AS;
        for (int i = blockIdx.y * 256 + threadIdx.x; i < n; i += gridDim.y * 256) {
            const int a = S.cand[tb + i];
            const int cls = S.ccls[tb + i];
            float x, y, ww, hh, t;
            decode_anchor(head + ((int64_t)b * A + a) * no, a, nc, h, w, x, y, ww, hh, t);
            const float c = (float)cls * kMaxWh;
            S.rb[tb + i] = make_rbox(x + c, y + c, ww, hh, t);
            float *sb = S.sbox + (tb + i) * 5;
            sb[0] = x; sb[1] = y; sb[2] = ww; sb[3] = hh; sb[4] = t;
        }
    }
}

static constexpr int kHeavyRowsThreads = 512;
static constexpr int kHeavyBins = 4096;
// grid: resident workgroups (one per CU: the LDS), each walks work items w = blockIdx.x, + gridDim.x, ... -- no workgroup is launched for nothing
__global__ __launch_bounds__(kHeavyRowsThreads) void k_heavy_rows(int A, float conf_thres, float iou_thres, int max_det, float kq, float bdmax, HeavyScratch S,
                                                              float *__restrict__ out, int32_t *__restrict__ count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
    __shared__ int wave_tot[16];
    __shared__ int s_last, s_tbin, s_next;
    constexpr int NT = kHeavyRowsThreads, NWV = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t AS = (A + 3) & ~3;
    float *lx = reinterpret_cast<float *>(hsm), *ly = lx + AS, *lA = ly + AS, *lB = lA + AS, *lC = lB + AS, *ldet = lC + AS, *ltr = ldet + AS, *ls = ltr + AS;
    int32_t *surv = reinterpret_cast<int32_t *>(ls + AS);
    uint8_t *lcls = reinterpret_cast<uint8_t *>(surv + AS);
    const bool skip_other_cls = iou_thres > 1e-3f;
    const int nflag = *S.nflag;
    const int total = S.wbase[nflag];
    for (int wi = blockIdx.x; wi < total; wi += gridDim.x) {
        int lo = 0, hi = nflag;  // the tile of item wi: the last f with wbase[f] <= wi
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (S.wbase[mid] <= wi) lo = mid; else hi = mid; }
        const int f = lo, part = wi - S.wbase[f], G = S.wbase[f + 1] - S.wbase[f];
        const int b = S.flist[f];
        const int n = S.ncand[b];
        const int64_t tb = (int64_t)b * AS;
        __syncthreads();  // the previous item's LDS is no longer read
        if (tid == 0) s_next = 0;
        for (int i = tid; i < n; i += NT) {
            const RBox q = S.rb[tb + i];
            lx[i] = q.x; ly[i] = q.y; lA[i] = q.A; lB[i] = q.B; lC[i] = q.C; ldet[i] = q.det; ltr[i] = q.tr;
            ls[i] = S.cscore[tb + i];
            lcls[i] = S.ccls[tb + i];
        }
        __syncthreads();
        uint8_t *keep = S.keep + tb;
        constexpr int U = 4, STEP = 64 * U;
        const int nblk = (n + STEP - 1) / STEP;
        // this workgroup's rows are r = part, part + G, ...; its waves take them one at a time from a shared counter (a kept row scans the whole
        // tile, a suppressed one stops at its first hit: dealt out in a fixed pattern the slowest wave took 1.6x the average)
        auto claim = [&]() -> int {  // the next row of this workgroup, the same in every lane of the wave
            int j = 0;
            if (lane == 0) j = __hip_atomic_fetch_add(&s_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return part + G * __shfl(j, 0);
        };
        for (int r = claim(); r < n; r = claim()) {
            const RBox q = {lx[r], ly[r], lA[r], lB[r], lC[r], ldet[r], ltr[r]};
            const float sr = ls[r];
            const uint8_t cq = lcls[r];
            bool hit = false;
            int blk = r / STEP;
            for (int k = 0; k < nblk; ++k) {
                // stage 1, branch-free: score / class / centre / trace of 256 candidates in one round of LDS reads -> who precedes r, has its
                // class and is not far apart; stage 2 (a second round of reads + the pair test) for those lanes only -- about every other step
                RBox p[U];
                bool near[U];
                bool some = false;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = blk * STEP + u * 64 + lane, ii = min(i, n - 1);
                    const float si = ls[ii];
                    const uint8_t ci = lcls[ii];
                    p[u].x = lx[ii]; p[u].y = ly[ii]; p[u].tr = ltr[ii];
                    const bool pred = (si > sr) | ((si == sr) & (i < r));  // i precedes r in the stable descending order
                    near[u] = (i < n) & pred & !(skip_other_cls & (ci != cq)) & !far_apart(p[u], q, kq);
                    some |= near[u];
                }
                bool hh = false;
                if (__any(some)) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (near[u]) {
                            const int ii = blk * STEP + u * 64 + lane;
                            p[u].A = lA[ii]; p[u].B = lB[ii]; p[u].C = lC[ii]; p[u].det = ldet[ii];
                            const int dec = bdmax > 0.0f ? probiou_fast_decision(p[u], q, bdmax) : 0;
                            hh = hh || dec > 0 || (dec == 0 && probiou(p[u], q) >= iou_thres);
                        }
                    }
                }
                if (__ballot(hh)) { hit = true; break; }
                if (++blk == nblk) blk = 0;
            }
            if (lane == 0) __hip_atomic_store(keep + r, (uint8_t)!hit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (written through: see below)
        }
        // ---- the last of the tile's G workgroups writes its rows.  The flags are the only data that crosses workgroups (and XCDs, each with
        // its own L2): they are stored and loaded as device-scope atomics (written through / read past the L2), so that the hand-over needs
        // no __threadfence() -- its write-back of the whole L2 cost 29 of this kernel's 105 us -- only "my stores have completed" before
        // the arrival counter (a device-scope atomic itself).
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        if (tid == 0) s_last = atomicAdd(&S.nkept[b], 1) == G - 1;
        __syncthreads();
        if (!s_last) continue;
        // survivors in candidate order; with more than max_det of them only those in the top score bins can be output: a histogram of the
        // scores' upper bits (positive floats order like their bit patterns) finds the bin where the count from the top reaches max_det, and
        // exact ranks -- (score descending, position ascending) among ALL survivors = among those at or above that bin -- are counted for
        // those only (a 2000-candidate tile of random boxes keeps 1300 rows: 1300^2 comparisons otherwise)
        int32_t *hist = reinterpret_cast<int32_t *>(hsm + ((AS * 37 + 15) & ~(int64_t)15));  // (lx: the survivors' scores from here on -- the pair tests are over)
        const unsigned bits0 = __float_as_uint(fmaxf(conf_thres, 0.0f));
        const unsigned range = 0x3f800000u > bits0 ? 0x3f800000u - bits0 : 1u;
        const int shift = max(0, 32 - __clz((int)range) - 12);  // (range >> shift) < 4096
        for (int i = tid; i < kHeavyBins + 8; i += NT) hist[i] = 0;
        int m = 0;
        const uint32_t *keep32 = reinterpret_cast<const uint32_t *>(keep);  // (tb and the scratch base are multiples of 4)
        for (int r0 = 0; r0 < n; r0 += 4 * NT) {  // four consecutive rows per thread and round: one 32-bit load of their flags
            const int r4 = r0 + 4 * tid;
            const uint32_t wv = r4 < n ? __hip_atomic_load(keep32 + (r4 >> 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            bool fl[4];
            int c = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { fl[q] = (r4 + q < n) && ((wv >> (8 * q)) & 0xffu); c += fl[q]; }
            int v = c;  // exclusive prefix of the counts over the workgroup (the ordered_slot pattern)
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o); if (lane >= o) v += t; }
            if (lane == 63) wave_tot[wave] = v;
            __syncthreads();
            int off = 0, tot = 0;
            for (int q = 0; q < NWV; ++q) { if (q < wave) off += wave_tot[q]; tot += wave_tot[q]; }
            int slot = m + off + v - c;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (fl[q]) { surv[slot] = r4 + q; lx[slot] = ls[r4 + q]; ++slot; }
            m += tot;
            __syncthreads();
        }
        __syncthreads();
        int m2 = m;
        if (m > max_det) {
            for (int k = tid; k < m; k += NT) {
                const unsigned sb = __float_as_uint(lx[k]);
                atomicAdd(&hist[min(kHeavyBins, (int)((sb > bits0 ? sb - bits0 : 0u) >> shift))], 1);
            }
            __syncthreads();
            {   // thread t owns the PER bins below kHeavyBins - PER t (from the top): where does the count from the top reach max_det?
                constexpr int PER = (kHeavyBins + 1 + NT - 1) / NT;
                const int top = kHeavyBins - tid * PER;
                int mine = 0;
#pragma unroll
                for (int j = 0; j < PER; ++j) { const int bin = top - j; if (bin >= 0) mine += hist[bin]; }
                int incl = mine;
                for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
                if (lane == 63) wave_tot[wave] = incl;
                __syncthreads();
                int off = 0;
                for (int q = 0; q < NWV; ++q) if (q < wave) off += wave_tot[q];
                incl += off;
                const int before = incl - mine;  // survivors in the bins above this thread's
                if (before < max_det && incl >= max_det) {
                    int acc = before, tb_ = 0;
                    for (int j = 0; j < PER; ++j) { const int bin = top - j; if (bin < 0) break; acc += hist[bin]; tb_ = bin; if (acc >= max_det) break; }
                    s_tbin = tb_;
                }
                __syncthreads();
            }
            const int tbin = s_tbin;
            // second compaction (in place: slot <= k): the survivors at or above the threshold bin, still in candidate order
            m2 = 0;
            for (int k0 = 0; k0 < m; k0 += NT) {
                const int k = k0 + tid;
                int r = 0; float sc = 0.f; bool flag = false;
                if (k < m) {
                    r = surv[k]; sc = lx[k];
                    const unsigned sb = __float_as_uint(sc);
                    flag = min(kHeavyBins, (int)((sb > bits0 ? sb - bits0 : 0u) >> shift)) >= tbin;
                }
                const int slot = ordered_slot(flag, wave_tot, m2);  // (barriers inside: every read of this chunk is done before a write)
                if (flag) { surv[slot] = r; lx[slot] = sc; }
            }
            __syncthreads();
        }
        // exact ranks among the m2 rows left, four scores per LDS read (padding: -inf never precedes anything)
        for (int k = m2 + tid; k < ((m2 + 3) & ~3); k += NT) lx[k] = -INFINITY;
        __syncthreads();
        float *ob = out + (int64_t)b * max_det * 7;
        for (int k = tid; k < m2; k += NT) {
            const int r = surv[k];
            const float sr = lx[k];
            int rank = 0;
            for (int j = 0; j < m2; j += 4) {
                const float4 sj = *reinterpret_cast<const float4 *>(lx + j);
                rank += (sj.x > sr) | ((sj.x == sr) & (j < k));
                rank += (sj.y > sr) | ((sj.y == sr) & (j + 1 < k));
                rank += (sj.z > sr) | ((sj.z == sr) & (j + 2 < k));
                rank += (sj.w > sr) | ((sj.w == sr) & (j + 3 < k));
            }
            if (rank < max_det) {
                const float *sb = S.sbox + (tb + r) * 5;
                float *o = ob + (int64_t)rank * 7;
                o[0] = sb[0]; o[1] = sb[1]; o[2] = sb[2]; o[3] = sb[3];
                o[4] = sr; o[5] = (float)lcls[r]; o[6] = sb[4];
            }
        }
        if (tid == 0) count[b] = m < max_det ? m : max_det;
    }
}

// stand-alone Fast-NMS on a caller-provided candidate list (parity tap): boxes [n,5], scores [n]
__global__ __launch_bounds__(256) void k_probiou_nms_list(const float *__restrict__ boxes, const float *__restrict__ scores, int n, float thr,
                                                         int32_t *__restrict__ order, uint8_t *__restrict__ keep, RBox *__restrict__ rb) {
    // single workgroup, grid-stride over candidates
    for (int i = threadIdx.x; i < n; i += 256) {
        float si = scores[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) { float sj = scores[j]; rank += (sj > si) | ((sj == si) & (j < i)); }
        order[rank] = i;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < n; r += 256) {
        const float *p = boxes + (int64_t)order[r] * 5;
        rb[r] = make_rbox(p[0], p[1], p[2], p[3], p[4]);
    }
    __syncthreads();
    for (int r = threadIdx.x; r < n; r += 256) {
        RBox q = rb[r];
        bool k = true;
        for (int i = 0; i < r; ++i)
            if (probiou(rb[i], q) >= thr) { k = false; break; }
        keep[r] = (uint8_t)k;
    }
}

// ---------------------------------------------------------------------------------------------- results
__global__ __launch_bounds__(256) void k_results(const float *__restrict__ det, const float *__restrict__ lb, int64_t n,
                                                float *__restrict__ xywhr, float *__restrict__ pts) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float r5[5], p[8];
    results_row(det + i * 7, lb ? lb + i * 3 : nullptr, r5, p);  // post_device.h
    for (int k = 0; k < 5; ++k) xywhr[i * 5 + k] = r5[k];
    for (int k = 0; k < 8; ++k) pts[i * 8 + k] = p[k];
}

// ---------------------------------------------------------------------------------------------- tiler crops / letterbox
__global__ __launch_bounds__(256) void k_gather_tiles(const uint8_t *__restrict__ img, int H, int W, int C, const int32_t *__restrict__ rects,
                                                     int tile, uint8_t *__restrict__ out) {
    // grid: (row chunks, tile rows, ntiles); a thread copies 4 consecutive bytes of a tile row
    int t = blockIdx.z, row = blockIdx.y;
    int rowbytes = tile * C;
    int x0 = rects[4 * t], y0 = rects[4 * t + 1];
    const uint8_t *src = img + ((int64_t)(y0 + row) * W + x0) * C;
    uint8_t *dst = out + ((int64_t)t * tile + row) * rowbytes;
    int i = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < rowbytes) {
        uint32_t v = (uint32_t)src[i] | ((uint32_t)src[i + 1] << 8) | ((uint32_t)src[i + 2] << 16) | ((uint32_t)src[i + 3] << 24);
        *reinterpret_cast<uint32_t *>(dst + i) = v;
    } else {
        for (int k = i; k < rowbytes; ++k) dst[k] = src[k];
    }
}

struct LbParams { int x, y, cw, ch, new_w, new_h, top, left, out_h, out_w, resize; double sx, sy; };

__device__ __forceinline__ void lin_coeff(int d, double scale, int n_src, int &s, int &a1) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);  // OpenCV: double expression, stored to float
    int fl = (int)floorf(f);
    f -= (float)fl;
    if (fl < 0) { fl = 0; f = 0.f; }
    if (fl >= n_src - 1) { fl = n_src - 1; f = 0.f; }
    s = fl;
    a1 = (int)rintf(f * 2048.0f);
}

__global__ __launch_bounds__(256) void k_letterbox(const uint8_t *__restrict__ img, int H, int W, int C, LbParams P, uint8_t *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)P.out_h * P.out_w) return;
    int ox = (int)(i % P.out_w), oy = (int)(i / P.out_w);
    int rx = ox - P.left, ry = oy - P.top;
    uint8_t *o = out + i * C;
    if (rx < 0 || ry < 0 || rx >= P.new_w || ry >= P.new_h) {
        for (int c = 0; c < C; ++c) o[c] = 114;
        return;
    }
    if (!P.resize) {
        const uint8_t *s = img + ((int64_t)(P.y + ry) * W + P.x + rx) * C;
        for (int c = 0; c < C; ++c) o[c] = s[c];
        return;
    }
    // cv2.resize INTER_LINEAR for 8-bit: 11-bit fixed-point coefficients, half-pixel centres (restated; cv2 absent)
    int sx0, ax1, sy0, ay1;
    lin_coeff(rx, P.sx, P.cw, sx0, ax1);
    lin_coeff(ry, P.sy, P.ch, sy0, ay1);
    int sx1 = min(sx0 + 1, P.cw - 1), sy1 = min(sy0 + 1, P.ch - 1);
    int ax0 = 2048 - ax1, ay0 = 2048 - ay1;
    const uint8_t *r0 = img + ((int64_t)(P.y + sy0) * W + P.x) * C, *r1 = img + ((int64_t)(P.y + sy1) * W + P.x) * C;
    for (int c = 0; c < C; ++c) {
        long long top = (long long)r0[sx0 * C + c] * ax0 + (long long)r0[sx1 * C + c] * ax1;
        long long bot = (long long)r1[sx0 * C + c] * ax0 + (long long)r1[sx1 * C + c] * ax1;
        long long v = (top * ay0 + bot * ay1 + (1ll << 21)) >> 22;
        o[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

static int nms_scratch(obb_ctx *ctx, int B, int A, NmsScratch &S) {
    size_t rows = (size_t)B * ((A + 3) & ~3);
    S.cand = (int32_t *)ctx->workspace(WS_NMS_A, rows * 4);
    S.cscore = (float *)ctx->workspace(WS_NMS_B, rows * 4);
    S.ccls = (int32_t *)ctx->workspace(WS_NMS_C, rows * 4);
    S.order = (int32_t *)ctx->workspace(WS_NMS_D, rows * 4);
    S.rb = (RBox *)ctx->workspace(WS_GEOM_A, rows * sizeof(RBox));
    S.keep = (uint8_t *)ctx->workspace(WS_GEOM_B, rows * 3);
    if (!S.cand || !S.cscore || !S.ccls || !S.order || !S.rb || !S.keep) return set_error(ctx, OBB_ERR_HIP, "NMS workspace allocation failed");
    return OBB_OK;
}

}  // namespace obb

using namespace obb;

extern "C" {

int obb_decode(obb_ctx *ctx, const float *head, int32_t B, int32_t h, int32_t w, float *pred, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && B >= 0 && h > 0 && w > 0 && h % 32 == 0 && w % 32 == 0, "obb_decode: bad arguments");
    int32_t nc = 0, A = 0;
    int rc = obb_model_info(ctx, h, w, &nc, nullptr, &A, nullptr);
    if (rc) return rc;
    if (B == 0) return OBB_OK;
    OBB_REQUIRE(ctx, head && pred, "obb_decode: NULL buffer");
    hipLaunchKernelGGL(k_decode, dim3((unsigned)cdiv((int64_t)B * A, 256)), dim3(256), 0, (hipStream_t)s, head, B, A, nc, h, w, pred, (const int32_t *)nullptr);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

int obb_decode_nms(obb_ctx *ctx, const float *head, int32_t B, int32_t h, int32_t w, float conf_thres, float iou_thres, int32_t max_det,
                   float *out, int32_t *count, obb_stream_t s) {
    return obb_decode_nms_gate(ctx, head, nullptr, B, h, w, conf_thres, iou_thres, max_det, out, count, s);
}

int obb_decode_nms_gate(obb_ctx *ctx, const float *head, const float *cmax, int32_t B, int32_t h, int32_t w, float conf_thres, float iou_thres, int32_t max_det,
                        float *out, int32_t *count, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && B >= 0 && max_det > 0, "obb_decode_nms: bad arguments");
    int32_t nc = 0, A = 0;
    int rc = obb_model_info(ctx, h, w, &nc, nullptr, &A, nullptr);
    if (rc) return rc;
    if (B == 0) return OBB_OK;
    OBB_REQUIRE(ctx, head && out && count, "obb_decode_nms: NULL buffer");
    OBB_REQUIRE(ctx, nc <= 255, "obb_decode_nms: nc > 255 unsupported");
    hipStream_t st = (hipStream_t)s;
    const size_t rows = (size_t)B * ((A + 3) & ~3);
    HeavyScratch S;
    S.cand = (int32_t *)ctx->workspace(WS_NMS_A, rows * 4);
    S.cscore = (float *)ctx->workspace(WS_NMS_B, rows * 4);
    S.sscore = (float *)ctx->workspace(WS_NMS_C, rows * 4);
    S.sbox = (float *)ctx->workspace(WS_NMS_D, rows * 20);
    S.rb = (RBox *)ctx->workspace(WS_GEOM_A, rows * sizeof(RBox));
    uint8_t *bytes = (uint8_t *)ctx->workspace(WS_GEOM_B, rows * 3);
    int32_t *ints = (int32_t *)ctx->workspace(WS_GEOM_C, sizeof(int32_t) * ((size_t)5 * B + 128));
    S.sidx = (int32_t *)ctx->workspace(WS_GEOM_D, rows * 4);
    if (!S.cand || !S.cscore || !S.sscore || !S.sbox || !S.rb || !bytes || !ints || !S.sidx) return set_error(ctx, OBB_ERR_HIP, "obb_decode_nms: workspace allocation failed");
    S.ccls = bytes; S.scls = bytes + rows; S.keep = bytes + 2 * rows;
    S.ncand = ints; S.flist = ints + B; S.nkept = ints + 2 * (size_t)B; S.done = ints + 3 * (size_t)B; S.wbase = ints + 4 * (size_t)B; S.nflag = ints + 5 * (size_t)B + 64;
    OBB_HIP(ctx, hipMemsetAsync(S.nflag, 0, sizeof(int32_t), st));
    // largest logit an anchor needs to be worth the exact class scores: logit(conf) minus a guard band far above the error of sigmoid_f
    float gate = -INFINITY;
    if (conf_thres > 0.0f && conf_thres < 1.0f) {
        const double L = log((double)conf_thres / (1.0 - (double)conf_thres));
        gate = (float)(L - 1e-3 - 1e-4 * fabs(L));
    } else if (conf_thres >= 1.0f) gate = INFINITY;
    const float kq = far_apart_factor(iou_thres), bdmax = probiou_bdmax(iou_thres);
    if (nc <= 16) hipLaunchKernelGGL(k_cand_nms<true>, dim3((unsigned)B), dim3(256), 0, st, head, cmax, A, nc, h, w, conf_thres, gate, iou_thres, max_det, kq, bdmax, S, out, count);
    else hipLaunchKernelGGL(k_cand_nms<false>, dim3((unsigned)B), dim3(256), 0, st, head, (const float *)nullptr, A, nc, h, w, conf_thres, gate, iou_thres, max_det, kq, bdmax, S, out, count);
    OBB_LAUNCH_CHECK(ctx);
    // tiles above kCandCap candidates (device-side list; every block of these launches exits at once when the list is empty)
    const int slots = std::min<int>(kHeavySlots, B);
    int NPmax = 1024;
    while (NPmax < A) NPmax <<= 1;
    // Two forms for the flagged tiles, both bit-identical to the full path (tests): at predict-mode thresholds few tiles are flagged and
    // they rarely reach max_det survivors, so the all-rows form (two launches) is the cheaper one (1024 tiles at conf 0.25: 375 us against
    // 457 us with the round form's seven launches); at metrics-mode thresholds (conf 0.001: every tile flagged and saturated) the round
    // form stops each tile at its max_det-th survivor (3.87 -> 1.1 ms).  The threshold is the caller's conf, the only thing the host knows.
    if ((size_t)NPmax * 8 <= 128 * 1024 && conf_thres < 0.05f) {
        // round form: sort once, then rows in score order, kRoundRows at a time, until every tile has its max_det survivors
        static bool attr_set = false;
        if (!attr_set) {
            OBB_HIP(ctx, hipFuncSetAttribute((const void *)k_heavy_bitonic, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
            attr_set = true;
        }
        hipLaunchKernelGGL(k_heavy_bitonic, dim3((unsigned)std::min<int>(B, 256)), dim3(1024), (size_t)NPmax * 8, st, A, max_det, S);
        OBB_LAUNCH_CHECK(ctx);
        // equal rounds of kRoundRows (a multiple of k_heavy_out's 256-row chunks), one workgroup column per tile: a saturated tile has its
        // max_det survivors after ~600-1000 rows, i.e. two rounds; growing rounds (512, 1024, 2048 on a 256-column grid) made it walk 1536
        // rows against all their predecessors (0.76 -> 1.1 ms per 1024 tiles)
        const int gx = std::min<int>(B, 65535);
        for (int r0 = 0; r0 < A;) {
            const int r1 = std::min(A, r0 + kRoundRows);
            hipLaunchKernelGGL(k_heavy_decode_rows, dim3((unsigned)gx, (unsigned)cdiv(r1 - r0, 256)), dim3(256), 0, st, head, A, nc, h, w, r0, r1, max_det, S);
            hipLaunchKernelGGL(k_heavy_nms_rows, dim3((unsigned)gx, 16), dim3(256), 0, st, A, iou_thres, r0, r1, kq, bdmax, S);
            OBB_LAUNCH_CHECK(ctx);
            r0 = r1;
        }
    } else {  // the all-rows form
        const size_t AS = (size_t)((A + 3) & ~3);
        const size_t lds = ((AS * 37 + 15) & ~(size_t)15) + (kHeavyBins + 8) * 4;  // seven covariance terms + score + survivor list + class per candidate | score histogram
        if (lds <= 156 * 1024) {  // every candidate of a tile fits the LDS: two launches, the output included
            static std::mutex mu;
            static size_t attr_lds = 0;
            {
                std::lock_guard<std::mutex> lock(mu);
                if (lds > attr_lds) {
                    OBB_HIP(ctx, hipFuncSetAttribute((const void *)k_heavy_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    attr_lds = lds;
                }
            }
            hipLaunchKernelGGL(k_heavy_prep, dim3((unsigned)slots, (unsigned)std::min<int64_t>(16, cdiv(A, 256))), dim3(256), 0, st, head, A, nc, h, w, S);
            OBB_LAUNCH_CHECK(ctx);
            int ncu = 0, dev = 0;
            OBB_HIP(ctx, hipGetDevice(&dev));
            OBB_HIP(ctx, hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
            const int resident = std::max(1, ncu) * (int)std::max<size_t>(1, (160 * 1024) / (lds + 128));
            hipLaunchKernelGGL(k_heavy_rows, dim3((unsigned)std::min<int64_t>(resident, (int64_t)B * kHeavyShareMax)), dim3(kHeavyRowsThreads), lds, st, A, conf_thres, iou_thres,
                               max_det, kq, bdmax, S, out, count);
            OBB_LAUNCH_CHECK(ctx);
            return OBB_OK;
        }
        // (more anchors than that -- inputs above ~500 x 500: rank sort from global scores, rows spread over 64 x 4 waves per tile)
        hipLaunchKernelGGL(k_heavy_sort, dim3((unsigned)slots, (unsigned)std::min<int64_t>(16, cdiv(A, 256))), dim3(256), 0, st, head, A, nc, h, w, 30000, S);
        OBB_LAUNCH_CHECK(ctx);
        hipLaunchKernelGGL(k_heavy_nms, dim3((unsigned)slots, 64), dim3(256), 0, st, A, iou_thres, 30000, kq, bdmax, S);
        OBB_LAUNCH_CHECK(ctx);
    }
    hipLaunchKernelGGL(k_heavy_out, dim3((unsigned)std::min<int>(B, 256)), dim3(256), 0, st, A, max_det, 30000, S, out, count);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

// The full form: decode every anchor, then one workgroup per tile with every per-candidate array resident in LDS (or in global scratch for
// inputs too large).  obb_decode_nms used to run this; it stays as an independent implementation that the parity tests compare the
// candidate-first path with (same rows, bit for bit).
int obb_decode_nms_full(obb_ctx *ctx, const float *head, int32_t B, int32_t h, int32_t w, float conf_thres, float iou_thres, int32_t max_det,
                        float *out, int32_t *count, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && B >= 0 && max_det > 0, "obb_decode_nms_full: bad arguments");
    int32_t nc = 0, A = 0;
    int rc = obb_model_info(ctx, h, w, &nc, nullptr, &A, nullptr);
    if (rc) return rc;
    if (B == 0) return OBB_OK;
    OBB_REQUIRE(ctx, head && out && count, "obb_decode_nms_full: NULL buffer");
    OBB_REQUIRE(ctx, nc <= 255, "obb_decode_nms_full: nc > 255 unsupported");
    float *pred = (float *)ctx->workspace(WS_GEOM_D, sizeof(float) * (size_t)B * A * (4 + nc + 1));
    if (!pred) return set_error(ctx, OBB_ERR_HIP, "obb_decode_nms_full: workspace allocation failed");
    NmsScratch S;
    rc = nms_scratch(ctx, B, A, S);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)s;
    hipLaunchKernelGGL(k_decode, dim3((unsigned)cdiv((int64_t)B * A, 256)), dim3(256), 0, st, head, B, A, nc, h, w, pred, (const int32_t *)nullptr);
    OBB_LAUNCH_CHECK(ctx);
    size_t lds = (size_t)A * (sizeof(RBox) + 12 + 3) + 96;
    if (lds <= 159 * 1024) {
        static bool attr_set = false;
        if (!attr_set) {
            OBB_HIP(ctx, hipFuncSetAttribute((const void *)k_nms_tile<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
            attr_set = true;
        }
        hipLaunchKernelGGL(k_nms_tile<true>, dim3((unsigned)B), dim3(1024), lds, st, (const float *)pred, A, nc, conf_thres, iou_thres, max_det,
                           30000, far_apart_factor(iou_thres), probiou_bdmax(iou_thres), S, out, count, 0);
    } else {
        hipLaunchKernelGGL(k_nms_tile<false>, dim3((unsigned)B), dim3(1024), 0, st, (const float *)pred, A, nc, conf_thres, iou_thres, max_det,
                           30000, far_apart_factor(iou_thres), probiou_bdmax(iou_thres), S, out, count, 0);
    }
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

int obb_probiou_nms(obb_ctx *ctx, const float *boxes, const float *scores, int64_t n, float iou_thres, int32_t *order, uint8_t *keep,
                    obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && n >= 0 && n < (1 << 24), "obb_probiou_nms: bad n");
    if (n == 0) return OBB_OK;
    OBB_REQUIRE(ctx, boxes && scores && order && keep, "obb_probiou_nms: NULL buffer");
    RBox *rb = (RBox *)ctx->workspace(WS_GEOM_A, sizeof(RBox) * (size_t)n);
    if (!rb) return set_error(ctx, OBB_ERR_HIP, "obb_probiou_nms: workspace allocation failed");
    hipLaunchKernelGGL(k_probiou_nms_list, dim3(1), dim3(256), 0, (hipStream_t)s, boxes, scores, (int)n, iou_thres, order, keep, rb);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

int obb_results(obb_ctx *ctx, const float *det, const float *lb, int64_t n, float *xywhr, float *pts, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && n >= 0, "obb_results: bad arguments");
    if (n == 0) return OBB_OK;
    OBB_REQUIRE(ctx, det && xywhr && pts, "obb_results: NULL buffer");
    hipLaunchKernelGGL(k_results, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, det, lb, n, xywhr, pts);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

int obb_gather_tiles(obb_ctx *ctx, const uint8_t *image, int32_t H, int32_t W, int32_t C, const int32_t *rects, int32_t ntiles,
                     int32_t tile, uint8_t *tiles_out, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && H > 0 && W > 0 && (C == 3 || C == 4) && ntiles >= 0 && tile > 0 && tile <= H && tile <= W,
                "obb_gather_tiles: bad arguments");
    if (ntiles == 0) return OBB_OK;
    OBB_REQUIRE(ctx, image && rects && tiles_out, "obb_gather_tiles: NULL buffer");
    OBB_REQUIRE(ctx, ntiles <= 65535 && tile <= 65535, "obb_gather_tiles: too many tiles in one call");
    int rowbytes = tile * C;
    dim3 grid((unsigned)cdiv(cdiv(rowbytes, 4), 256), (unsigned)tile, (unsigned)ntiles);
    hipLaunchKernelGGL(k_gather_tiles, grid, dim3(256), 0, (hipStream_t)s, image, H, W, C, rects, tile, tiles_out);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

int obb_letterbox(obb_ctx *ctx, const uint8_t *image, int32_t H, int32_t W, int32_t C, int32_t x, int32_t y, int32_t x2, int32_t y2,
                  int32_t imgsz, uint8_t *out, int32_t out_h, int32_t out_w, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && image && out && (C == 3 || C == 4) && x >= 0 && y >= 0 && x2 > x && y2 > y && x2 <= W && y2 <= H && imgsz > 0,
                "obb_letterbox: bad arguments");
    int ch = y2 - y, cw = x2 - x;
    double r = std::min((double)imgsz / ch, (double)imgsz / cw);
    LbParams P;
    P.x = x; P.y = y; P.cw = cw; P.ch = ch;
    P.new_w = (int)std::nearbyint(cw * r);  // Python round(): ties to even
    P.new_h = (int)std::nearbyint(ch * r);
    int dw = ((imgsz - P.new_w) % 32 + 32) % 32, dh = ((imgsz - P.new_h) % 32 + 32) % 32;
    double hw = dw / 2.0, hh = dh / 2.0;
    P.top = (int)std::nearbyint(hh - 0.1); P.left = (int)std::nearbyint(hw - 0.1);
    int bottom = (int)std::nearbyint(hh + 0.1), right = (int)std::nearbyint(hw + 0.1);
    P.out_h = P.new_h + P.top + bottom; P.out_w = P.new_w + P.left + right;
    OBB_REQUIRE(ctx, P.out_h == out_h && P.out_w == out_w, "obb_letterbox: output must be %dx%d for this crop, got %dx%d", P.out_h, P.out_w,
                out_h, out_w);
    P.resize = (P.new_w != cw || P.new_h != ch);
    P.sx = (double)cw / P.new_w; P.sy = (double)ch / P.new_h;
    hipLaunchKernelGGL(k_letterbox, dim3((unsigned)cdiv((int64_t)out_h * out_w, 256)), dim3(256), 0, (hipStream_t)s, image, H, W, C, P, out);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

}  // extern "C"
