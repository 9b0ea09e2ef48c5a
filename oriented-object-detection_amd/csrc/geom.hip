// Post-processing geometry kernels (fp64, wave64) + their C-ABI entry points.
//
//   obb_poly_iou_pairs / _matrix   <- compute_polygon_iou                 Detect_OBB.py:144-154
//   obb_sort_desc_stable           <- detections.sort(key=conf, reverse)   Detect_OBB.py:183
//   obb_nms_mask / obb_nms_reduce  <- merge_detections' greedy double loop Detect_OBB.py:186-198
//   obb_merge_detections / _segments  (whole function; per-tile batched form for Detect_OBB.py:264)
//   obb_consensus                  <- cross_scale_consensus_filter         Detect_OBB.py:347-423
//   obb_tile_postprocess           <- per-detection body of detect_symbols Detect_OBB.py:229-262
//
// All of this is byte/compare work on KB..MB inputs: the kernels are HBM/latency bound, so the design rules are
// coalesced SoA streams, LDS-resident segments, wave64 ballots/shuffles for the serial scans, and no GEMM shaping.
#include <algorithm>
#include <mutex>

#include "ctx.h"
#include "geom_device.h"
#include "post_device.h"

namespace obb {


// ------------------------------------------------------------------------------------------------ pair list / matrix

__global__ __launch_bounds__(256) void k_iou_pairs(const double *__restrict__ a, const double *__restrict__ b, int64_t m,
                                                  double *__restrict__ out) {
    // stage 256 pairs x 2 x 64 B through LDS: global reads are contiguous 8 B/lane, LDS rows padded to 9 doubles so
    // that the per-thread row reads (stride 72 B) are bank-conflict free
    // (a lane's row of 18 doubles = quad a | pad | quad b | pad; once both quads are in registers the row is the lane's clip buffer:
    // kClipCap vertices of 16 bytes -- see clip_area_convex)
    __shared__ __attribute__((aligned(16))) double sab[256 * 18];
    int64_t base = (int64_t)blockIdx.x * 256;
    int nloc = (int)((m - base) < 256 ? (m - base) : 256);
    const double *ga = a + base * 8, *gb = b + base * 8;
    for (int e = threadIdx.x; e < nloc * 8; e += 256) {
        sab[(e >> 3) * 18 + (e & 7)] = ga[e];
        sab[(e >> 3) * 18 + 9 + (e & 7)] = gb[e];
    }
    __syncthreads();
    if ((int)threadIdx.x < nloc) {
        double pa[8], pb[8];
        for (int k = 0; k < 8; ++k) { pa[k] = sab[threadIdx.x * 18 + k]; pb[k] = sab[threadIdx.x * 18 + 9 + k]; }
        out[base + threadIdx.x] = poly_iou_lds(pa, pb, reinterpret_cast<P2 *>(sab + threadIdx.x * 18), 1);
    }
}

// out[i*nq + j] = 1 iff point i lies strictly inside the valid quad j (and their classes agree when both class arrays are given)
__global__ __launch_bounds__(256) void k_points_in_quads(const double *__restrict__ pts, const int32_t *__restrict__ cp, int64_t np,
                                                        const double *__restrict__ quads, const int32_t *__restrict__ cq, int64_t nq,
                                                        uint8_t *__restrict__ out) {
    int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= np * nq) return;
    int64_t i = t / nq, j = t - i * nq;
    bool hit = false;
    if (!(cp && cq) || cp[i] == cq[j]) {
        P2 q[4];
        for (int k = 0; k < 4; ++k) { q[k].x = quads[j * 8 + 2 * k]; q[k].y = quads[j * 8 + 2 * k + 1]; }
        P2 c;
        c.x = pts[i * 2]; c.y = pts[i * 2 + 1];
        hit = point_in_quad(q, c);
    }
    out[t] = hit ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_iou_matrix(const double *__restrict__ a, const int32_t *__restrict__ ca, int64_t na,
                                                   const double *__restrict__ b, const int32_t *__restrict__ cb, int64_t nb,
                                                   double *__restrict__ out) {
    // block = 16 rows x 16.. cols tile: thread (ty, tx)
    __shared__ double sa[16 * 8], sb[16 * 8];
    __shared__ int32_t sca[16], scb[16];
    __shared__ __attribute__((aligned(16))) P2 sclip[kClipCap * 256];  // lane-private clip buffers (clip_area_convex)
    int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    int64_t i0 = (int64_t)blockIdx.y * 16, j0 = (int64_t)blockIdx.x * 16;
    if (threadIdx.x < 128) {
        int r = threadIdx.x >> 3, k = threadIdx.x & 7;
        sa[threadIdx.x] = (i0 + r < na) ? a[(i0 + r) * 8 + k] : 0.0;
    } else {
        int t = threadIdx.x - 128;
        int r = t >> 3, k = t & 7;
        sb[t] = (j0 + r < nb) ? b[(j0 + r) * 8 + k] : 0.0;
    }
    if (threadIdx.x < 16) sca[threadIdx.x] = (ca && i0 + threadIdx.x < na) ? ca[i0 + threadIdx.x] : 0;
    else if (threadIdx.x < 32) scb[threadIdx.x - 16] = (cb && j0 + threadIdx.x - 16 < nb) ? cb[j0 + threadIdx.x - 16] : 0;
    __syncthreads();
    int64_t i = i0 + ty, j = j0 + tx;
    if (i < na && j < nb) {
        double v = 0.0;
        if (!(ca && cb) || sca[ty] == scb[tx]) v = poly_iou_lds(&sa[ty * 8], &sb[tx * 8], sclip + threadIdx.x, 256);
        out[i * nb + j] = v;
    }
}

// ------------------------------------------------------------------------------------------------ stable descending rank sort

__device__ __forceinline__ double sort_key(double k) { return isnan(k) ? -INFINITY : k; }

// order[rank(i)] = i, rank(i) = #{j: key_j > key_i} + #{j < i: key_j == key_i}  (stable, descending).
// 2-D decomposition: block (x, y) counts, for its 256 rows i, the keys of j-slice y that precede them; partial counts are
// summed with integer atomics (exact, order-independent), a second pass scatters.  Keys are compared as monotone uint64
// images of the doubles (integer compares; -0.0 == +0.0 and NaN -> -inf handled in the mapping).
__device__ __forceinline__ unsigned long long sort_image(double k) {
    k = sort_key(k);
    if (k == 0.0) k = 0.0;  // -0.0 -> +0.0
    unsigned long long u = (unsigned long long)__double_as_longlong(k);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

static constexpr int kRankJSlice = 2048;

__global__ __launch_bounds__(256) void k_rank_count(const double *__restrict__ key, int64_t n, int32_t *__restrict__ rank) {
    __shared__ unsigned long long sk[kRankJSlice];
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t j0 = (int64_t)blockIdx.y * kRankJSlice;
    int cnt = (int)((n - j0) < kRankJSlice ? (n - j0) : kRankJSlice);
    for (int t = threadIdx.x; t < cnt; t += 256) sk[t] = sort_image(key[j0 + t]);
    __syncthreads();
    if (i >= n) return;
    unsigned long long ki = sort_image(key[i]);
    int32_t r = 0;
    // ties: j < i counts.  The slice lies entirely before i, entirely after, or straddles it.
    if (j0 + cnt <= i) { for (int t = 0; t < cnt; ++t) r += (sk[t] >= ki); }
    else if (j0 > i) { for (int t = 0; t < cnt; ++t) r += (sk[t] > ki); }
    else { for (int t = 0; t < cnt; ++t) r += (sk[t] > ki) | ((sk[t] == ki) & (j0 + t < i)); }
    if (r) atomicAdd(&rank[i], r);
}

__global__ __launch_bounds__(256) void k_rank_scatter(const int32_t *__restrict__ rank, int64_t n, int32_t *__restrict__ order) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) order[rank[i]] = (int32_t)i;
}

// ------------------------------------------------------------------------------------------------ dense NMS (n > kSegMax)

struct BoxMeta { double x0, y0, x1, y1; };  // envelope; invalid quads get an empty envelope (x0 > x1)

// gather into sorted order + envelope/validity precompute
__global__ __launch_bounds__(256) void k_prep_sorted(const double *__restrict__ boxes, const int32_t *__restrict__ cls,
                                                    const int32_t *__restrict__ order, int64_t n,
                                                    double *__restrict__ sboxes, int32_t *__restrict__ scls,
                                                    BoxMeta *__restrict__ meta) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int64_t src = order ? order[i] : i;
    P2 p[4];
    for (int k = 0; k < 4; ++k) { p[k].x = boxes[src * 8 + 2 * k]; p[k].y = boxes[src * 8 + 2 * k + 1]; }
    if (sboxes)
        for (int k = 0; k < 4; ++k) { sboxes[i * 8 + 2 * k] = p[k].x; sboxes[i * 8 + 2 * k + 1] = p[k].y; }
    if (scls) scls[i] = cls[src];
    BoxMeta m;
    if (quad_valid(p)) { Aabb a = quad_aabb(p); m.x0 = a.x0; m.y0 = a.y0; m.x1 = a.x1; m.y1 = a.y1; }
    else { m.x0 = 1.0; m.x1 = -1.0; m.y0 = 1.0; m.y1 = -1.0; }
    meta[i] = m;
}

__device__ __forceinline__ bool meta_overlap(const BoxMeta &a, const BoxMeta &b) {
    // both valid (x0 <= x1) and envelopes not strictly disjoint
    return a.x0 <= a.x1 && b.x0 <= b.x1 && !(a.x1 < b.x0 || b.x1 < a.x0 || a.y1 < b.y0 || b.y1 < a.y0);
}

// ---- bucketed form of the same sort (n >= kBucketSortMin): O(n) instead of O(n^2) compares.
// The monotone uint64 images are spread over kSortBuckets equal key ranges between the batch's smallest and largest image (a
// data-adaptive most-significant digit: confidences live in [0.25, 1], a fixed digit of the float format would use a handful of
// buckets).  rank(i) = (elements in higher buckets) + (elements of i's own bucket that precede it: larger key, or equal key and smaller
// index) -- the same stable descending order, element for element.  Buckets hold n / 65536 elements on average, so the in-bucket count
// is a few compares; a batch of all-equal keys degenerates to one bucket = the O(n^2) count it replaces.
static constexpr int kSortBuckets = 65536;
static constexpr int64_t kBucketSortMin = 8192;

struct SortInfo { unsigned long long lo; int shift; };

// smallest / largest key image in two steps (kSortParts workgroups reduce a slice each, one thread combines: exact, order-independent)
static constexpr int kSortParts = 64;
__global__ __launch_bounds__(1024) void k_sort_range_part(const double *__restrict__ key, int64_t n, unsigned long long *__restrict__ part /* [kSortParts][2] */) {
    __shared__ unsigned long long smin[1024], smax[1024];
    unsigned long long lo = ~0ull, hi = 0ull;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x, b0 = (int64_t)blockIdx.x * per, b1 = b0 + per < n ? b0 + per : n;
    for (int64_t i = b0 + threadIdx.x; i < b1; i += 1024) { const unsigned long long u = sort_image(key[i]); lo = u < lo ? u : lo; hi = u > hi ? u : hi; }
    smin[threadIdx.x] = lo; smax[threadIdx.x] = hi;
    __syncthreads();
    for (int d = 512; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) {
            if (smin[threadIdx.x + d] < smin[threadIdx.x]) smin[threadIdx.x] = smin[threadIdx.x + d];
            if (smax[threadIdx.x + d] > smax[threadIdx.x]) smax[threadIdx.x] = smax[threadIdx.x + d];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = smin[0]; part[2 * blockIdx.x + 1] = smax[0]; }
}

__global__ __launch_bounds__(64) void k_sort_range(const unsigned long long *__restrict__ part, int nparts, SortInfo *__restrict__ info) {
    if (threadIdx.x != 0) return;
    unsigned long long lo = ~0ull, hi = 0ull;
    for (int k = 0; k < nparts; ++k) { lo = part[2 * k] < lo ? part[2 * k] : lo; hi = part[2 * k + 1] > hi ? part[2 * k + 1] : hi; }
    const unsigned long long range = hi >= lo ? hi - lo : 0ull;
    int sh = 0;
    while (sh < 63 && (range >> sh) >= (unsigned long long)kSortBuckets) ++sh;
    info->lo = lo; info->shift = sh;
}

__device__ __forceinline__ int sort_bucket(unsigned long long u, const SortInfo &I) {
    return kSortBuckets - 1 - (int)((u - I.lo) >> I.shift);  // bucket 0 holds the LARGEST keys (descending order)
}

__global__ __launch_bounds__(256) void k_sort_count(const double *__restrict__ key, int64_t n, const SortInfo *__restrict__ info, int32_t *__restrict__ hist) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    atomicAdd(&hist[sort_bucket(sort_image(key[i]), *info)], 1);
}

// exclusive prefix sum of `cnt` cells (one workgroup); start[cnt] = total; cursor = copy of start (scatter positions; may alias hist).
// Each of the 16 waves owns a contiguous segment and walks it 64 cells at a time (coalesced): segment sums -> scan of the 16 sums ->
// a second walk with a wave prefix scan per 64 cells and a running carry.
__global__ __launch_bounds__(1024) void k_cells_scan(const int32_t *hist, int cnt, int32_t *__restrict__ start, int32_t *cursor) {
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // a lane takes FOUR consecutive cells per step (one 16-byte load when the segment is aligned: cnt is 65 536 in both callers): a quarter of
    // the steps of the one-cell-per-lane walk, whose 64 dependent load -> scan -> store rounds per wave were 38 us for 65 536 cells
    const int per = ((cnt + 15) / 16 + 255) / 256 * 256;  // cells per wave, a multiple of 256
    const int c0 = wave * per, c1 = min(c0 + per, cnt);
    const bool vec = ((reinterpret_cast<uintptr_t>(hist) | reinterpret_cast<uintptr_t>(start) | reinterpret_cast<uintptr_t>(cursor)) & 15) == 0;
    auto load4 = [&](int c, int v[4]) {
        if (vec && c + 3 < c1) { const int4 q = *reinterpret_cast<const int4 *>(hist + c); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
        else for (int k = 0; k < 4; ++k) v[k] = (c + k < c1) ? hist[c + k] : 0;
    };
    int local = 0;
    for (int c = c0 + lane * 4; c < c1; c += 256) { int v[4]; load4(c, v); local += v[0] + v[1] + v[2] + v[3]; }
    for (int d = 32; d >= 1; d >>= 1) local += __shfl_xor(local, d);
    if (lane == 0) wsum[wave] = local;
    __syncthreads();
    int carry = 0;
    for (int w = 0; w < wave; ++w) carry += wsum[w];
    for (int cb = c0; cb < c1; cb += 256) {
        const int c = cb + lane * 4;
        int v[4];
        load4(c, v);  // (hist may alias cursor: every cell of this step is read before any is written -- the wave runs in lockstep)
        const int s4 = v[0] + v[1] + v[2] + v[3];
        int incl = s4;
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
        int e = carry + incl - s4;  // exclusive prefix of this lane's first cell
        int o[4];
        for (int k = 0; k < 4; ++k) { o[k] = e; e += v[k]; }
        if (vec && c + 3 < c1) {
            *reinterpret_cast<int4 *>(start + c) = make_int4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<int4 *>(cursor + c) = make_int4(o[0], o[1], o[2], o[3]);
        } else for (int k = 0; k < 4; ++k) if (c + k < c1) { start[c + k] = o[k]; cursor[c + k] = o[k]; }
        carry += __shfl(incl, 63);
    }
    if (tid == 0) { int tot = 0; for (int w = 0; w < 16; ++w) tot += wsum[w]; start[cnt] = tot; }
}

__global__ __launch_bounds__(256) void k_sort_scatter(const double *__restrict__ key, int64_t n, const SortInfo *__restrict__ info, int32_t *__restrict__ cursor,
                                                     int32_t *__restrict__ members) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    members[atomicAdd(&cursor[sort_bucket(sort_image(key[i]), *info)], 1)] = (int32_t)i;
}

// rank inside the bucket: a thread per row walks its bucket's members.  Buckets above kSortHot members (many EQUAL keys: confidences that
// went through 16-bit arithmetic take a few thousand distinct values) would cost members^2 thread-serial steps: those are listed instead
// (by the row that is the bucket's first member) and ranked by k_sort_hot, a whole workgroup per bucket.
static constexpr int kSortHot = 192, kSortHotMax = 4096, kSortHotTile = 4096;
__global__ __launch_bounds__(256) void k_sort_rank(const double *__restrict__ key, int64_t n, const SortInfo *__restrict__ info, const int32_t *__restrict__ start,
                                                  const int32_t *__restrict__ members, int32_t *__restrict__ order, int32_t *__restrict__ hot /* [0] = count, then bucket ids */) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long ki = sort_image(key[i]);
    const int b = sort_bucket(ki, *info);
    const int lo = start[b], hi = start[b + 1];
    if (hi - lo > kSortHot) {
        if (members[lo] == (int32_t)i) {  // one work item per 1024 members of the bucket
            const int nchunk = (hi - lo + 1023) >> 10;
            const int slot = atomicAdd(&hot[0], nchunk);
            for (int c = 0; c < nchunk && slot + c < kSortHotMax; ++c) { hot[1 + 2 * (slot + c)] = b; hot[2 + 2 * (slot + c)] = c; }
        }
        return;
    }
    int r = lo;
    for (int m = lo; m < hi; ++m) {
        const int j = members[m];
        const unsigned long long kj = sort_image(key[j]);
        r += (kj > ki) | ((kj == ki) & (j < (int)i));
    }
    order[r] = (int32_t)i;
}

// one workgroup per listed work item = 1024 members of a hot bucket (grid-stride over the list): the bucket's (key image, row) pairs go
// through LDS in tiles and every thread ranks its member against each tile (LDS broadcasts).  The list cannot overflow: an item stands
// for more than kSortHot rows or a full 1024, and the host takes this path for n <= kSortHot * kSortHotMax only.
__global__ __launch_bounds__(1024) void k_sort_hot(const double *__restrict__ key, const SortInfo *__restrict__ info, const int32_t *__restrict__ start,
                                                  const int32_t *__restrict__ members, int32_t *__restrict__ order, const int32_t *__restrict__ hot) {
    __shared__ unsigned long long tk[kSortHotTile];
    __shared__ int32_t tj[kSortHotTile];
    const int nhot = min(hot[0], kSortHotMax);
    for (int h = blockIdx.x; h < nhot; h += gridDim.x) {
        const int b = hot[1 + 2 * h], lo = start[b], hi = start[b + 1];
        const int m = lo + hot[2 + 2 * h] * 1024 + (int)threadIdx.x;
        const int ii = m < hi ? members[m] : -1;
        const unsigned long long ki = ii >= 0 ? sort_image(key[ii]) : 0ull;
        int r = lo;
        for (int t0 = lo; t0 < hi; t0 += kSortHotTile) {
            __syncthreads();
            for (int t = threadIdx.x; t < kSortHotTile && t0 + t < hi; t += 1024) {
                const int j = members[t0 + t];
                tj[t] = j; tk[t] = sort_image(key[j]);
            }
            __syncthreads();
            const int cnt = min(kSortHotTile, hi - t0);
#pragma unroll 4
            for (int t = 0; t < cnt; ++t) {
                const unsigned long long kj = tk[t];
                r += (kj > ki) | ((kj == ki) & (tj[t] < ii));
            }
        }
        if (ii >= 0) order[r] = ii;
        __syncthreads();
    }
}

// ---- suppression pairs of the dense merge through a uniform grid (n >= kGridMin, thr > 0): two envelopes can only overlap if their
// centres are closer than the larger of the two extents in both axes, so with cells at least as large as every SMALL box's extent every
// small partner of a small box sits in the 3 x 3 cells around it.  The cell is min(largest extent, 4 x mean extent) (never below
// span / 255): boxes larger than the cell ("large", at most a quarter of the rows by Markov, a handful on real maps) stay out of the
// grid and are tested against every row instead, so a few outsized quads cannot blow the cells up to hundreds of members.  Replaces the all-pairs envelope test (n^2 / 2) of k_nms_mask<EDGES> by ~9 cells x occupancy tests per box;
// the exact IoU runs on the same candidate set (same class, envelopes not disjoint), so the edge list is the same set of pairs.
static constexpr int kGridDim = 256;  // cells per axis at most
static constexpr int64_t kGridMin = 8192;

struct GridInfo { double ox, oy, inv_cell, small_max; int gw, gh, n_large, pad; };

// extent statistics of the envelopes in two steps: kGridParts workgroups reduce a slice each (one workgroup reading all 2 MB of a 65 536-row
// set took 42 us), a wave combines the partial results in a fixed order.  (The sum of the extents only sizes the cells: the candidate set, and
// with it every result, does not depend on the cell size -- see the comment on the grid below.)
static constexpr int kGridParts = 64;
__global__ __launch_bounds__(1024) void k_grid_info_part(const BoxMeta *__restrict__ meta, int64_t n, double *__restrict__ part /* [kGridParts][7] */) {
    __shared__ double s0[1024], s1[1024], s2[1024], s3[1024], s4[1024], s5[1024], s6[1024];
    double x0 = INFINITY, y0 = INFINITY, x1 = -INFINITY, y1 = -INFINITY, ext = 0.0, esum = 0.0, ecnt = 0.0;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x, lo = (int64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 1024) {
        const BoxMeta m = meta[i];
        if (!(m.x0 <= m.x1)) continue;  // invalid quad: empty envelope, never a partner
        x0 = fmin(x0, m.x0); y0 = fmin(y0, m.y0); x1 = fmax(x1, m.x1); y1 = fmax(y1, m.y1);
        const double e = fmax(m.x1 - m.x0, m.y1 - m.y0);
        ext = fmax(ext, e); esum += e; ecnt += 1.0;
    }
    s0[threadIdx.x] = x0; s1[threadIdx.x] = y0; s2[threadIdx.x] = x1; s3[threadIdx.x] = y1; s4[threadIdx.x] = ext;
    s5[threadIdx.x] = esum; s6[threadIdx.x] = ecnt;
    __syncthreads();
    for (int d = 512; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) {
            s0[threadIdx.x] = fmin(s0[threadIdx.x], s0[threadIdx.x + d]); s1[threadIdx.x] = fmin(s1[threadIdx.x], s1[threadIdx.x + d]);
            s2[threadIdx.x] = fmax(s2[threadIdx.x], s2[threadIdx.x + d]); s3[threadIdx.x] = fmax(s3[threadIdx.x], s3[threadIdx.x + d]);
            s4[threadIdx.x] = fmax(s4[threadIdx.x], s4[threadIdx.x + d]);
            s5[threadIdx.x] += s5[threadIdx.x + d]; s6[threadIdx.x] += s6[threadIdx.x + d];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double *o = part + (size_t)blockIdx.x * 7;
        o[0] = s0[0]; o[1] = s1[0]; o[2] = s2[0]; o[3] = s3[0]; o[4] = s4[0]; o[5] = s5[0]; o[6] = s6[0];
    }
}

__global__ __launch_bounds__(64) void k_grid_info(const double *__restrict__ part, int nparts, GridInfo *__restrict__ info) {
    if (threadIdx.x != 0) return;
    double x0 = INFINITY, y0 = INFINITY, x1 = -INFINITY, y1 = -INFINITY, ext = 0.0, esum = 0.0, ecnt = 0.0;
    for (int k = 0; k < nparts; ++k) {  // (fixed order: deterministic)
        const double *o = part + (size_t)k * 7;
        x0 = fmin(x0, o[0]); y0 = fmin(y0, o[1]); x1 = fmax(x1, o[2]); y1 = fmax(y1, o[3]); ext = fmax(ext, o[4]); esum += o[5]; ecnt += o[6];
    }
    GridInfo g;
    g.n_large = 0; g.pad = 0; g.small_max = 0.0;
    if (!(x0 <= x1)) { g.ox = g.oy = 0.0; g.inv_cell = 0.0; g.gw = g.gh = 1; }
    else {
        const double wx = x1 - x0, wy = y1 - y0;
        double cell = fmax(fmin(ext, 4.0 * esum / ecnt), fmax(wx, wy) / (double)(kGridDim - 1));
        g.small_max = cell;  // rows with a larger extent are "large"
        cell = cell > 0.0 ? cell * (1.0 + 1e-9) : 1.0;  // (a hair larger than the largest small extent: centre distance < cell for overlapping small envelopes)
        g.ox = x0; g.oy = y0; g.inv_cell = 1.0 / cell;
        g.gw = min(kGridDim, (int)(wx / cell) + 1); g.gh = min(kGridDim, (int)(wy / cell) + 1);
    }
    *info = g;
}

__device__ __forceinline__ int grid_cell(const BoxMeta &m, const GridInfo &g, int &cx, int &cy) {
    cx = (int)(((m.x0 + m.x1) * 0.5 - g.ox) * g.inv_cell); cy = (int)(((m.y0 + m.y1) * 0.5 - g.oy) * g.inv_cell);
    cx = cx < 0 ? 0 : (cx >= g.gw ? g.gw - 1 : cx); cy = cy < 0 ? 0 : (cy >= g.gh ? g.gh - 1 : cy);
    return cy * g.gw + cx;
}

__device__ __forceinline__ bool grid_is_large(const BoxMeta &m, const GridInfo &g) { return fmax(m.x1 - m.x0, m.y1 - m.y0) > g.small_max; }

__global__ __launch_bounds__(256) void k_grid_count(const BoxMeta *__restrict__ meta, int64_t n, GridInfo *__restrict__ info, int32_t *__restrict__ hist,
                                                   int32_t *__restrict__ large) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const BoxMeta m = meta[i];
    if (!(m.x0 <= m.x1)) return;
    if (grid_is_large(m, *info)) { large[atomicAdd(&info->n_large, 1)] = (int32_t)i; return; }
    int cx, cy;
    atomicAdd(&hist[grid_cell(m, *info, cx, cy)], 1);
}

__global__ __launch_bounds__(256) void k_grid_scatter(const BoxMeta *__restrict__ meta, int64_t n, const GridInfo *__restrict__ info, int32_t *__restrict__ cursor,
                                                     int32_t *__restrict__ members) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const BoxMeta m = meta[i];
    if (!(m.x0 <= m.x1) || grid_is_large(m, *info)) return;
    int cx, cy;
    members[atomicAdd(&cursor[grid_cell(m, *info, cx, cy)], 1)] = (int32_t)i;
}

// WAVE = box j (sorted position).  Small j: small partners i < j in the 3 x 3 cells around it.  Every j: the large rows (a small j takes
// each large row, a large j the large rows before it, so every pair with a large member is seen once).  The 64 lanes walk a cell's member
// list together (class + envelope test each); survivors are compacted into the wave's LDS queue (ballot + prefix count, no atomics) and
// clipped 64 at a time, one pair per lane: the exact IoU is ~100x a cheap test and must not run in one lane of a wave, and on dense maps
// a box has hundreds of candidates, which a thread per box would walk serially.  The queue carries over from box to box.
// Edge (lo, hi) when IoU(lo, hi) >= thr, the earlier row always the first operand (as k_nms_mask evaluates it).
__global__ __launch_bounds__(256) void k_grid_pairs(const double *__restrict__ sboxes, const int32_t *__restrict__ scls, const BoxMeta *__restrict__ meta, int64_t n,
                                                   double thr, const GridInfo *__restrict__ info, const int32_t *__restrict__ start,
                                                   const int32_t *__restrict__ members, const int32_t *__restrict__ large,
                                                   unsigned long long *__restrict__ edges, unsigned int *__restrict__ edge_count, unsigned int edge_cap) {
    __shared__ unsigned long long queue[4][128];
    __shared__ __attribute__((aligned(16))) P2 sclip[kClipCap * 256];  // lane-private clip buffers (clip_area_convex)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const GridInfo g = *info;
    unsigned long long *q = queue[wave];
    int qn = 0;  // wave-uniform fill of the queue (< 64 between pushes)
    auto clip64 = [&](int cnt) {  // clips the last `cnt` (<= 64) queued pairs, one per lane
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (lane < cnt) {
            const unsigned long long pr = q[qn - cnt + lane];
            const int64_t lo = (int64_t)(pr >> 32), hi = (int64_t)(pr & 0xffffffffull);
            P2 a[4], b[4];
            for (int k = 0; k < 4; ++k) {
                a[k].x = sboxes[lo * 8 + 2 * k]; a[k].y = sboxes[lo * 8 + 2 * k + 1];
                b[k].x = sboxes[hi * 8 + 2 * k]; b[k].y = sboxes[hi * 8 + 2 * k + 1];
            }
            if (poly_iou_core_lds(a, b, sclip + threadIdx.x, 256) >= thr) {
                const unsigned int slot = atomicAdd(edge_count, 1u);
                if (slot < edge_cap) edges[slot] = pr;
            }
        }
        qn -= cnt;
        __builtin_amdgcn_wave_barrier();
    };
    for (int64_t j = (int64_t)blockIdx.x * 4 + wave; j < n; j += (int64_t)gridDim.x * 4) {
        const BoxMeta mj = meta[j];
        if (!(mj.x0 <= mj.x1)) continue;
        const int cj = scls[j];
        const bool j_large = grid_is_large(mj, g);
        int cx = 0, cy = 0;
        grid_cell(mj, g, cx, cy);
        // The members of the 3 x 3 cells around j (small j only) and the large rows as ONE index space: lane k < 9 fetches its cell's range, the
        // ranges are laid end to end and walked 64 candidates at a time -- one round of (member, class + envelope) loads per 64 candidates
        // instead of one per cell (a row's nine cells hold a handful of members each: nine dependent load chains per row were the kernel's time)
        int cs = 0, len = 0;
        if (!j_large && lane < 9) {
            const int yy = cy - 1 + lane / 3, xx = cx - 1 + lane % 3;
            if (yy >= 0 && yy < g.gh && xx >= 0 && xx < g.gw) { const int c = yy * g.gw + xx; cs = start[c]; len = start[c + 1] - cs; }
        }
        int CS[9], PR[9], ncell_m = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) { CS[k] = __shfl(cs, k); PR[k] = ncell_m; ncell_m += __shfl(len, k); }
        const int total = ncell_m + g.n_large;
        for (int base = 0; base < total; base += 64) {
            const int f = base + lane;
            int i = -1;
            bool take = false;
            if (f < ncell_m) {
                int m = CS[0] + f;
#pragma unroll
                for (int k = 1; k < 9; ++k) if (f >= PR[k]) m = CS[k] + (f - PR[k]);
                i = members[m];
                take = i < (int)j;
            } else if (f < total) {
                i = large[f - ncell_m];
                take = i != (int)j && (!j_large || i < (int)j);
            }
            bool pass = false;
            if (i >= 0) pass = take && scls[i] == cj && meta_overlap(meta[i], mj);
            const unsigned long long bal = __ballot(pass);
            if (bal) {
                if (pass) {
                    const unsigned long long lo = i < (int)j ? (unsigned long long)i : (unsigned long long)j, hi = i < (int)j ? (unsigned long long)j : (unsigned long long)i;
                    q[qn + __popcll(bal & ((1ull << lane) - 1ull))] = (lo << 32) | hi;
                }
                qn += __popcll(bal);
                if (qn >= 64) clip64(64);
            }
        }
    }
    if (qn > 0) clip64(qn);
}

// One wave per 64x64 block of the (row i, col j>i) pair matrix.  Phase 1: every lane runs the 64 cheap
// class+envelope tests of its row.  Phase 2: the surviving pairs are compacted through LDS and clipped one per lane.
// Output word layout is column-block-major: word (row i, block jb) lives at mask[jb * n + i] so that both this
// kernel's stores and the scan kernel's loads are 512 B contiguous per wave.
// EDGES: instead of the dense matrix, append one (i, j) record per suppression pair to an edge list (sparse form used by
// obb_merge_detections; the pair set is tiny compared with n^2 / 64 words).
template <bool EDGES>
__global__ __launch_bounds__(64) void k_nms_mask(const double *__restrict__ sboxes, const int32_t *__restrict__ scls,
                                                const BoxMeta *__restrict__ meta, int64_t n, double thr,
                                                unsigned long long *__restrict__ mask, unsigned long long *__restrict__ edges,
                                                unsigned int *__restrict__ edge_count, unsigned int edge_cap) {
    int jb = blockIdx.x, ib = blockIdx.y;
    if (jb < ib) return;
    __shared__ BoxMeta cm[64];
    __shared__ int32_t cc[64];
    __shared__ unsigned long long bits[64];
    __shared__ unsigned short plist[4096];
    __shared__ __attribute__((aligned(16))) P2 sclip[kClipCap * 64];  // lane-private clip buffers (clip_area_convex)
    int lane = threadIdx.x;
    int64_t i = (int64_t)ib * 64 + lane, j = (int64_t)jb * 64 + lane;
    BoxMeta rm;
    int32_t rc = -1;
    if (i < n) { rm = meta[i]; rc = scls[i]; } else { rm.x0 = 1.0; rm.x1 = -1.0; rm.y0 = rm.y1 = 0.0; }
    if (j < n) { cm[lane] = meta[j]; cc[lane] = scls[j]; } else { cm[lane].x0 = 1.0; cm[lane].x1 = -1.0; cm[lane].y0 = cm[lane].y1 = 0.0; cc[lane] = -2; }
    bits[lane] = 0ull;
    __syncthreads();
    unsigned long long cand = 0ull, direct = 0ull;
    bool all_hit = !(thr > 0.0);  // IoU >= thr holds for every same-class pair when thr <= 0 (IoU is never negative)
    for (int c = 0; c < 64; ++c) {
        int64_t jj = (int64_t)jb * 64 + c;
        bool upper = (jj > i) && (jj < n) && (i < n);
        bool same = upper && (cc[c] == rc);
        if (all_hit) direct |= (unsigned long long)same << c;
        else cand |= (unsigned long long)(same && meta_overlap(rm, cm[c])) << c;
    }
    // wave-wide exclusive prefix of popcounts -> compact (row, col) list
    int cnt = __popcll(cand);
    int incl = cnt;
    for (int d = 1; d < 64; d <<= 1) {
        int v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    int total = __shfl(incl, 63);
    int pos = incl - cnt;
    unsigned long long cw = cand;
    while (cw) {
        int c = __ffsll((long long)cw) - 1;
        cw &= cw - 1;
        plist[pos++] = (unsigned short)((lane << 6) | c);
    }
    __syncthreads();
    for (int base = 0; base < total; base += 64) {
        int t = base + lane;
        if (t < total) {
            int r = plist[t] >> 6, c = plist[t] & 63;
            const double *pa = sboxes + ((int64_t)ib * 64 + r) * 8;
            const double *pb = sboxes + ((int64_t)jb * 64 + c) * 8;
            P2 p[4], q[4];
            for (int k = 0; k < 4; ++k) { p[k].x = pa[2 * k]; p[k].y = pa[2 * k + 1]; q[k].x = pb[2 * k]; q[k].y = pb[2 * k + 1]; }
            double iou = poly_iou_core_lds(p, q, sclip + lane, 64);
            if (iou >= thr) {
                if constexpr (EDGES) {
                    unsigned int e = atomicAdd(edge_count, 1u);
                    if (e < edge_cap) edges[e] = ((unsigned long long)((int64_t)ib * 64 + r) << 32) | (unsigned long long)((int64_t)jb * 64 + c);
                } else atomicOr(&bits[r], 1ull << c);
            }
        }
    }
    if constexpr (EDGES) {
        unsigned long long dw = direct;  // thr <= 0: every same-class pair is an edge
        while (dw) {
            int c = __ffsll((long long)dw) - 1;
            dw &= dw - 1;
            unsigned int e = atomicAdd(edge_count, 1u);
            if (e < edge_cap) edges[e] = ((unsigned long long)i << 32) | (unsigned long long)((int64_t)jb * 64 + c);
        }
    } else {
        __syncthreads();
        if (i < n) mask[(int64_t)jb * n + i] = bits[lane] | direct;
    }
}

// Greedy resolution on the sparse suppression graph (edges i -> j, i < j in confidence order):
//   keep[j] <=> every predecessor of j is dropped;  drop[j] <=> some predecessor is kept.
// The graph is a DAG ordered by index, so relaxing all undecided nodes in parallel reaches the greedy fixed point of
// Detect_OBB.py:186-198 in (longest suppression chain) rounds.  Single workgroup; node states live in LDS, 1 byte each:
// bits 0-1 = state (0 undecided, 1 keep, 2 drop), bit 2 = "a kept predecessor seen", bit 3 = "an undecided predecessor seen".
__global__ __launch_bounds__(1024) void k_nms_resolve(const unsigned long long *__restrict__ edges, const unsigned int *__restrict__ edge_count, unsigned int edge_cap,
                                                     int64_t n, uint8_t *__restrict__ keep, int32_t *__restrict__ n_keep) {
    extern __shared__ unsigned int st_words[];  // ceil(n/4) words
    __shared__ int undecided_s, total_s;
    const int tid = threadIdx.x;
    const int nw = (int)((n + 3) / 4);
    const unsigned int E = *edge_count;
    if (E > edge_cap) return;  // the pair list overflowed (edges past the capacity were dropped): k_nms_lazy, launched right behind, takes over
    for (int w = tid; w < nw; w += 1024) st_words[w] = 0u;
    if (tid == 0) total_s = 0;
    __syncthreads();
    for (int round = 0; round <= n; ++round) {
        if (tid == 0) undecided_s = 0;
        for (unsigned int e = tid; e < E; e += 1024) {
            unsigned long long ed = edges[e];
            unsigned int i = (unsigned int)(ed >> 32), j = (unsigned int)ed;
            unsigned int sj = (st_words[j >> 2] >> ((j & 3) * 8)) & 3u;
            if (sj != 0u) continue;
            unsigned int si = (st_words[i >> 2] >> ((i & 3) * 8)) & 3u;
            if (si == 1u) atomicOr(&st_words[j >> 2], 4u << ((j & 3) * 8));
            else if (si == 0u) atomicOr(&st_words[j >> 2], 8u << ((j & 3) * 8));
        }
        __syncthreads();
        int local_und = 0;
        for (int w = tid; w < nw; w += 1024) {
            unsigned int v = st_words[w], o = 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                unsigned int s = (v >> (k * 8)) & 0xffu;
                if ((s & 3u) == 0u) {
                    if (s & 4u) s = 2u;
                    else if (!(s & 8u)) s = 1u;
                    else { s = 0u; if ((int64_t)w * 4 + k < n) local_und = 1; }
                }
                o |= s << (k * 8);
            }
            st_words[w] = o;
        }
        if (local_und) undecided_s = 1;  // benign race: all writers store 1
        __syncthreads();
        if (!undecided_s) break;
        __syncthreads();
    }
    int cnt = 0;
    for (int64_t i = tid; i < n; i += 1024) {
        unsigned int s = (st_words[i >> 2] >> ((i & 3) * 8)) & 3u;
        keep[i] = (uint8_t)(s == 1u);
        cnt += (s == 1u);
    }
    atomicAdd(&total_s, cnt);
    __syncthreads();
    if (tid == 0 && n_keep) *n_keep = total_s;
}

// Greedy scan (single workgroup, 1024 threads = 16 waves).  Chunk rb of 64 rows: wave 0 resolves the diagonal block
// serially with scalar ops, then all waves OR the kept rows' words into `removed` for the later column blocks.
__global__ __launch_bounds__(1024) void k_nms_reduce(const unsigned long long *__restrict__ mask, int64_t n,
                                                    uint8_t *__restrict__ keep, int32_t *__restrict__ n_keep) {
    extern __shared__ unsigned long long removed[];  // W words
    __shared__ unsigned long long keepmask_s;
    __shared__ int total_s;
    int W = (int)((n + 63) / 64);
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int w = threadIdx.x; w < W; w += 1024) removed[w] = 0ull;
    if (threadIdx.x == 0) total_s = 0;
    __syncthreads();
    for (int rb = 0; rb < W; ++rb) {
        if (wave == 0) {
            int64_t i = (int64_t)rb * 64 + lane;
            unsigned long long d = (i < n) ? mask[(int64_t)rb * n + i] : 0ull;
            unsigned long long rem = removed[rb];
            int nvalid = (int)((n - (int64_t)rb * 64) < 64 ? (n - (int64_t)rb * 64) : 64);
            unsigned long long km = 0ull;
            for (int r = 0; r < nvalid; ++r) {
                unsigned long long dr = __shfl(d, r);  // wave-uniform
                if (!((rem >> r) & 1ull)) { km |= 1ull << r; rem |= dr; }
            }
            if (i < n) keep[i] = (uint8_t)((km >> lane) & 1ull);
            if (lane == 0) { keepmask_s = km; total_s += __popcll(km); }
        }
        __syncthreads();
        unsigned long long km = keepmask_s;
        for (int cb = rb + 1 + wave; cb < W; cb += 16) {
            int64_t i = (int64_t)rb * 64 + lane;
            unsigned long long w = (i < n && ((km >> lane) & 1ull)) ? mask[(int64_t)cb * n + i] : 0ull;
            for (int d = 32; d >= 1; d >>= 1) w |= __shfl_xor(w, d);
            if (lane == 0) removed[cb] |= w;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && n_keep) *n_keep = total_s;
}

// Greedy scan WITHOUT a stored pair matrix: the fall-back of the sparse form for the inputs it cannot take -- a pair list that overflowed its
// capacity (more than 64 n + 4096 suppressing pairs: piles of near-identical boxes), thr <= 0 (every same-class pair suppresses), or more rows
// than k_nms_resolve's LDS states hold.  Predicated on the DEVICE (`force`, or *edge_count > edge_cap): the host launches it unconditionally
// behind k_nms_resolve and never reads the pair count, so obb_merge_detections has no host synchronisation and can be stream-captured.
// One workgroup walks the rows in blocks of 64 (sorted order): (1) the block's 64 x 64 upper-triangular pair tests (two per thread) -> 64
// suppression words in LDS; (2) wave 0 resolves the block serially against `removed` (as k_nms_reduce does); (3) all 1024 threads test
// the not-yet-removed later rows against the block's KEPT rows only and mark the hits in `removed` (1 bit per row, LDS: n <= 1.2 M).
// Same pair predicate as k_nms_mask / k_grid_pairs (class, envelope, exact IoU with the earlier row as first operand), so the result is
// the same greedy fixed point; the cost is kept-rows x later-rows cheap tests, paid only by the rare inputs named above.
__global__ __launch_bounds__(1024) void k_nms_lazy(const double *__restrict__ sboxes, const int32_t *__restrict__ scls, const BoxMeta *__restrict__ meta, int64_t n,
                                                  double thr, const unsigned int *__restrict__ edge_count, unsigned int edge_cap, int force,
                                                  uint8_t *__restrict__ keep, int32_t *__restrict__ n_keep) {
    if (!force && *edge_count <= edge_cap) return;
    extern __shared__ unsigned long long removed[];  // ceil(n / 64) words
    __shared__ BoxMeta bm[64];
    __shared__ int32_t bc[64];
    __shared__ double bq[64][8];
    __shared__ unsigned long long diag[64];
    __shared__ unsigned long long keepmask_s;
    __shared__ int total_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int W = (int)((n + 63) / 64);
    const bool all_hit = !(thr > 0.0);
    for (int w = tid; w < W; w += 1024) removed[w] = 0ull;
    if (tid == 0) total_s = 0;
    __syncthreads();
    auto hit = [&](int r, const BoxMeta &mj, int32_t cj, int64_t j) -> bool {  // does block row r (earlier) suppress row j?
        if (bc[r] != cj) return false;
        if (all_hit) return true;
        if (!meta_overlap(bm[r], mj)) return false;
        P2 a[4], b[4];
        for (int k = 0; k < 4; ++k) { a[k].x = bq[r][2 * k]; a[k].y = bq[r][2 * k + 1]; b[k].x = sboxes[j * 8 + 2 * k]; b[k].y = sboxes[j * 8 + 2 * k + 1]; }
        return poly_iou_core(a, b) >= thr;
    };
    for (int rb = 0; rb < W; ++rb) {
        const int64_t i0 = (int64_t)rb * 64;
        const int nvalid = (int)((n - i0) < 64 ? (n - i0) : 64);
        if (tid < 64) {
            const int64_t i = i0 + tid;
            if (tid < nvalid) { bm[tid] = meta[i]; bc[tid] = scls[i]; } else { bm[tid].x0 = 1.0; bm[tid].x1 = -1.0; bm[tid].y0 = 1.0; bm[tid].y1 = -1.0; bc[tid] = -2; }
            diag[tid] = 0ull;
        }
        if (tid < 512) { const int r = tid >> 3, q = tid & 7; bq[r][q] = r < nvalid ? sboxes[(i0 + r) * 8 + q] : 0.0; }
        __syncthreads();
        for (int pidx = tid; pidx < 64 * 64; pidx += 1024) {  // pairs (r < c) inside the block
            const int r = pidx >> 6, c = pidx & 63;
            if (r < c && c < nvalid && hit(r, bm[c], bc[c], i0 + c)) atomicOr(&diag[r], 1ull << c);
        }
        __syncthreads();
        if (wave == 0) {
            const unsigned long long d = diag[lane];
            unsigned long long rem = removed[rb], km = 0ull;
            for (int r = 0; r < nvalid; ++r) {
                const unsigned long long dr = __shfl(d, r);  // wave-uniform
                if (!((rem >> r) & 1ull)) { km |= 1ull << r; rem |= dr; }
            }
            if (lane < nvalid) keep[i0 + lane] = (uint8_t)((km >> lane) & 1ull);
            if (lane == 0) { keepmask_s = km; total_s += __popcll(km); }
        }
        __syncthreads();
        const unsigned long long km = keepmask_s;
        for (int64_t j = i0 + 64 + tid; j < n; j += 1024) {
            if ((removed[j >> 6] >> (j & 63)) & 1ull) continue;
            const BoxMeta mj = meta[j];
            const int32_t cj = scls[j];
            unsigned long long w = km;
            while (w) {
                const int r = __ffsll((long long)w) - 1;
                w &= w - 1;
                if (hit(r, mj, cj, j)) { atomicOr(&removed[j >> 6], 1ull << (j & 63)); break; }
            }
        }
        __syncthreads();
    }
    if (tid == 0 && n_keep) *n_keep = total_s;
}

// ------------------------------------------------------------------------------------------------ LDS-resident segments (n <= kSegMax)

static constexpr int kSegMax = 512;
static constexpr int kSegWords = kSegMax / 64;

// One workgroup (1024 threads) per segment: rank sort, pair tests, greedy scan -- everything in LDS.
// The cheap class + envelope tests emit a compact candidate-pair list; the expensive fp64 clipping is then spread evenly
// over all lanes (a row-per-thread loop would leave most of the group idle behind the few crowded rows).
static constexpr int kSegPairCap = 8192;
static constexpr int kSegMid = 256, kSegMidPairCap = 4096;

// Segment `seg` = rows [seg_lo[seg], seg_hi[seg]) (a prefix-offset array passes (off, off + 1); the fused per-tile path passes fixed-stride
// slots with their fill counts).
// SEGMAX / NT / PAIRCAP: the 512-row form (1024 threads, ~104 KB of LDS: one segment per CU) and a 256-row form (256 threads, 36 KB) for
// the single-segment call.  (Measured, round 4: giving the 65 .. 256-row segments of a batch the small form does NOT help -- the kernel's time
// is the latency of its slowest segment, i.e. the exact fp64 clips of a saturated tile's same-class pairs spread over the workgroup's
// threads: 256 threads took 372 us for what 1024 do in < 483 us, and the two launches add up.)  `skip_above` > 0: longer segments are
// left to another launch
template <int SEGMAX, int NT, int PAIRCAP>
__device__ __forceinline__ void merge_segment(const int seg, const double *__restrict__ boxes, const int32_t *__restrict__ cls, const double *__restrict__ conf,
                                              const int32_t *__restrict__ seg_lo, const int32_t *__restrict__ seg_hi, double thr, int32_t *__restrict__ order,
                                              uint8_t *__restrict__ keep, int32_t *__restrict__ n_keep, int32_t *__restrict__ status, int skip_upto, int skip_above) {
    constexpr int kSegMax = SEGMAX, kSegWords = SEGMAX / 64, kSegPairCap = PAIRCAP;
    __shared__ double skey[kSegMax];
    __shared__ int32_t sord[kSegMax];
    __shared__ int32_t scl[kSegMax];
    __shared__ BoxMeta smeta[kSegMax];
    __shared__ __attribute__((aligned(16))) unsigned long long sbits[kSegMax * kSegWords];
    __shared__ unsigned int spairs[kSegPairCap];
    __shared__ unsigned int npairs_s;
    // lane-private clip buffers (clip_area_convex) for CLIPT lanes: the 1024-thread form clips with its first 512 threads (128 bytes per lane)
    constexpr int CLIPT = NT < 512 ? NT : 512;
    __shared__ __attribute__((aligned(16))) P2 sclip[kClipCap * CLIPT];
    int32_t s0 = seg_lo[seg], s1 = seg_hi[seg];
    int n = s1 - s0;
    if (n <= skip_upto) { if (n <= 0 && skip_upto == 0 && threadIdx.x == 0 && n_keep) n_keep[seg] = 0; return; }  // (short ones: k_merge_segments_wave)
    if (skip_above > 0 && n > skip_above) return;
    if (n > kSegMax) {  // too long for LDS: flag it and write a DEFINED result (identity order, nothing kept); outside a stream capture the
        // host then reruns such segments through the dense path, inside one (no host read possible) the caller sees them dropped, not garbage
        if (threadIdx.x == 0) atomicExch(status, 1);
        for (int t = threadIdx.x; t < n; t += NT) { order[s0 + t] = s0 + t; keep[s0 + t] = 0; }
        return;
    }
    if (threadIdx.x == 0) npairs_s = 0;
    for (int t = threadIdx.x; t < n; t += NT) skey[t] = sort_key(conf[s0 + t]);
    __syncthreads();
    for (int t = threadIdx.x; t < n; t += NT) {
        double ki = skey[t];
        int rank = 0;
        for (int u = 0; u < n; ++u) rank += (skey[u] > ki) | ((skey[u] == ki) & (u < t));
        sord[rank] = t;
    }
    __syncthreads();
    double *sarea = skey;  // (the keys are spent: the rows' areas from here on)
    int W = (n + 63) / 64;
    for (int t = threadIdx.x; t < n; t += NT) {
        int src = s0 + sord[t];
        order[s0 + t] = src;
        P2 p[4];
        for (int k = 0; k < 4; ++k) { p[k].x = boxes[(int64_t)src * 8 + 2 * k]; p[k].y = boxes[(int64_t)src * 8 + 2 * k + 1]; }
        BoxMeta m;
        double area = 0.0;
        if (quad_valid(p)) { Aabb a = quad_aabb(p); m.x0 = a.x0; m.y0 = a.y0; m.x1 = a.x1; m.y1 = a.y1; area = fabs(shoelace2<4>(p, 4)) * 0.5; }
        else { m.x0 = 1.0; m.x1 = -1.0; m.y0 = 1.0; m.y1 = -1.0; }
        smeta[t] = m;
        sarea[t] = area;
        scl[t] = cls[src];
        for (int w = 0; w < kSegWords; ++w) sbits[t * kSegWords + w] = 0ull;
    }
    __syncthreads();
    const bool all_hit = !(thr > 0.0);
    // phase A: cheap tests (class, envelope) over the strict upper triangle -> candidate pairs.  Thread (j, part): row j against the earlier
    // rows of slice `part` -- the row's own class and envelope stay in registers and a wave's lanes read the SAME earlier row (one LDS
    // broadcast per test).  (The flattened triangle this replaces spent 60 us of a 290-row segment's 216 on inverting pair indices.)
    {
        const int npad = (n + 63) & ~63;               // whole waves per part
        const int parts = max(1, NT / npad), chunk = (n + parts - 1) / parts;
        for (int idx = threadIdx.x; idx < npad * parts; idx += NT) {
            const int j = idx % npad, part = idx / npad;
            if (j >= n) continue;
            const int i0 = part * chunk, i1 = min(min(i0 + chunk, n), j);
            const int cj = scl[j];
            const BoxMeta mj = smeta[j];
            const double aj = sarea[j];
            for (int i = i0; i < i1; ++i) {
                if (scl[i] != cj) continue;
                if (all_hit) { atomicOr(&sbits[i * kSegWords + (j >> 6)], 1ull << (j & 63)); continue; }
                const BoxMeta mi = smeta[i];
                if (!meta_overlap(mi, mj)) continue;
                // the intersection lies inside both envelopes and inside either quad: IoU <= ub / (a_i + a_j - ub) with ub = min(area of
                // the envelopes' overlap, a_i, a_j).  A pair whose bound is below the threshold by more than any rounding of the exact
                // evaluation (relative 1e-9 against ~1e-15) cannot be a hit: it never reaches the clip -- the decisions are unchanged.
                {
                    const double ai = sarea[i];
                    const double ub = fmin(fmin(ai, aj), (fmin(mi.x1, mj.x1) - fmax(mi.x0, mj.x0)) * (fmin(mi.y1, mj.y1) - fmax(mi.y0, mj.y0)));
                    const double den = ai + aj - ub;
                    if (den > 0.0 && ub * (1.0 + 1e-9) < thr * den) continue;
                }
                const unsigned int slot = atomicAdd(&npairs_s, 1u);
                if (slot < (unsigned)kSegPairCap) spairs[slot] = ((unsigned)i << 16) | (unsigned)j;
                else {  // list full (pathologically crowded tile): clip right here (general routine: no LDS buffer for every thread)
                    P2 p[4], q[4];
                    const double *pa = boxes + (int64_t)(s0 + sord[i]) * 8, *pb = boxes + (int64_t)(s0 + sord[j]) * 8;
                    for (int k = 0; k < 4; ++k) { p[k].x = pa[2 * k]; p[k].y = pa[2 * k + 1]; q[k].x = pb[2 * k]; q[k].y = pb[2 * k + 1]; }
                    if (poly_iou_core(p, q) >= thr) atomicOr(&sbits[i * kSegWords + (j >> 6)], 1ull << (j & 63));
                }
            }
        }
    }
    __syncthreads();
    // phase B: exact fp64 clipping, one candidate pair per lane
    int npairs = (int)min(npairs_s, (unsigned)kSegPairCap);
    for (int t = threadIdx.x; t < npairs && (int)threadIdx.x < CLIPT; t += CLIPT) {
        int i = spairs[t] >> 16, j = spairs[t] & 0xffff;
        P2 p[4], q[4];
        const double *pa = boxes + (int64_t)(s0 + sord[i]) * 8, *pb = boxes + (int64_t)(s0 + sord[j]) * 8;
        for (int k = 0; k < 4; ++k) { p[k].x = pa[2 * k]; p[k].y = pa[2 * k + 1]; q[k].x = pb[2 * k]; q[k].y = pb[2 * k + 1]; }
        if (poly_iou_core_lds(p, q, sclip + threadIdx.x, CLIPT) >= thr) atomicOr(&sbits[i * kSegWords + (j >> 6)], 1ull << (j & 63));
    }
    __syncthreads();
    // greedy scan, wave 0, 64 rows at a time (one row per step -- shuffle, test, LDS read, or -- was 36 us for 290 rows).  Lane r holds the
    // diagonal word of row i0 + r (whom it suppresses inside the block).  The kept set K of the block is the unique solution of
    // K = alive & ~OR_{j in K} diag_j (the pair graph is acyclic: only earlier rows suppress), reached by iterating from K = alive -- row i is
    // final after i + 1 rounds at the latest, in practice after the depth of the longest suppression chain (a few); then the kept rows' later
    // words are OR-reduced over the wave into `removed` (lane w owns word w).
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        auto wave_or = [](unsigned long long v) -> unsigned long long {
            unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
#pragma unroll
            for (int o = 32; o; o >>= 1) { lo |= (unsigned)__shfl_xor((int)lo, o); hi |= (unsigned)__shfl_xor((int)hi, o); }
            return ((unsigned long long)hi << 32) | lo;
        };
        unsigned long long rem = 0ull;
        int kept = 0;
        for (int b = 0; b < W; ++b) {
            const int i0 = b * 64, r = i0 + lane;
            const int nb = min(64, n - i0);
            const unsigned long long valid = nb == 64 ? ~0ull : ((1ull << nb) - 1ull);
            const unsigned long long diag = r < n ? sbits[r * kSegWords + b] : 0ull;
            const unsigned rlo = (unsigned)__shfl((int)(unsigned)rem, b), rhi = (unsigned)__shfl((int)(unsigned)(rem >> 32), b);
            const unsigned long long alive = ~(((unsigned long long)rhi << 32) | rlo) & valid;  // not removed by an earlier block's rows
            unsigned long long K = alive;
            for (int it = 0; it < 64; ++it) {
                const unsigned long long R = wave_or(((K >> lane) & 1ull) ? diag : 0ull);
                const unsigned long long Kn = alive & ~R;
                if (Kn == K) break;  // (uniform)
                K = Kn;
            }
            kept += __popcll(K);
            const bool mine = (K >> lane) & 1ull;
            if (r < n) keep[s0 + r] = (uint8_t)mine;
            for (int w = b + 1; w < W; ++w) {  // later words: OR of the kept rows' bits
                const unsigned long long v = wave_or(mine ? sbits[r * kSegWords + w] : 0ull);
                if (lane == w) rem |= v;
            }
        }
        if (lane == 0 && n_keep) n_keep[seg] = kept;
    }
}

// grid: one workgroup per segment (work_list == nullptr), or resident workgroups walking the device-side list of the segments longer than a
// wave takes (k_merge_segments_wave builds it): no 1024-thread, 155-KB workgroup is launched for a short segment only to return.
template <int SEGMAX, int NT, int PAIRCAP>
__global__ __launch_bounds__(NT) void k_merge_segments(const double *__restrict__ boxes, const int32_t *__restrict__ cls,
                                                        const double *__restrict__ conf, const int32_t *__restrict__ seg_lo, const int32_t *__restrict__ seg_hi,
                                                        double thr, int32_t *__restrict__ order, uint8_t *__restrict__ keep,
                                                        int32_t *__restrict__ n_keep, int32_t *__restrict__ status, int skip_upto, int skip_above,
                                                        const int32_t *__restrict__ work_list, const int32_t *__restrict__ work_count) {
    if (!work_list) {
        merge_segment<SEGMAX, NT, PAIRCAP>(blockIdx.x, boxes, cls, conf, seg_lo, seg_hi, thr, order, keep, n_keep, status, skip_upto, skip_above);
        return;
    }
    const int nw = *work_count;
    for (int w = blockIdx.x; w < nw; w += gridDim.x) {
        merge_segment<SEGMAX, NT, PAIRCAP>(work_list[w], boxes, cls, conf, seg_lo, seg_hi, thr, order, keep, n_keep, status, skip_upto, skip_above);
        __syncthreads();
    }
}

// Segments of at most 64 rows (the usual tile: a dozen symbols): one wave per segment, a few KB of LDS, so a CU holds dozens of
// segments at once instead of one 1024-thread workgroup each.  Same phases and the same arithmetic as k_merge_segments.
static constexpr int kSegWave = 64;
__global__ __launch_bounds__(64) void k_merge_segments_wave(const double *__restrict__ boxes, const int32_t *__restrict__ cls,
                                                           const double *__restrict__ conf, const int32_t *__restrict__ seg_lo, const int32_t *__restrict__ seg_hi,
                                                           double thr, int32_t *__restrict__ order, uint8_t *__restrict__ keep,
                                                           int32_t *__restrict__ n_keep, int32_t *__restrict__ long_list, int32_t *__restrict__ long_count) {
    __shared__ double skey[kSegWave];
    __shared__ int32_t sord[kSegWave];
    __shared__ int32_t scl[kSegWave];
    __shared__ BoxMeta smeta[kSegWave];
    __shared__ double sbox[kSegWave * 8];
    __shared__ unsigned long long sbits[kSegWave];
    __shared__ unsigned short spairs[kSegWave * (kSegWave - 1) / 2];
    __shared__ unsigned int npairs_s;
    __shared__ __attribute__((aligned(16))) P2 sclip[kClipCap * 64];  // lane-private clip buffers (clip_area_convex)
    const int seg = blockIdx.x, t = threadIdx.x;
    const int32_t s0 = seg_lo[seg], s1 = seg_hi[seg];
    const int n = s1 - s0;
    if (n <= 0) { if (t == 0 && n_keep) n_keep[seg] = 0; return; }
    if (n > kSegWave) { if (t == 0 && long_list) long_list[atomicAdd(long_count, 1)] = seg; return; }  // k_merge_segments takes it (from this list)
    if (t == 0) npairs_s = 0;
    if (t < n) skey[t] = sort_key(conf[s0 + t]);
    sbits[t] = 0ull;
    __syncthreads();
    if (t < n) {
        const double ki = skey[t];
        int rank = 0;
        for (int u = 0; u < n; ++u) rank += (skey[u] > ki) | ((skey[u] == ki) & (u < t));
        sord[rank] = t;
    }
    __syncthreads();
    if (t < n) {
        const int src = s0 + sord[t];
        order[s0 + t] = src;
        P2 p[4];
        for (int k = 0; k < 4; ++k) { p[k].x = boxes[(int64_t)src * 8 + 2 * k]; p[k].y = boxes[(int64_t)src * 8 + 2 * k + 1]; }
        for (int k = 0; k < 4; ++k) { sbox[t * 8 + 2 * k] = p[k].x; sbox[t * 8 + 2 * k + 1] = p[k].y; }
        BoxMeta m;
        if (quad_valid(p)) { Aabb a = quad_aabb(p); m.x0 = a.x0; m.y0 = a.y0; m.x1 = a.x1; m.y1 = a.y1; }
        else { m.x0 = 1.0; m.x1 = -1.0; m.y0 = 1.0; m.y1 = -1.0; }
        smeta[t] = m;
        scl[t] = cls[src];
    }
    __syncthreads();
    const bool all_hit = !(thr > 0.0);
    // cheap tests: lane = column j against the rows i < j (the row data are LDS broadcasts)
    if (t < n) {
        const BoxMeta mj = smeta[t];
        const int cj = scl[t];
        for (int i = 0; i < t; ++i) {
            if (scl[i] != cj) continue;
            if (all_hit) atomicOr(&sbits[i], 1ull << t);
            else if (meta_overlap(smeta[i], mj)) spairs[atomicAdd(&npairs_s, 1u)] = (unsigned short)((i << 8) | t);
        }
    }
    __syncthreads();
    const int npairs = (int)npairs_s;
    for (int e = t; e < npairs; e += 64) {
        const int i = spairs[e] >> 8, j = spairs[e] & 0xff;
        P2 p[4], q[4];
        for (int k = 0; k < 4; ++k) { p[k].x = sbox[i * 8 + 2 * k]; p[k].y = sbox[i * 8 + 2 * k + 1]; q[k].x = sbox[j * 8 + 2 * k]; q[k].y = sbox[j * 8 + 2 * k + 1]; }
        if (poly_iou_core_lds(p, q, sclip + t, 64) >= thr) atomicOr(&sbits[i], 1ull << j);
    }
    __syncthreads();
    unsigned long long rem = 0ull;  // wave-uniform greedy scan
    for (int i = 0; i < n; ++i)
        if (!((rem >> i) & 1ull)) rem |= sbits[i];
    if (t < n) keep[s0 + t] = (uint8_t)(!((rem >> t) & 1ull));
    if (t == 0 && n_keep) n_keep[seg] = n - __popcll(rem & (n == 64 ? ~0ull : ((1ull << n) - 1ull)));
}

// ------------------------------------------------------------------------------------------------ consensus (single workgroup)

struct ConsOff { int64_t v[17]; };  // scale k owns rows [v[k], v[k + 1]): passed BY VALUE (a host array copied to the device per call would be a
                                    // pageable host-to-device copy inside the entry point -- host-side latency, and host memory referenced by a captured graph)

__global__ __launch_bounds__(256) void k_consensus(const double *__restrict__ boxes, const int32_t *__restrict__ cls,
                                                  const double *__restrict__ conf, const ConsOff offs,
                                                  int32_t nscales, double iou_partner, double cons_low, double cons_high,
                                                  uint8_t *__restrict__ state /* [total]: bit0 alive, bit1 visited */,
                                                  BoxMeta *__restrict__ meta, int32_t *__restrict__ out_idx,
                                                  int32_t *__restrict__ n_out, const int32_t *__restrict__ only_if /* nullptr, or run only when *only_if != 0 */) {
    const int64_t *off = offs.v;
    if (only_if && *only_if == 0) return;
    __shared__ double r_conf[256], r_iou[256];
    __shared__ int32_t r_idx[256];
    __shared__ int32_t nout_s;
    __shared__ __attribute__((aligned(16))) P2 sclip[kClipCap * 256];  // lane-private clip buffers (clip_area_convex)
    int64_t total = off[nscales];
    int tid = threadIdx.x;
    if (tid == 0) nout_s = 0;
    if (nscales == 1) {  // Detect_OBB.py:357-358 passthrough
        for (int64_t i = tid; i < total; i += 256) out_idx[i] = (int32_t)i;
        if (tid == 0) *n_out = (int32_t)total;
        return;
    }
    for (int64_t i = tid; i < total; i += 256) {
        state[i] = (conf[i] >= cons_low) ? 1 : 0;  // :361-364
        P2 p[4];
        for (int k = 0; k < 4; ++k) { p[k].x = boxes[i * 8 + 2 * k]; p[k].y = boxes[i * 8 + 2 * k + 1]; }
        BoxMeta m;
        if (quad_valid(p)) { Aabb a = quad_aabb(p); m.x0 = a.x0; m.y0 = a.y0; m.x1 = a.x1; m.y1 = a.y1; }
        else { m.x0 = 1.0; m.x1 = -1.0; m.y0 = 1.0; m.y1 = -1.0; }
        meta[i] = m;
    }
    __syncthreads();
    for (int32_t s = 0; s < nscales; ++s) {
        for (int64_t i = off[s]; i < off[s + 1]; ++i) {
            if (state[i] != 1) continue;  // not alive or already visited (uniform: every thread reads the same byte)
            int ci = cls[i];
            BoxMeta mi = meta[i];
            P2 p[4];
            for (int k = 0; k < 4; ++k) { p[k].x = boxes[i * 8 + 2 * k]; p[k].y = boxes[i * 8 + 2 * k + 1]; }
            double bc = -1.0, bi = 0.0;
            int32_t bj = -1;
            for (int64_t j = tid; j < total; j += 256) {
                if (j >= off[s] && j < off[s + 1]) continue;  // same scale
                if (state[j] != 1 || cls[j] != ci) continue;
                double iou = 0.0;
                if (meta_overlap(mi, meta[j])) {
                    P2 q[4];
                    for (int k = 0; k < 4; ++k) { q[k].x = boxes[j * 8 + 2 * k]; q[k].y = boxes[j * 8 + 2 * k + 1]; }
                    iou = poly_iou_core_lds(p, q, sclip + tid, 256);
                }
                if (iou >= iou_partner) {
                    double cp = conf[j];
                    // thread-local scan is in ascending j: strict improvements only (Detect_OBB.py:397-399)
                    if (cp > bc || (cp == bc && iou > bi)) { bc = cp; bi = iou; bj = (int32_t)j; }
                }
            }
            r_conf[tid] = bc; r_iou[tid] = bi; r_idx[tid] = bj;
            __syncthreads();
            for (int d = 128; d >= 1; d >>= 1) {
                if (tid < d) {
                    double c2 = r_conf[tid + d], i2 = r_iou[tid + d];
                    int32_t j2 = r_idx[tid + d];
                    double c1 = r_conf[tid], i1 = r_iou[tid];
                    int32_t j1 = r_idx[tid];
                    // prefer higher conf, then higher iou, then the earlier pool position (first encountered wins ties)
                    bool take2 = (j2 >= 0) && (j1 < 0 || c2 > c1 || (c2 == c1 && (i2 > i1 || (i2 == i1 && j2 < j1))));
                    if (take2) { r_conf[tid] = c2; r_iou[tid] = i2; r_idx[tid] = j2; }
                }
                __syncthreads();
            }
            if (tid == 0) {
                int32_t best = r_idx[0];
                double bconf = r_conf[0];
                if (best < 0 || bconf < cons_low) {  // :406-410
                    if (conf[i] >= cons_high) out_idx[nout_s++] = (int32_t)i;
                    state[i] = 3;
                } else {  // :412-421
                    out_idx[nout_s++] = (conf[i] >= bconf) ? (int32_t)i : best;
                    state[i] = 3;
                    state[best] = 3;
                }
            }
            __syncthreads();
        }
    }
    if (tid == 0) *n_out = nout_s;
}

// ---- parallel form of cross_scale_consensus_filter (Detect_OBB.py:347-423)
// The reference walks the detections in flat order (scales ascending, list order); an unvisited detection takes its best unvisited partner
// from the other scales (same class, polygon IoU >= 0.40; best = higher confidence, then higher IoU, then first in pool order) and both
// become visited.  That is a greedy matching on a SPARSE graph: (1) the candidate edges are found by the whole chip (k_cons_edges: every
// cross-scale same-class pair, envelope test first, exact IoU for the few that pass); (2) the greedy order is then resolved in rounds by one
// workgroup (k_cons_resolve): a detection decides as soon as every earlier detection within two hops (a neighbour, or a rival for one of
// its neighbours) has decided -- exactly the information the sequential walk would have had at its turn -- so each round decides many
// independent clusters at once and the result is identical to the walk.  Adjacency lists are fixed-capacity (kConsK per detection); if one
// overflows, the single-workgroup walk (k_consensus) runs instead.
static constexpr int kConsK = 32;

__global__ __launch_bounds__(256) void k_cons_prep(const double *__restrict__ boxes, const double *__restrict__ conf, const ConsOff offs,
                                                  int32_t nscales, double cons_low, uint8_t *__restrict__ state, uint8_t *__restrict__ scale_id,
                                                  BoxMeta *__restrict__ meta, int32_t *__restrict__ deg, int32_t *__restrict__ flags) {
    const int64_t *off = offs.v;
    const int64_t total = off[nscales];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i == 0) { flags[0] = 0; flags[1] = 0; }
    if (i >= total) return;
    state[i] = (conf[i] >= cons_low) ? 1 : 0;  // :361-364
    int sc = 0;
    while (sc + 1 < nscales && i >= off[sc + 1]) ++sc;
    scale_id[i] = (uint8_t)sc;
    deg[i] = 0;
    P2 p[4];
    for (int k = 0; k < 4; ++k) { p[k].x = boxes[i * 8 + 2 * k]; p[k].y = boxes[i * 8 + 2 * k + 1]; }
    BoxMeta m;
    if (quad_valid(p)) { Aabb a = quad_aabb(p); m.x0 = a.x0; m.y0 = a.y0; m.x1 = a.x1; m.y1 = a.y1; }
    else { m.x0 = 1.0; m.x1 = -1.0; m.y0 = 1.0; m.y1 = -1.0; }
    meta[i] = m;
}

// One launch per scale b >= 1: its rows j in [j_base, j_end) against every row i of the EARLIER scales, [0, j_base) -- only pairs of different
// scales are partners and a scale's rows are contiguous, so nothing else is enumerated (all pairs i < j: 2.07 ms for 51 k + 12 k rows; with
// the same-scale blocks returning at once 1.03 ms; this grid: see profiles/r04_summary.md).  grid (ceil((j_end - j_base) / 256),
// ceil(j_base / 256)): thread = one j, block row = 256 i's held in LDS (class in one word, -1 = takes no part: one compare per pair).
template <int IB /* i rows per block: 256, or 64 when that fills the chip better */>
__global__ __launch_bounds__(256) void k_cons_edges(const double *__restrict__ boxes, const int32_t *__restrict__ cls, const uint8_t *__restrict__ state,
                                                   const uint8_t *__restrict__ scale_id, const BoxMeta *__restrict__ meta, int64_t j_base, int64_t j_end, double iou_thr,
                                                   int32_t *__restrict__ adj_idx, double *__restrict__ adj_iou, int32_t *__restrict__ deg,
                                                   int32_t *__restrict__ flags) {
    __shared__ BoxMeta sm[IB];
    __shared__ int32_t sc[IB];  // class, or -1 for a detection that takes no part (below CONS_LOW): ONE compare per candidate i in the loop
    __shared__ __attribute__((aligned(16))) P2 sclip[kClipCap * 256];  // lane-private clip buffers (clip_area_convex)
    const int64_t i0 = (int64_t)blockIdx.y * IB, j0 = j_base + (int64_t)blockIdx.x * 256;
    const int64_t total = j_base;  // (bound of the i rows: every one of them belongs to an earlier scale than j's)
    if ((int)threadIdx.x < IB) {
        const int64_t i = i0 + threadIdx.x;
        const bool ok = i < total && state[i] == 1;
        sc[threadIdx.x] = ok ? cls[i] : -1;
        if (i < total) sm[threadIdx.x] = meta[i];
    }
    __syncthreads();
    const int64_t j = j0 + threadIdx.x;
    if (j >= j_end || state[j] != 1) return;
    const BoxMeta mj = meta[j];
    const int cj = cls[j];
    const int nr = (int)(total - i0 < IB ? total - i0 : IB);
    for (int r = 0; r < nr; ++r) {
        if (sc[r] != cj) continue;
        if (!meta_overlap(sm[r], mj)) continue;
        const int64_t i = i0 + r;
        P2 p[4], q[4];
        for (int k = 0; k < 4; ++k) {
            p[k].x = boxes[i * 8 + 2 * k]; p[k].y = boxes[i * 8 + 2 * k + 1];
            q[k].x = boxes[j * 8 + 2 * k]; q[k].y = boxes[j * 8 + 2 * k + 1];
        }
        // the walk evaluates compute_polygon_iou(d, p) with d = the detection being processed: the earlier one (i) in every pair that
        // can matter (a later detection never looks back at a visited one), so (i, j) is the argument order to reproduce bit for bit
        const double iou = poly_iou_core_lds(p, q, sclip + threadIdx.x, 256);
        if (!(iou >= iou_thr)) continue;
        const int a = atomicAdd(&deg[i], 1), b = atomicAdd(&deg[j], 1);
        if (a < kConsK) { adj_idx[i * kConsK + a] = (int32_t)j; adj_iou[i * kConsK + a] = iou; } else flags[0] = 1;
        if (b < kConsK) { adj_idx[j * kConsK + b] = (int32_t)i; adj_iou[j * kConsK + b] = iou; } else flags[0] = 1;
    }
}

__global__ __launch_bounds__(1024) void k_cons_resolve(const double *__restrict__ conf, int64_t total, double cons_high, uint8_t *__restrict__ state,
                                                      const int32_t *__restrict__ adj_idx, const double *__restrict__ adj_iou, const int32_t *__restrict__ deg,
                                                      int32_t *__restrict__ decision, int32_t *__restrict__ emit, int32_t *__restrict__ flags,
                                                      int32_t *__restrict__ work, int32_t *__restrict__ out_idx, int32_t *__restrict__ n_out) {
    if (flags[0]) return;  // adjacency overflow: k_consensus (launched behind this kernel) does the walk
    __shared__ int s_cnt, s_base, s_wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // rows that still wait for a decision, two lists in turn (after the first round a few percent of the rows are left: every later round
    // walks those instead of all `total` rows -- 63 k rows: 0.59 -> see profiles/r04_summary.md)
    int32_t *lists[2] = {work, work + total};
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (int64_t i = tid; i < total; i += 1024) {
        int e = -1;
        if (state[i] == 1) {
            if (deg[i] == 0) { e = conf[i] >= cons_high ? (int32_t)i : -1; state[i] = 3; }  // no candidate partner at all: :406-410
            else lists[0][atomicAdd(&s_cnt, 1)] = (int32_t)i;
        }
        emit[i] = e;
    }
    __syncthreads();
    int n_cur = s_cnt;
    for (int64_t round = 0; round <= total && n_cur > 0; ++round) {
        const int32_t *cur = lists[round & 1];
        int32_t *nxt = lists[(round + 1) & 1];
        __syncthreads();
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        for (int q = tid; q < n_cur; q += 1024) {  // decisions against the states of the round's start (nothing is written here but the lists)
            const int i = cur[q];
            int dec = -2;  // not this round
            if (state[i] != 1) { decision[i] = dec; continue; }  // taken as a partner while it waited: gone from the lists
            const int di = deg[i];
            bool ready = true;
            for (int a = 0; a < di && ready; ++a) {
                const int j = adj_idx[(int64_t)i * kConsK + a];
                if (state[j] != 1) continue;  // visited: no longer a candidate, and it blocks nothing
                if (j < i) { ready = false; break; }  // j is walked before i and may or may not take i
                const int dj = deg[j];
                for (int b = 0; b < dj; ++b) {
                    const int k = adj_idx[(int64_t)j * kConsK + b];
                    if (k < i && state[k] == 1) { ready = false; break; }  // an earlier rival for j has not decided yet
                }
            }
            if (ready) {
                double bc = -1.0, bi = 0.0;
                int bj = -1;
                for (int a = 0; a < di; ++a) {
                    const int j = adj_idx[(int64_t)i * kConsK + a];
                    if (state[j] != 1) continue;
                    const double cp = conf[j], iou = adj_iou[(int64_t)i * kConsK + a];
                    // :397-399 with the pool scanned in ascending position: higher confidence, then higher IoU, then the earlier one
                    if (bj < 0 || cp > bc || (cp == bc && (iou > bi || (iou == bi && j < bj)))) { bc = cp; bi = iou; bj = j; }
                }
                dec = bj;
            } else nxt[atomicAdd(&s_cnt, 1)] = i;
            decision[i] = dec;
        }
        __syncthreads();
        for (int q = tid; q < n_cur; q += 1024) {
            const int i = cur[q];
            const int dec = decision[i];
            if (dec == -2) continue;
            if (dec < 0) emit[i] = conf[i] >= cons_high ? (int32_t)i : -1;  // :406-410
            else { emit[i] = conf[i] >= conf[dec] ? (int32_t)i : dec; state[dec] = 3; }  // :412-421
            state[i] = 3;
        }
        __syncthreads();
        n_cur = s_cnt;
    }
    __syncthreads();
    // kept list = the emitting detections in walk order: ordered compaction of emit[]
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int64_t c0 = 0; c0 < total; c0 += 1024) {
        const int64_t i = c0 + tid;
        const int e = i < total ? emit[i] : -1;
        const unsigned long long bal = __ballot(e >= 0);
        if (lane == 0) s_wsum[wave] = __popcll(bal);
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave; ++w) before += s_wsum[w];
        if (e >= 0) out_idx[s_base + before + __popcll(bal & ((1ull << lane) - 1ull))] = e;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += s_wsum[w]; s_base += t; }
        __syncthreads();
    }
    if (tid == 0) { *n_out = s_base; flags[1] = 1; }
}

// ------------------------------------------------------------------------------------------------ detect_symbols per-detection body

__global__ __launch_bounds__(256) void k_tile_post(const float *__restrict__ lp, const int32_t *__restrict__ cls,
                                                  const int32_t *__restrict__ det_tile, int64_t n,
                                                  const int32_t *__restrict__ rects, int32_t margin, int32_t strike_cls,
                                                  double *__restrict__ gb, double *__restrict__ angle,
                                                  uint8_t *__restrict__ inside) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int t = det_tile[i];
    double g[8], a;
    bool in;
    tile_post_row(lp + i * 8, cls[i], rects[4 * t], rects[4 * t + 1], rects[4 * t + 2], rects[4 * t + 3], margin, strike_cls, g, a, in);  // post_device.h
    for (int k = 0; k < 8; ++k) gb[i * 8 + k] = g[k];
    inside[i] = (uint8_t)in;
    angle[i] = a;
}

// ------------------------------------------------------------------------------------------------ fused per-tile survivors path
// obb_tile_survivors: everything between the NMS output and the exchange records on the device, tile by tile, with no host-visible
// count in between (the stand-alone chain glued k_results / k_tile_post / the segment merge with host-side compactions).
struct SurvStage {  // fixed-stride staging rows: slot t * md + j = the j-th detection of tile t that passed the border filter
    double *gb;      // [B * md][8] global corners
    double *conf;    // [B * md] float32 confidence widened (the merge's sort key, exactly float(det.conf[0]))
    int32_t *cls;    // [B * md]
    float *pts;      // [B * md][8] local corners
    float *conf32;   // [B * md]
    int32_t *lo, *hi;  // [B] segment bounds in slot space
};

// One workgroup per tile: construct_result + the per-detection body for each of its NMS rows (post_device.h: the same arithmetic as
// k_results / k_tile_post), ordered compaction of the rows inside the border (ballot + per-wave sums: NMS order is kept).
__global__ __launch_bounds__(256) void k_tile_stage(const float *__restrict__ det, const int32_t *__restrict__ count, int md, const float *__restrict__ lb,
                                                   const int32_t *__restrict__ tile_ids, const int32_t *__restrict__ rects, int margin, int strike_cls,
                                                   SurvStage S) {
    __shared__ int wsum[4];
    __shared__ int base_s;
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = min(count[t], md);
    const int tile = tile_ids[t];
    const int x = rects[4 * tile], y = rects[4 * tile + 1], x2 = rects[4 * tile + 2], y2 = rects[4 * tile + 3];
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int k0 = 0; k0 < n; k0 += 256) {
        const int k = k0 + tid;
        bool in = false;
        float p[8];
        double g[8], ang;
        int c = 0;
        float cf = 0.f;
        if (k < n) {
            const float *d = det + ((int64_t)t * md + k) * 7;
            results_row(d, lb ? lb + (int64_t)t * 3 : nullptr, nullptr, p);
            c = (int)d[5];
            cf = d[4];
            tile_post_row(p, c, x, y, x2, y2, margin, strike_cls, g, ang, in);
        }
        const unsigned long long bal = __ballot(in);
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int before = base_s;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (in) {
            const int64_t slot = (int64_t)t * md + before + __popcll(bal & ((1ull << lane) - 1ull));
            for (int q = 0; q < 8; ++q) { S.gb[slot * 8 + q] = g[q]; S.pts[slot * 8 + q] = p[q]; }
            S.cls[slot] = c; S.conf[slot] = (double)cf; S.conf32[slot] = cf;
        }
        __syncthreads();
        if (tid == 0) base_s += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (tid == 0) { S.lo[t] = t * md; S.hi[t] = t * md + base_s; }
}

// exclusive scan of cnt[0..n) -> off[0..n], off[n] = total (also stored to *total); one workgroup
__global__ __launch_bounds__(1024) void k_scan_counts(const int32_t *__restrict__ cnt, int n, int32_t *__restrict__ off, int32_t *__restrict__ total) {
    __shared__ int wtot[16];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + tid;
        const int v = i < n ? cnt[i] : 0;
        int incl = v;
        for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(incl, d); if (lane >= d) incl += u; }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int before = base_s;
        for (int w = 0; w < wave; ++w) before += wtot[w];
        if (i < n) off[i] = before + incl - v;
        __syncthreads();
        if (tid == 0) { int s = 0; for (int w = 0; w < 16; ++w) s += wtot[w]; base_s += s; }
        __syncthreads();
    }
    if (tid == 0) { off[n] = base_s; if (total) *total = base_s; }
}

// One wave per tile: the kept rows of its merged segment, in merge (confidence) order, as 48-byte exchange records at off[tile]
__global__ __launch_bounds__(64) void k_tile_emit(SurvStage S, const int32_t *__restrict__ order, const uint8_t *__restrict__ keep, const int32_t *__restrict__ off,
                                                 const int32_t *__restrict__ tile_ids, int32_t *__restrict__ rec) {
    const int t = blockIdx.x, lane = threadIdx.x;
    const int s0 = S.lo[t], n = S.hi[t] - s0;
    int base = off[t];
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const bool k = i < n && keep[s0 + i];
        const unsigned long long bal = __ballot(k);
        if (k) {
            const int src = order[s0 + i];
            int32_t *r = rec + ((int64_t)base + __popcll(bal & ((1ull << lane) - 1ull))) * 12;
            r[0] = tile_ids[t]; r[1] = S.cls[src]; r[2] = __float_as_int(S.conf32[src]); r[3] = 0;
            for (int q = 0; q < 8; ++q) r[4 + q] = __float_as_int(S.pts[(int64_t)src * 8 + q]);
        }
        base += __popcll(bal);
    }
}

// Ordered compaction of a merge result: out row r = input row order[i] for the r-th kept sorted position i.  Three small launches: kept
// rows per 1024-row block, exclusive scan of the block counts (k_scan_counts, which also leaves the total in *n_out), ordered scatter.
__global__ __launch_bounds__(1024) void k_kept_count(const uint8_t *__restrict__ keep, int64_t n, int32_t *__restrict__ cnt) {
    __shared__ int wtot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t i = (int64_t)blockIdx.x * 1024 + tid;
    const unsigned long long bal = __ballot(i < n && keep[i]);
    if (lane == 0) wtot[wave] = __popcll(bal);
    __syncthreads();
    if (tid == 0) { int s_ = 0; for (int w = 0; w < 16; ++w) s_ += wtot[w]; cnt[blockIdx.x] = s_; }
}

__global__ __launch_bounds__(1024) void k_kept_scatter(const int32_t *__restrict__ order, const uint8_t *__restrict__ keep, int64_t n, const int32_t *__restrict__ off,
                                                      const double *__restrict__ boxes, const int32_t *__restrict__ cls, const double *__restrict__ conf,
                                                      const double *__restrict__ angle, double *__restrict__ oboxes, int32_t *__restrict__ ocls,
                                                      double *__restrict__ oconf, double *__restrict__ oangle) {
    __shared__ int wtot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t i = (int64_t)blockIdx.x * 1024 + tid;
    const bool k = i < n && keep[i];
    const unsigned long long bal = __ballot(k);
    if (lane == 0) wtot[wave] = __popcll(bal);
    __syncthreads();
    if (!k) return;
    int before = off[blockIdx.x];
    for (int w = 0; w < wave; ++w) before += wtot[w];
    const int64_t src = order[i], dst = before + __popcll(bal & ((1ull << lane) - 1ull));
    for (int q = 0; q < 8; ++q) oboxes[dst * 8 + q] = boxes[src * 8 + q];
    ocls[dst] = cls[src]; oconf[dst] = conf[src];
    if (angle) oangle[dst] = angle[src];
}

// exchange records -> SoA detections: global float64 corners + strike angle re-derived from the float32 local corners and the integer
// tile offset (exact), confidence widened to float64 (what float(det.conf[0]) gives): the consumer side of the 48-byte records
__global__ __launch_bounds__(256) void k_records_to_dets(const int32_t *__restrict__ rec, int64_t n, const int32_t *__restrict__ rects, int strike_cls,
                                                        double *__restrict__ gb, int32_t *__restrict__ cls, double *__restrict__ conf, double *__restrict__ angle) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t *r = rec + i * 12;
    const int t = r[0], c = r[1];
    float lp[8];
    for (int q = 0; q < 8; ++q) lp[q] = __int_as_float(r[4 + q]);
    double g[8], a;
    bool in;
    tile_post_row(lp, c, rects[4 * t], rects[4 * t + 1], rects[4 * t + 2], rects[4 * t + 3], 0, strike_cls, g, a, in);
    for (int q = 0; q < 8; ++q) gb[i * 8 + q] = g[q];
    cls[i] = c; conf[i] = (double)__int_as_float(r[2]); angle[i] = a;
}

// valid rows of a fixed-capacity all-gather buffer [world][cap + 1][12] (row 0 of a rank's block: its count) -> dense [total][12] in
// rank order; counts_out[w] = rank w's count (may exceed cap: the caller then repeats the exchange), counts_out[world] = rows written
__global__ __launch_bounds__(256) void k_gather_compact(const int32_t *__restrict__ recv, int world, int cap, int32_t *__restrict__ out, int32_t *__restrict__ counts_out) {
    const int w = blockIdx.y;
    int before = 0, mine = 0, total = 0;
    for (int r = 0; r < world; ++r) {
        const int c = min(recv[(int64_t)r * (cap + 1) * 12], cap);
        if (r < w) before += c;
        if (r == w) mine = c;
        total += c;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        counts_out[w] = recv[(int64_t)w * (cap + 1) * 12];
        if (w == 0) counts_out[world] = total;
    }
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // 16-byte piece i of this rank's rows (3 per row)
    if (i >= (int64_t)mine * 3) return;
    const int4 *src = reinterpret_cast<const int4 *>(recv + ((int64_t)w * (cap + 1) + 1) * 12);
    reinterpret_cast<int4 *>(out + (int64_t)before * 12)[i] = src[i];
}

__global__ void k_set_segment(int32_t *segoff, int32_t n, int32_t *status) {
    if (threadIdx.x == 0) { segoff[0] = 0; segoff[1] = n; status[0] = 0; }
}

__global__ void k_add_offset(int32_t *v, int64_t n, int32_t add) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] += add;
}


// ---- f4: label assignment of the training-set tiler (Train_OBB.py:87-112, one thread per (tile, label)): a label belongs to a tile when
// the midpoint of its first and fourth corner lies inside the tile (half-open) and at least `thr` of its axis-aligned box does; its
// corners are then shifted to the tile, clipped to [0, tile] and divided by the tile size.  Same float64 operations in the same order
// as the pandas expressions, so the rows are bit-identical to the reference's label files before their text formatting.
__global__ __launch_bounds__(256) void k_tile_labels(const double *__restrict__ labels, int64_t n, const int32_t *__restrict__ rects, int32_t ntiles,
                                                    double thr, uint8_t *__restrict__ mask, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)ntiles * n) return;
    const int64_t t = i / n, l = i - t * n;
    const double x = (double)rects[4 * t], y = (double)rects[4 * t + 1];
    const double ts = (double)(rects[4 * t + 2] - rects[4 * t]);
    double px[4], py[4];
    for (int k = 0; k < 4; ++k) { px[k] = labels[l * 8 + 2 * k]; py[k] = labels[l * 8 + 2 * k + 1]; }
    const double cx = (px[0] + px[3]) / 2, cy = (py[0] + py[3]) / 2;
    bool in = cx >= x && cx < x + ts && cy >= y && cy < y + ts;
    if (in) {  // _cov_frac :60-70
        const double bx1 = fmin(fmin(px[0], px[1]), fmin(px[2], px[3])), bx2 = fmax(fmax(px[0], px[1]), fmax(px[2], px[3]));
        const double by1 = fmin(fmin(py[0], py[1]), fmin(py[2], py[3])), by2 = fmax(fmax(py[0], py[1]), fmax(py[2], py[3]));
        const double ax = fmax(0.0, fmin(bx2, x + ts) - fmax(bx1, x)), ay = fmax(0.0, fmin(by2, y + ts) - fmax(by1, y));
        const double inter = ax * ay, area = fmax(1e-6, (bx2 - bx1) * (by2 - by1));
        in = inter / area >= thr;
    }
    mask[i] = in ? 1 : 0;
    if (!in) return;
    for (int k = 0; k < 4; ++k) {
        out[i * 8 + 2 * k] = fmin(fmax(px[k] - x, 0.0), ts) / ts;
        out[i * 8 + 2 * k + 1] = fmin(fmax(py[k] - y, 0.0), ts) / ts;
    }
}

}  // namespace obb

using namespace obb;

extern "C" {

int obb_poly_iou_pairs(obb_ctx *ctx, const double *a, const double *b, int64_t m, double *out, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && m >= 0, "obb_poly_iou_pairs: bad arguments");
    if (m == 0) return OBB_OK;
    OBB_REQUIRE(ctx, a && b && out, "obb_poly_iou_pairs: NULL buffer");
    hipLaunchKernelGGL(k_iou_pairs, dim3((unsigned)cdiv(m, 256)), dim3(256), 0, (hipStream_t)s, a, b, m, out);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

int obb_poly_iou_matrix(obb_ctx *ctx, const double *a, const int32_t *cls_a, int64_t na, const double *b,
                        const int32_t *cls_b, int64_t nb, double *out, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && na >= 0 && nb >= 0, "obb_poly_iou_matrix: bad arguments");
    if (na == 0 || nb == 0) return OBB_OK;
    OBB_REQUIRE(ctx, a && b && out, "obb_poly_iou_matrix: NULL buffer");
    OBB_REQUIRE(ctx, cdiv(na, 16) <= 65535, "obb_poly_iou_matrix: na too large (%lld)", (long long)na);
    hipLaunchKernelGGL(k_iou_matrix, dim3((unsigned)cdiv(nb, 16), (unsigned)cdiv(na, 16)), dim3(256), 0, (hipStream_t)s, a,
                       cls_a, na, b, cls_b, nb, out);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

int obb_points_in_quads(obb_ctx *ctx, const double *pts, const int32_t *cls_p, int64_t np, const double *quads, const int32_t *cls_q,
                        int64_t nq, uint8_t *out, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && np >= 0 && nq >= 0, "obb_points_in_quads: bad arguments");
    if (np == 0 || nq == 0) return OBB_OK;
    OBB_REQUIRE(ctx, pts && quads && out, "obb_points_in_quads: NULL buffer");
    OBB_REQUIRE(ctx, np * nq < (1ll << 40), "obb_points_in_quads: matrix too large");
    hipLaunchKernelGGL(k_points_in_quads, dim3((unsigned)cdiv(np * nq, 256)), dim3(256), 0, (hipStream_t)s, pts, cls_p, np, quads, cls_q, nq, out);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

int obb_sort_desc_stable(obb_ctx *ctx, const double *key, int64_t n, int32_t *order, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && n >= 0 && n < (1ll << 31), "obb_sort_desc_stable: bad n");
    if (n == 0) return OBB_OK;
    OBB_REQUIRE(ctx, key && order, "obb_sort_desc_stable: NULL buffer");
    if (n >= kBucketSortMin && n <= (int64_t)kSortHot * kSortHotMax) {  // bucketed O(n) form, same order element for element (the bound: every hot bucket fits the list)
        hipStream_t st = (hipStream_t)s;
        int32_t *buf = (int32_t *)ctx->workspace(WS_GEOM_C, sizeof(int32_t) * ((size_t)n + 2 * (size_t)kSortBuckets + 64 + 2 * kSortHotMax + 8 + 4 * kSortParts + 4));
        if (!buf) return set_error(ctx, OBB_ERR_HIP, "obb_sort_desc_stable: workspace allocation failed");
        int32_t *hist = buf, *start = buf + kSortBuckets + 8, *members = buf + 2 * kSortBuckets + 32;
        SortInfo *info = (SortInfo *)(buf + 2 * kSortBuckets + 16);
        int32_t *cursor = hist;  // the histogram is dead once scanned: reuse it as the scatter cursor
        OBB_HIP(ctx, hipMemsetAsync(hist, 0, sizeof(int32_t) * kSortBuckets, st));
        unsigned long long *rpart = (unsigned long long *)(buf + ((2 * (size_t)kSortBuckets + 32 + (size_t)n + 8 + 2 * kSortHotMax + 8 + 1) & ~(size_t)1));  // behind the hot list
        hipLaunchKernelGGL(k_sort_range_part, dim3(kSortParts), dim3(1024), 0, st, key, n, rpart);
        hipLaunchKernelGGL(k_sort_range, dim3(1), dim3(64), 0, st, (const unsigned long long *)rpart, kSortParts, info);
        hipLaunchKernelGGL(k_sort_count, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, key, n, (const SortInfo *)info, hist);
        hipLaunchKernelGGL(k_cells_scan, dim3(1), dim3(1024), 0, st, (const int32_t *)hist, kSortBuckets, start, cursor);
        hipLaunchKernelGGL(k_sort_scatter, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, key, n, (const SortInfo *)info, cursor, members);
        int32_t *hot = members + n + 8;  // [0] = number of buckets above kSortHot members, then their ids
        OBB_HIP(ctx, hipMemsetAsync(hot, 0, sizeof(int32_t), st));
        hipLaunchKernelGGL(k_sort_rank, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, key, n, (const SortInfo *)info, (const int32_t *)start,
                           (const int32_t *)members, order, hot);
        hipLaunchKernelGGL(k_sort_hot, dim3(256), dim3(1024), 0, st, key, (const SortInfo *)info, (const int32_t *)start, (const int32_t *)members, order,
                           (const int32_t *)hot);
        OBB_LAUNCH_CHECK(ctx);
        return OBB_OK;
    }
    int32_t *rank = (int32_t *)ctx->workspace(WS_GEOM_C, sizeof(int32_t) * (size_t)n);
    if (!rank) return set_error(ctx, OBB_ERR_HIP, "obb_sort_desc_stable: workspace allocation failed");
    OBB_REQUIRE(ctx, cdiv(n, kRankJSlice) <= 65535, "obb_sort_desc_stable: n too large");
    OBB_HIP(ctx, hipMemsetAsync(rank, 0, sizeof(int32_t) * (size_t)n, (hipStream_t)s));
    hipLaunchKernelGGL(k_rank_count, dim3((unsigned)cdiv(n, 256), (unsigned)cdiv(n, kRankJSlice)), dim3(256), 0, (hipStream_t)s, key, n, rank);
    OBB_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(k_rank_scatter, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, (const int32_t *)rank, n, order);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

static int nms_mask_impl(obb_ctx *ctx, const double *sboxes, const int32_t *scls, const BoxMeta *meta, int64_t n, double thr,
                         uint64_t *mask, hipStream_t st) {
    int64_t W = cdiv(n, 64);
    OBB_REQUIRE(ctx, W <= 65535, "nms: n=%lld exceeds the 4.19M-box grid limit", (long long)n);
    hipLaunchKernelGGL(k_nms_mask<false>, dim3((unsigned)W, (unsigned)W), dim3(64), 0, st, sboxes, scls, meta, n, thr,
                       (unsigned long long *)mask, (unsigned long long *)nullptr, (unsigned int *)nullptr, 0u);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

int obb_nms_mask(obb_ctx *ctx, const double *boxes_sorted, const int32_t *cls_sorted, int64_t n, double thr, uint64_t *mask,
                 obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && n >= 0, "obb_nms_mask: bad arguments");
    if (n == 0) return OBB_OK;
    OBB_REQUIRE(ctx, boxes_sorted && cls_sorted && mask, "obb_nms_mask: NULL buffer");
    BoxMeta *meta = (BoxMeta *)ctx->workspace(WS_NMS_A, sizeof(BoxMeta) * (size_t)n);
    if (!meta) return set_error(ctx, OBB_ERR_HIP, "obb_nms_mask: workspace allocation failed");
    hipLaunchKernelGGL(k_prep_sorted, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, boxes_sorted, cls_sorted,
                       (const int32_t *)nullptr, n, (double *)nullptr, (int32_t *)nullptr, meta);
    OBB_LAUNCH_CHECK(ctx);
    return nms_mask_impl(ctx, boxes_sorted, cls_sorted, meta, n, thr, mask, (hipStream_t)s);
}

int obb_nms_reduce(obb_ctx *ctx, const uint64_t *mask, int64_t n, uint8_t *keep, int32_t *n_keep, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && n >= 0, "obb_nms_reduce: bad arguments");
    if (n == 0) {
        if (n_keep) OBB_HIP(ctx, hipMemsetAsync(n_keep, 0, sizeof(int32_t), (hipStream_t)s));
        return OBB_OK;
    }
    OBB_REQUIRE(ctx, mask && keep, "obb_nms_reduce: NULL buffer");
    size_t lds = sizeof(unsigned long long) * (size_t)cdiv(n, 64);
    OBB_REQUIRE(ctx, lds <= 128 * 1024, "obb_nms_reduce: n=%lld too large for the LDS-resident scan (max 1M boxes)", (long long)n);
    hipLaunchKernelGGL(k_nms_reduce, dim3(1), dim3(1024), lds, (hipStream_t)s, (const unsigned long long *)mask, n, keep, n_keep);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

int obb_merge_detections(obb_ctx *ctx, const double *boxes, const int32_t *cls, const double *conf, int64_t n, double thr,
                         int32_t *order, uint8_t *keep, int32_t *n_keep, obb_stream_t s);

// workgroups of the 512-row form the chip holds at once (its LDS: one per CU)
static int merge_resident(obb_ctx *) {
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 1) ncu = 64;
    return ncu;
}

int obb_merge_segments(obb_ctx *ctx, const double *boxes, const int32_t *cls, const double *conf, const int32_t *seg_off,
                       int32_t nseg, int64_t n, double thr, int32_t *order, uint8_t *keep, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && nseg >= 0 && n >= 0, "obb_merge_segments: bad arguments");
    if (nseg == 0 || n == 0) return OBB_OK;
    OBB_REQUIRE(ctx, boxes && cls && conf && seg_off && order && keep, "obb_merge_segments: NULL buffer");
    hipStream_t st = (hipStream_t)s;
    // status word | count of segments longer than a wave takes | their list (built by k_merge_segments_wave, walked by k_merge_segments)
    int32_t *status = (int32_t *)ctx->workspace(WS_GEOM_E, 256 + sizeof(int32_t) * (size_t)nseg);
    if (!status) return set_error(ctx, OBB_ERR_HIP, "obb_merge_segments: workspace allocation failed");
    OBB_HIP(ctx, hipMemsetAsync(status, 0, 2 * sizeof(int32_t), st));
    hipLaunchKernelGGL(k_merge_segments_wave, dim3((unsigned)nseg), dim3(64), 0, st, boxes, cls, conf, seg_off, seg_off + 1, thr, order, keep, (int32_t *)nullptr,
                       status + 64, status + 1);
    if (n > kSegWave)  // (some segment may be longer than a wave takes)
        hipLaunchKernelGGL((k_merge_segments<kSegMax, 1024, kSegPairCap>), dim3((unsigned)std::min<int>(nseg, merge_resident(ctx))), dim3(1024), 0, st, boxes, cls, conf,
                           seg_off, seg_off + 1, thr, order, keep, (int32_t *)nullptr, status, kSegWave, 0, status + 64, status + 1);
    OBB_LAUNCH_CHECK(ctx);
    if (n <= kSegMax) return OBB_OK;  // no segment can be longer than the LDS-resident kernel takes
    // Segments above kSegMax rows were flagged and left untouched by the kernel (a tile with more than 512 detections: max_det > 512, or
    // a foreign model without a cap): those go through the dense path of obb_merge_detections, one by one.  (synchronises, 4 bytes;
    // skipped inside a stream capture, where the caller must keep segments within 512 rows: a longer one replays as "identity order, nothing
    // kept" -- defined, and visible in the status word of workspace WS_GEOM_E)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return OBB_OK;
    int32_t flagged = 0;
    OBB_HIP(ctx, hipMemcpyAsync(&flagged, status, sizeof flagged, hipMemcpyDeviceToHost, st));
    OBB_HIP(ctx, hipStreamSynchronize(st));
    if (!flagged) return OBB_OK;
    std::vector<int32_t> off((size_t)nseg + 1);
    OBB_HIP(ctx, hipMemcpy(off.data(), seg_off, sizeof(int32_t) * off.size(), hipMemcpyDeviceToHost));
    for (int32_t k = 0; k < nseg; ++k) {
        const int64_t lo = off[k], len = (int64_t)off[k + 1] - lo;
        if (len <= kSegMax) continue;
        OBB_REQUIRE(ctx, lo >= 0 && lo + len <= n, "obb_merge_segments: segment %d out of range", k);
        int rc = obb_merge_detections(ctx, boxes + 8 * lo, cls + lo, conf + lo, len, thr, order + lo, keep + lo, nullptr, s);
        if (rc) return rc;
        hipLaunchKernelGGL(k_add_offset, dim3((unsigned)cdiv(len, 256)), dim3(256), 0, st, order + lo, len, (int32_t)lo);  // local -> global rows
        OBB_LAUNCH_CHECK(ctx);
    }
    return OBB_OK;
}

int obb_merge_detections(obb_ctx *ctx, const double *boxes, const int32_t *cls, const double *conf, int64_t n, double thr,
                         int32_t *order, uint8_t *keep, int32_t *n_keep, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && n >= 0 && n < (1ll << 31), "obb_merge_detections: bad n");
    hipStream_t st = (hipStream_t)s;
    if (n == 0) {
        if (n_keep) OBB_HIP(ctx, hipMemsetAsync(n_keep, 0, sizeof(int32_t), st));
        return OBB_OK;
    }
    OBB_REQUIRE(ctx, boxes && cls && conf && order && keep, "obb_merge_detections: NULL buffer");
    if (n <= kSegMax) {
        int32_t *segoff = (int32_t *)ctx->workspace(WS_GEOM_D, 256);
        int32_t *status = (int32_t *)ctx->workspace(WS_GEOM_E, 256);
        if (!segoff || !status) return set_error(ctx, OBB_ERR_HIP, "obb_merge_detections: workspace allocation failed");
        hipLaunchKernelGGL(k_set_segment, dim3(1), dim3(64), 0, st, segoff, (int32_t)n, status);  // (a kernel, not a host copy: capturable, no host memory referenced at replay)
        if (n <= kSegMid)
            hipLaunchKernelGGL((k_merge_segments<kSegMid, 256, kSegMidPairCap>), dim3(1), dim3(256), 0, st, boxes, cls, conf, segoff, segoff + 1, thr, order, keep, n_keep, status, 0, 0,
                               (const int32_t *)nullptr, (const int32_t *)nullptr);
        else
            hipLaunchKernelGGL((k_merge_segments<kSegMax, 1024, kSegPairCap>), dim3(1), dim3(1024), 0, st, boxes, cls, conf, segoff, segoff + 1, thr, order, keep, n_keep, status, 0, 0,
                               (const int32_t *)nullptr, (const int32_t *)nullptr);
        OBB_LAUNCH_CHECK(ctx);
        return OBB_OK;
    }
    int64_t W = cdiv(n, 64);
    OBB_REQUIRE(ctx, W <= 65535, "nms: n=%lld exceeds the 4.19M-box grid limit", (long long)n);
    double *sboxes = (double *)ctx->workspace(WS_GEOM_A, sizeof(double) * 8 * (size_t)n);
    int32_t *scls = (int32_t *)ctx->workspace(WS_GEOM_B, sizeof(int32_t) * (size_t)n);
    BoxMeta *meta = (BoxMeta *)ctx->workspace(WS_NMS_A, sizeof(BoxMeta) * (size_t)n);
    if (!sboxes || !scls || !meta) return set_error(ctx, OBB_ERR_HIP, "obb_merge_detections: workspace allocation failed");
    int rc = obb_sort_desc_stable(ctx, conf, n, order, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_prep_sorted, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, boxes, cls, (const int32_t *)order, n, sboxes,
                       scls, meta);
    OBB_LAUNCH_CHECK(ctx);
    // sparse form: suppression pairs as an edge list + parallel DAG relaxation (node states in LDS: n <= 600k).  The dense fall-back
    // (k_nms_lazy) is predicated on the device: no host read of the pair count, no synchronisation -- the call is capturable.
    const size_t lds_states = (size_t)cdiv(n, 4) * 4;
    const size_t lds_lazy = (size_t)W * 8 + 64;
    OBB_REQUIRE(ctx, lds_lazy <= 150 * 1024, "obb_merge_detections: n=%lld exceeds the 1.2M rows the LDS-resident scans hold", (long long)n);
    static std::once_flag attr_once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(attr_once, [] {
        attr_err = hipFuncSetAttribute((const void *)k_nms_resolve, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (attr_err == hipSuccess) attr_err = hipFuncSetAttribute((const void *)k_nms_lazy, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    });
    OBB_HIP(ctx, attr_err);
    unsigned int *ecount = (unsigned int *)ctx->workspace(WS_NMS_D, 256);
    if (!ecount) return set_error(ctx, OBB_ERR_HIP, "obb_merge_detections: workspace allocation failed");
    OBB_HIP(ctx, hipMemsetAsync(ecount, 0, sizeof(unsigned int), st));
    const bool sparse = thr > 0.0 && lds_states <= 150 * 1024;
    unsigned int cap = 0;
    if (sparse) {
        cap = (unsigned int)std::min<int64_t>(64 * n + 4096, 1ll << 28);
        unsigned long long *edges = (unsigned long long *)ctx->workspace(WS_NMS_C, sizeof(unsigned long long) * (size_t)cap);
        if (!edges) return set_error(ctx, OBB_ERR_HIP, "obb_merge_detections: workspace allocation failed");
        if (n >= kGridMin) {  // candidate partners through a uniform grid instead of all pairs
            const int ncell = kGridDim * kGridDim;
            int32_t *gbuf = (int32_t *)ctx->workspace(WS_GEOM_D, sizeof(int32_t) * (2 * (size_t)n + 2 * (size_t)ncell + 64 + 2 * 7 * kGridParts + 2));
            if (!gbuf) return set_error(ctx, OBB_ERR_HIP, "obb_merge_detections: workspace allocation failed");
            int32_t *hist = gbuf, *start = gbuf + ncell + 8, *members = gbuf + 2 * ncell + 32, *large = members + n;
            GridInfo *ginfo = (GridInfo *)(gbuf + 2 * ncell + 16);
            OBB_HIP(ctx, hipMemsetAsync(hist, 0, sizeof(int32_t) * ncell, st));
            double *gpart = (double *)(gbuf + ((2 * (size_t)ncell + 32 + 2 * (size_t)n + 1) & ~(size_t)1));  // behind members | large, 8-byte aligned
            hipLaunchKernelGGL(k_grid_info_part, dim3(kGridParts), dim3(1024), 0, st, (const BoxMeta *)meta, n, gpart);
            hipLaunchKernelGGL(k_grid_info, dim3(1), dim3(64), 0, st, (const double *)gpart, kGridParts, ginfo);
            hipLaunchKernelGGL(k_grid_count, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, (const BoxMeta *)meta, n, ginfo, hist, large);
            hipLaunchKernelGGL(k_cells_scan, dim3(1), dim3(1024), 0, st, (const int32_t *)hist, ncell, start, hist);
            hipLaunchKernelGGL(k_grid_scatter, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, (const BoxMeta *)meta, n, (const GridInfo *)ginfo, hist, members);
            hipLaunchKernelGGL(k_grid_pairs, dim3((unsigned)std::min<int64_t>(cdiv(n, 4), 8192)), dim3(256), 0, st, (const double *)sboxes, (const int32_t *)scls, (const BoxMeta *)meta, n,
                               thr, (const GridInfo *)ginfo, (const int32_t *)start, (const int32_t *)members, (const int32_t *)large, edges, ecount, cap);
        } else {
            hipLaunchKernelGGL(k_nms_mask<true>, dim3((unsigned)W, (unsigned)W), dim3(64), 0, st, (const double *)sboxes, (const int32_t *)scls,
                               (const BoxMeta *)meta, n, thr, (unsigned long long *)nullptr, edges, ecount, cap);
        }
        hipLaunchKernelGGL(k_nms_resolve, dim3(1), dim3(1024), lds_states, st, (const unsigned long long *)edges, (const unsigned int *)ecount, cap, n, keep, n_keep);
    }
    hipLaunchKernelGGL(k_nms_lazy, dim3(1), dim3(1024), lds_lazy, st, (const double *)sboxes, (const int32_t *)scls, (const BoxMeta *)meta, n, thr,
                       (const unsigned int *)ecount, cap, sparse ? 0 : 1, keep, n_keep);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

int obb_consensus(obb_ctx *ctx, const double *boxes, const int32_t *cls, const double *conf, const int64_t *off_host,
                  int32_t nscales, double iou_partner, double cons_low, double cons_high, int32_t *out_idx, int32_t *n_out,
                  obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && off_host && nscales >= 1 && nscales <= 16 && n_out, "obb_consensus: bad arguments");
    hipStream_t st = (hipStream_t)s;
    int64_t total = off_host[nscales];
    for (int k = 0; k < nscales; ++k) OBB_REQUIRE(ctx, off_host[k] <= off_host[k + 1], "obb_consensus: offsets must ascend");
    if (total == 0) { OBB_HIP(ctx, hipMemsetAsync(n_out, 0, sizeof(int32_t), st)); return OBB_OK; }
    OBB_REQUIRE(ctx, boxes && cls && conf && out_idx, "obb_consensus: NULL buffer");
    ConsOff off;
    for (int k = 0; k <= 16; ++k) off.v[k] = off_host[std::min<int>(k, nscales)];
    uint8_t *state = (uint8_t *)ctx->workspace(WS_GEOM_C, (size_t)total * 2 + 256);
    BoxMeta *meta = (BoxMeta *)ctx->workspace(WS_NMS_A, sizeof(BoxMeta) * (size_t)total);
    if (!state || !meta) return set_error(ctx, OBB_ERR_HIP, "obb_consensus: workspace allocation failed");
    if (nscales == 1 || total < 512) {  // passthrough (:357-358), or small enough for the single-workgroup walk
        hipLaunchKernelGGL(k_consensus, dim3(1), dim3(256), 0, st, boxes, cls, conf, off, nscales, iou_partner,
                           cons_low, cons_high, state, meta, out_idx, n_out, (const int32_t *)nullptr);
        OBB_LAUNCH_CHECK(ctx);
        return OBB_OK;
    }
    OBB_REQUIRE(ctx, total < (1ll << 31) && cdiv(total, 64) <= 65535, "obb_consensus: %lld detections exceed the grid limit", (long long)total);
    uint8_t *scale_id = state + (((size_t)total + 127) & ~(size_t)127);
    int32_t *adj_idx = (int32_t *)ctx->workspace(WS_NMS_B, sizeof(int32_t) * (size_t)total * kConsK);
    double *adj_iou = (double *)ctx->workspace(WS_NMS_C, sizeof(double) * (size_t)total * kConsK);
    int32_t *ibuf = (int32_t *)ctx->workspace(WS_GEOM_A, sizeof(int32_t) * ((size_t)total * 5 + 64));  // deg | decision | emit | flags (64) | two work lists
    if (!adj_idx || !adj_iou || !ibuf) return set_error(ctx, OBB_ERR_HIP, "obb_consensus: workspace allocation failed");
    int32_t *deg = ibuf, *decision = ibuf + total, *emit = ibuf + 2 * total, *flags = ibuf + 3 * total;
    hipLaunchKernelGGL(k_cons_prep, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, boxes, conf, off, nscales, cons_low, state, scale_id, meta,
                       deg, flags);
    OBB_LAUNCH_CHECK(ctx);
    for (int b = 1; b < nscales; ++b) {
        const int64_t jb = off_host[b], je = off_host[b + 1];
        if (je <= jb || jb <= 0) continue;
        if (cdiv(je - jb, 256) * cdiv(jb, 256) >= 2048)
            hipLaunchKernelGGL(k_cons_edges<256>, dim3((unsigned)cdiv(je - jb, 256), (unsigned)cdiv(jb, 256)), dim3(256), 0, st, boxes, cls, (const uint8_t *)state,
                               (const uint8_t *)scale_id, (const BoxMeta *)meta, jb, je, iou_partner, adj_idx, adj_iou, deg, flags);
        else
            hipLaunchKernelGGL(k_cons_edges<64>, dim3((unsigned)cdiv(je - jb, 256), (unsigned)cdiv(jb, 64)), dim3(256), 0, st, boxes, cls, (const uint8_t *)state,
                               (const uint8_t *)scale_id, (const BoxMeta *)meta, jb, je, iou_partner, adj_idx, adj_iou, deg, flags);
        OBB_LAUNCH_CHECK(ctx);
    }
    hipLaunchKernelGGL(k_cons_resolve, dim3(1), dim3(1024), 0, st, conf, total, cons_high, state, (const int32_t *)adj_idx, (const double *)adj_iou,
                       (const int32_t *)deg, decision, emit, flags, flags + 64, out_idx, n_out);
    OBB_LAUNCH_CHECK(ctx);
    // a detection with more than kConsK candidate partners (flags[0]): the walk itself, on untouched state (it re-derives state and meta)
    hipLaunchKernelGGL(k_consensus, dim3(1), dim3(256), 0, st, boxes, cls, conf, off, nscales, iou_partner, cons_low, cons_high, state,
                       meta, out_idx, n_out, (const int32_t *)flags);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

int obb_tile_grid(int32_t H, int32_t W, int32_t tile, int32_t overlap, int32_t *rects_host, int64_t max_tiles, int64_t *n_tiles) {
    if (H < 0 || W < 0 || tile <= 0 || !n_tiles) return set_error(nullptr, OBB_ERR_INVALID, "obb_tile_grid: bad arguments");
    int32_t step = tile - overlap;
    if (step < 1) step = 1;  // Detect_OBB.py:211
    int64_t n = 0;
    for (int32_t y = 0; y < H; y += step)
        for (int32_t x = 0; x < W; x += step) {
            int32_t y2 = y + tile < H ? y + tile : H, x2 = x + tile < W ? x + tile : W;
            if (y2 - y == 0 || x2 - x == 0) continue;  // :222
            if (rects_host && n < max_tiles) { rects_host[4 * n] = x; rects_host[4 * n + 1] = y; rects_host[4 * n + 2] = x2; rects_host[4 * n + 3] = y2; }
            ++n;
        }
    *n_tiles = n;
    return OBB_OK;
}

int obb_tile_postprocess(obb_ctx *ctx, const float *local_pts, const int32_t *cls, const int32_t *det_tile, int64_t n,
                         const int32_t *rects, int32_t ntiles, int32_t margin, int32_t strike_cls, double *gboxes, double *angle,
                         uint8_t *inside, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && n >= 0 && ntiles >= 0, "obb_tile_postprocess: bad arguments");
    if (n == 0) return OBB_OK;
    OBB_REQUIRE(ctx, local_pts && cls && det_tile && rects && gboxes && angle && inside, "obb_tile_postprocess: NULL buffer");
    hipLaunchKernelGGL(k_tile_post, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, local_pts, cls, det_tile, n, rects,
                       margin, strike_cls, gboxes, angle, inside);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

int obb_tile_survivors(obb_ctx *ctx, const float *det, const int32_t *count, int32_t B, int32_t max_det, const float *lb, const int32_t *tile_ids,
                       const int32_t *rects, int32_t margin, int32_t strike_cls, double iou_thr, int32_t *records, int32_t *tile_off, int32_t *n_records,
                       obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && B >= 0 && max_det >= 1, "obb_tile_survivors: bad arguments");
    OBB_REQUIRE(ctx, max_det <= kSegMax, "obb_tile_survivors: max_det %d exceeds the LDS-resident segment merge (%d rows)", max_det, kSegMax);
    hipStream_t st = (hipStream_t)s;
    OBB_REQUIRE(ctx, n_records, "obb_tile_survivors: NULL n_records");
    if (B == 0) { OBB_HIP(ctx, hipMemsetAsync(n_records, 0, sizeof(int32_t), st)); return OBB_OK; }
    OBB_REQUIRE(ctx, det && count && tile_ids && rects && records && tile_off, "obb_tile_survivors: NULL buffer");
    const int64_t cap = (int64_t)B * max_det;
    OBB_REQUIRE(ctx, cap < (1ll << 31), "obb_tile_survivors: B * max_det too large");
    // scratch: staging rows + merge outputs (grow-only workspace slots of the context)
    const size_t a_bytes = (size_t)cap * (64 + 8 + 32 + 4 + 4) + 256, b_bytes = (size_t)cap * (4 + 1) + (size_t)B * 12 + 1024;
    char *wa = (char *)ctx->workspace(WS_SURV_A, a_bytes), *wb = (char *)ctx->workspace(WS_SURV_B, b_bytes);
    int32_t *status = (int32_t *)ctx->workspace(WS_GEOM_E, 256 + sizeof(int32_t) * (size_t)B);  // status | long-segment count | their list
    if (!wa || !wb || !status) return set_error(ctx, OBB_ERR_HIP, "obb_tile_survivors: workspace allocation failed");
    SurvStage S;
    S.gb = (double *)wa; S.conf = (double *)(wa + (size_t)cap * 64); S.pts = (float *)(wa + (size_t)cap * 72); S.cls = (int32_t *)(wa + (size_t)cap * 104);
    S.conf32 = (float *)(wa + (size_t)cap * 108);
    int32_t *order = (int32_t *)wb;
    S.lo = (int32_t *)(wb + (size_t)cap * 4); S.hi = S.lo + B;
    int32_t *nkeep = S.hi + B;
    uint8_t *keep = (uint8_t *)(nkeep + B + 4);
    OBB_HIP(ctx, hipMemsetAsync(status, 0, 2 * sizeof(int32_t), st));
    hipLaunchKernelGGL(k_tile_stage, dim3((unsigned)B), dim3(256), 0, st, det, count, (int)max_det, lb, tile_ids, rects, (int)margin, (int)strike_cls, S);
    hipLaunchKernelGGL(k_merge_segments_wave, dim3((unsigned)B), dim3(64), 0, st, S.gb, S.cls, S.conf, S.lo, S.hi, iou_thr, order, keep, nkeep, status + 64, status + 1);
    if (max_det > kSegWave)  // (tiles with more than 64 survivors: resident workgroups walk the list the wave kernel left)
        hipLaunchKernelGGL((k_merge_segments<kSegMax, 1024, kSegPairCap>), dim3((unsigned)std::min<int>(B, merge_resident(ctx))), dim3(1024), 0, st, S.gb, S.cls, S.conf, S.lo,
                           S.hi, iou_thr, order, keep, nkeep, status, kSegWave, 0, status + 64, status + 1);
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, st, nkeep, (int)B, tile_off, n_records);
    hipLaunchKernelGGL(k_tile_emit, dim3((unsigned)B), dim3(64), 0, st, S, order, keep, tile_off, tile_ids, records);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

int obb_select_kept(obb_ctx *ctx, const int32_t *order, const uint8_t *keep, int64_t n, const double *boxes, const int32_t *cls, const double *conf,
                    const double *angle, double *out_boxes, int32_t *out_cls, double *out_conf, double *out_angle, int32_t *n_out, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && n >= 0 && n_out, "obb_select_kept: bad arguments");
    if (n == 0) { OBB_HIP(ctx, hipMemsetAsync(n_out, 0, sizeof(int32_t), (hipStream_t)s)); return OBB_OK; }
    OBB_REQUIRE(ctx, order && keep && boxes && cls && conf && out_boxes && out_cls && out_conf && (!angle || out_angle), "obb_select_kept: NULL buffer");
    const int nb = (int)cdiv(n, 1024);
    int32_t *cnt = (int32_t *)ctx->workspace(WS_SEL, (size_t)(2 * nb + 2) * 4);
    if (!cnt) return set_error(ctx, OBB_ERR_HIP, "obb_select_kept: workspace allocation failed");
    int32_t *off = cnt + nb;
    hipStream_t st = (hipStream_t)s;
    hipLaunchKernelGGL(k_kept_count, dim3((unsigned)nb), dim3(1024), 0, st, keep, n, cnt);
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, st, cnt, nb, off, n_out);
    hipLaunchKernelGGL(k_kept_scatter, dim3((unsigned)nb), dim3(1024), 0, st, order, keep, n, off, boxes, cls, conf, angle, out_boxes, out_cls, out_conf, out_angle);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

int obb_records_to_dets(obb_ctx *ctx, const int32_t *records, int64_t n, const int32_t *rects, int32_t strike_cls, double *gboxes, int32_t *cls,
                        double *conf, double *angle, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && n >= 0, "obb_records_to_dets: bad arguments");
    if (n == 0) return OBB_OK;
    OBB_REQUIRE(ctx, records && rects && gboxes && cls && conf && angle, "obb_records_to_dets: NULL buffer");
    hipLaunchKernelGGL(k_records_to_dets, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, records, n, rects, (int)strike_cls, gboxes, cls, conf, angle);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

int obb_gather_compact(obb_ctx *ctx, const int32_t *recv, int32_t world, int32_t capacity, int32_t *out, int32_t *counts, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && world >= 1 && world <= 4096 && capacity >= 1, "obb_gather_compact: bad arguments");
    OBB_REQUIRE(ctx, recv && out && counts, "obb_gather_compact: NULL buffer");
    hipLaunchKernelGGL(k_gather_compact, dim3((unsigned)cdiv((int64_t)capacity * 3, 256), (unsigned)world), dim3(256), 0, (hipStream_t)s, recv, (int)world, (int)capacity, out, counts);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

int obb_tile_labels(obb_ctx *ctx, const double *labels, int64_t n, const int32_t *rects, int32_t ntiles, double min_fraction, uint8_t *mask,
                    double *out, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && n >= 0 && ntiles >= 0, "obb_tile_labels: bad arguments");
    if (n == 0 || ntiles == 0) return OBB_OK;
    OBB_REQUIRE(ctx, labels && rects && mask && out, "obb_tile_labels: NULL buffer");
    const int64_t total = (int64_t)ntiles * n;
    OBB_REQUIRE(ctx, cdiv(total, 256) < (1ll << 31), "obb_tile_labels: too many (tile, label) pairs");
    hipLaunchKernelGGL(k_tile_labels, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)s, labels, n, rects, ntiles, min_fraction, mask, out);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

}  // extern "C"
