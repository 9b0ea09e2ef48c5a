// 4-channel network input: RGB + distance-transform edge channel, built per crop on the device.
// Replaces `build_multich(crop_bgr, out_channels=4)` of Detect_OBB.py:87-133 (== Train_OBB.py:615-653),
// i.e. the OpenCV / numpy sequence  BGR2GRAY -> GaussianBlur x3 -> Scharr magnitude, max over scales -> 90th percentile threshold ->
// 3x3 cross opening -> 3x3 chamfer distance transform -> [1, 99] percentile normalisation -> 0.7 exp(-d/3) + 0.3 minmax(acc) -> uint8.
// Every cv2 call is restated from its documented algorithm (fixed-point grey / blur / chamfer arithmetic, REFLECT_101 borders); the
// numpy steps (float64 linear-interpolation percentiles, float64 promotion afterwards, truncating cast) are reproduced operation by
// operation.  OpenCV is absent offline: parity with cv2 itself is unpinned; the test-suite checks this file against a numpy restatement.
//
// One workgroup (1024 threads) per crop walks the whole sequence; intermediates live in a per-crop global scratch area (L2 resident),
// percentiles are exact order statistics found by a 4-pass radix select on the float bit patterns with LDS histograms, the distance
// transform's row recurrences d[x] = min(c[x], d[x-1] + a) are prefix minima of c[k] - k a (wave shuffles + one LDS combine per row).
#include <hip/hip_runtime.h>

#include "ctx.h"

namespace obb {

constexpr int kDtThreads = 1024;
constexpr int kHV = 62587, kDIAG = 89738, kINIT = 0x7fffffff >> 2;  // round(0.955 * 2^16), round(1.3693 * 2^16), INT_MAX >> 2

struct DtKernels { int n[3]; int q[3][15]; };  // Gaussian kernels in 1/256 units (5, 9, 15 taps)

__device__ __forceinline__ int reflect101(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

// make this workgroup's global-memory writes visible to all of its waves (the vector L1 is not coherent with stores)
__device__ __forceinline__ void tile_sync() {
    __threadfence();
    __syncthreads();
    __threadfence();
}

__device__ __forceinline__ float scharr_mag_at(const uint8_t *img, int y, int x, int h, int w) {
    const int ym = reflect101(y - 1, h), yp = reflect101(y + 1, h), xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
    const float a00 = img[ym * w + xm], a01 = img[ym * w + x], a02 = img[ym * w + xp];
    const float a10 = img[y * w + xm], a12 = img[y * w + xp];
    const float a20 = img[yp * w + xm], a21 = img[yp * w + x], a22 = img[yp * w + xp];
    const float dx = 3.f * (a02 - a00) + 10.f * (a12 - a10) + 3.f * (a22 - a20);
    const float dy = 3.f * (a20 - a00) + 10.f * (a21 - a01) + 3.f * (a22 - a02);
    return sqrtf(dx * dx + dy * dy);
}

// k-th and (k+1)-th smallest of n non-negative floats (bit patterns are order preserving): 4-pass radix select, LDS histogram
__device__ void select_pair(const float *v, int n, int k, float *out2, unsigned *hist /*256*/, unsigned *sh /*8*/) {
    const int tid = threadIdx.x;
    unsigned prefix = 0, mask = 0;
    int kk = k;
    unsigned cnt_eq = 0;
    for (int pass = 3; pass >= 0; --pass) {
        for (int i = tid; i < 256; i += kDtThreads) hist[i] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += kDtThreads) {
            unsigned key = __float_as_uint(v[i]);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> (8 * pass)) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned cum = 0;
            int b = 0;
            for (; b < 256; ++b) {
                if (cum + hist[b] > (unsigned)kk) break;
                cum += hist[b];
            }
            sh[0] = (unsigned)b; sh[1] = cum; sh[2] = hist[b];
        }
        __syncthreads();
        prefix |= sh[0] << (8 * pass);
        mask |= 0xffu << (8 * pass);
        kk -= (int)sh[1];
        cnt_eq = sh[2];
        __syncthreads();
    }
    // prefix = key of the k-th element; kk = its rank among the cnt_eq equal elements
    float next = __uint_as_float(prefix);
    if (kk + 1 >= (int)cnt_eq) {  // the next order statistic is the smallest element above it
        if (tid == 0) sh[3] = 0x7f800000u;
        __syncthreads();
        unsigned best = 0x7f800000u;
        for (int i = tid; i < n; i += kDtThreads) {
            unsigned key = __float_as_uint(v[i]);
            if (key > prefix && key < best) best = key;
        }
        atomicMin(&sh[3], best);
        __syncthreads();
        next = sh[3] == 0x7f800000u ? __uint_as_float(prefix) : __uint_as_float(sh[3]);
        __syncthreads();
    }
    out2[0] = __uint_as_float(prefix);
    out2[1] = next;
}

// numpy.percentile(a, [q])[0] with method="linear" for a float32 array: float32 difference, float64 interpolation
__device__ __forceinline__ double np_percentile(const float *v, int n, double q, unsigned *hist, unsigned *sh) {
    const double vi = (double)(n - 1) * (q / 100.0);
    const int i = (int)floor(vi);
    const double t = vi - (double)i;
    float ab[2];
    select_pair(v, n, i, ab, hist, sh);
    if (i + 1 > n - 1) ab[1] = ab[0];
    const float diff = ab[1] - ab[0];
    double r = (double)ab[0] + (double)diff * t;
    if (t >= 0.5) r = (double)ab[1] - (double)diff * (1.0 - t);
    return r;
}

// inclusive prefix minimum over the first w threads (w <= 1024); `part` = 16 ints of LDS
__device__ __forceinline__ int block_prefix_min(int v, int x, int w, int *part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (x >= w) v = 0x7fffffff;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int u = __shfl_up(v, o);
        if (lane >= o) v = min(v, u);
    }
    if (lane == 63) part[wave] = v;
    __syncthreads();
    int pre = 0x7fffffff;
    for (int k = 0; k < wave; ++k) pre = min(pre, part[k]);
    __syncthreads();
    return min(v, pre);
}

__global__ __launch_bounds__(kDtThreads) void k_build_multich(const uint8_t *__restrict__ bgr, int B, int h, int w, uint8_t *__restrict__ out4,
                                                               char *__restrict__ scratch, size_t scratch_per_tile, DtKernels K) {
    __shared__ unsigned s_hist[256];
    __shared__ unsigned s_sh[8];
    __shared__ int s_part[16];
    __shared__ int s_row[2][1024 + 2];
    __shared__ float s_red[2][16];
    const int tid = threadIdx.x, n = h * w;
    const uint8_t *src = bgr + (size_t)blockIdx.x * n * 3;
    uint8_t *dst = out4 + (size_t)blockIdx.x * n * 4;
    char *sc = scratch + (size_t)blockIdx.x * scratch_per_tile;
    float *acc = reinterpret_cast<float *>(sc);                     // n floats
    float *dist = acc + n;                                          // n floats
    int *tt = reinterpret_cast<int *>(dist + n);                    // n ints (chamfer distances, 16.16)
    unsigned short *rowt = reinterpret_cast<unsigned short *>(tt + n);  // n u16 (horizontal blur, 8.8)
    uint8_t *gray = reinterpret_cast<uint8_t *>(rowt + n), *blur = gray + n, *edges = blur + n, *er = edges + n;

    // ---- grey (8-bit fixed point) and the unblurred scale
    for (int i = tid; i < n; i += kDtThreads) {
        const int b = src[i * 3], g = src[i * 3 + 1], r = src[i * 3 + 2];
        gray[i] = (uint8_t)((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14);
    }
    tile_sync();
    for (int i = tid; i < n; i += kDtThreads) acc[i] = scharr_mag_at(gray, i / w, i % w, h, w);
    // ---- three Gaussian scales: separable 8-bit fixed-point blur, Scharr magnitude, running maximum
    for (int s = 0; s < 3; ++s) {
        const int nk = K.n[s], r = nk / 2;
        for (int i = tid; i < n; i += kDtThreads) {
            const int y = i / w, x = i - y * w;
            int a = 0;
            for (int k = 0; k < nk; ++k) a += K.q[s][k] * (int)gray[y * w + reflect101(x + k - r, w)];
            rowt[i] = (unsigned short)a;
        }
        tile_sync();
        for (int i = tid; i < n; i += kDtThreads) {
            const int y = i / w, x = i - y * w;
            int a = 0;
            for (int k = 0; k < nk; ++k) a += K.q[s][k] * (int)rowt[reflect101(y + k - r, h) * w + x];
            blur[i] = (uint8_t)((a + (1 << 15)) >> 16);
        }
        tile_sync();
        for (int i = tid; i < n; i += kDtThreads) acc[i] = fmaxf(acc[i], scharr_mag_at(blur, i / w, i % w, h, w));
        tile_sync();
    }
    // ---- edges = acc >= percentile(acc, 90)   (numpy: float64 threshold, float32 values promoted for the comparison)
    const double thr = np_percentile(acc, n, 90.0, s_hist, s_sh);
    for (int i = tid; i < n; i += kDtThreads) edges[i] = ((double)acc[i] >= thr) ? 255 : 0;
    tile_sync();
    // ---- opening with the 3x3 cross (pixels outside the image do not take part)
    for (int i = tid; i < n; i += kDtThreads) {
        const int y = i / w, x = i - y * w;
        uint8_t m = edges[i];
        if (y > 0) m = min(m, edges[i - w]);
        if (y < h - 1) m = min(m, edges[i + w]);
        if (x > 0) m = min(m, edges[i - 1]);
        if (x < w - 1) m = min(m, edges[i + 1]);
        er[i] = m;
    }
    tile_sync();
    for (int i = tid; i < n; i += kDtThreads) {
        const int y = i / w, x = i - y * w;
        uint8_t m = er[i];
        if (y > 0) m = max(m, er[i - w]);
        if (y < h - 1) m = max(m, er[i + w]);
        if (x > 0) m = max(m, er[i - 1]);
        if (x < w - 1) m = max(m, er[i + 1]);
        edges[i] = m;
    }
    tile_sync();
    // ---- 3x3 chamfer distance to the nearest edge pixel, 16.16 fixed point: forward then backward sweep, one row per step
    {
        const int x = tid;
        int *prev = s_row[0], *cur = s_row[1];
        for (int i = tid; i < w + 2; i += kDtThreads) prev[i] = kINIT;
        __syncthreads();
        for (int y = 0; y < h; ++y) {
            int c = 0x7fffffff;
            if (x < w) {
                c = min(min(prev[x] + kDIAG, prev[x + 1] + kHV), prev[x + 2] + kDIAG);
                if (edges[y * w + x]) c = 0;
            }
            int m = block_prefix_min(x < w ? c - x * kHV : 0x7fffffff, x, w, s_part);
            if (x < w) {
                int d = min(m + x * kHV, kINIT + kHV + x * kHV);
                cur[x + 1] = d;
                tt[y * w + x] = d;
            }
            if (tid == 0) { cur[0] = kINIT; cur[w + 1] = kINIT; }
            __syncthreads();
            int *t = prev; prev = cur; cur = t;
        }
        tile_sync();  // the backward sweep reads the forward distances written by other threads
        for (int i = tid; i < w + 2; i += kDtThreads) prev[i] = kINIT;
        __syncthreads();
        for (int y = h - 1; y >= 0; --y) {
            const int xr = w - 1 - x;  // position counted from the right end
            int c = 0x7fffffff;
            if (x < w) c = min(min(tt[y * w + xr], prev[xr + 2] + kDIAG), min(prev[xr + 1] + kHV, prev[xr] + kDIAG));
            int m = block_prefix_min(x < w ? c - x * kHV : 0x7fffffff, x, w, s_part);
            if (x < w) {
                int d = min(m + x * kHV, kINIT + kHV + x * kHV);
                cur[xr + 1] = d;
                dist[y * w + xr] = (float)d * (1.0f / 65536.0f);
            }
            if (tid == 0) { cur[0] = kINIT; cur[w + 1] = kINIT; }
            __syncthreads();
            int *t = prev; prev = cur; cur = t;
        }
    }
    tile_sync();
    // ---- [1, 99] percentiles of the distance, min / max of acc
    const double lo = np_percentile(dist, n, 1.0, s_hist, s_sh);
    const double hi = np_percentile(dist, n, 99.0, s_hist, s_sh);
    float mn = INFINITY, mx = -INFINITY;
    for (int i = tid; i < n; i += kDtThreads) { mn = fminf(mn, acc[i]); mx = fmaxf(mx, acc[i]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
    if ((tid & 63) == 0) { s_red[0][tid >> 6] = mn; s_red[1][tid >> 6] = mx; }
    __syncthreads();
    mn = s_red[0][0]; mx = s_red[1][0];
    for (int k = 1; k < kDtThreads / 64; ++k) { mn = fminf(mn, s_red[0][k]); mx = fmaxf(mx, s_red[1][k]); }
    const double scale_d = (mx > mn) ? 1.0 / ((double)mx - (double)mn) : 0.0;
    const float scale = (float)scale_d, shift = (float)(-(double)mn * scale_d);
    const double den = fmax(1e-6, hi - lo);
    // ---- 0.7 exp(-d / 3) + 0.3 minmax(acc), clipped, * 255, truncated; output RGB + that channel
    for (int i = tid; i < n; i += kDtThreads) {
        double d = ((double)dist[i] - lo) / den;
        d = fmin(fmax(d, 0.0), 1.0);
        double soft = exp(-d / 3.0);
        const float nrm = acc[i] * scale + shift;
        soft = 0.7 * soft + (double)(0.3f * nrm);  // numpy: python scalar * float32 array stays float32, then promotes in the sum
        soft = fmin(fmax(soft, 0.0), 1.0);
        dst[i * 4 + 0] = src[i * 3 + 2];
        dst[i * 4 + 1] = src[i * 3 + 1];
        dst[i * 4 + 2] = src[i * 3 + 0];
        dst[i * 4 + 3] = (uint8_t)(soft * 255.0);
    }
}

static void gauss_q8(double sigma, int *n_out, int *q) {
    int n = (int)lrint(sigma * 6.0 + 1.0) | 1;
    double g[15], sum = 0;
    for (int i = 0; i < n; ++i) { double d = i - (n - 1) / 2.0; g[i] = exp(-(d * d) / (2.0 * sigma * sigma)); sum += g[i]; }
    int tot = 0;
    for (int i = 0; i < n; ++i) { q[i] = (int)nearbyint(g[i] / sum * 256.0); tot += q[i]; }
    q[n / 2] += 256 - tot;
    *n_out = n;
}

}  // namespace obb

using namespace obb;

extern "C" int obb_build_multich(obb_ctx *ctx, const uint8_t *bgr, int32_t B, int32_t h, int32_t w, uint8_t *out4, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && B >= 0 && h > 0 && w > 0, "obb_build_multich: bad arguments");
    if (B == 0) return OBB_OK;
    OBB_REQUIRE(ctx, bgr && out4, "obb_build_multich: NULL buffer");
    OBB_REQUIRE(ctx, w <= kDtThreads && h <= 4096 && (int64_t)h * w >= 2, "obb_build_multich: crop %dx%d unsupported (width <= 1024)", h, w);
    const size_t n = (size_t)h * w;
    const size_t per_tile = (n * (4 + 4 + 4 + 2 + 4) + 255) / 256 * 256;
    char *scratch = (char *)ctx->workspace(WS_GEOM_E, per_tile * (size_t)B);
    if (!scratch) return set_error(ctx, OBB_ERR_HIP, "obb_build_multich: workspace allocation failed");
    DtKernels K;
    const double sig[3] = {0.6, 1.2, 2.4};  // MS_SIGMAS without the unblurred scale 0 (Detect_OBB.py:29)
    for (int i = 0; i < 3; ++i) gauss_q8(sig[i], &K.n[i], K.q[i]);
    hipLaunchKernelGGL(k_build_multich, dim3((unsigned)B), dim3(kDtThreads), 0, (hipStream_t)s, bgr, B, h, w, out4, scratch, per_tile, K);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}
