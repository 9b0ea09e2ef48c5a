// 4-channel network input: RGB + distance-transform edge channel, built per crop on the device.
// Replaces `build_multich(crop_bgr, out_channels=4)` of Detect_OBB.py:87-133 (== Train_OBB.py:615-653),
// i.e. the OpenCV / numpy sequence  BGR2GRAY -> GaussianBlur x3 -> Scharr magnitude, max over scales -> 90th percentile threshold ->
// 3x3 cross opening -> 3x3 chamfer distance transform -> [1, 99] percentile normalisation -> 0.7 exp(-d/3) + 0.3 minmax(acc) -> uint8.
// Every cv2 call is restated from its documented algorithm (fixed-point grey / blur / chamfer arithmetic, REFLECT_101 borders); the
// numpy steps (float64 linear-interpolation percentiles, float64 promotion afterwards, truncating cast) are reproduced operation by
// operation.  OpenCV is absent offline: parity with cv2 itself is unpinned; the test-suite checks this file against a numpy restatement.
//
// Six launches per batch of crops, every intermediate in a per-crop scratch image of pitch P = 64 * PX floats (PX = 2 / 4 / 8 / 16):
//   k_dt_acc      64 x 32 pixel tiles: grey tile (+ halo, reflected at load time) in LDS, the three Gaussian scales as separable
//                 fixed-point passes on v_dot4_u32_u8 / v_dot2_u32_u16, Scharr in integers; max over scales of dx^2 + dy^2 is kept as an
//                 int (sqrt and int -> float are monotone, so one correctly rounded sqrt of the maximum equals the maximum of the roots)
//   k_dt_select   one workgroup per crop: exact order statistics by a 3-pass (11 + 10 + 10 bit) radix select on the float bit patterns,
//                 the successor of the selected key tracked on the way; numpy's float64 interpolation; min / max of acc
//   k_dt_edges    threshold + 3x3 cross opening on LDS tiles
//   k_dt_chamfer  ONE WAVE per crop: the row recurrence d[x] = min(c[x], d[x-1] + a) is a prefix minimum of c[k] - k a, done as a
//                 per-lane scan over PX pixels + a DPP wave scan; previous row in registers, rows prefetched 4 ahead, branch-free
//   k_dt_select   [1, 99] percentiles of the distance (both ranks in one walk)
//   k_dt_blend    float64 blend, uint8 output
#include <hip/hip_runtime.h>

#include "ctx.h"

namespace obb {

constexpr int kHV = 62587, kDIAG = 89738, kINIT = 0x7fffffff >> 2;  // round(0.955 * 2^16), round(1.3693 * 2^16), INT_MAX >> 2
constexpr int kTW = 64, kTH = 32;                                    // pixel tile of k_dt_acc / k_dt_edges

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// Gaussian kernels in 1/256 units (5, 9, 15 taps; every tap < 256), packed for the dot instructions
struct DtTaps {
    unsigned h[3][4];   // horizontal pass: taps 4d .. 4d+3 as bytes
    unsigned ve[3][8];  // vertical pass, window starting on an even row: (q[2d], q[2d+1]) as u16 pairs
    unsigned vo[3][8];  // window starting on an odd row: (0, q[0]), (q[1], q[2]), ...
};

struct DtStats { double thr, lo, hi; float mn, mx; int pad[2]; };

// BORDER_REFLECT_101 for any offset (cv2's borderInterpolate folds until the index is inside): the extension is even and periodic
// with period 2 (n - 1).  SMALL = false: one reflection, branch-free -- all that a crop of 16+ pixels asks for (7-tap radius + Scharr;
// cells further out only exist in tiles that overhang the image and feed no valid pixel: clamped).  SMALL = true: the general fold.
template <bool SMALL>
__device__ __forceinline__ int reflect_clamp(int i, int n) {
    if constexpr (SMALL) {
        if (n == 1) return 0;
        const int period = 2 * (n - 1);
        int r = i % period;
        r = r < 0 ? r + period : r;
        return r < n ? r : period - r;
    } else {
        int r = i < 0 ? -i : i;
        r = r >= n ? 2 * (n - 1) - r : r;
        return min(max(r, 0), n - 1);
    }
}

// ------------------------------------------------------------------------------------------------------------------ k_dt_acc

// One horizontal output: taps c-R .. c+R of the grey bytes in D (bytes 0 .. 19), c = 8 + I
template <int R, int I>
__device__ __forceinline__ unsigned hsum(const unsigned (&D)[5], const unsigned (&wq)[4]) {
    constexpr int sb = 8 + I - R, di = sb >> 2, sh = sb & 3, nd = (2 * R + 1 + 3) / 4;
    unsigned s = 0;
#pragma unroll
    for (int d = 0; d < nd; ++d) {
        unsigned win;
        if constexpr (sh == 0) win = D[di + d];
        else win = __builtin_amdgcn_alignbyte(D[di + d + 1], D[di + d], sh);
        s = __builtin_amdgcn_udot4(win, wq[d], s, false);
    }
    return s;
}

// Scharr of four horizontally adjacent pixels from three rows of bytes; `p` -> the dword holding the bytes left of the quad's dword,
// in the row above.  Returns dx^2 + dy^2 (exact integers) folded into s[] by max.
template <int PITCH>
__device__ __forceinline__ void scharr_quad_max(const unsigned *p, int (&s)[4]) {
    unsigned D[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) D[r][k] = p[r * PITCH + k];
    constexpr unsigned kW = 0x00030a03u;  // (3, 10, 3, 0)
    // column sums S(k) = 3 a0[k] + 10 a1[k] + 3 a2[k] for bytes k = 3 .. 8
    int S[6];
#pragma unroll
    for (int k = 3; k < 9; ++k) {
        const int dk = k >> 2, bk = k & 3;
        const unsigned sel1 = (unsigned)bk | ((unsigned)(4 + bk) << 8) | 0x0c0c0000u;
        const unsigned sel2 = 0x0c000100u | ((unsigned)(4 + bk) << 16);
        unsigned t = __builtin_amdgcn_perm(D[1][dk], D[0][dk], sel1);
        t = __builtin_amdgcn_perm(D[2][dk], t, sel2);
        S[k - 3] = (int)__builtin_amdgcn_udot4(t, kW, 0u, false);
    }
    // row sums T(i) = 3 a[3+i] + 10 a[4+i] + 3 a[5+i] for rows 0 and 2
    int T0[4], T2[4];
    {
        const unsigned w0[4] = {__builtin_amdgcn_alignbyte(D[0][1], D[0][0], 3), D[0][1], __builtin_amdgcn_alignbyte(D[0][2], D[0][1], 1),
                                __builtin_amdgcn_alignbyte(D[0][2], D[0][1], 2)};
        const unsigned w2[4] = {__builtin_amdgcn_alignbyte(D[2][1], D[2][0], 3), D[2][1], __builtin_amdgcn_alignbyte(D[2][2], D[2][1], 1),
                                __builtin_amdgcn_alignbyte(D[2][2], D[2][1], 2)};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            T0[i] = (int)__builtin_amdgcn_udot4(w0[i], kW, 0u, false);
            T2[i] = (int)__builtin_amdgcn_udot4(w2[i], kW, 0u, false);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int dx = S[i + 2] - S[i], dy = T2[i] - T0[i];
        s[i] = max(s[i], __mul24(dx, dx) + __mul24(dy, dy));
    }
}

constexpr int kGR = kTH + 16;   // grey rows: virtual y0-8 .. y0+TH+7
constexpr int kGPD = 22;        // grey dwords per row: virtual x0-12 .. x0+TW+11
constexpr int kHRP = kGR / 2;   // row pairs of the horizontally blurred tile (same rows as the grey tile)
constexpr int kHC = 72;         // its columns: virtual x0-4 .. x0+TW+3  (also the blurred tile's)
constexpr int kBR = kTH + 2;    // blurred rows: virtual y0-1 .. y0+TH

template <int R>
__device__ __forceinline__ void hpass(const unsigned *s_g, unsigned *s_hb, const unsigned (&wq)[4]) {
    constexpr int rp0 = (7 - R) >> 1, rp1 = (kTH + 8 + R) >> 1, nrp = rp1 - rp0 + 1;
    for (int it = threadIdx.x; it < nrp * 18; it += 256) {
        const int rp = rp0 + it / 18, g = it % 18;
        unsigned lo[4], hi[4];
        {
            unsigned D[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) D[k] = s_g[(2 * rp) * kGPD + g + k];
            lo[0] = hsum<R, 0>(D, wq); lo[1] = hsum<R, 1>(D, wq); lo[2] = hsum<R, 2>(D, wq); lo[3] = hsum<R, 3>(D, wq);
        }
        {
            unsigned D[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) D[k] = s_g[(2 * rp + 1) * kGPD + g + k];
            hi[0] = hsum<R, 0>(D, wq); hi[1] = hsum<R, 1>(D, wq); hi[2] = hsum<R, 2>(D, wq); hi[3] = hsum<R, 3>(D, wq);
        }
        uint4 o;
        o.x = lo[0] | (hi[0] << 16); o.y = lo[1] | (hi[1] << 16); o.z = lo[2] | (hi[2] << 16); o.w = lo[3] | (hi[3] << 16);
        *reinterpret_cast<uint4 *>(&s_hb[rp * kHC + 4 * g]) = o;
    }
}

template <int R>
__device__ __forceinline__ void vpass(const unsigned *s_hb, uint8_t *s_bv, const unsigned (&ve)[8], const unsigned (&vo)[8]) {
    constexpr int c7 = 7 - R, NW = ((c7 + 5) >> 1) - (c7 >> 1) + R + 1;
    for (int it = threadIdx.x; it < kHC * 6; it += 256) {
        const int gq = it / kHC, cb = it - gq * kHC, t0 = 3 * gq, pb = t0 + (c7 >> 1);
        unsigned W[NW];
#pragma unroll
        for (int k = 0; k < NW; ++k) W[k] = s_hb[min(pb + k, kHRP - 1) * kHC + cb];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int w0 = ((c7 + u) >> 1) - (c7 >> 1);
            const bool odd = ((c7 + u) & 1) != 0;
            unsigned s = 1u << 15;
#pragma unroll
            for (int d = 0; d <= R; ++d)
                s = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, W[w0 + d]), __builtin_bit_cast(u16x2, odd ? vo[d] : ve[d]), s, false);
            const int rb = 2 * t0 + u;
            if (rb < kBR) s_bv[rb * kHC + cb] = (uint8_t)(s >> 16);
        }
    }
}

template <bool SMALL>
__global__ __launch_bounds__(256) void k_dt_acc(const uint8_t *__restrict__ bgr, int h, int w, float *__restrict__ acc, size_t crop_stride, int pitch,
                                                DtTaps K) {
    __shared__ unsigned s_g[kGR * kGPD];
    __shared__ __attribute__((aligned(16))) unsigned s_hb[kHRP * kHC];
    __shared__ unsigned s_bv[kBR * kHC / 4];
    const int tid = threadIdx.x, x0 = blockIdx.x * kTW, y0 = blockIdx.y * kTH;
    const uint8_t *src = bgr + (size_t)blockIdx.z * h * w * 3;
    // ---- grey (8-bit fixed point) of the tile and its halo; reflection happens here, so every later pass is border-free
    for (int q = tid; q < kGR * kGPD; q += 256) {
        const int row = q / kGPD, dq = q - row * kGPD;
        const int iy = reflect_clamp<SMALL>(y0 - 8 + row, h);
        unsigned packed = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ix = reflect_clamp<SMALL>(x0 - 12 + dq * 4 + i, w);
            const uint8_t *p = src + (iy * w + ix) * 3;  // (h * w * 3 < 2^31: checked by the launcher)
            const unsigned g = ((unsigned)p[0] * 1868u + (unsigned)p[1] * 9617u + (unsigned)p[2] * 4899u + (1u << 13)) >> 14;
            packed |= g << (8 * i);
        }
        s_g[q] = packed;
    }
    __syncthreads();
    constexpr int NQ = kTW * kTH / 4 / 256;  // pixel quads per thread
    int s[NQ][4];
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const int q = tid + 256 * k, ty = q >> 4, j = q & 15;
        s[k][0] = s[k][1] = s[k][2] = s[k][3] = 0;
        scharr_quad_max<kGPD>(s_g + (ty + 7) * kGPD + j + 2, s[k]);  // the unblurred scale
    }
    uint8_t *bv8 = reinterpret_cast<uint8_t *>(s_bv);
#define OBB_DT_SCALE(IDX, R)                                                   \
    hpass<R>(s_g, s_hb, K.h[IDX]);                                             \
    __syncthreads();                                                           \
    vpass<R>(s_hb, bv8, K.ve[IDX], K.vo[IDX]);                                 \
    __syncthreads();                                                           \
    _Pragma("unroll") for (int k = 0; k < NQ; ++k) {                           \
        const int q = tid + 256 * k, ty = q >> 4, j = q & 15;                  \
        scharr_quad_max<kHC / 4>(s_bv + ty * (kHC / 4) + j, s[k]);             \
    }
    OBB_DT_SCALE(0, 2)
    OBB_DT_SCALE(1, 4)
    OBB_DT_SCALE(2, 7)
#undef OBB_DT_SCALE
    float *dst = acc + (size_t)blockIdx.z * crop_stride;
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const int q = tid + 256 * k, ty = q >> 4, j = q & 15, y = y0 + ty, x = x0 + 4 * j;
        if (y < h && x < pitch) {
            float4 o;
            o.x = sqrtf((float)s[k][0]); o.y = sqrtf((float)s[k][1]); o.z = sqrtf((float)s[k][2]); o.w = sqrtf((float)s[k][3]);
            *reinterpret_cast<float4 *>(dst + (size_t)y * pitch + x) = o;
        }
    }
}

// --------------------------------------------------------------------------------------------------------------- k_dt_select

constexpr int kSelThreads = 1024;

// histogram increment with a fast path for a wave whose participating lanes all hit one bin (flat images, sparse distances)
__device__ __forceinline__ void hist_add(unsigned *hist, unsigned bin, bool pred) {
    const unsigned long long m = __ballot(pred);
    if (m == 0) return;
    const int first = __ffsll((long long)m) - 1;
    const unsigned b0 = (unsigned)__builtin_amdgcn_readlane((int)bin, first);
    const unsigned long long same = __ballot(pred && bin == b0);
    if (same == m) {
        if ((int)(threadIdx.x & 63) == first) atomicAdd(&hist[b0], (unsigned)__popcll(m));
    } else if (pred) {
        atomicAdd(&hist[bin], 1u);
    }
}

// Walk every valid element of the crop once.  PASS 0: histogram of key >> 20 (+ min / max);  PASS 1: of (key >> 10) & 1023 among the
// keys whose top digit is the selected one, min key above that group;  PASS 2: of key & 1023 within the selected 21-bit prefix, min key
// of the same top digit above it.
template <int PASS, int NQ, bool MINMAX>
__device__ __forceinline__ void select_walk(const float *v, int h, int w, int pitch, unsigned (*hist)[2048], const unsigned (&pref)[NQ], unsigned (&above)[NQ],
                                            float &mn, float &mx) {
    const int qpr = (w + 3) >> 2, total = h * qpr;
    constexpr int U = 4;
    for (int it0 = threadIdx.x; it0 < total; it0 += kSelThreads * U) {
        float4 val[U];
        int xq[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int it = min(it0 + u * kSelThreads, total - 1);
            const int y = it / qpr;
            xq[u] = it - y * qpr;
            val[u] = *reinterpret_cast<const float4 *>(v + (size_t)y * pitch + 4 * xq[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool item_ok = it0 + u * kSelThreads < total;
            const float e[4] = {val[u].x, val[u].y, val[u].z, val[u].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool ok = item_ok && 4 * xq[u] + c < w;
                const unsigned key = __float_as_uint(e[c]);
                if (MINMAX && ok) { mn = fminf(mn, e[c]); mx = fmaxf(mx, e[c]); }
                if (PASS == 0) {
                    hist_add(hist[0], key >> 20, ok);
                } else {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const unsigned hi = PASS == 1 ? key >> 20 : key >> 10;
                        hist_add(hist[q], PASS == 1 ? (key >> 10) & 1023u : key & 1023u, ok && hi == pref[q]);
                        const bool ab = PASS == 1 ? hi > pref[q] : (hi > pref[q] && (key >> 20) == (pref[q] >> 10));
                        if (ok && ab) above[q] = min(above[q], key);
                    }
                }
            }
        }
    }
}

// one wave: the bin of `hist[0 .. nb)` holding rank k; returns (bin, elements before it, its count) to every lane
__device__ __forceinline__ void find_bin(const unsigned *hist, int nb, unsigned k, unsigned &bin, unsigned &before, unsigned &count) {
    const int lane = threadIdx.x & 63, per = nb / 64;
    unsigned sum = 0;
    for (int i = 0; i < per; ++i) sum += hist[lane * per + i];
    unsigned inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    const unsigned exc = inc - sum;
    const bool mine = k >= exc && k < inc;
    unsigned b = 0, bf = 0, ct = 0;
    if (mine) {
        unsigned cum = exc;
        int i = 0;
        for (; i < per - 1; ++i) {
            const unsigned c = hist[lane * per + i];
            if (cum + c > k) break;
            cum += c;
        }
        b = (unsigned)(lane * per + i); bf = cum; ct = hist[lane * per + i];
    }
    const unsigned long long m = __ballot(mine);
    const int src = m ? __ffsll((long long)m) - 1 : 0;
    bin = (unsigned)__builtin_amdgcn_readlane((int)b, src);
    before = (unsigned)__builtin_amdgcn_readlane((int)bf, src);
    count = (unsigned)__builtin_amdgcn_readlane((int)ct, src);
}

// numpy.percentile(a, [q])[0], method "linear", float32 input: float32 difference of the two order statistics, float64 interpolation
__device__ __forceinline__ double np_lerp(float a, float b, double t) {
    const float diff = b - a;
    double r = (double)a + (double)diff * t;
    if (t >= 0.5) r = (double)b - (double)diff * (1.0 - t);
    return r;
}

// mode 0: stats.thr = percentile(v, qa), stats.mn / mx = min / max (NQ = 1);  mode 1: stats.lo / hi = percentile(v, [qa, qb]) (NQ = 2)
template <int NQ>
__global__ __launch_bounds__(kSelThreads) void k_dt_select(const float *__restrict__ base, size_t crop_stride, int h, int w, int pitch, DtStats *__restrict__ stats,
                                                           double qa, double qb) {
    __shared__ unsigned s_hist[NQ][2048];
    __shared__ unsigned s_res[NQ][4];
    __shared__ unsigned s_above[NQ][2];
    __shared__ float s_mm[2][kSelThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = h * w;
    const float *v = base + (size_t)blockIdx.x * crop_stride;
    int rank[NQ];
    double frac[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const double vi = (double)(n - 1) * ((q == 0 ? qa : qb) / 100.0);
        rank[q] = (int)floor(vi);
        frac[q] = vi - (double)rank[q];
    }
    unsigned pref[NQ], kk[NQ], above[NQ];
    float mn = INFINITY, mx = -INFINITY;
#pragma unroll
    for (int q = 0; q < NQ; ++q) { pref[q] = 0; kk[q] = (unsigned)rank[q]; above[q] = 0xffffffffu; }
    if (tid < NQ * 2) s_above[tid >> 1][tid & 1] = 0xffffffffu;

    // ---- pass 0: top 11 bits (shared by both ranks)
    for (int i = tid; i < 2048; i += kSelThreads) s_hist[0][i] = 0;
    __syncthreads();
    select_walk<0, NQ, NQ == 1>(v, h, w, pitch, s_hist, pref, above, mn, mx);
    __syncthreads();
    if (wave < NQ) {
        unsigned b, bf, ct;
        find_bin(s_hist[0], 2048, kk[wave == 0 ? 0 : NQ - 1], b, bf, ct);
        if (lane == 0) { s_res[wave][0] = b; s_res[wave][1] = bf; s_res[wave][2] = ct; }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) { pref[q] = s_res[q][0]; kk[q] -= s_res[q][1]; }
    __syncthreads();
    // ---- pass 1: middle 10 bits
    for (int i = tid; i < NQ * 2048; i += kSelThreads) (&s_hist[0][0])[i] = 0;
    __syncthreads();
    select_walk<1, NQ, false>(v, h, w, pitch, s_hist, pref, above, mn, mx);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        unsigned a = above[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a = min(a, (unsigned)__shfl_xor((int)a, o));
        if (lane == 0 && a != 0xffffffffu) atomicMin(&s_above[q][1], a);
        above[q] = 0xffffffffu;
    }
    __syncthreads();
    if (wave < NQ) {
        unsigned b, bf, ct;
        find_bin(s_hist[wave], 1024, kk[wave == 0 ? 0 : NQ - 1], b, bf, ct);
        if (lane == 0) { s_res[wave][0] = b; s_res[wave][1] = bf; s_res[wave][2] = ct; }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) { pref[q] = (pref[q] << 10) | s_res[q][0]; kk[q] -= s_res[q][1]; }
    __syncthreads();
    // ---- pass 2: low 10 bits
    for (int i = tid; i < NQ * 2048; i += kSelThreads) (&s_hist[0][0])[i] = 0;
    __syncthreads();
    select_walk<2, NQ, false>(v, h, w, pitch, s_hist, pref, above, mn, mx);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        unsigned a = above[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a = min(a, (unsigned)__shfl_xor((int)a, o));
        if (lane == 0 && a != 0xffffffffu) atomicMin(&s_above[q][0], a);
    }
    if (NQ == 1) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
        if (lane == 0) { s_mm[0][wave] = mn; s_mm[1][wave] = mx; }
    }
    __syncthreads();
    if (wave < NQ) {
        const int q = wave;
        unsigned b, bf, ct;
        find_bin(s_hist[q], 1024, kk[wave == 0 ? 0 : NQ - 1], b, bf, ct);
        // successor inside the same 22-bit prefix: first non-empty bin above b
        unsigned cand = 0xffffffffu;
        for (int i = 0; i < 16; ++i) {
            const unsigned bi = (unsigned)(lane * 16 + i);
            if (bi > b && s_hist[q][bi] != 0) { cand = bi; break; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cand = min(cand, (unsigned)__shfl_xor((int)cand, o));
        if (lane == 0) {
            const unsigned p = wave == 0 ? pref[0] : pref[NQ - 1];
            const unsigned k2 = (wave == 0 ? kk[0] : kk[NQ - 1]) - bf;
            const unsigned key = (p << 10) | b;
            unsigned nxt = key;
            if (k2 + 1 >= ct) {  // the next order statistic is the smallest key above this one
                if (cand != 0xffffffffu) nxt = (p << 10) | cand;
                else if (s_above[q][0] != 0xffffffffu) nxt = s_above[q][0];
                else if (s_above[q][1] != 0xffffffffu) nxt = s_above[q][1];
            }
            const int r = wave == 0 ? rank[0] : rank[NQ - 1];
            if (r + 1 > n - 1) nxt = key;
            const double val = np_lerp(__uint_as_float(key), __uint_as_float(nxt), wave == 0 ? frac[0] : frac[NQ - 1]);
            DtStats *st = stats + blockIdx.x;
            if (NQ == 1) {
                st->thr = val;
                float a = s_mm[0][0], c = s_mm[1][0];
                for (int k = 1; k < kSelThreads / 64; ++k) { a = fminf(a, s_mm[0][k]); c = fmaxf(c, s_mm[1][k]); }
                st->mn = a; st->mx = c;
            } else if (q == 0) {
                st->lo = val;
            } else {
                st->hi = val;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- k_dt_edges

// edges = acc >= thr (the reference's numpy 1.26.4: the float64 percentile is a SCALAR, cast to the array's float32 before the compare --
// value-based casting, changed by NEP 50 in numpy 2), then opening with the 3x3 cross; pixels outside the image
// do not take part (erosion sees 255 there, dilation 0).  Output: 255 / 0 per pixel, pitch P.
__global__ __launch_bounds__(256) void k_dt_edges(const float *__restrict__ acc, size_t crop_stride, int h, int w, int pitch, const DtStats *__restrict__ stats,
                                                  uint8_t *__restrict__ edges, size_t edge_stride) {
    constexpr int EW = kTW + 4, EH = kTH + 4, RW = kTW + 2, RH = kTH + 2;
    __shared__ uint8_t s_e[EH * EW];
    __shared__ uint8_t s_r[RH * RW];
    const int tid = threadIdx.x, x0 = blockIdx.x * kTW, y0 = blockIdx.y * kTH;
    const float *a = acc + (size_t)blockIdx.z * crop_stride;
    const float thr = (float)stats[blockIdx.z].thr;
    for (int i = tid; i < EH * EW; i += 256) {
        const int r = i / EW, c = i - r * EW, y = y0 - 2 + r, x = x0 - 2 + c;
        uint8_t e = 255;
        if (y >= 0 && y < h && x >= 0 && x < w) e = (a[(size_t)y * pitch + x] >= thr) ? 255 : 0;
        s_e[i] = e;
    }
    __syncthreads();
    for (int i = tid; i < RH * RW; i += 256) {
        const int r = i / RW, c = i - r * RW, y = y0 - 1 + r, x = x0 - 1 + c;
        const uint8_t *p = s_e + (r + 1) * EW + (c + 1);
        uint8_t m = min(min(min(p[0], p[-EW]), min(p[EW], p[-1])), p[1]);
        if (y < 0 || y >= h || x < 0 || x >= w) m = 0;
        s_r[i] = m;
    }
    __syncthreads();
    uint8_t *out = edges + (size_t)blockIdx.z * edge_stride;
    for (int i = tid; i < kTH * kTW; i += 256) {
        const int r = i >> 6, c = i & 63, y = y0 + r, x = x0 + c;
        const uint8_t *p = s_r + (r + 1) * RW + (c + 1);
        const uint8_t m = max(max(max(p[0], p[-RW]), max(p[RW], p[-1])), p[1]);
        if (y < h && x < pitch) out[(size_t)y * pitch + x] = m;
    }
}

// -------------------------------------------------------------------------------------------------------------- k_dt_chamfer

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or(int fill, int v) {
    return __builtin_amdgcn_update_dpp(fill, v, CTRL, ROW_MASK, 0xf, false);
}

// minimum over the lanes below this one (INT_MAX for lane 0)
__device__ __forceinline__ int wave_excl_prefix_min(int v) {
    constexpr int I = 0x7fffffff;
    v = min(v, dpp_or<0x111, 0xf>(I, v));  // row_shr:1
    v = min(v, dpp_or<0x112, 0xf>(I, v));  // row_shr:2
    v = min(v, dpp_or<0x114, 0xf>(I, v));  // row_shr:4
    v = min(v, dpp_or<0x118, 0xf>(I, v));  // row_shr:8
    v = min(v, dpp_or<0x142, 0xa>(I, v));  // row_bcast:15 into rows 1 and 3
    v = min(v, dpp_or<0x143, 0xc>(I, v));  // row_bcast:31 into rows 2 and 3
    return dpp_or<0x138, 0xf>(I, v);       // wave_shr:1
}

template <int PX> struct PxVec;
template <> struct PxVec<2>  { typedef unsigned short E; typedef int2 T; static constexpr int NT = 1; };
template <> struct PxVec<4>  { typedef unsigned E; typedef int4 T; static constexpr int NT = 1; };
template <> struct PxVec<8>  { typedef uint2 E; typedef int4 T; static constexpr int NT = 2; };
template <> struct PxVec<16> { typedef uint4 E; typedef int4 T; static constexpr int NT = 4; };

template <int PX>
__device__ __forceinline__ bool edge_at(const typename PxVec<PX>::E &e, int j) {
    if constexpr (PX == 2) return ((e >> (8 * j)) & 0xffu) != 0;
    else if constexpr (PX == 4) return ((e >> (8 * j)) & 0xffu) != 0;
    else if constexpr (PX == 8) return (((j < 4 ? e.x : e.y) >> (8 * (j & 3))) & 0xffu) != 0;
    else return (((j < 4 ? e.x : j < 8 ? e.y : j < 12 ? e.z : e.w) >> (8 * (j & 3))) & 0xffu) != 0;
}

template <int PX>
__device__ __forceinline__ void load_row(const int *p, int (&d)[PX]) {
    typedef typename PxVec<PX>::T T;
    constexpr int per = sizeof(T) / 4;
#pragma unroll
    for (int k = 0; k < PxVec<PX>::NT; ++k) {
        const T t = reinterpret_cast<const T *>(p)[k];
        const int *ti = reinterpret_cast<const int *>(&t);
#pragma unroll
        for (int i = 0; i < per; ++i) d[k * per + i] = ti[i];
    }
}

template <int PX, typename S>
__device__ __forceinline__ void store_row(S *p, const S (&d)[PX]) {
    typedef typename PxVec<PX>::T T;
    constexpr int per = sizeof(T) / 4;
#pragma unroll
    for (int k = 0; k < PxVec<PX>::NT; ++k) {
        T t;
        S *ti = reinterpret_cast<S *>(&t);
#pragma unroll
        for (int i = 0; i < per; ++i) ti[i] = d[k * per + i];
        reinterpret_cast<T *>(p)[k] = t;
    }
}

// 3x3 chamfer distance to the nearest edge pixel, 16.16 fixed point: forward then backward sweep, one row per step, one wave per crop.
// Lane l owns pixels [PX l, PX l + PX) in the forward sweep and the mirrored chunk in the backward sweep, so that both row scans run
// towards higher lanes.  Pixels x >= w behave as the image border (kINIT).  `tt` has h + 1 rows: row h takes the stores of the
// unrolled loop's overshoot, which keeps every load and store unconditional (one in-order vmcnt: see stem.hip).
template <int PX>
__global__ __launch_bounds__(64) void k_dt_chamfer(const uint8_t *__restrict__ edges, size_t edge_stride, int *__restrict__ tt_all, size_t tt_stride, int h, int w,
                                                   int pitch) {
    typedef typename PxVec<PX>::E E;
    constexpr int D = 4;
    const int lane = threadIdx.x;
    const uint8_t *ed = edges + (size_t)blockIdx.x * edge_stride;
    int *tt = tt_all + (size_t)blockIdx.x * tt_stride;
    constexpr int kBorder = kINIT + kHV;
    {  // ---- forward: rows top to bottom, scan left to right
        const int x0 = lane * PX;
        int prev[PX];
#pragma unroll
        for (int j = 0; j < PX; ++j) prev[j] = kINIT;
        E ring[D];
#pragma unroll
        for (int u = 0; u < D; ++u) ring[u] = *reinterpret_cast<const E *>(ed + (size_t)min(u, h - 1) * pitch + x0);
        for (int yb = 0; yb < h; yb += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int y = yb + u;
                const bool live = y < h;
                const E e = ring[u];
                ring[u] = *reinterpret_cast<const E *>(ed + (size_t)min(y + D, h - 1) * pitch + x0);
                const int left = dpp_or<0x138, 0xf>(kINIT, prev[PX - 1]);   // wave_shr:1
                const int right = dpp_or<0x130, 0xf>(kINIT, prev[0]);       // wave_shl:1
                int m[PX];
#pragma unroll
                for (int j = 0; j < PX; ++j) {
                    const int ul = j ? prev[j - 1] : left, ur = j < PX - 1 ? prev[j + 1] : right;
                    int c = min(min(ul + kDIAG, prev[j] + kHV), ur + kDIAG);
                    if (edge_at<PX>(e, j)) c = 0;
                    if (x0 + j >= w) c = kINIT;
                    c -= (x0 + j) * kHV;
                    m[j] = j ? min(m[j - 1], c) : c;
                }
                const int carry = min(wave_excl_prefix_min(m[PX - 1]), kBorder);
                int d[PX];
#pragma unroll
                for (int j = 0; j < PX; ++j) {
                    d[j] = min(m[j], carry) + (x0 + j) * kHV;
                    if (x0 + j >= w) d[j] = kINIT;
                }
                store_row<PX, int>(tt + (size_t)(live ? y : h) * pitch + x0, d);
#pragma unroll
                for (int j = 0; j < PX; ++j) prev[j] = live ? d[j] : prev[j];
            }
        }
    }
    __threadfence_block();  // the mirrored lane mapping below reads what other lanes OF THIS WAVE stored: their completion is all it needs (the
                            // CU's L1 is coherent for its own waves; a device-scope fence writes back the whole L2 -- measured at 29 us in k_heavy_rows)
    {  // ---- backward: rows bottom to top, scan right to left (lane 0 owns the rightmost chunk)
        const int x0 = (63 - lane) * PX;
        int prev[PX];
#pragma unroll
        for (int j = 0; j < PX; ++j) prev[j] = kINIT;
        int ring[D][PX];
#pragma unroll
        for (int u = 0; u < D; ++u) load_row<PX>(tt + (size_t)max(h - 1 - u, 0) * pitch + x0, ring[u]);
        float *dist = reinterpret_cast<float *>(tt);
        for (int yb = 0; yb < h; yb += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int yy = yb + u, y = h - 1 - yy;
                const bool live = yy < h;
                int cur[PX];
#pragma unroll
                for (int j = 0; j < PX; ++j) cur[j] = ring[u][j];
                load_row<PX>(tt + (size_t)max(y - D, 0) * pitch + x0, ring[u]);
                const int hi_n = dpp_or<0x138, 0xf>(kINIT, prev[0]);        // pixel x0 + PX of the row below: lane - 1
                const int lo_n = dpp_or<0x130, 0xf>(kINIT, prev[PX - 1]);   // pixel x0 - 1: lane + 1
                int m[PX];
#pragma unroll
                for (int j = PX - 1; j >= 0; --j) {
                    const int dl = j ? prev[j - 1] : lo_n, dr = j < PX - 1 ? prev[j + 1] : hi_n;
                    int c = min(min(cur[j], dr + kDIAG), min(prev[j] + kHV, dl + kDIAG));
                    if (x0 + j >= w) c = kINIT;
                    c -= (pitch - 1 - (x0 + j)) * kHV;
                    m[j] = j < PX - 1 ? min(m[j + 1], c) : c;
                }
                const int carry = min(wave_excl_prefix_min(m[0]), kBorder);
                int d[PX];
                float f[PX];
#pragma unroll
                for (int j = 0; j < PX; ++j) {
                    d[j] = min(m[j], carry) + (pitch - 1 - (x0 + j)) * kHV;
                    if (x0 + j >= w) d[j] = kINIT;
                    f[j] = (float)d[j] * (1.0f / 65536.0f);
                }
                store_row<PX, float>(dist + (size_t)(live ? y : h) * pitch + x0, f);
#pragma unroll
                for (int j = 0; j < PX; ++j) prev[j] = live ? d[j] : prev[j];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- k_dt_blend

// 0.7 exp(-clip((d - lo) / max(1e-6, hi - lo)) / 3) + 0.3 minmax(acc), clipped, * 255, truncated; output RGB + that channel.
// Arithmetic as the reference's pinned numpy 1.26.4 runs it (Detect_OBB.py:126-133): lo / hi are float64 SCALARS, and a float32 array
// combined with a float64 scalar stays float32 there (value-based casting; numpy >= 2 would promote), so every step below is one
// correctly rounded float32 operation (no contraction: the build uses -ffp-contract=off).  exp: float32(exp64(x)) with exp64 the degree-14
// Taylor polynomial in Horner form, double multiply + add per step -- the same IEEE operations as the CPU oracle (dtedge.py: exp32), hence the
// same bytes; numpy's own float32 SIMD exp may differ from it by an ulp (unpinned).
__device__ __forceinline__ float dt_exp32(float xf) {  // xf in [-0.5, 0]
    constexpr double C[15] = {1.0, 1.0, 1.0 / 2.0, 1.0 / 6.0, 1.0 / 24.0, 1.0 / 120.0, 1.0 / 720.0, 1.0 / 5040.0, 1.0 / 40320.0, 1.0 / 362880.0, 1.0 / 3628800.0,
                              1.0 / 39916800.0, 1.0 / 479001600.0, 1.0 / 6227020800.0, 1.0 / 87178291200.0};
    const double x = (double)xf;
    double p = C[14];
#pragma unroll
    for (int k = 13; k >= 0; --k) {
        p = p * x;
        p = p + C[k];
    }
    return (float)p;
}

__global__ __launch_bounds__(256) void k_dt_blend(const uint8_t *__restrict__ bgr, const float *__restrict__ acc, size_t acc_stride, const float *__restrict__ dist,
                                                  size_t dist_stride, const DtStats *__restrict__ stats, int h, int w, int pitch, uint8_t *__restrict__ out4) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const DtStats st = stats[blockIdx.z];
    const size_t n = (size_t)h * w, i = (size_t)y * w + x, ip = (size_t)y * pitch + x;
    const uint8_t *src = bgr + blockIdx.z * n * 3;
    const double scale_d = (st.mx > st.mn) ? 1.0 / ((double)st.mx - (double)st.mn) : 0.0;
    const float scale = (float)scale_d, shift = (float)(-(double)st.mn * scale_d);
    const float lo = (float)st.lo, den = (float)fmax(1e-6, st.hi - st.lo);  // (hi - lo and the max in float64: both scalars)
    float d = (dist[blockIdx.z * dist_stride + ip] - lo) / den;
    d = fminf(fmaxf(d, 0.0f), 1.0f);
    float soft = dt_exp32(-d / 3.0f);
    const float nrm = acc[blockIdx.z * acc_stride + ip] * scale + shift;
    const float a7 = 0.7f * soft, a3 = 0.3f * nrm;
    soft = a7 + a3;
    soft = fminf(fmaxf(soft, 0.0f), 1.0f);
    const unsigned o = (unsigned)src[i * 3 + 2] | ((unsigned)src[i * 3 + 1] << 8) | ((unsigned)src[i * 3] << 16) | ((unsigned)(uint8_t)(soft * 255.0f) << 24);
    reinterpret_cast<unsigned *>(out4)[blockIdx.z * n + i] = o;
}

static void gauss_taps(double sigma, int idx, DtTaps *K) {
    int q[16] = {0};
    const int n = (int)lrint(sigma * 6.0 + 1.0) | 1;
    double g[15], sum = 0;
    for (int i = 0; i < n; ++i) { double d = i - (n - 1) / 2.0; g[i] = exp(-(d * d) / (2.0 * sigma * sigma)); sum += g[i]; }
    int tot = 0;
    for (int i = 0; i < n; ++i) { q[i] = (int)nearbyint(g[i] / sum * 256.0); tot += q[i]; }
    q[n / 2] += 256 - tot;
    for (int d = 0; d < 4; ++d) K->h[idx][d] = (unsigned)q[4 * d] | ((unsigned)q[4 * d + 1] << 8) | ((unsigned)q[4 * d + 2] << 16) | ((unsigned)q[4 * d + 3] << 24);
    for (int d = 0; d < 8; ++d) {
        const int e0 = 2 * d, e1 = 2 * d + 1, o0 = 2 * d - 1, o1 = 2 * d;
        K->ve[idx][d] = (unsigned)(e0 < n ? q[e0] : 0) | ((unsigned)(e1 < n ? q[e1] : 0) << 16);
        K->vo[idx][d] = (unsigned)(o0 >= 0 && o0 < n ? q[o0] : 0) | ((unsigned)(o1 < n ? q[o1] : 0) << 16);
    }
}

template <int PX>
static void launch_chamfer(int B, hipStream_t s, const uint8_t *edges, size_t es, int *tt, size_t ts, int h, int w, int pitch) {
    hipLaunchKernelGGL(k_dt_chamfer<PX>, dim3((unsigned)B), dim3(64), 0, s, edges, es, tt, ts, h, w, pitch);
}

}  // namespace obb

using namespace obb;

extern "C" int obb_build_multich(obb_ctx *ctx, const uint8_t *bgr, int32_t B, int32_t h, int32_t w, uint8_t *out4, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && B >= 0 && h > 0 && w > 0, "obb_build_multich: bad arguments");
    if (B == 0) return OBB_OK;
    OBB_REQUIRE(ctx, bgr && out4, "obb_build_multich: NULL buffer");
    // one lane per <= 16 pixels of a row; grid.z carries the crop index
    OBB_REQUIRE(ctx, w >= 2 && h >= 2 && w <= 1024 && h <= 4096 && B <= 65535, "obb_build_multich: crop %dx%d unsupported (2 <= width <= 1024, 2 <= height <= 4096)", h, w);
    const int px = w <= 128 ? 2 : w <= 256 ? 4 : w <= 512 ? 8 : 16, pitch = 64 * px;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t acc_bytes = up((size_t)h * pitch * 4), tt_bytes = up((size_t)(h + 1) * pitch * 4), edge_bytes = up((size_t)h * pitch);
    const size_t stats_bytes = up(sizeof(DtStats) * (size_t)B);
    char *scratch = (char *)ctx->workspace(WS_DT, (acc_bytes + tt_bytes + edge_bytes) * (size_t)B + stats_bytes);
    if (!scratch) return set_error(ctx, OBB_ERR_HIP, "obb_build_multich: workspace allocation failed");
    float *acc = reinterpret_cast<float *>(scratch);
    int *tt = reinterpret_cast<int *>(scratch + acc_bytes * (size_t)B);
    uint8_t *edges = reinterpret_cast<uint8_t *>(scratch + (acc_bytes + tt_bytes) * (size_t)B);
    DtStats *stats = reinterpret_cast<DtStats *>(scratch + (acc_bytes + tt_bytes + edge_bytes) * (size_t)B);
    const size_t acc_stride = acc_bytes / 4, tt_stride = tt_bytes / 4;
    DtTaps K;
    const double sig[3] = {0.6, 1.2, 2.4};  // MS_SIGMAS without the unblurred scale 0 (Detect_OBB.py:29)
    for (int i = 0; i < 3; ++i) gauss_taps(sig[i], i, &K);
    hipStream_t st = (hipStream_t)s;
    const dim3 tiles((unsigned)((w + kTW - 1) / kTW), (unsigned)((h + kTH - 1) / kTH), (unsigned)B);
    if (h < 16 || w < 16) hipLaunchKernelGGL(k_dt_acc<true>, tiles, dim3(256), 0, st, bgr, h, w, acc, acc_stride, pitch, K);
    else hipLaunchKernelGGL(k_dt_acc<false>, tiles, dim3(256), 0, st, bgr, h, w, acc, acc_stride, pitch, K);
    hipLaunchKernelGGL(k_dt_select<1>, dim3((unsigned)B), dim3(kSelThreads), 0, st, acc, acc_stride, h, w, pitch, stats, 90.0, 0.0);  // DT_P_HI (Detect_OBB.py:31)
    hipLaunchKernelGGL(k_dt_edges, tiles, dim3(256), 0, st, acc, acc_stride, h, w, pitch, stats, edges, edge_bytes);
    switch (px) {
        case 2: launch_chamfer<2>(B, st, edges, edge_bytes, tt, tt_stride, h, w, pitch); break;
        case 4: launch_chamfer<4>(B, st, edges, edge_bytes, tt, tt_stride, h, w, pitch); break;
        case 8: launch_chamfer<8>(B, st, edges, edge_bytes, tt, tt_stride, h, w, pitch); break;
        default: launch_chamfer<16>(B, st, edges, edge_bytes, tt, tt_stride, h, w, pitch); break;
    }
    hipLaunchKernelGGL(k_dt_select<2>, dim3((unsigned)B), dim3(kSelThreads), 0, st, reinterpret_cast<const float *>(tt), tt_stride, h, w, pitch, stats, 1.0, 99.0);
    hipLaunchKernelGGL(k_dt_blend, dim3((unsigned)((w + 63) / 64), (unsigned)((h + 3) / 4), (unsigned)B), dim3(256), 0, st, bgr, acc, acc_stride,
                       reinterpret_cast<const float *>(tt), tt_stride, stats, h, w, pitch, out4);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}
