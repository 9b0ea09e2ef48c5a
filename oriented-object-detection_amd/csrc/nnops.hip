// The non-GEMM layers of the YOLO11-OBB forward: depthwise 3x3 (DWConv in the OBB head's
// cls branch and the attention positional encoding), MaxPool2d(5,1,2) of SPPF, nearest x2 Upsample, and the
// 2-head C2PSA attention.  All are HBM/LDS-bound VALU kernels on NHWC bf16 with 16 B (8-channel) accesses; each
// writes straight into the channel slice of its consumer's concat buffer.
#include "nnops.h"

#include <cstdlib>

namespace obb {

// ---------------------------------------------------------------- depthwise 3x3, stride 1, pad 1 (+bias, SiLU, +residual)
// w16: 16-bit [9][C] (the weights in the storage type).  A thread owns a 1x4 strip of pixels x 8 channels: the 3x6
// input window and the 9 weight vectors are loaded once per strip (4.5 loads per output instead of 27).
template <bool F16>
__global__ __launch_bounds__(256) void k_dwconv3(TensorRef in, TensorRef out, TensorRef res, const bf16_t *__restrict__ w16,
                                                const float *__restrict__ bias, int B, int H, int W, int C, int act) {
    const int c8n = C >> 3;
    const int W4 = (W + 3) >> 2;
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t total = (int64_t)B * H * W4 * c8n;
    if (idx >= total) return;
    int c8 = (int)(idx % c8n);
    int64_t pix = idx / c8n;
    int x0 = (int)(pix % W4) * 4;
    int y = (int)((pix / W4) % H);
    int b = (int)(pix / ((int64_t)W4 * H));
    const bf16_t *ip = (const bf16_t *)in.p + (int64_t)b * in.bs + in.co + c8 * 8;
    // every load of the strip is issued before the first use: 18 activation vectors (clamped address, zeroed when outside the
    // image: exact zeros leave the sum of the taps that exist unchanged), 9 weight vectors and the bias -> one memory latency
    uint4 raw[3][6];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        int yy = y + ky - 1;
        bool yok = yy >= 0 && yy < H;
        int yc = min(max(yy, 0), H - 1);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            int xx = x0 + k - 1;
            bool ok = yok && xx >= 0 && xx < W;
            int xc = min(max(xx, 0), W - 1);
            uint4 v = *reinterpret_cast<const uint4 *>(ip + ((int64_t)yc * W + xc) * in.cs);
            raw[ky][k] = ok ? v : make_uint4(0, 0, 0, 0);
        }
    }
    // weights as 16-bit values (what they were rounded to anyway): the products run as mixed-precision FMAs (v_fma_mix_f32 takes
    // the fp16 operands directly, fp32 accumulate), so no conversion instructions are spent on either operand
    uint4 wraw[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wraw[t] = *reinterpret_cast<const uint4 *>(w16 + t * C + c8 * 8);
    float acc[4][8];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[p][j] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int p = 0; p < 4; ++p) fma8_mixed<F16>(acc[p], raw[ky][p + kx], wraw[ky * 3 + kx]);
    const float4 *bp = reinterpret_cast<const float4 *>(bias + c8 * 8);
    float4 b0 = bp[0], b1 = bp[1];
    float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int x = x0 + p;
        if (x >= W) break;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = acc[p][j] + bb[j];
            if (act) v = silu_f(v);
            acc[p][j] = v;
        }
        int64_t opix = (int64_t)y * W + x;
        if (res.p) {
            uint4 rv = *reinterpret_cast<const uint4 *>((const bf16_t *)res.p + (int64_t)b * res.bs + opix * res.cs + res.co + c8 * 8);
            float rf[8];
            unpack8<F16>(rv, rf);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[p][j] += rf[j];
        }
        *reinterpret_cast<uint4 *>((bf16_t *)out.p + (int64_t)b * out.bs + opix * out.cs + out.co + c8 * 8) = pack8<F16>(acc[p]);
    }
}

// ---------------------------------------------------------------- MaxPool2d(k=5, s=1, p=2)  (implicit -inf padding)
template <bool F16>
__global__ __launch_bounds__(256) void k_maxpool5(TensorRef in, TensorRef out, int B, int H, int W, int C) {
    const int c8n = C >> 3;
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t total = (int64_t)B * H * W * c8n;
    if (idx >= total) return;
    int c8 = (int)(idx % c8n);
    int64_t pix = idx / c8n;
    int x = (int)(pix % W);
    int y = (int)((pix / W) % H);
    int b = (int)(pix / ((int64_t)W * H));
    const bf16_t *ip = (const bf16_t *)in.p + (int64_t)b * in.bs + in.co + c8 * 8;
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
    for (int yy = max(0, y - 2); yy <= min(H - 1, y + 2); ++yy)
        for (int xx = max(0, x - 2); xx <= min(W - 1, x + 2); ++xx) {
            uint4 v = *reinterpret_cast<const uint4 *>(ip + ((int64_t)yy * W + xx) * in.cs);
            float f[8];
            unpack8<F16>(v, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], f[j]);
        }
    *reinterpret_cast<uint4 *>((bf16_t *)out.p + (int64_t)b * out.bs + ((int64_t)y * W + x) * out.cs + out.co + c8 * 8) = pack8<F16>(m);
}

// ---------------------------------------------------------------- SPPF: the three chained MaxPool2d(5,1,2) in one launch
// out slices 1..3 of the concat buffer = pool(x), pool(pool(x)), pool(pool(pool(x))).  One workgroup owns (image, 32 channels): the
// H x W plane lives in LDS and is pooled three times (separable 5x1 / 1x5 max), each result stored as it appears.  `cat` is the concat
// buffer [x | y1 | y2 | y3] with C channels per member.
// max of two 8-channel chunks in the storage type (max is exact in any format: no conversion to fp32 needed for fp16)
template <bool F16>
__device__ __forceinline__ uint4 max8(uint4 a, uint4 b) {
    if constexpr (F16) {
        typedef __attribute__((ext_vector_type(8))) _Float16 h8;
        h8 x, y;
        __builtin_memcpy(&x, &a, 16);
        __builtin_memcpy(&y, &b, 16);
        h8 m = __builtin_elementwise_max(x, y);  // v_pk_max_f16
        uint4 o;
        __builtin_memcpy(&o, &m, 16);
        return o;
    } else {
        float fa[8], fb[8];
        unpack8<false>(a, fa);
        unpack8<false>(b, fb);
#pragma unroll
        for (int j = 0; j < 8; ++j) fa[j] = fmaxf(fa[j], fb[j]);
        return pack8<false>(fa);
    }
}

template <bool F16>
__global__ __launch_bounds__(256) void k_sppf_pools(TensorRef cat, int H, int W, int C) {
    extern __shared__ __attribute__((aligned(16))) uint4 sp[];  // two planes [H*W][4 chunks]: current tensor, row-max scratch
    const int groups = C >> 5;
    const int b = blockIdx.x / groups, cg = blockIdx.x % groups;
    const int n = H * W * 4;
    bf16_t *base = (bf16_t *)cat.p + (int64_t)b * cat.bs + cat.co + cg * 32;
    uint4 *cur = sp, *tmp = sp + n;
    for (int i = threadIdx.x; i < n; i += 256) cur[i] = *reinterpret_cast<const uint4 *>(base + (int64_t)(i >> 2) * cat.cs + (i & 3) * 8);
    __syncthreads();
    for (int pass = 1; pass <= 3; ++pass) {
        // 5x5 max = 1x5 max of the 5x1 max (window clipped at the border = implicit -inf padding)
        for (int i = threadIdx.x; i < n; i += 256) {
            const int pix = i >> 2, c = i & 3;
            const int y = pix / W, x = pix - y * W;
            uint4 m = cur[i];
            for (int xx = max(0, x - 2); xx <= min(W - 1, x + 2); ++xx) m = max8<F16>(m, cur[(y * W + xx) * 4 + c]);
            tmp[i] = m;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) {
            const int pix = i >> 2, c = i & 3;
            const int y = pix / W, x = pix - y * W;
            uint4 m = tmp[i];
            for (int yy = max(0, y - 2); yy <= min(H - 1, y + 2); ++yy) m = max8<F16>(m, tmp[(yy * W + x) * 4 + c]);
            cur[i] = m;  // (element i of `cur` is read only by this thread in this loop: in place)
            *reinterpret_cast<uint4 *>(base + (int64_t)pass * C + (int64_t)pix * cat.cs + c * 8) = m;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- nearest-neighbour x2 upsample (pure copy)
__global__ __launch_bounds__(256) void k_upsample2(TensorRef in, TensorRef out, int B, int H, int W, int C) {
    const int c8n = C >> 3;
    const int Ho = H * 2, Wo = W * 2;
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t total = (int64_t)B * Ho * Wo * c8n;
    if (idx >= total) return;
    int c8 = (int)(idx % c8n);
    int64_t pix = idx / c8n;
    int x = (int)(pix % Wo);
    int y = (int)((pix / Wo) % Ho);
    int b = (int)(pix / ((int64_t)Wo * Ho));
    uint4 v = *reinterpret_cast<const uint4 *>((const bf16_t *)in.p + (int64_t)b * in.bs + ((int64_t)(y >> 1) * W + (x >> 1)) * in.cs + in.co + c8 * 8);
    *reinterpret_cast<uint4 *>((bf16_t *)out.p + (int64_t)b * out.bs + ((int64_t)y * Wo + x) * out.cs + out.co + c8 * 8) = v;
}

// ---------------------------------------------------------------- C2PSA attention core
// qkv slice layout per token (channels permuted by the weight loader): [q: nh*KD][k: nh*KD][v: nh*HD].
// One workgroup per (tile, head): K and V of all N tokens live in LDS as fp32; a thread owns query rows.
// out[n, h*HD + d] = sum_m softmax_m(q_n . k_m * scale) * v_m[d]   (two-pass softmax in fp32, like torch.softmax)
template <int KD, int HD, bool F16>
__global__ __launch_bounds__(256) void k_attention(TensorRef qkv, TensorRef out, int N, int nh, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *sk = sm;                   // [N][KD]  (every lane reads the same row -> LDS broadcast, no padding needed)
    float *sv = sm + (size_t)N * KD;  // [N][HD]
    const int b = blockIdx.x / nh, h = blockIdx.x % nh;
    const bf16_t *base = (const bf16_t *)qkv.p + (int64_t)b * qkv.bs + qkv.co;
    for (int i = threadIdx.x; i < N * (KD / 8); i += 256) {
        int n = i / (KD / 8), c = i % (KD / 8);
        uint4 v = *reinterpret_cast<const uint4 *>(base + (int64_t)n * qkv.cs + nh * KD + h * KD + c * 8);
        float f[8];
        unpack8<F16>(v, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) sk[n * KD + c * 8 + j] = f[j];
    }
    for (int i = threadIdx.x; i < N * (HD / 8); i += 256) {
        int n = i / (HD / 8), c = i % (HD / 8);
        uint4 v = *reinterpret_cast<const uint4 *>(base + (int64_t)n * qkv.cs + 2 * nh * KD + h * HD + c * 8);
        float f[8];
        unpack8<F16>(v, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) sv[n * HD + c * 8 + j] = f[j];
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += 256) {
        float q[KD];
#pragma unroll
        for (int c = 0; c < KD / 8; ++c) {
            uint4 v = *reinterpret_cast<const uint4 *>(base + (int64_t)n * qkv.cs + h * KD + c * 8);
            unpack8<F16>(v, &q[c * 8]);
        }
        float mx = -INFINITY;
        for (int m = 0; m < N; ++m) {
            float s = 0.f;
            const float4 *kr = reinterpret_cast<const float4 *>(sk + m * KD);
#pragma unroll
            for (int d = 0; d < KD / 4; ++d) { float4 k4 = kr[d]; s += q[4 * d] * k4.x; s += q[4 * d + 1] * k4.y; s += q[4 * d + 2] * k4.z; s += q[4 * d + 3] * k4.w; }
            mx = fmaxf(mx, s * scale);
        }
        float acc[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] = 0.f;
        float den = 0.f;
        for (int m = 0; m < N; ++m) {
            float s = 0.f;
            const float4 *kr = reinterpret_cast<const float4 *>(sk + m * KD);
#pragma unroll
            for (int d = 0; d < KD / 4; ++d) { float4 k4 = kr[d]; s += q[4 * d] * k4.x; s += q[4 * d + 1] * k4.y; s += q[4 * d + 2] * k4.z; s += q[4 * d + 3] * k4.w; }
            float p = __expf(s * scale - mx);
            den += p;
            const float4 *vr = reinterpret_cast<const float4 *>(sv + m * HD);
#pragma unroll
            for (int d = 0; d < HD / 4; ++d) { float4 v4 = vr[d]; acc[4 * d] += p * v4.x; acc[4 * d + 1] += p * v4.y; acc[4 * d + 2] += p * v4.z; acc[4 * d + 3] += p * v4.w; }
        }
        float inv = 1.0f / den;
        bf16_t *op = (bf16_t *)out.p + (int64_t)b * out.bs + (int64_t)n * out.cs + out.co + h * HD;
#pragma unroll
        for (int c = 0; c < HD / 8; ++c) {
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = acc[c * 8 + j] * inv;
            *reinterpret_cast<uint4 *>(op + c * 8) = pack8<F16>(f);
        }
    }
}


// Matrix-core form for N <= 192 tokens (every tile up to 416x448): one workgroup per (tile, head), a wave owns 16-query blocks.
//   S^T[key][query] = K Q^T     : A = K block (LDS, 80-B rows), B = Q block (straight from global), one 16x16x32 MFMA per key block;
//                                 the lane then holds, for ITS query (lane & 15), the scores of keys 16*jb + 4*(lane >> 4) + r.
//   softmax over keys           : registers + two cross-lane steps (lanes l, l^16, l^32, l^48 share a query), fp32, exp(x - max).
//   O^T[d][query] = V^T P^T     : the probabilities never leave registers: they ARE the B operand, with the k index of a 32-key
//                                 step enumerating keys in the order the lanes hold them; V is staged in LDS in that same order.
//                                 P is split into hi + lo 16-bit parts (two MFMAs) so that no precision is lost to its rounding.
template <bool F16>
__global__ __launch_bounds__(256) void k_attention_mfma(TensorRef qkv, TensorRef out, int N, int nh, float scale) {
    constexpr int KD = 32, HD = 64, MAXKB = 12;
    typedef typename HX<F16>::vec8 hx8;
    typedef typename HX<F16>::elem hel;
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    extern __shared__ __attribute__((aligned(16))) char smraw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, pl = lane & 15;
    const int nkb = (N + 15) >> 4;   // key blocks of 16
    const int nks = (nkb + 1) >> 1;  // k-steps of 32 keys
    char *sK = smraw;                // [nks * 32][80 B]   (64 B of key + 16 B pad: conflict-free 16-B row reads)
    char *sV = smraw + nks * 32 * 80;  // [nks][HD / 16][64 lanes][16 B]: A-operand fragments of V^T
    const int b = blockIdx.x / nh, h = blockIdx.x % nh;
    const bf16_t *base = (const bf16_t *)qkv.p + (int64_t)b * qkv.bs + qkv.co;
    for (int i = tid; i < nks * 32 * (KD / 8); i += 256) {
        int key = i >> 2, c = i & 3;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (key < N) v = *reinterpret_cast<const uint4 *>(base + (int64_t)key * qkv.cs + nh * KD + h * KD + c * 8);
        *reinterpret_cast<uint4 *>(sK + key * 80 + c * 16) = v;
    }
    for (int i = tid; i < nks * 32 * (HD / 8); i += 256) {
        int key = i >> 3, c8 = i & 7;
        uint4 v = make_uint4(0, 0, 0, 0);  // keys beyond N: zeros (their probability is 0, but 0 * garbage could be NaN)
        if (key < N) v = *reinterpret_cast<const uint4 *>(base + (int64_t)key * qkv.cs + 2 * nh * KD + h * HD + c8 * 8);
        const int st = key >> 5, j = ((key >> 4) & 1) * 4 + (key & 3), kg = (key & 15) >> 2;
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int d = c8 * 8 + e;
            uint16_t hv = (uint16_t)(e & 1 ? w[e >> 1] >> 16 : w[e >> 1] & 0xffffu);
            *reinterpret_cast<uint16_t *>(sV + (((st * (HD / 16) + (d >> 4)) * 64 + kg * 16 + (d & 15)) * 16) + j * 2) = hv;
        }
    }
    __syncthreads();
    for (int qb = wave; qb < nkb; qb += 4) {
        const int q = qb * 16 + pl;
        const int qc = q < N ? q : N - 1;
        const hx8 qf = *reinterpret_cast<const hx8 *>(base + (int64_t)qc * qkv.cs + h * KD + g * 8);
        f32x4 s[MAXKB];
        float mx = -INFINITY;
#pragma unroll
        for (int jb = 0; jb < MAXKB; ++jb) {
            if (jb < nkb) {
                hx8 kf = *reinterpret_cast<const hx8 *>(sK + (jb * 16 + pl) * 80 + g * 16);
                s[jb] = HX<F16>::mfma(kf, qf, f32x4{0.f, 0.f, 0.f, 0.f});
            } else {
                s[jb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = (jb * 16 + g * 4 + r < N) ? s[jb][r] * scale : -INFINITY;
                s[jb][r] = v;
                mx = fmaxf(mx, v);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float den = 0.f;
        f32x4 acc[HD / 16];
#pragma unroll
        for (int db = 0; db < HD / 16; ++db) acc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < MAXKB / 2; ++st) {
            if (st >= nks) break;
            hx8 phi, plo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float p = __expf(s[2 * st + (j >> 2)][j & 3] - mx);
                den += p;
                hel hi = (hel)p;
                phi[j] = hi;
                plo[j] = (hel)(p - (float)hi);
            }
#pragma unroll
            for (int db = 0; db < HD / 16; ++db) {
                hx8 vf = *reinterpret_cast<const hx8 *>(sV + ((st * (HD / 16) + db) * 64 + lane) * 16);
                acc[db] = HX<F16>::mfma(vf, phi, acc[db]);
                acc[db] = HX<F16>::mfma(vf, plo, acc[db]);
            }
        }
        den += __shfl_xor(den, 16);
        den += __shfl_xor(den, 32);
        if (q < N) {
            const float inv = 1.0f / den;
            bf16_t *op = (bf16_t *)out.p + (int64_t)b * out.bs + (int64_t)q * out.cs + out.co + h * HD + g * 4;
#pragma unroll
            for (int db = 0; db < HD / 16; ++db) {
                uint2 o;
                o.x = HX<F16>::pack2(acc[db][0] * inv, acc[db][1] * inv);
                o.y = HX<F16>::pack2(acc[db][2] * inv, acc[db][3] * inv);
                *reinterpret_cast<uint2 *>(op + db * 16) = o;
            }
        }
    }
}

static inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

hipError_t launch_dwconv3(const TensorRef &in, const TensorRef &out, const TensorRef &res, const bf16_t *w, const float *bias, int B,
                          int H, int W, int C, int act, bool f16, hipStream_t st) {
    if (C % 8) return hipErrorInvalidValue;
    dim3 grid(blocks_for((int64_t)B * H * ((W + 3) / 4) * (C / 8)));
    if (f16) hipLaunchKernelGGL(k_dwconv3<true>, grid, dim3(256), 0, st, in, out, res, w, bias, B, H, W, C, act);
    else hipLaunchKernelGGL(k_dwconv3<false>, grid, dim3(256), 0, st, in, out, res, w, bias, B, H, W, C, act);
    return hipGetLastError();
}

hipError_t launch_maxpool5(const TensorRef &in, const TensorRef &out, int B, int H, int W, int C, bool f16, hipStream_t st) {
    if (C % 8) return hipErrorInvalidValue;
    dim3 grid(blocks_for((int64_t)B * H * W * (C / 8)));
    if (f16) hipLaunchKernelGGL(k_maxpool5<true>, grid, dim3(256), 0, st, in, out, B, H, W, C);
    else hipLaunchKernelGGL(k_maxpool5<false>, grid, dim3(256), 0, st, in, out, B, H, W, C);
    return hipGetLastError();
}

hipError_t launch_sppf_pools(const TensorRef &cat, int B, int H, int W, int C, bool f16, hipStream_t st) {
    size_t lds = (size_t)H * W * 4 * 16 * 2;
    if (C % 32 || lds > 64 * 1024 || cat.cpb) return hipErrorInvalidValue;
    dim3 grid((unsigned)(B * (C / 32)));
    if (f16) hipLaunchKernelGGL(k_sppf_pools<true>, grid, dim3(256), lds, st, cat, H, W, C);
    else hipLaunchKernelGGL(k_sppf_pools<false>, grid, dim3(256), lds, st, cat, H, W, C);
    return hipGetLastError();
}

hipError_t launch_upsample2(const TensorRef &in, const TensorRef &out, int B, int H, int W, int C, hipStream_t st) {
    if (C % 8) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_upsample2, dim3(blocks_for((int64_t)B * H * W * 4 * (C / 8))), dim3(256), 0, st, in, out, B, H, W, C);
    return hipGetLastError();
}

template <bool F16>
static hipError_t launch_attention_t(const TensorRef &qkv, const TensorRef &out, int B, int N, int nh, int kd, size_t lds, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)k_attention<32, 64, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    float scale = (float)(1.0 / sqrt((double)kd));  // python: key_dim ** -0.5 evaluated in double, applied to an fp32 tensor
    hipLaunchKernelGGL((k_attention<32, 64, F16>), dim3((unsigned)(B * nh)), dim3(256), lds, st, qkv, out, N, nh, scale);
    return hipGetLastError();
}

hipError_t launch_attention(const TensorRef &qkv, const TensorRef &out, int B, int N, int nh, int kd, int hd, bool f16, bool use_mfma, hipStream_t st) {
    if (kd != 32 || hd != 64) return hipErrorInvalidValue;  // head_dim is 64 for every YOLO11 scale (heads = c/64)
    if (use_mfma && N <= 192 && N >= 1) {
        const int nks = ((N + 15) / 16 + 1) / 2;
        size_t lds_m = (size_t)nks * 32 * 80 + (size_t)nks * 4 * 1024;
        float scale = (float)(1.0 / sqrt((double)kd));
        if (f16) hipLaunchKernelGGL(k_attention_mfma<true>, dim3((unsigned)(B * nh)), dim3(256), lds_m, st, qkv, out, N, nh, scale);
        else hipLaunchKernelGGL(k_attention_mfma<false>, dim3((unsigned)(B * nh)), dim3(256), lds_m, st, qkv, out, N, nh, scale);
        return hipGetLastError();
    }
    size_t lds = sizeof(float) * ((size_t)N * 32 + (size_t)N * 64);
    if (lds > 160 * 1024) return hipErrorInvalidValue;  // N <= 422 tokens (input up to 640x640)
    return f16 ? launch_attention_t<true>(qkv, out, B, N, nh, kd, lds, st) : launch_attention_t<false>(qkv, out, B, N, nh, kd, lds, st);
}

}  // namespace obb
