// Network input layer (model.0: Conv 3x3 stride 2 on the uint8 tile) as full-width row stripes.
//
// Replaces, for this one layer, the predictor's preprocess (BGR->RGB, HWC->CHW, /255) + Conv + SiLU
// (Detect_OBB.py:81-83 -> OBBModel layer 0).  The generic implicit-GEMM kernel spends its time on this layer in per-tile
// latency (13x13 tiles, byte loads, four barriers per 169 pixels); the layer itself is pure streaming: 0.5 MB of uint8 in,
// 1.4 MB of 16-bit activations out per 416x416 tile.  Here a workgroup owns 4 output rows x the whole width:
//   * the 9 input rows it needs are whole contiguous byte runs of the image -> 16-B loads, prefetched a stripe ahead,
//   * bytes -> 16-bit v/255 (v * (1/255) rounds to the same 16-bit value as v/255 for all 256 inputs; checked by the host
//     before this kernel is selected) into an LDS image [row][1 + W + 1][4 channels] (left / top zero padding included),
//   * k = (dy, dx-pair, channel): the two horizontally adjacent taps of a stride-2 window are 16 contiguous LDS bytes, so a
//     3x3x(3|4) window is 6 chunks = two 16x16x32 MFMA steps (weights: two A fragments per 16 couts, held in registers),
//   * D^T[cout][pixel] puts 16 consecutive pixels x all couts in one wave: the NHWC store is one contiguous 512*NF-byte run.
#include "stem.h"

#include <algorithm>
#include <cstdlib>

namespace obb {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct StemParams {
    const uint8_t *in; int64_t in_bs;
    bf16_t *out; int64_t out_bs; int out_cs, out_co;
    const bf16_t *wpk; const float *bias;
    int Hin, Win, Hout, Wout, act;
    int stripes_y, nstripes, spw;  // stripes per image, total, per workgroup
    int rb, cpr, nchunk;           // bytes per input row, 16-B chunks per row, chunks per stripe
    int cvt_off, pitch;            // LDS offset of the converted image; its row pitch in bytes
    int gpr; float inv_gpr;        // 4-pixel groups per input row
    int fpr;                       // 16-pixel fragments per output row
};

constexpr int kStemRows = 4, kStemInRows = 2 * kStemRows + 1;

template <int NF, int CH, bool F16>
__global__ __launch_bounds__(256) void k_stem_conv(const StemParams P) {
    typedef typename HX<F16>::vec8 hx8;
    constexpr int MAXPF = CH == 3 ? 3 : 4;  // ceil(9 rows * 416 px * CH / 16 / 256)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, pl = lane & 15;
    const int s0 = blockIdx.x * P.spw;
    const int s1 = min(s0 + P.spw, P.nstripes);
    if (s0 >= s1) return;
    char *raw = smem, *cvt = smem + P.cvt_off;

    // weights: 2 k-steps x NF fragments, resident in registers; bias of the lane's 4*NF couts
    hx8 wf[2][NF];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int f = 0; f < NF; ++f) wf[ks][f] = *reinterpret_cast<const hx8 *>(P.wpk + ((ks * NF + f) * 64 + lane) * 8);
    float bias[NF * 4];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        float4 bv = *reinterpret_cast<const float4 *>(P.bias + g * 4 * NF + f * 4);
        bias[f * 4 + 0] = bv.x; bias[f * 4 + 1] = bv.y; bias[f * 4 + 2] = bv.z; bias[f * 4 + 3] = bv.w;
    }
    // zero padding columns of the converted image (never overwritten): pixel index 0 (x = -1) and W + 1 (read with zero weights)
    for (int i = tid; i < kStemInRows * 2; i += 256) {
        int r = i >> 1, side = i & 1;
        *reinterpret_cast<uint2 *>(cvt + r * P.pitch + (side ? (P.Win + 1) * 8 : 0)) = make_uint2(0, 0);
    }

    u32x4 pre[MAXPF];
    auto issue = [&](int s) {
        const int b = s / P.stripes_y, oy0 = (s - b * P.stripes_y) * kStemRows;
        const uint8_t *src = P.in + (int64_t)b * P.in_bs;
#pragma unroll
        for (int k = 0; k < MAXPF; ++k) {
            int idx = tid + k * 256;
            int r = idx / P.cpr, c = idx - r * P.cpr;
            int gy = 2 * oy0 - 1 + r;
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
            if (idx < P.nchunk && gy >= 0 && gy < P.Hin) v = *reinterpret_cast<const u32x4 *>(src + (int64_t)gy * P.rb + c * 16);
            pre[k] = v;
        }
    };
    issue(s0);
    for (int s = s0; s < s1; ++s) {
        const int b = s / P.stripes_y, oy0 = (s - b * P.stripes_y) * kStemRows;
        __syncthreads();  // the previous stripe's fragments are done with the LDS image
#pragma unroll
        for (int k = 0; k < MAXPF; ++k) {
            int idx = tid + k * 256;
            if (idx < P.nchunk) *reinterpret_cast<u32x4 *>(raw + idx * 16) = pre[k];
        }
        __syncthreads();
        if (s + 1 < s1) issue(s + 1);
        // ---- bytes -> 16-bit v/255, four pixels per work item
        for (int i = tid; i < kStemInRows * P.gpr; i += 256) {
            int r = (int)(((float)i + 0.5f) * P.inv_gpr);
            int k4 = i - r * P.gpr;
            const unsigned *sp = reinterpret_cast<const unsigned *>(raw + r * P.rb + k4 * 4 * CH);
            unsigned bytes[CH * 4];
#pragma unroll
            for (int d = 0; d < CH; ++d) {
                unsigned w = sp[d];
                bytes[d * 4 + 0] = w & 0xffu; bytes[d * 4 + 1] = (w >> 8) & 0xffu; bytes[d * 4 + 2] = (w >> 16) & 0xffu; bytes[d * 4 + 3] = w >> 24;
            }
            char *dp = cvt + r * P.pitch + (1 + 4 * k4) * 8;
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                float c0 = (float)bytes[px * CH + 0] * (1.0f / 255.0f), c1 = (float)bytes[px * CH + 1] * (1.0f / 255.0f);
                float c2 = (float)bytes[px * CH + 2] * (1.0f / 255.0f), c3 = CH == 4 ? (float)bytes[px * CH + (CH - 1)] * (1.0f / 255.0f) : 0.f;
                uint2 o;
                o.x = HX<F16>::pack2(c0, c1);
                o.y = HX<F16>::pack2(c2, c3);
                *reinterpret_cast<uint2 *>(dp + px * 8) = o;
            }
        }
        __syncthreads();
        // ---- MFMA + epilogue + store, one 16-pixel fragment at a time
        const int nfrag = kStemRows * P.fpr;
        for (int fr = wave; fr < nfrag; fr += 4) {
            const int ry = fr / P.fpr, ox = (fr - ry * P.fpr) * 16 + pl;
            f32x4 acc[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                int q = ks * 4 + g;
                q = q < 6 ? q : 5;  // padding chunks: valid address, zero weights
                const int dy = q >> 1, part = q & 1;
                hx8 a = *reinterpret_cast<const hx8 *>(cvt + (ry * 2 + dy) * P.pitch + (ox + part) * 16);
#pragma unroll
                for (int f = 0; f < NF; ++f) acc[f] = HX<F16>::mfma(wf[ks][f], a, acc[f]);
            }
            const int oy = oy0 + ry;
            if (oy >= P.Hout) continue;
            float v[NF * 4];
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = acc[f][r] + bias[f * 4 + r];
                    if (P.act) x = silu_f(x);
                    v[f * 4 + r] = x;
                }
            bf16_t *op = P.out + (int64_t)b * P.out_bs + ((int64_t)oy * P.Wout + ox) * P.out_cs + P.out_co + g * 4 * NF;
            if constexpr (NF == 1) {
                uint2 o;
                o.x = HX<F16>::pack2(v[0], v[1]); o.y = HX<F16>::pack2(v[2], v[3]);
                *reinterpret_cast<uint2 *>(op) = o;
            } else {
#pragma unroll
                for (int h = 0; h < NF / 2; ++h) {
                    uint4 o;
                    o.x = HX<F16>::pack2(v[h * 8 + 0], v[h * 8 + 1]); o.y = HX<F16>::pack2(v[h * 8 + 2], v[h * 8 + 3]);
                    o.z = HX<F16>::pack2(v[h * 8 + 4], v[h * 8 + 5]); o.w = HX<F16>::pack2(v[h * 8 + 6], v[h * 8 + 7]);
                    *reinterpret_cast<uint4 *>(op + h * 8) = o;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side

bool stem_supported(int cin, int cout, int ks, int stride, int Hin, int Win) {
    if (ks != 3 || stride != 2 || (cin != 3 && cin != 4)) return false;
    if (cout != 16 && cout != 32 && cout != 64) return false;
    if (Win % 32 || Hin % 2 || Win > 416 || Win < 32) return false;
    return true;
}

bool stem_scale_is_exact(bool f16) {
    for (int v = 0; v < 256; ++v)
        if (host_to_half((float)v / 255.0f, f16) != host_to_half((float)v * (1.0f / 255.0f), f16)) return false;
    return true;
}

std::vector<bf16_t> pack_stem_weights(const float *w, int cout, int cin, bool flip_bgr, bool f16) {
    const int NF = cout / 16;
    std::vector<bf16_t> out((size_t)2 * NF * 64 * 8, 0);
    for (int ks = 0; ks < 2; ++ks)
        for (int f = 0; f < NF; ++f)
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 15, g = lane >> 4;
                const int co = (r >> 2) * 4 * NF + f * 4 + (r & 3);  // a lane of D owns 4*NF contiguous couts (same permutation as conv.hip)
                const int q = ks * 4 + g;
                if (q >= 6) continue;
                const int dy = q >> 1, part = q & 1;
                for (int j = 0; j < 8; ++j) {
                    const int dx = part ? 2 : (j >> 2), cm = j & 3;
                    if ((part && j >= 4) || cm >= cin) continue;
                    const int c = flip_bgr ? 2 - cm : cm;  // memory order B,G,R -> model order R,G,B
                    out[((size_t)(ks * NF + f) * 64 + lane) * 8 + j] = host_to_half(w[(((size_t)co * cin + c) * 3 + dy) * 3 + dx], f16);
                }
            }
    return out;
}

template <int NF, int CH>
static hipError_t launch_t(const StemLaunch &L, const StemParams &P, dim3 grid, size_t lds, hipStream_t st) {
    if (L.f16) hipLaunchKernelGGL((k_stem_conv<NF, CH, true>), grid, dim3(256), lds, st, P);
    else hipLaunchKernelGGL((k_stem_conv<NF, CH, false>), grid, dim3(256), lds, st, P);
    return hipGetLastError();
}

hipError_t launch_stem(const StemLaunch &L, hipStream_t st) {
    if (!stem_supported(L.cin, L.cout, 3, 2, L.Hin, L.Win)) return hipErrorInvalidValue;
    StemParams P;
    P.in = L.in; P.in_bs = (int64_t)L.Hin * L.Win * L.cin;
    P.out = (bf16_t *)L.out.p; P.out_bs = L.out.bs; P.out_cs = L.out.cs; P.out_co = L.out.co;
    P.wpk = L.wpk; P.bias = L.bias;
    P.Hin = L.Hin; P.Win = L.Win; P.Hout = L.Hin / 2; P.Wout = L.Win / 2; P.act = L.act;
    P.stripes_y = (P.Hout + kStemRows - 1) / kStemRows;
    int64_t ns = (int64_t)L.B * P.stripes_y;
    if (ns <= 0 || ns >= (1ll << 31)) return hipErrorInvalidValue;
    P.nstripes = (int)ns;
    P.rb = L.Win * L.cin; P.cpr = P.rb / 16; P.nchunk = kStemInRows * P.cpr;
    if (P.rb % 16 || P.nchunk > (L.cin == 3 ? 3 : 4) * 256) return hipErrorInvalidValue;
    P.cvt_off = (kStemInRows * P.rb + 15) / 16 * 16;
    P.pitch = (L.Win + 2) * 8;
    P.gpr = L.Win / 4; P.inv_gpr = 1.0f / (float)P.gpr;
    P.fpr = P.Wout / 16;
    size_t lds = (size_t)P.cvt_off + (size_t)kStemInRows * P.pitch;
    static const int spw_max = getenv("OBB_STEM_SPW") ? std::max(1, atoi(getenv("OBB_STEM_SPW"))) : 4;
    int64_t spw = ns / (256 * 3 * 2);
    P.spw = (int)std::max<int64_t>(1, std::min<int64_t>(spw, spw_max));
    dim3 grid((unsigned)((ns + P.spw - 1) / P.spw));
    const int NF = L.cout / 16;
    if (L.cin == 3) {
        if (NF == 1) return launch_t<1, 3>(L, P, grid, lds, st);
        if (NF == 2) return launch_t<2, 3>(L, P, grid, lds, st);
        if (NF == 4) return launch_t<4, 3>(L, P, grid, lds, st);
    } else {
        if (NF == 1) return launch_t<1, 4>(L, P, grid, lds, st);
        if (NF == 2) return launch_t<2, 4>(L, P, grid, lds, st);
        if (NF == 4) return launch_t<4, 4>(L, P, grid, lds, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace obb
