// Network input layer (model.0: Conv 3x3 stride 2 on the uint8 tile) as full-width row stripes.
//
// Replaces, for this one layer, the predictor's preprocess (BGR->RGB, HWC->CHW, /255) + Conv + SiLU
// (Detect_OBB.py:81-83 -> OBBModel layer 0).  The generic implicit-GEMM kernel spends its time on this layer in per-tile
// latency (13x13 tiles, byte loads, four barriers per 169 pixels); the layer itself is pure streaming: 0.5 MB of uint8 in,
// 1.4 MB of 16-bit activations out per 416x416 tile.  Here a workgroup owns 4 output rows x the whole width:
//   * the 9 input rows it needs are whole contiguous byte runs of the image: a thread owns 16 consecutive pixels (CH 16-B
//     loads, prefetched a stripe ahead into registers),
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
    int Hin, Win, Hout, Wout;
    int stripes_y, nstripes, spw;  // stripes per image, total, per workgroup
    int rb;                        // bytes per input row
    int pitch;                     // row pitch of the converted LDS image in bytes
    int ipr, nitem; float inv_ipr; // 16-pixel work items per input row, per stripe (<= 256: one per thread)
    int fpr;                       // 16-pixel fragments per output row
};

constexpr int vmcnt_imm(int n) { return (n & 15) | (7 << 4) | (15 << 8) | ((n >> 4) << 14); }  // s_waitcnt vmcnt(n), other counters untouched (gfx9 encoding)
constexpr int kStemRows = 4, kStemInRows = 2 * kStemRows + 1;  // 4 output rows per stripe = one per wave

// FPR > 0: fragments per output row known at compile time -> the fragment loop is fully unrolled, which lets the compiler wait for
// the prefetched loads with an exact vmcnt (loads and stores share one in-order counter on gfx9: a loop of unknown length forces
// vmcnt(0), i.e. a full drain of this stripe's stores before the next stripe can start) and overlap consecutive fragments.
template <int NF, int CH, bool F16, int FPR>
__global__ __launch_bounds__(256) void k_stem_conv(const StemParams P) {
    typedef typename HX<F16>::vec8 hx8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, pl = lane & 15;
    const int s0 = blockIdx.x * P.spw;
    const int s1 = min(s0 + P.spw, P.nstripes);
    if (s0 >= s1) return;
    char *cvt = smem;

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
    // loads and stores share one in-order counter (vmcnt): settle the loop-invariant loads here, otherwise their first use inside the
    // stripe loop would also wait for the stripe prefetch issued just before it
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
    // zero padding columns of the converted image (never overwritten): pixel index 0 (x = -1) and W + 1 (read with zero weights)
    for (int i = tid; i < kStemInRows * 2; i += 256) {
        int r = i >> 1, side = i & 1;
        *reinterpret_cast<uint2 *>(cvt + r * P.pitch + (side ? (P.Win + 1) * 8 : 0)) = make_uint2(0, 0);
    }

    // work item = 16 consecutive pixels of one input row = CH contiguous 16-B chunks; a stripe has at most 256 items, so a thread
    // owns one item: its bytes are prefetched a stripe ahead into registers and converted straight into the LDS image
    // (everything below is branch-free on purpose: loads, conversions and LDS writes are unconditional, so the compiler's vmcnt
    // bookkeeping stays exact and the wait for the prefetch does not have to drain the stores issued after it)
    const int my_r0 = (int)(((float)tid + 0.5f) * P.inv_ipr);
    const bool mine = tid < P.nitem;
    const int my_r = mine ? my_r0 : 0, my_k = mine ? tid - my_r0 * P.ipr : 0;
    char *const dp = mine ? cvt + my_r * P.pitch + (1 + 16 * my_k) * 8 : cvt + kStemInRows * P.pitch;  // others: 128-B dummy slot
    u32x4 pre[CH];
    bool pre_ok = false;
    auto issue = [&](int s) {
        const int b = s / P.stripes_y, oy0 = (s - b * P.stripes_y) * kStemRows;
        const int gy = 2 * oy0 - 1 + my_r;
        pre_ok = mine && gy >= 0 && gy < P.Hin;
        const uint8_t *src = P.in + (int64_t)b * P.in_bs + (int64_t)min(max(gy, 0), P.Hin - 1) * P.rb + my_k * 16 * CH;
#pragma unroll
        for (int k = 0; k < CH; ++k) pre[k] = *reinterpret_cast<const u32x4 *>(src + k * 16);
    };
    issue(s0);
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));  // first stripe: settled before the loop, so that inside the loop the prefetch is always "FPR stores old"
    for (int s = s0; s < s1; ++s) {
        const int b = s / P.stripes_y, oy0 = (s - b * P.stripes_y) * kStemRows;
        __syncthreads();  // the previous stripe's fragments are done with the LDS image
        {  // ---- bytes -> 16-bit v/255 (rows outside the image: zeros)
            unsigned bytes[CH * 16];
#pragma unroll
            for (int d = 0; d < CH * 4; ++d) {
                unsigned w = pre_ok ? pre[d >> 2][d & 3] : 0u;
                bytes[d * 4 + 0] = w & 0xffu; bytes[d * 4 + 1] = (w >> 8) & 0xffu; bytes[d * 4 + 2] = (w >> 16) & 0xffu; bytes[d * 4 + 3] = w >> 24;
            }
#pragma unroll
            for (int px = 0; px < 16; ++px) {
                float c0 = (float)bytes[px * CH + 0] * (1.0f / 255.0f), c1 = (float)bytes[px * CH + 1] * (1.0f / 255.0f);
                float c2 = (float)bytes[px * CH + 2] * (1.0f / 255.0f), c3 = CH == 4 ? (float)bytes[px * CH + (CH - 1)] * (1.0f / 255.0f) : 0.f;
                uint2 o;
                o.x = HX<F16>::pack2(c0, c1);
                o.y = HX<F16>::pack2(c2, c3);
                *reinterpret_cast<uint2 *>(dp + px * 8) = o;
            }
        }
        __syncthreads();
        issue(min(s + 1, s1 - 1));  // unconditional (the last stripe re-reads its own rows): keeps the outstanding-op count exact
        // ---- MFMA + epilogue + store: wave w owns output row w of the stripe, one 16-pixel fragment at a time
        const int ry = wave;
        const int oy = oy0 + ry;
#pragma unroll
        for (int i = 0; i < (FPR > 0 ? FPR : P.fpr); ++i) {
            const int ox = i * 16 + pl;
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
            float v[NF * 4];
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = acc[f][r] + bias[f * 4 + r];
                    v[f * 4 + r] = silu_f(x);
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
    if (Win % 32 || Hin % 8 || Win > 416 || Win < 32) return false;  // whole stripes of 4 output rows, whole 16-pixel fragments
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
    if (P.fpr == 13) {  // 416-px tiles
        if (L.f16) hipLaunchKernelGGL((k_stem_conv<NF, CH, true, 13>), grid, dim3(256), lds, st, P);
        else hipLaunchKernelGGL((k_stem_conv<NF, CH, false, 13>), grid, dim3(256), lds, st, P);
    } else {
        if (L.f16) hipLaunchKernelGGL((k_stem_conv<NF, CH, true, 0>), grid, dim3(256), lds, st, P);
        else hipLaunchKernelGGL((k_stem_conv<NF, CH, false, 0>), grid, dim3(256), lds, st, P);
    }
    return hipGetLastError();
}

hipError_t launch_stem(const StemLaunch &L, hipStream_t st) {
    if (!stem_supported(L.cin, L.cout, 3, 2, L.Hin, L.Win) || !L.act) return hipErrorInvalidValue;
    StemParams P;
    P.in = L.in; P.in_bs = (int64_t)L.Hin * L.Win * L.cin;
    P.out = (bf16_t *)L.out.p; P.out_bs = L.out.bs; P.out_cs = L.out.cs; P.out_co = L.out.co;
    P.wpk = L.wpk; P.bias = L.bias;
    P.Hin = L.Hin; P.Win = L.Win; P.Hout = L.Hin / 2; P.Wout = L.Win / 2;
    P.stripes_y = (P.Hout + kStemRows - 1) / kStemRows;
    int64_t ns = (int64_t)L.B * P.stripes_y;
    if (ns <= 0 || ns >= (1ll << 31)) return hipErrorInvalidValue;
    P.nstripes = (int)ns;
    P.rb = L.Win * L.cin;
    P.ipr = L.Win / 16; P.nitem = kStemInRows * P.ipr; P.inv_ipr = 1.0f / (float)P.ipr;
    if (P.rb % 16 || P.nitem > 256) return hipErrorInvalidValue;
    P.pitch = (L.Win + 2) * 8;
    P.fpr = P.Wout / 16;
    size_t lds = (size_t)kStemInRows * P.pitch + 128;  // + dummy slot written by threads without a work item
    static const int spw_max = getenv("OBB_STEM_SPW") ? std::max(1, atoi(getenv("OBB_STEM_SPW"))) : 4;
    int64_t spw = ns / (256 * 5 * 2);
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
