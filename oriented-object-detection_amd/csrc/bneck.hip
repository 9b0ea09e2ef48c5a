// The Bottleneck inside a C3k2 block (ultralytics Bottleneck: 3x3 c -> c/2, 3x3 c/2 -> c, + shortcut) as ONE
// kernel over full-width row stripes, for the high-resolution levels where the two convs are far too small to hide memory latency
// as separate launches (16 -> 8 -> 16 channels at 104x104, 32 -> 16 -> 32 at 52x52).
//
// y1 and y2 are members of the block's channel-blocked concat buffer, i.e. dense [image][pixel][C] planes (conv.h TensorRef::cpb).
// A workgroup owns 4 output rows x the whole width:
//   * the 8 input rows it needs are contiguous byte runs: 16-B loads prefetched a stripe ahead into registers, committed to an LDS
//     image [row][1 + W + 1][C] (zero columns / rows = the convs' zero padding),
//   * conv 1 (3x3, C -> C/2, SiLU) runs on 6 rows and leaves its result in LDS (rounded to the storage type exactly like the
//     separate kernel would store it; rows outside the image are zero because conv 2 zero-pads ITS input),
//   * conv 2 (3x3, C/2 -> C, SiLU) + shortcut (y1 re-read from LDS) stores y2 straight from registers: 16 pixels x C channels per
//     wave instruction are one contiguous run of the dense plane.
// CO > 0: the block's closing 1x1 (cv2 over [y0 | y1 | y2], C3k2 with one Bottleneck) runs on each 16-pixel fragment right behind
// conv 2: y0 comes straight from global memory as the MFMA B operand (used once: no reason to stage it), y1 is re-read from the LDS
// image, and y2 never leaves the registers -- with the cout permutation of the weight packing a lane's conv-2 results ARE the 8 (C = 32)
// or 4 (C = 16: the other half of that k step's slots is zero) consecutive k values the next GEMM wants from it.  The 1x1's weights live in registers (two
// workgroups per CU leave each wave 256 VGPRs).  y2 is then never written and the 3C-channel concat is never read back.
// Same MFMA mapping, weight-fragment order, rounding points and k order as k_conv_igemm with a single channel stage.  The stripe loop
// is branch-free with compile-time trip counts so that the wait for the prefetch is an exact vmcnt (see stem.hip).
#include "bneck.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace obb {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct BneckParams {
    const bf16_t *y1; bf16_t *y2; int64_t bs;  // plane bases of the two members, batch stride (elements)
    const bf16_t *w1pk, *w2pk; const float *bias1, *bias2;
    int H, stripes_y, nstripes, spw;
    // closing 1x1 (CO > 0): y0 plane, packed weights (k = 32-wide steps, then for C = 16 the 16-wide step over y2), bias, output tensor
    const bf16_t *y0; const bf16_t *wc32, *wc16; const float *biasc;
    bf16_t *out; int64_t out_bs, out_ps; int out_cs, out_bsh, out_bmask;
};

constexpr int kBnRows = 4;

template <int C, int W, int CO, bool F16>
__global__ __launch_bounds__(256, 2) void k_bneck_stripe(const BneckParams P) {
    typedef typename HX<F16>::vec8 hx8;
    constexpr int CH = C / 2, R = kBnRows, XR = R + 4, TR = R + 2, XW = W + 2;
    constexpr int XP = C * 2 + 16, TP = CH == 8 ? 16 : CH * 2 + 16;  // LDS bytes per pixel (+16 B: conflict-free 16-B row reads)
    constexpr int XB = XR * XW * XP, TB = TR * XW * TP;
    constexpr int CPK1 = C / 8, NQ1 = 9 * CPK1, KST1 = (NQ1 + 3) / 4;
    constexpr int CPK2 = CH / 8, NQ2 = 9 * CPK2, KST2 = (NQ2 + 3) / 4, NF2 = C / 16;
    constexpr int NPX1 = TR * W, FPW1 = ((NPX1 + 15) / 16 + 3) / 4;  // fragments per wave (padded: surplus fragments recompute the last one)
    constexpr int NPX2 = R * W, FPW2 = ((NPX2 + 15) / 16 + 3) / 4;
    constexpr int CPR = W * CPK1, NCHUNK = XR * CPR, MAXPF = (NCHUNK + 255) / 256;  // 16-B chunks per input row / stripe / thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool WCL = CO == 128;  // 24 weight fragments do not fit the register file next to everything else: the y2 step reads LDS
    char *X = smem, *T = smem + XB, *W1 = T + TB, *W2 = W1 + KST1 * 1024, *WC = W2 + KST2 * NF2 * 1024, *dummy = WC + (WCL ? CO / 16 * 1024 : 0);
    __shared__ __attribute__((aligned(16))) float s_b1[16], s_b2[16 * NF2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, pl = lane & 15;
    const int s0 = blockIdx.x * P.spw;
    const int s1 = min(s0 + P.spw, P.nstripes);
    if (s0 >= s1) return;

    // loop-invariant state: weights and bias in LDS, zero padding columns of both images (never overwritten)
    for (int i = tid; i < KST1 * 64; i += 256) reinterpret_cast<u32x4 *>(W1)[i] = reinterpret_cast<const u32x4 *>(P.w1pk)[i];
    for (int i = tid; i < KST2 * NF2 * 64; i += 256) reinterpret_cast<u32x4 *>(W2)[i] = reinterpret_cast<const u32x4 *>(P.w2pk)[i];
    if (tid < 16) s_b1[tid] = P.bias1[tid];
    if (tid < 16 * NF2) s_b2[tid] = P.bias2[tid];
    constexpr int NFC = CO / 16, KSC = C == 32 ? 3 : 1;  // closing 1x1: cout fragments, 32-wide k steps
    __shared__ __attribute__((aligned(16))) float s_bc[CO > 0 ? CO : 4];
    constexpr int KSR = WCL ? KSC - 1 : KSC;  // k steps whose weights stay in registers
    hx8 wc[CO > 0 ? KSR : 1][CO > 0 ? NFC : 1];
    hx8 wcy2[CO > 0 && C == 16 ? NFC : 1];  // C = 16: the k step over y2, half of its slots zero (a lane holds only 4 y2 channels)
    if constexpr (CO > 0) {
        if (tid < CO) s_bc[tid] = P.biasc[tid];
#pragma unroll
        for (int ks = 0; ks < KSR; ++ks)
#pragma unroll
            for (int f = 0; f < NFC; ++f) wc[ks][f] = *reinterpret_cast<const hx8 *>(P.wc32 + ((size_t)(ks * NFC + f) * 64 + lane) * 8);
        if constexpr (WCL)
            for (int i = tid; i < NFC * 64; i += 256) reinterpret_cast<u32x4 *>(WC)[i] = reinterpret_cast<const u32x4 *>(P.wc32 + (size_t)KSR * NFC * 512)[i];
        if constexpr (C == 16) {
#pragma unroll
            for (int f = 0; f < NFC; ++f) wcy2[f] = *reinterpret_cast<const hx8 *>(P.wc16 + ((size_t)f * 64 + lane) * 8);
        }
    }
    for (int i = tid; i < XR * 2 * (XP / 16); i += 256) {
        int r = i / (2 * (XP / 16)), rem = i - r * (2 * (XP / 16)), side = rem / (XP / 16), c = rem - side * (XP / 16);
        *reinterpret_cast<u32x4 *>(X + (r * XW + (side ? W + 1 : 0)) * XP + c * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    for (int i = tid; i < TR * 2 * (TP / 16); i += 256) {
        int r = i / (2 * (TP / 16)), rem = i - r * (2 * (TP / 16)), side = rem / (TP / 16), c = rem - side * (TP / 16);
        *reinterpret_cast<u32x4 *>(T + (r * XW + (side ? W + 1 : 0)) * TP + c * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    __builtin_amdgcn_s_waitcnt((0 & 15) | (7 << 4) | (15 << 8));  // vmcnt(0): settle the weight / bias loads outside the stripe loop

    // input chunks of this thread: idx = tid + k*256 -> (row, chunk in row); surplus slots re-read the last chunk into a dummy LDS slot
    int src_row[MAXPF], src_off[MAXPF], lds_off[MAXPF];
#pragma unroll
    for (int k = 0; k < MAXPF; ++k) {
        int idx = tid + k * 256;
        bool real = idx < NCHUNK;
        idx = real ? idx : NCHUNK - 1;
        int r = idx / CPR, j = idx - r * CPR;
        src_row[k] = r; src_off[k] = j * 8;
        lds_off[k] = real ? (r * XW + 1 + j / CPK1) * XP + (j % CPK1) * 16 : (int)(dummy - X) + (tid & 63) * 16;
    }
    u32x4 pre[MAXPF];
    unsigned pre_ok = 0;
    auto issue = [&](int s) {
        const int b = s / P.stripes_y, oy0 = (s - b * P.stripes_y) * R;
        const bf16_t *src = P.y1 + (int64_t)b * P.bs;
        pre_ok = 0;
#pragma unroll
        for (int k = 0; k < MAXPF; ++k) {
            int gy = oy0 - 2 + src_row[k];
            if (gy >= 0 && gy < P.H) pre_ok |= 1u << k;
            gy = min(max(gy, 0), P.H - 1);
            pre[k] = *reinterpret_cast<const u32x4 *>(src + ((int64_t)gy * W) * C + src_off[k]);
        }
    };
    issue(s0);
    __builtin_amdgcn_s_waitcnt((0 & 15) | (7 << 4) | (15 << 8));
    for (int s = s0; s < s1; ++s) {
        const int b = s / P.stripes_y, oy0 = (s - b * P.stripes_y) * R;
        __syncthreads();  // the previous stripe is done with both LDS images
#pragma unroll
        for (int k = 0; k < MAXPF; ++k) *reinterpret_cast<u32x4 *>(X + lds_off[k]) = ((pre_ok >> k) & 1u) ? pre[k] : u32x4{0u, 0u, 0u, 0u};
        __syncthreads();
        u32x4 y0r[CO > 0 ? FPW2 : 1];
        if constexpr (CO > 0) {  // y0 of this stripe's pixels: the B operand of the closing 1x1, needed only after conv 1
            const bf16_t *y0b = P.y0 + (int64_t)b * P.bs;
#pragma unroll
            for (int i = 0; i < FPW2; ++i) {
                int p = (wave + 4 * i) * 16 + pl;
                p = p < NPX2 ? p : NPX2 - 1;
                y0r[i] = *reinterpret_cast<const u32x4 *>(y0b + ((int64_t)oy0 * W + p) * C + (g & (C / 8 - 1)) * 8);
            }
        }
        issue(min(s + 1, s1 - 1));  // unconditional (the last stripe re-reads its own rows): keeps the outstanding-op count exact

        // ---- conv 1: 3x3, C -> C/2, SiLU, rows oy0-1 .. oy0+R of the level -> T (zero outside the image)
        // (no global memory traffic in this loop: it need not be unrolled for the vmcnt bookkeeping, and rolling it keeps registers down)
#pragma unroll 2
        for (int i = 0; i < FPW1; ++i) {
            int p = (wave + 4 * i) * 16 + pl;
            p = p < NPX1 ? p : NPX1 - 1;
            const int tr = p / W, x = p - tr * W;
            const char *xb = X + (tr * XW + x) * XP;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KST1; ++ks) {
                hx8 wf = *reinterpret_cast<const hx8 *>(W1 + (ks * 64 + lane) * 16);
                int q = ks * 4 + g;
                q = q < NQ1 ? q : NQ1 - 1;  // padding chunks: valid address, zero weights
                const int tap = q / CPK1, c0 = q - tap * CPK1, dy = tap / 3, dx = tap - dy * 3;
                hx8 a = *reinterpret_cast<const hx8 *>(xb + (dy * XW + dx) * XP + c0 * 16);
                acc = HX<F16>::mfma(wf, a, acc);
            }
            const int gy = oy0 - 1 + tr;
            const bool inside = gy >= 0 && gy < P.H;
            float4 bv = *reinterpret_cast<const float4 *>(s_b1 + g * 4);
            float v0 = silu_f(acc[0] + bv.x), v1 = silu_f(acc[1] + bv.y), v2 = silu_f(acc[2] + bv.z), v3 = silu_f(acc[3] + bv.w);
            uint2 o;
            o.x = inside ? HX<F16>::pack2(v0, v1) : 0u;
            o.y = inside ? HX<F16>::pack2(v2, v3) : 0u;
            if (g * 4 < CH) *reinterpret_cast<uint2 *>(T + (tr * XW + x + 1) * TP + g * 8) = o;
        }
        __syncthreads();
        // ---- conv 2: 3x3, C/2 -> C, SiLU, + y1 (shortcut), rows oy0 .. oy0+R-1 -> y2
        bf16_t *dst = P.y2 + (int64_t)b * P.bs;
#pragma unroll
        for (int i = 0; i < FPW2; ++i) {
            int p = (wave + 4 * i) * 16 + pl;
            p = p < NPX2 ? p : NPX2 - 1;
            const int r = p / W, x = p - r * W;
            const char *tb = T + (r * XW + x) * TP;
            f32x4 acc[NF2];
#pragma unroll
            for (int f = 0; f < NF2; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KST2; ++ks) {
                int q = ks * 4 + g;
                q = q < NQ2 ? q : NQ2 - 1;
                const int tap = q / CPK2, c0 = q - tap * CPK2, dy = tap / 3, dx = tap - dy * 3;
                hx8 a = *reinterpret_cast<const hx8 *>(tb + (dy * XW + dx) * TP + c0 * 16);
#pragma unroll
                for (int f = 0; f < NF2; ++f) {
                    hx8 wf = *reinterpret_cast<const hx8 *>(W2 + ((ks * NF2 + f) * 64 + lane) * 16);
                    acc[f] = HX<F16>::mfma(wf, a, acc[f]);
                }
            }
            const char *rp = X + ((r + 2) * XW + x + 1) * XP + g * 8 * NF2;  // shortcut: y1 at the output pixel, this lane's 4*NF2 channels
            float v[NF2 * 4];
#pragma unroll
            for (int f = 0; f < NF2; ++f) {
                float4 bv = *reinterpret_cast<const float4 *>(s_b2 + g * 4 * NF2 + f * 4);
                uint2 rv = *reinterpret_cast<const uint2 *>(rp + f * 8);
                v[f * 4 + 0] = silu_f(acc[f][0] + bv.x) + HX<F16>::lo(rv.x);
                v[f * 4 + 1] = silu_f(acc[f][1] + bv.y) + HX<F16>::hi(rv.x);
                v[f * 4 + 2] = silu_f(acc[f][2] + bv.z) + HX<F16>::lo(rv.y);
                v[f * 4 + 3] = silu_f(acc[f][3] + bv.w) + HX<F16>::hi(rv.y);
            }
            if constexpr (CO == 0) {
                bf16_t *op = dst + ((int64_t)(oy0 + r) * W + x) * C + g * 4 * NF2;
                if constexpr (NF2 == 1) {
                    uint2 o;
                    o.x = HX<F16>::pack2(v[0], v[1]); o.y = HX<F16>::pack2(v[2], v[3]);
                    *reinterpret_cast<uint2 *>(op) = o;
                } else {
                    uint4 o;
                    o.x = HX<F16>::pack2(v[0], v[1]); o.y = HX<F16>::pack2(v[2], v[3]);
                    o.z = HX<F16>::pack2(v[4], v[5]); o.w = HX<F16>::pack2(v[6], v[7]);
                    *reinterpret_cast<uint4 *>(op) = o;
                }
            } else {
                // ---- closing 1x1 over [y0 | y1 | y2] of these 16 pixels, SiLU, 16-bit rows of the block's output tensor
                f32x4 acc3[NFC];
#pragma unroll
                for (int f = 0; f < NFC; ++f) acc3[f] = f32x4{0.f, 0.f, 0.f, 0.f};
                const char *y1p = X + ((r + 2) * XW + x + 1) * XP;
                if constexpr (C == 32) {
                    const hx8 b0 = __builtin_bit_cast(hx8, y0r[i]);
                    const hx8 b1 = *reinterpret_cast<const hx8 *>(y1p + g * 16);
                    u32x4 y2u;
                    y2u.x = HX<F16>::pack2(v[0], v[1]); y2u.y = HX<F16>::pack2(v[2], v[3]);
                    y2u.z = HX<F16>::pack2(v[4], v[5]); y2u.w = HX<F16>::pack2(v[6], v[7]);
                    const hx8 b2 = __builtin_bit_cast(hx8, y2u);
#pragma unroll
                    for (int f = 0; f < NFC; ++f) acc3[f] = HX<F16>::mfma(wc[0][f], b0, acc3[f]);
#pragma unroll
                    for (int f = 0; f < NFC; ++f) acc3[f] = HX<F16>::mfma(wc[1][f], b1, acc3[f]);
#pragma unroll
                    for (int f = 0; f < NFC; ++f) {
                        if constexpr (WCL) acc3[f] = HX<F16>::mfma(*reinterpret_cast<const hx8 *>(WC + (f * 64 + lane) * 16), b2, acc3[f]);
                        else acc3[f] = HX<F16>::mfma(wc[KSR - 1][f], b2, acc3[f]);
                    }
                } else {
                    const u32x4 l1 = *reinterpret_cast<const u32x4 *>(y1p + (g & 1) * 16);
                    const u32x4 bu = g < 2 ? y0r[i] : l1;  // k chunks 0, 1 = y0, chunks 2, 3 = y1
                    const hx8 b01 = __builtin_bit_cast(hx8, bu);
                    u32x4 y2u;
                    y2u.x = HX<F16>::pack2(v[0], v[1]); y2u.y = HX<F16>::pack2(v[2], v[3]); y2u.z = 0u; y2u.w = 0u;
                    const hx8 b2 = __builtin_bit_cast(hx8, y2u);
#pragma unroll
                    for (int f = 0; f < NFC; ++f) acc3[f] = HX<F16>::mfma(wc[0][f], b01, acc3[f]);
#pragma unroll
                    for (int f = 0; f < NFC; ++f) acc3[f] = HX<F16>::mfma(wcy2[f], b2, acc3[f]);
                }
                bf16_t *ob = P.out + (int64_t)b * P.out_bs + ((int64_t)(oy0 + r) * W + x) * P.out_cs;
#pragma unroll
                for (int h = 0; h < NFC / 2; ++h) {
                    float4 ba = *reinterpret_cast<const float4 *>(s_bc + g * 4 * NFC + h * 8), bb = *reinterpret_cast<const float4 *>(s_bc + g * 4 * NFC + h * 8 + 4);
                    uint4 o;
                    o.x = HX<F16>::pack2(silu_f(acc3[2 * h][0] + ba.x), silu_f(acc3[2 * h][1] + ba.y));
                    o.y = HX<F16>::pack2(silu_f(acc3[2 * h][2] + ba.z), silu_f(acc3[2 * h][3] + ba.w));
                    o.z = HX<F16>::pack2(silu_f(acc3[2 * h + 1][0] + bb.x), silu_f(acc3[2 * h + 1][1] + bb.y));
                    o.w = HX<F16>::pack2(silu_f(acc3[2 * h + 1][2] + bb.z), silu_f(acc3[2 * h + 1][3] + bb.w));
                    const int occ = g * (NFC / 2) + h;  // 8-channel chunk of the output slice
                    *reinterpret_cast<uint4 *>(ob + (int64_t)(occ >> P.out_bsh) * P.out_ps + ((occ & P.out_bmask) << 3)) = o;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side

// (W = 104 / 52: the 416-px tile's levels; W = 32 / 16: the same blocks of a 128-px tile -- the dual-scale default of Detect_OBB.py:24-28)
bool bneck_supported(int C, int H, int W) { return ((C == 16 && (W == 104 || W == 32)) || (C == 32 && (W == 52 || W == 16))) && H > 0 && H % kBnRows == 0; }

bool bneck_cv2_supported(int C, int CO) { return (C == 16 && CO == 64) || (C == 32 && (CO == 64 || CO == 128)); }

// A-operand fragments of the closing 1x1's k step over y2 for C = 16: [cout fragment][lane][8], same cout permutation as
// pack_conv_weights (lane's row r -> cout (r >> 2) * 4 * NF + f * 4 + (r & 3)).  The B operand of that step is a lane's own four conv-2
// results followed by four zeros, so k slot 8 (lane >> 4) + e carries input channel c0 + 4 (lane >> 4) + e for e < 4 and nothing above.
std::vector<bf16_t> pack_bneck_k16(const float *w, int cout, int cin, int c0, bool f16) {
    const int NF = cout / 16;
    std::vector<bf16_t> out((size_t)NF * 64 * 8, 0);
    for (int f = 0; f < NF; ++f)
        for (int lane = 0; lane < 64; ++lane) {
            const int r = lane & 15, gq = lane >> 4, co = (r >> 2) * 4 * NF + f * 4 + (r & 3);
            for (int j = 0; j < 4; ++j) out[((size_t)f * 64 + lane) * 8 + j] = host_to_half(w[(size_t)co * cin + c0 + 4 * gq + j], f16);
        }
    return out;
}

template <int C, int W, int CO>
static hipError_t launch_t(const BneckLaunch &L, const BneckParams &P, dim3 grid, hipStream_t st) {
    constexpr int CH = C / 2, XW = W + 2, XP = C * 2 + 16, TP = CH == 8 ? 16 : CH * 2 + 16;
    constexpr int KST1 = (9 * (C / 8) + 3) / 4, KST2 = (9 * (CH / 8) + 3) / 4, NF2 = C / 16;
    size_t lds = (size_t)(kBnRows + 4) * XW * XP + (size_t)(kBnRows + 2) * XW * TP + (size_t)(KST1 + KST2 * NF2 + (CO == 128 ? CO / 16 : 0)) * 1024 + 1024;
    if (L.f16) hipLaunchKernelGGL((k_bneck_stripe<C, W, CO, true>), grid, dim3(256), lds, st, P);
    else hipLaunchKernelGGL((k_bneck_stripe<C, W, CO, false>), grid, dim3(256), lds, st, P);
    return hipGetLastError();
}

hipError_t launch_bneck(const BneckLaunch &L, hipStream_t st) {
    if (!bneck_supported(L.C, L.H, L.W)) return hipErrorInvalidValue;
    // both members must be whole blocks of a channel-blocked buffer with block size C: dense [image][pixel][C] planes
    if (L.y1.cpb * 8 != L.C || L.y2.cpb * 8 != L.C || L.y1.co % L.C || L.y2.co % L.C || L.y1.bs != L.y2.bs || L.y1.bs != (int64_t)L.H * L.W * L.C)
        return hipErrorInvalidValue;
    BneckParams P;
    P.y1 = (const bf16_t *)L.y1.p + (int64_t)(L.y1.co / L.C) * L.y1.ps;
    P.y2 = (bf16_t *)L.y2.p + (int64_t)(L.y2.co / L.C) * L.y2.ps;
    P.bs = L.y1.bs;
    P.w1pk = L.w1pk; P.w2pk = L.w2pk; P.bias1 = L.bias1; P.bias2 = L.bias2;
    P.H = L.H; P.stripes_y = L.H / kBnRows;
    P.y0 = nullptr; P.wc32 = P.wc16 = nullptr; P.biasc = nullptr; P.out = nullptr; P.out_bs = P.out_ps = 0; P.out_cs = 0; P.out_bsh = 31; P.out_bmask = 0x7fffffff;
    if (L.CO > 0) {
        if (!bneck_cv2_supported(L.C, L.CO) || !L.wc32pk || (L.C == 16 && !L.wc16pk) || !L.biasc || !L.out.p) return hipErrorInvalidValue;
        if (L.y0.cpb * 8 != L.C || L.y0.co % L.C || L.y0.bs != L.y1.bs) return hipErrorInvalidValue;
        P.y0 = (const bf16_t *)L.y0.p + (int64_t)(L.y0.co / L.C) * L.y0.ps;
        P.wc32 = L.wc32pk; P.wc16 = L.wc16pk; P.biasc = L.biasc;
        P.out = (bf16_t *)L.out.p + L.out.co; P.out_bs = L.out.bs; P.out_cs = L.out.cs;
        if (L.out.cpb > 0) {  // channel-blocked output: 8-channel chunk cc lives at (cc >> bsh) * ps + pixel * cs + (cc & bmask) * 8
            const int blk = 8 * L.out.cpb;
            if ((L.out.cpb & (L.out.cpb - 1)) || L.out.co % blk || L.out.cs != blk) return hipErrorInvalidValue;
            P.out_bsh = 0;
            while ((1 << P.out_bsh) < L.out.cpb) ++P.out_bsh;
            P.out_bmask = L.out.cpb - 1;
            P.out_ps = L.out.ps;
            P.out = (bf16_t *)L.out.p + (int64_t)(L.out.co / blk) * L.out.ps;
        } else if ((L.out.co | L.out.cs) & 7) return hipErrorInvalidValue;
    }
    int64_t ns = (int64_t)L.B * P.stripes_y;
    if (ns <= 0 || ns >= (1ll << 31)) return hipErrorInvalidValue;
    P.nstripes = (int)ns;
    static const int spw_max = getenv("OBB_BNECK_SPW") ? std::max(1, atoi(getenv("OBB_BNECK_SPW"))) : 4;
    int64_t spw = ns / (256 * 3 * 2);
    P.spw = (int)std::max<int64_t>(1, std::min<int64_t>(spw, spw_max));
    dim3 grid((unsigned)((ns + P.spw - 1) / P.spw));
    if (L.C == 16 && L.W == 104) return L.CO ? launch_t<16, 104, 64>(L, P, grid, st) : launch_t<16, 104, 0>(L, P, grid, st);
    if (L.C == 16) return L.CO ? launch_t<16, 32, 64>(L, P, grid, st) : launch_t<16, 32, 0>(L, P, grid, st);
    if (L.W == 52) {
        if (L.CO == 128) return launch_t<32, 52, 128>(L, P, grid, st);
        if (L.CO == 64) return launch_t<32, 52, 64>(L, P, grid, st);
        return launch_t<32, 52, 0>(L, P, grid, st);
    }
    if (L.CO == 128) return launch_t<32, 16, 128>(L, P, grid, st);
    if (L.CO == 64) return launch_t<32, 16, 64>(L, P, grid, st);
    return launch_t<32, 16, 0>(L, P, grid, st);
}

}  // namespace obb
