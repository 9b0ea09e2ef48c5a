// The class branch of the OBB head is a chain of  DWConv 3x3 (+SiLU) -> Conv 1x1 (+SiLU)  pairs (ultralytics Detect.cv3, SURVEY
// Appendix A): as separate launches each depthwise layer writes a tensor that the following 1x1 reads straight back, and the depthwise
// kernel alone is VALU-bound at a tenth of the 1x1's arithmetic.  Here one workgroup owns R output rows x the full width (row stripes as
// in bneck.hip): the R + 2 input rows it needs are contiguous runs, prefetched a stripe ahead into registers and committed to an LDS
// image with the zero padding built in.  For every 16-pixel fragment the lanes then compute the depthwise result in exactly the shape
// the matrix core wants as its B operand -- lane (g, pl) owns pixel pl and the 8 channels of k chunk 4 ks + g -- so the depthwise
// tensor never exists anywhere but in registers; the 1x1 runs on MFMA with its weights resident in LDS.  TAIL: the branch's last plain
// 1x1 (64 -> nc, no activation, fp32 rows of the head tensor) follows on the same fragment: a lane's 16 output channels are two k
// chunks of that GEMM (the weight packing is permuted to match), so that tensor stays in registers as well.
// Same arithmetic as the separate kernels: depthwise taps accumulated in fp32 in (ky, kx) order with mixed-precision FMAs, + bias, SiLU,
// one rounding to the 16-bit storage type; 1x1 with the k order of k_conv_igemm's 64-channel stages.
#include "dwpw.h"

#include <algorithm>
#include <cstdlib>

namespace obb {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct DwPwParams {
    const bf16_t *in; int64_t in_bs; int in_cs;
    bf16_t *out; int64_t out_bs; int out_cs;
    float *out2; int64_t out2_bs; int out2_cs, out2_co, cout2; float *sink;
    const bf16_t *dww, *pww, *tlw;
    const float *dwb, *pwb, *tlb;
    int H, stripes_y, nstripes, spw;
};

template <int CIN, int W, int R, bool TAIL, bool F16>
__global__ __launch_bounds__(256, 2) void k_dwpw_stripe(const DwPwParams P) {
    typedef typename HX<F16>::vec8 hx8;
    typedef typename HX<F16>::elem hel;
    constexpr int XR = R + 2, XW = W + 2, XP = CIN * 2 + 16, XB = XR * XW * XP;
    constexpr int KS = CIN / 32, NF = 4, CPK = CIN / 8;
    constexpr int NPX = R * W, FPW = ((NPX + 15) / 16 + 3) / 4;            // fragments per wave (surplus ones recompute the last pixel)
    constexpr int CPR = W * CPK, NCHUNK = XR * CPR, MAXPF = (NCHUNK + 255) / 256;  // 16-B input chunks per row / stripe / thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *X = smem, *PW = smem + XB, *DW = PW + KS * NF * 1024, *TL = DW + 9 * CIN * 2, *dummy = TL + (TAIL ? 2048 : 0);
    __shared__ __attribute__((aligned(16))) float s_dwb[CIN], s_pwb[64], s_tlb[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, pl = lane & 15;
    const int s0 = blockIdx.x * P.spw;
    const int s1 = min(s0 + P.spw, P.nstripes);
    if (s0 >= s1) return;

    // loop-invariant state: weights and biases in LDS, zero padding columns of the image (never overwritten)
    for (int i = tid; i < KS * NF * 64; i += 256) reinterpret_cast<u32x4 *>(PW)[i] = reinterpret_cast<const u32x4 *>(P.pww)[i];
    for (int i = tid; i < 9 * CIN / 8; i += 256) reinterpret_cast<u32x4 *>(DW)[i] = reinterpret_cast<const u32x4 *>(P.dww)[i];
    if constexpr (TAIL) {
        for (int i = tid; i < 2 * 64; i += 256) reinterpret_cast<u32x4 *>(TL)[i] = reinterpret_cast<const u32x4 *>(P.tlw)[i];
        if (tid < 16) s_tlb[tid] = P.tlb[tid];
    }
    for (int i = tid; i < CIN; i += 256) s_dwb[i] = P.dwb[i];
    if (tid < 64) s_pwb[tid] = P.pwb[tid];
    for (int i = tid; i < XR * 2 * (XP / 16); i += 256) {
        int r = i / (2 * (XP / 16)), rem = i - r * (2 * (XP / 16)), side = rem / (XP / 16), c = rem - side * (XP / 16);
        *reinterpret_cast<u32x4 *>(X + (r * XW + (side ? W + 1 : 0)) * XP + c * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    // depthwise weights of this lane's chunks in registers when they fit (64 input channels: 2 k steps x 9 taps x 16 B): the stripe
    // loop's barriers would otherwise force an LDS re-read per fragment
    constexpr bool DWREG = false;  // (measured: 72 weight registers push the 52-wide variants into spills)
    u32x4 dwr[DWREG ? KS : 1][DWREG ? 9 : 1];
    if constexpr (DWREG) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int t = 0; t < 9; ++t) dwr[ks][t] = *reinterpret_cast<const u32x4 *>(P.dww + t * CIN + (ks * 4 + g) * 8);
    }
    __builtin_amdgcn_s_waitcnt((0 & 15) | (7 << 4) | (15 << 8));  // vmcnt(0): settle the weight / bias loads outside the stripe loop

    // input chunks of this thread: idx = tid + k*256 -> (row, chunk in row); surplus slots re-read the last chunk into a dummy LDS slot
    int src_row[MAXPF], src_off[MAXPF], lds_off[MAXPF];
#pragma unroll
    for (int k = 0; k < MAXPF; ++k) {
        int idx = tid + k * 256;
        bool real = idx < NCHUNK;
        idx = real ? idx : NCHUNK - 1;
        int r = idx / CPR, j = idx - r * CPR, x = j / CPK, c = j - x * CPK;
        src_row[k] = r; src_off[k] = x * P.in_cs + c * 8;
        lds_off[k] = real ? (r * XW + 1 + x) * XP + c * 16 : (int)(dummy - X) + (tid & 63) * 16;
    }
    u32x4 pre[MAXPF];
    unsigned pre_ok = 0;
    auto issue = [&](int s) {
        const int b = s / P.stripes_y, oy0 = (s - b * P.stripes_y) * R;
        const bf16_t *src = P.in + (int64_t)b * P.in_bs;
        pre_ok = 0;
#pragma unroll
        for (int k = 0; k < MAXPF; ++k) {
            int gy = oy0 - 1 + src_row[k];
            if (gy >= 0 && gy < P.H) pre_ok |= 1u << k;
            gy = min(max(gy, 0), P.H - 1);
            pre[k] = *reinterpret_cast<const u32x4 *>(src + ((int64_t)gy * W) * P.in_cs + src_off[k]);
        }
    };
    issue(s0);
    __builtin_amdgcn_s_waitcnt((0 & 15) | (7 << 4) | (15 << 8));
    for (int s = s0; s < s1; ++s) {
        const int b = s / P.stripes_y, oy0 = (s - b * P.stripes_y) * R;
        __syncthreads();  // the previous stripe is done with the LDS image
#pragma unroll
        for (int k = 0; k < MAXPF; ++k) *reinterpret_cast<u32x4 *>(X + lds_off[k]) = ((pre_ok >> k) & 1u) ? pre[k] : u32x4{0u, 0u, 0u, 0u};
        __syncthreads();
        issue(min(s + 1, s1 - 1));  // unconditional (the last stripe re-reads its own rows): keeps the outstanding-op count exact

#pragma unroll
        for (int i = 0; i < FPW; ++i) {
            int p = (wave + 4 * i) * 16 + pl;
            p = p < NPX ? p : NPX - 1;
            const int r = p / W, x = p - r * W;
            const char *xb = X + (r * XW + x) * XP;  // top-left of the pixel's 3x3 window (image row oy0 - 1 + r, column x - 1)
            f32x4 acc[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int q = ks * 4 + g;  // this lane's 8-channel chunk of the k step
                float a8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) a8[j] = 0.f;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const u32x4 xv = *reinterpret_cast<const u32x4 *>(xb + (ky * XW + kx) * XP + q * 16);
                        u32x4 wv;
                        if constexpr (DWREG) wv = dwr[ks][ky * 3 + kx];
                        else wv = *reinterpret_cast<const u32x4 *>(DW + ((ky * 3 + kx) * CIN + q * 8) * 2);
                        fma8_mixed<F16>(a8, uint4{xv.x, xv.y, xv.z, xv.w}, uint4{wv.x, wv.y, wv.z, wv.w});
                    }
                const float4 b0 = *reinterpret_cast<const float4 *>(s_dwb + q * 8), b1 = *reinterpret_cast<const float4 *>(s_dwb + q * 8 + 4);
                u32x4 du;
                du.x = HX<F16>::pack2(silu_f(a8[0] + b0.x), silu_f(a8[1] + b0.y));
                du.y = HX<F16>::pack2(silu_f(a8[2] + b0.z), silu_f(a8[3] + b0.w));
                du.z = HX<F16>::pack2(silu_f(a8[4] + b1.x), silu_f(a8[5] + b1.y));
                du.w = HX<F16>::pack2(silu_f(a8[6] + b1.z), silu_f(a8[7] + b1.w));
                const hx8 bop = __builtin_bit_cast(hx8, du);
#pragma unroll
                for (int f = 0; f < NF; ++f) acc[f] = HX<F16>::mfma(*reinterpret_cast<const hx8 *>(PW + ((ks * NF + f) * 64 + lane) * 16), bop, acc[f]);
            }
            // lane owns couts g * 16 + f * 4 + j of pixel p: + bias, SiLU, 16 bit
            u32x4 y[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 ba = *reinterpret_cast<const float4 *>(s_pwb + g * 16 + h * 8), bb = *reinterpret_cast<const float4 *>(s_pwb + g * 16 + h * 8 + 4);
                y[h].x = HX<F16>::pack2(silu_f(acc[2 * h][0] + ba.x), silu_f(acc[2 * h][1] + ba.y));
                y[h].y = HX<F16>::pack2(silu_f(acc[2 * h][2] + ba.z), silu_f(acc[2 * h][3] + ba.w));
                y[h].z = HX<F16>::pack2(silu_f(acc[2 * h + 1][0] + bb.x), silu_f(acc[2 * h + 1][1] + bb.y));
                y[h].w = HX<F16>::pack2(silu_f(acc[2 * h + 1][2] + bb.z), silu_f(acc[2 * h + 1][3] + bb.w));
            }
            const int64_t pix = (int64_t)(oy0 + r) * W + x;
            if constexpr (!TAIL) {
                bf16_t *op = P.out + (int64_t)b * P.out_bs + pix * P.out_cs + g * 16;
                *reinterpret_cast<u32x4 *>(op) = y[0];
                *reinterpret_cast<u32x4 *>(op + 8) = y[1];
            } else {
                // trailing plain 1x1 (64 -> cout2): k step s takes this lane's channels 16 g + 8 s .. + 7 (weights packed to match)
                f32x4 acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
                acc2 = HX<F16>::mfma(*reinterpret_cast<const hx8 *>(TL + lane * 16), __builtin_bit_cast(hx8, y[0]), acc2);
                acc2 = HX<F16>::mfma(*reinterpret_cast<const hx8 *>(TL + (64 + lane) * 16), __builtin_bit_cast(hx8, y[1]), acc2);
                const float4 tb = *reinterpret_cast<const float4 *>(s_tlb + g * 4);
                const float v0 = acc2[0] + tb.x, v1 = acc2[1] + tb.y, v2 = acc2[2] + tb.z, v3 = acc2[3] + tb.w;
                // one float4 per lane on every path (the host guarantees 16-B aligned rows and cout2 % 4 == 0): lanes beyond the last cout store
                // into the sink, so the compiler can count the stores and the wait for the next stripe's prefetch leaves them in flight
                float *op = P.out2 + (int64_t)b * P.out2_bs + pix * P.out2_cs + P.out2_co + g * 4;
                if (g * 4 + 4 > P.cout2) op = P.sink + lane * 4;
                *reinterpret_cast<float4 *>(op) = make_float4(v0, v1, v2, v3);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side

static int dwpw_rows(int W) {
    static const int r52 = getenv("OBB_DWPW_R52") ? atoi(getenv("OBB_DWPW_R52")) : 2;  // measured: 2-row stripes (41 KB of LDS, three groups per CU) 99.6 k vs 4-row 98.9 k tiles/s
    if (W == 16) return 4;  // (128-px tiles: 64 pixels = one fragment per wave)
    if (W == 8) return 8;   // (the whole 8 x 8 map)
    return W == 52 ? (r52 == 4 ? 4 : 2) : 2;
}

bool dwpw_supported(int cin, int cout, int H, int W, int tail_cout) {
    if (cout != 64 || tail_cout < 0 || tail_cout > 16) return false;
    const bool shape = (cin == 64 && (W == 52 || W == 16)) || ((cin == 64 || cin == 128) && (W == 26 || W == 8));  // 416-px levels; 128-px levels (Detect_OBB.py:24-28)
    if (!shape || H <= 0 || H % dwpw_rows(W)) return false;
    return tail_cout == 0 || (cin == 64 && tail_cout % 4 == 0);  // the branch's second pair (64 -> 64 -> nc) carries the tail (whole float4s per lane)
}

std::vector<bf16_t> pack_dwpw_tail(const float *w, int cout2, bool f16) {
    std::vector<bf16_t> out((size_t)2 * 64 * 8, 0);
    for (int s = 0; s < 2; ++s)
        for (int lane = 0; lane < 64; ++lane) {
            const int r = lane & 15, gq = lane >> 4;
            if (r >= cout2) continue;
            for (int e = 0; e < 8; ++e) out[((size_t)s * 64 + lane) * 8 + e] = host_to_half(w[(size_t)r * 64 + 16 * gq + 8 * s + e], f16);
        }
    return out;
}

template <int CIN, int W, int R, bool TAIL>
static hipError_t launch_t(const DwPwLaunch &L, const DwPwParams &P, dim3 grid, hipStream_t st) {
    constexpr size_t lds = (size_t)(R + 2) * (W + 2) * (CIN * 2 + 16) + (size_t)(CIN / 32) * 4 * 1024 + 9 * CIN * 2 + (TAIL ? 2048 : 0) + 1024;
    static_assert(lds <= 64 * 1024, "dynamic LDS above 64 KB needs the function attribute");
    if (L.f16) hipLaunchKernelGGL((k_dwpw_stripe<CIN, W, R, TAIL, true>), grid, dim3(256), lds, st, P);
    else hipLaunchKernelGGL((k_dwpw_stripe<CIN, W, R, TAIL, false>), grid, dim3(256), lds, st, P);
    return hipGetLastError();
}

hipError_t launch_dwpw(const DwPwLaunch &L, hipStream_t st) {
    if (!dwpw_supported(L.cin, 64, L.H, L.W, L.tail_cout)) return hipErrorInvalidValue;
    if (L.in.cpb || !L.in.p || (L.in.cs | L.in.co) & 7 || !L.dw_w || !L.dw_b || !L.pw_w || !L.pw_b) return hipErrorInvalidValue;
    DwPwParams P;
    P.in = (const bf16_t *)L.in.p + L.in.co; P.in_bs = L.in.bs; P.in_cs = L.in.cs;
    P.out = nullptr; P.out_bs = 0; P.out_cs = 0;
    P.out2 = nullptr; P.out2_bs = 0; P.out2_cs = 0; P.out2_co = 0; P.cout2 = 0; P.sink = nullptr;
    P.dww = L.dw_w; P.dwb = L.dw_b; P.pww = L.pw_w; P.pwb = L.pw_b; P.tlw = L.tail_w; P.tlb = L.tail_b;
    if (L.tail_cout > 0) {
        if (!L.tail_out.p || !L.tail_w || !L.tail_b) return hipErrorInvalidValue;
        P.out2 = (float *)L.tail_out.p; P.out2_bs = L.tail_out.bs; P.out2_cs = L.tail_out.cs; P.out2_co = L.tail_out.co; P.cout2 = L.tail_cout;
        P.sink = (float *)L.sink;
        if (!L.sink || (L.tail_cout & 3) || ((L.tail_out.cs | L.tail_out.co) & 3)) return hipErrorInvalidValue;  // float4 rows, sink for the surplus lanes
    } else {
        if (L.out.cpb || !L.out.p || (L.out.cs | L.out.co) & 7) return hipErrorInvalidValue;
        P.out = (bf16_t *)L.out.p + L.out.co; P.out_bs = L.out.bs; P.out_cs = L.out.cs;
    }
    const int R = dwpw_rows(L.W);
    P.H = L.H; P.stripes_y = L.H / R;
    int64_t ns = (int64_t)L.B * P.stripes_y;
    if (ns <= 0 || ns >= (1ll << 31)) return hipErrorInvalidValue;
    P.nstripes = (int)ns;
    static const int spw_max = getenv("OBB_DWPW_SPW") ? std::max(1, atoi(getenv("OBB_DWPW_SPW"))) : 4;
    int64_t spw = ns / (256 * 3 * 2);
    P.spw = (int)std::max<int64_t>(1, std::min<int64_t>(spw, spw_max));
    dim3 grid((unsigned)((ns + P.spw - 1) / P.spw));
    if (L.W == 16) return L.tail_cout ? launch_t<64, 16, 4, true>(L, P, grid, st) : launch_t<64, 16, 4, false>(L, P, grid, st);
    if (L.W == 8 && L.cin == 128) return launch_t<128, 8, 8, false>(L, P, grid, st);
    if (L.W == 8) return L.tail_cout ? launch_t<64, 8, 8, true>(L, P, grid, st) : launch_t<64, 8, 8, false>(L, P, grid, st);
    if (L.W == 52 && R == 2) return L.tail_cout ? launch_t<64, 52, 2, true>(L, P, grid, st) : launch_t<64, 52, 2, false>(L, P, grid, st);
    if (L.W == 52) return L.tail_cout ? launch_t<64, 52, 4, true>(L, P, grid, st) : launch_t<64, 52, 4, false>(L, P, grid, st);
    if (L.cin == 128) return launch_t<128, 26, 2, false>(L, P, grid, st);
    return L.tail_cout ? launch_t<64, 26, 2, true>(L, P, grid, st) : launch_t<64, 26, 2, false>(L, P, grid, st);
}

}  // namespace obb
