// fp32-arithmetic OBBModel forward for gfx950 (obb_set_option "precision" = 32; Detect_OBB.py:79-83 calls the model with Ultralytics'
// default half=False, i.e. fp32 arithmetic: SURVEY.md section 8 row a5, section 7.2 item 5).
//
// Convolution = implicit GEMM on v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: bit-for-bit a k-ordered fmaf chain, 64 FLOP/clk/SIMD
// = 157 TFLOP/s on the chip, 1/16 of the fp16 rate -- the kernel is MFMA-bound by a wide margin, so everything else is kept simple):
//   * D^T[cout][pixel] = W[cout][k] * X[k][pixel]: A operand = weights (lane: cout = lane & 15, k slot = lane >> 4), B operand =
//     activations (lane: pixel = lane & 15, k slot = lane >> 4), D: pixel = lane & 15, couts 4 * (lane >> 4) .. + 3 (one float4 store).
//   * k runs over (tap, channel) in 4-channel chunks; a lane reads ONE 16-byte chunk of its pixel (ds_read_b128) and feeds element s to
//     MFMA step s, so four MFMAs consume 16 k values per lane-group set: the k order inside a step is (chunk g, element s), identical on
//     the weight side (pack_conv32_weights).
//   * workgroup = 4 waves = an output tile of up to 208 pixels (13 fragments: R full rows of a 208 / 104 / 52 / 26-wide level, or a run of
//     the flattened batch for 1x1 layers) x 16 * WC couts; WC waves split the couts, 4 / WC waves split the pixel fragments.
//   * the input tile (+ halo, zero padding written as zeros) is staged in LDS one channel stage at a time; weights stream from L2 in
//     fragment order (1 KiB per wave-instruction), prefetched one k step ahead.  No software pipelining of the staging: at 1/16 of the
//     fp16 MFMA rate a stage computes for 6-15 us, and two or three resident workgroups per CU cover each other's staging phases.
//   * measured and dropped: prefetching the next channel stage into registers under the MFMAs (176 VGPRs -> two waves per SIMD
//     instead of three: 40.1 -> 42.1 ms per 1024 tiles), and dword loads of 4-pixel groups + an LDS table for the uint8 input layer
//     (596 -> 771 us per 256 tiles).
//   * epilogue: + bias, SiLU (expf + IEEE division, like torch), + residual, fp32 store into a channel slice of the consumer's buffer.
#include "f32path.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace obb {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct C32Params {
    const void *in; int64_t in_bs; int in_cs, in_co;
    float *out; int64_t out_bs; int out_cs, out_co;
    const float *res; int64_t res_bs; int res_cs, res_co;
    const float *wpk, *bias, *lut;
    int Hin, Win, Hout, Wout, cin, cout, stride, act, flip_bgr;
    int TH, TW, CK, sh /*log2(CK/4)*/, tiles_x, tiles_y, ntiles, nstage, kst, out_hw, ncb;
    float inv_twin, inv_tw;
    unsigned in_span_bytes;  // buffer-descriptor range of one image's input slice (its check returns zeros past the end)
};

// SiLU with the hardware exp2 / reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each): relative error <= ~1e-6 for |x| <= 10, below the
// summation-order noise of an fp32 convolution (~sqrt(K) * 6e-8); the library expf + IEEE division cost 4x the epilogue time
__device__ __forceinline__ float silu32(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

template <int KS, int MFM, int WC, bool IN_U8>
__global__ __launch_bounds__(256, 3) void k_conv_f32(const C32Params P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PAD = KS / 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: the fragment guards below become scalar branches
    const int g = lane >> 4, pl = lane & 15;
    const int wc = wave % WC, wp = wave / WC;
    // XCD-aware order (as k_conv_igemm): the cout blocks of one pixel tile are consecutive on one XCD and share the tile through its L2
    const int xcd = blockIdx.x & 7, lin = blockIdx.x >> 3;
    const int cb = lin % P.ncb;
    const int t = (lin / P.ncb) * 8 + xcd;
    if (t >= P.ntiles) return;
    const int S = P.stride;
    const int THin = (P.TH - 1) * S + KS, TWin = (P.TW - 1) * S + KS;
    const int PST = P.CK * 4 + 16;  // bytes per staged pixel (+16 B spreads consecutive pixels over the banks)
    const int cpk = P.CK >> 2;
    const int nq = (KS == 3 ? 9 : 1) * cpk;
    const int in_px = THin * TWin;
    const int tx_i = t % P.tiles_x, r_ = t / P.tiles_x, ty_i = r_ % P.tiles_y, b = r_ / P.tiles_y;
    const int oy0 = ty_i * P.TH, ox0 = tx_i * P.TW;
    const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;
    const int npix = P.TH * P.TW;

    int pixbase[MFM];
#pragma unroll
    for (int mf = 0; mf < MFM; ++mf) {
        const int p = (wp * MFM + mf) * 16 + pl;
        const int ty = (int)(((float)p + 0.5f) * P.inv_tw);
        const int tx = p - ty * P.TW;
        pixbase[mf] = p < npix ? ((ty * S) * TWin + tx * S) * PST : 0;  // fragments past the tile compute on pixel 0 and are dropped
    }
    const int F = cb * WC + wc;  // cout fragment of this wave
    const float *wbase = P.wpk + (size_t)F * P.nstage * P.kst * 256 + lane * 4;

    f32x4 acc[MFM];
#pragma unroll
    for (int mf = 0; mf < MFM; ++mf) acc[mf] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging plan: this thread moves the 16-B chunks idx = tid + k * 256 of the [in_px][CK] tile.  Activations come through buffer
    // loads (descriptor in SGPRs, one 32-bit byte offset per chunk): an offset past the descriptor's range reads zeros, which is how
    // the zero padding is written
    constexpr int MAXLD = IN_U8 ? 5 : (KS == 1 ? 8 : 9);  // plan_conv32 keeps a stage within that many x 256 chunks of 16 B
    constexpr unsigned NOPIX = 0xffffffffu;
    const int nchunk = in_px << P.sh;
    unsigned goff[IN_U8 ? 1 : MAXLD];
    __amdgpu_buffer_rsrc_t in_rsrc;
    if constexpr (!IN_U8) {
        in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((const float *)P.in + (int64_t)b * P.in_bs + P.in_co), 0, (int)P.in_span_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < MAXLD; ++k) {
            const int idx = tid + k * 256;
            const int pix = idx >> P.sh, c = idx & (cpk - 1);
            const int iy = (int)(((float)pix + 0.5f) * P.inv_twin), ix = pix - iy * TWin;
            const int gy = iy0 + iy, gx = ix0 + ix;
            const bool ok = idx < nchunk && gy >= 0 && gy < P.Hin && gx >= 0 && gx < P.Win;
            goff[k] = ok ? (unsigned)((((int64_t)gy * P.Win + gx) * P.in_cs + c * 4) * 4) : NOPIX;
        }
    }

    for (int stage = 0; stage < P.nstage; ++stage) {
        __syncthreads();  // every wave is done reading the previous stage
        // every load of the stage is issued before the first LDS store (one memory latency per stage, not one per chunk)
        if constexpr (IN_U8) {
            float4 pre[MAXLD];
            const uint8_t *src = (const uint8_t *)P.in + (int64_t)b * P.in_bs;
#pragma unroll
            for (int k = 0; k < MAXLD; ++k) {
                const int pix = tid + k * 256;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pix < in_px) {
                    const int iy = (int)(((float)pix + 0.5f) * P.inv_twin), ix = pix - iy * TWin;
                    const int gy = iy0 + iy, gx = ix0 + ix;
                    if (gy >= 0 && gy < P.Hin && gx >= 0 && gx < P.Win) {
                        const uint8_t *sp = src + ((int64_t)gy * P.Win + gx) * P.in_cs;
                        const float c0 = P.lut[sp[0]], c1 = P.lut[sp[1]], c2 = P.lut[sp[2]];
                        v.x = P.flip_bgr ? c2 : c0; v.y = c1; v.z = P.flip_bgr ? c0 : c2;
                        if (P.cin == 4) v.w = P.lut[sp[3]];
                    }
                }
                pre[k] = v;
            }
#pragma unroll
            for (int k = 0; k < MAXLD; ++k) {
                const int pix = tid + k * 256;
                if (pix < in_px) *reinterpret_cast<float4 *>(smem + pix * PST) = pre[k];
            }
        } else {
            u32x4 pre[MAXLD];
#pragma unroll
            for (int k = 0; k < MAXLD; ++k)
                pre[k] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, goff[k] == NOPIX ? NOPIX : goff[k] + (unsigned)(stage * P.CK * 4), 0, 0);
#pragma unroll
            for (int k = 0; k < MAXLD; ++k) {
                const int idx = tid + k * 256;
                if (idx < nchunk) *reinterpret_cast<u32x4 *>(smem + (idx >> P.sh) * PST + (idx & (cpk - 1)) * 16) = pre[k];
            }
        }
        __syncthreads();
        const float *wst = wbase + (size_t)stage * P.kst * 256;
        f32x4 wnext = *reinterpret_cast<const f32x4 *>(wst);
        for (int ks = 0; ks < P.kst; ++ks) {
            const f32x4 w = wnext;
            wnext = *reinterpret_cast<const f32x4 *>(wst + (size_t)min(ks + 1, P.kst - 1) * 256);
            int q = ks * 4 + g;
            q = q < nq ? q : nq - 1;  // padding chunks: any valid address, their weights are zero
            int off;
            if constexpr (KS == 3) {
                const int tap = q >> P.sh, c0 = q & (cpk - 1);
                const int dy = (tap * 11) >> 5, dx = tap - dy * 3;
                off = (dy * TWin + dx) * PST + c0 * 16;
            } else {
                off = q * 16;
            }
            // the fragments go through in two halves: 4 * H MFMAs (>= 700 cycles) cover the other half's LDS reads, and only H operand
            // vectors are live at a time
            constexpr int H = MFM > 7 ? (MFM + 1) / 2 : MFM;
#pragma unroll
            for (int m0 = 0; m0 < MFM; m0 += H) {
                f32x4 a[H];
#pragma unroll
                for (int i = 0; i < H; ++i)
                    if (m0 + i < MFM) a[i] = *reinterpret_cast<const f32x4 *>(smem + pixbase[m0 + i] + off);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < H; ++i)
                        if (m0 + i < MFM) acc[m0 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], a[i][s], acc[m0 + i], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: lane owns couts [cbase, cbase + 4) of its pixels
    const int cbase = F * 16 + g * 4;
    if (cbase >= P.cout) return;
    const float4 bv = *reinterpret_cast<const float4 *>(P.bias + cbase);  // bias is padded: always readable
    const bool full = cbase + 4 <= P.cout;
#pragma unroll
    for (int mf = 0; mf < MFM; ++mf) {
        const int p = (wp * MFM + mf) * 16 + pl;
        if (p >= npix) continue;
        const int ty = (int)(((float)p + 0.5f) * P.inv_tw), tx = p - ty * P.TW;
        const int oy = oy0 + ty, ox = ox0 + tx;
        if (oy >= P.Hout || ox >= P.Wout) continue;
        const int64_t opix = (int64_t)oy * P.Wout + ox;
        float v[4] = {acc[mf][0] + bv.x, acc[mf][1] + bv.y, acc[mf][2] + bv.z, acc[mf][3] + bv.w};
        if (P.act) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = silu32(v[j]);
        }
        if (P.res) {
            const float *rp = P.res + (int64_t)b * P.res_bs + opix * P.res_cs + P.res_co + cbase;
            if (full) {
                const float4 rv = *reinterpret_cast<const float4 *>(rp);
                v[0] = rv.x + v[0]; v[1] = rv.y + v[1]; v[2] = rv.z + v[2]; v[3] = rv.w + v[3];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (cbase + j < P.cout) v[j] = rp[j] + v[j];
            }
        }
        int64_t ob = b, opx = opix;
        if (P.out_hw > 0) { ob = opx / P.out_hw; opx -= ob * P.out_hw; }
        float *op = P.out + ob * P.out_bs + opx * P.out_cs + P.out_co + cbase;
        if (full && ((P.out_cs | P.out_co) & 3) == 0) {
            *reinterpret_cast<float4 *>(op) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (cbase + j < P.cout) op[j] = v[j];
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
static int ilog2_(int v) { int s = 0; while ((1 << s) < v) ++s; return s; }
static int c32_ksteps(int ks, int CK) { return ((ks == 3 ? 9 : 1) * (CK / 4) + 3) / 4; }
static int c32_mfm(int WC) { return WC == 4 ? 13 : (WC == 2 ? 7 : 4); }

Conv32Tiling plan_conv32(int ks, int stride, int cin, int cout, int Hout, int Wout, bool in_u8) {
    Conv32Tiling t;
    t.WC = cout >= 64 ? 4 : (cout >= 32 ? 2 : 1);
    if (in_u8) t.CK = 4;
    else {
        const int cap = ks == 1 ? 32 : (stride == 2 ? 8 : 16);
        int ck = 4;
        while (ck * 2 <= cap && cin % (ck * 2) == 0) ck *= 2;
        t.CK = ck;
    }
    const int WP = 4 / t.WC, MFM = c32_mfm(t.WC);
    const int maxpix = 16 * WP * MFM;  // 208 / 224 / 256 pixels for WC = 4 / 2 / 1
    if (ks == 1) {  // 1-D: the caller flattens batch x pixels
        while (t.CK > 4 && (maxpix * (t.CK / 4) > 9 * 256 || (int64_t)maxpix * (t.CK * 4 + 16) > 64 * 1024)) t.CK /= 2;
        t.TH = 1; t.TW = maxpix;
        return t;
    }
    const int PST = t.CK * 4 + 16;
    t.TW = std::min(Wout, maxpix);
    const int tiles_x = (Wout + t.TW - 1) / t.TW;
    int best_th = 1;
    int64_t best_cost = -1;
    for (int th = 1; th <= std::min(Hout, maxpix / t.TW); ++th) {
        const int64_t in_px = (int64_t)((th - 1) * stride + ks) * ((t.TW - 1) * stride + ks);
        if (in_px * PST > 64 * 1024 || in_px * (t.CK / 4) > 9 * 256) break;
        // every wave computes its full set of fragments whatever the tile holds: MFMA time ~ tiles x fragments per wave
        const int mfm = (t.WC == 4 && th * t.TW <= 11 * 16) ? 11 : MFM;
        const int64_t cost = (int64_t)((Hout + th - 1) / th) * tiles_x * mfm;
        if (best_cost < 0 || cost <= best_cost) { best_cost = cost; best_th = th; }
    }
    t.TH = best_th;
    return t;
}

std::vector<float> pack_conv32_weights(const float *w, int cout, int cin, int ks, const Conv32Tiling &t, const int *perm, bool in_u8) {
    const int CK = t.CK, cpk = CK / 4;
    const int cin_eff = in_u8 ? 4 : cin;
    const int nstage = (cin_eff + CK - 1) / CK;
    const int kst = c32_ksteps(ks, CK);
    const int nf = (cout + 15) / 16;
    const int nfp = (nf + t.WC - 1) / t.WC * t.WC;  // whole cout blocks
    const int taps = ks * ks, nq = taps * cpk;
    std::vector<float> out((size_t)nfp * nstage * kst * 256, 0.f);
    size_t o = 0;
    for (int F = 0; F < nfp; ++F)
        for (int st = 0; st < nstage; ++st)
            for (int k = 0; k < kst; ++k)
                for (int lane = 0; lane < 64; ++lane) {
                    const int r = lane & 15, gq = lane >> 4;
                    const int co = F * 16 + r;
                    const int q = k * 4 + gq;
                    for (int s = 0; s < 4; ++s, ++o) {
                        if (q >= nq || co >= cout) continue;
                        const int tap = q / cpk, c = st * CK + (q % cpk) * 4 + s;
                        if (c >= cin) continue;
                        const int src = perm ? perm[co] : co;
                        out[o] = w[((size_t)src * cin + c) * taps + tap];
                    }
                }
    return out;
}

size_t conv32_lds_bytes(const Conv32Launch &L) {
    const int THin = (L.TH - 1) * L.stride + L.ks, TWin = (L.TW - 1) * L.stride + L.ks;
    return (size_t)THin * TWin * (L.CK * 4 + 16);
}

template <int KS, int MFM, int WC>
static hipError_t launch32_t(const Conv32Launch &L, const C32Params &P, dim3 grid, size_t lds, hipStream_t st) {
    if (L.in_u8) {
        if constexpr (KS == 3 && WC == 1) hipLaunchKernelGGL((k_conv_f32<KS, MFM, WC, true>), grid, dim3(256), lds, st, P);
        else return hipErrorInvalidValue;
    } else hipLaunchKernelGGL((k_conv_f32<KS, MFM, WC, false>), grid, dim3(256), lds, st, P);
    return hipGetLastError();
}

template <int KS>
static hipError_t launch32_wc(const Conv32Launch &L, const C32Params &P, dim3 grid, size_t lds, hipStream_t st) {
    switch (L.WC) {
        case 4:
            if (L.TH * L.TW <= 11 * 16) return launch32_t<KS, 11, 4>(L, P, grid, lds, st);  // a whole 13 x 13 level: 169 pixels
            return launch32_t<KS, 13, 4>(L, P, grid, lds, st);
        case 2: return launch32_t<KS, 7, 2>(L, P, grid, lds, st);
        case 1: return launch32_t<KS, 4, 1>(L, P, grid, lds, st);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_conv32(const Conv32Launch &L, hipStream_t st) {
    if (L.in.cpb || L.out.cpb || L.res.cpb) return hipErrorInvalidValue;  // plain NHWC only
    if ((L.ks != 1 && L.ks != 3) || (L.WC != 1 && L.WC != 2 && L.WC != 4)) return hipErrorInvalidValue;
    C32Params P;
    P.in = L.in.p; P.in_bs = L.in.bs; P.in_cs = L.in.cs; P.in_co = L.in.co;
    P.out = (float *)L.out.p; P.out_bs = L.out.bs; P.out_cs = L.out.cs; P.out_co = L.out.co;
    P.res = (const float *)L.res.p; P.res_bs = L.res.bs; P.res_cs = L.res.cs; P.res_co = L.res.co;
    P.wpk = L.wpk; P.bias = L.bias; P.lut = L.lut;
    P.Hin = L.Hin; P.Win = L.Win; P.Hout = L.Hout; P.Wout = L.Wout; P.cin = L.cin; P.cout = L.cout; P.stride = L.stride; P.act = L.act;
    P.flip_bgr = L.flip_bgr;
    P.TH = L.TH; P.TW = L.TW; P.CK = L.CK; P.sh = ilog2_(L.CK / 4);
    if ((4 << P.sh) != L.CK) return hipErrorInvalidValue;
    const int WP = 4 / L.WC;
    if (L.TH * L.TW > 16 * WP * c32_mfm(L.WC) || L.TH < 1 || L.TW < 1) return hipErrorInvalidValue;
    const int cin_eff = L.in_u8 ? 4 : L.cin;
    if (!L.in_u8 && (L.cin % L.CK || (L.in.cs & 3) || (L.in.co & 3))) return hipErrorInvalidValue;
    if (L.in_u8 && (L.CK != 4 || (L.cin != 3 && L.cin != 4) || !L.lut)) return hipErrorInvalidValue;
    if (L.res.p && ((L.res.cs | L.res.co) & 3)) return hipErrorInvalidValue;
    P.nstage = (cin_eff + L.CK - 1) / L.CK;
    P.kst = c32_ksteps(L.ks, L.CK);
    P.tiles_x = L.tiles_x; P.tiles_y = L.tiles_y; P.out_hw = L.out_hw;
    const int64_t ntiles = (int64_t)L.B * L.tiles_y * L.tiles_x;
    P.ncb = (L.cout + 16 * L.WC - 1) / (16 * L.WC);
    if (ntiles < 1 || (ntiles + 7) / 8 * 8 * P.ncb >= (1ll << 31)) return hipErrorInvalidValue;
    P.ntiles = (int)ntiles;
    const int TWin = (L.TW - 1) * L.stride + L.ks;
    P.inv_twin = 1.0f / (float)TWin;
    P.inv_tw = 1.0f / (float)L.TW;
    {
        const int64_t span = ((int64_t)L.Hin * L.Win * L.in.cs - L.in.co) * 4;  // from the slice's first element to the end of the image
        if (!L.in_u8 && (span <= 0 || span >= (1ll << 32) - 65536)) return hipErrorInvalidValue;  // 32-bit buffer offsets
        P.in_span_bytes = L.in_u8 ? 0u : (unsigned)span;
        if (L.in_u8 && (int64_t)((L.TH - 1) * L.stride + L.ks) * TWin > 5 * 256) return hipErrorInvalidValue;
    }
    const size_t lds = conv32_lds_bytes(L);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    {
        const int THin = (L.TH - 1) * L.stride + L.ks;
        if ((int64_t)THin * TWin * (L.CK / 4) > 9 * 256) return hipErrorInvalidValue;  // staging plan: chunks per thread
    }
    dim3 grid((unsigned)((ntiles + 7) / 8 * 8 * P.ncb));
    return L.ks == 3 ? launch32_wc<3>(L, P, grid, lds, st) : launch32_wc<1>(L, P, grid, lds, st);
}

// ------------------------------------------------------------------------------------------------ the non-GEMM layers in fp32
// depthwise 3x3, stride 1, pad 1 (+bias, SiLU, +residual); w: fp32 [9][C]; a thread owns one pixel x 4 channels.  Taps are summed in
// (ky, kx) order, zero padding contributes exact zeros.
__global__ __launch_bounds__(256) void k_dwconv3_f32(TensorRef in, TensorRef out, TensorRef res, const float *__restrict__ w, const float *__restrict__ bias,
                                                    int B, int H, int W, int C, int act) {
    const int c4n = C >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)B * H * W * c4n) return;
    const int c4 = (int)(idx % c4n);
    const int64_t pix = idx / c4n;
    const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
    const float *ip = (const float *)in.p + (int64_t)b * in.bs + in.co + c4 * 4;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int yy = y + ky - 1, xx = x + kx - 1;
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            const float4 v = *reinterpret_cast<const float4 *>(ip + ((int64_t)yy * W + xx) * in.cs);
            const float4 wv = *reinterpret_cast<const float4 *>(w + (ky * 3 + kx) * C + c4 * 4);
            a[0] = fmaf(v.x, wv.x, a[0]); a[1] = fmaf(v.y, wv.y, a[1]); a[2] = fmaf(v.z, wv.z, a[2]); a[3] = fmaf(v.w, wv.w, a[3]);
        }
    const float4 bv = *reinterpret_cast<const float4 *>(bias + c4 * 4);
    float v[4] = {a[0] + bv.x, a[1] + bv.y, a[2] + bv.z, a[3] + bv.w};
    if (act) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = silu32(v[j]);
    }
    const int64_t opix = (int64_t)y * W + x;
    if (res.p) {
        const float4 rv = *reinterpret_cast<const float4 *>((const float *)res.p + (int64_t)b * res.bs + opix * res.cs + res.co + c4 * 4);
        v[0] = rv.x + v[0]; v[1] = rv.y + v[1]; v[2] = rv.z + v[2]; v[3] = rv.w + v[3];
    }
    *reinterpret_cast<float4 *>((float *)out.p + (int64_t)b * out.bs + opix * out.cs + out.co + c4 * 4) = make_float4(v[0], v[1], v[2], v[3]);
}

__global__ __launch_bounds__(256) void k_maxpool5_f32(TensorRef in, TensorRef out, int B, int H, int W, int C) {
    const int c4n = C >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)B * H * W * c4n) return;
    const int c4 = (int)(idx % c4n);
    const int64_t pix = idx / c4n;
    const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
    const float *ip = (const float *)in.p + (int64_t)b * in.bs + in.co + c4 * 4;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int yy = max(0, y - 2); yy <= min(H - 1, y + 2); ++yy)
        for (int xx = max(0, x - 2); xx <= min(W - 1, x + 2); ++xx) {
            const float4 v = *reinterpret_cast<const float4 *>(ip + ((int64_t)yy * W + xx) * in.cs);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    *reinterpret_cast<float4 *>((float *)out.p + (int64_t)b * out.bs + ((int64_t)y * W + x) * out.cs + out.co + c4 * 4) = m;
}

__global__ __launch_bounds__(256) void k_upsample2_f32(TensorRef in, TensorRef out, int B, int H, int W, int C) {
    const int c4n = C >> 2, Ho = H * 2, Wo = W * 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)B * Ho * Wo * c4n) return;
    const int c4 = (int)(idx % c4n);
    const int64_t pix = idx / c4n;
    const int x = (int)(pix % Wo), y = (int)((pix / Wo) % Ho), b = (int)(pix / ((int64_t)Wo * Ho));
    const float4 v = *reinterpret_cast<const float4 *>((const float *)in.p + (int64_t)b * in.bs + ((int64_t)(y >> 1) * W + (x >> 1)) * in.cs + in.co + c4 * 4);
    *reinterpret_cast<float4 *>((float *)out.p + (int64_t)b * out.bs + ((int64_t)y * Wo + x) * out.cs + out.co + c4 * 4) = v;
}

// C2PSA attention core in fp32: qkv slice per token = [q: nh*KD][k: nh*KD][v: nh*HD] (channels permuted by the weight loader).  One
// workgroup per (tile, head, quarter of the queries): K and V of all N tokens in LDS; a thread owns one query row.  Two-pass softmax
// (max, then exp / sum) like torch.softmax; out[n, h*HD + d] = sum_m softmax_m(q_n . k_m * scale) * v_m[d].
template <int KD, int HD>
__global__ __launch_bounds__(64) void k_attention_f32(TensorRef qkv, TensorRef out, int N, int nh, int nsplit, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm32[];
    float *sk = sm32, *sv = sm32 + (size_t)N * KD;
    const int part = blockIdx.x % nsplit, bh = blockIdx.x / nsplit;
    const int b = bh / nh, h = bh % nh;
    const float *base = (const float *)qkv.p + (int64_t)b * qkv.bs + qkv.co;
    for (int i = threadIdx.x; i < N * (KD / 4); i += 64) {
        const int n = i / (KD / 4), c = i % (KD / 4);
        *reinterpret_cast<float4 *>(sk + n * KD + c * 4) = *reinterpret_cast<const float4 *>(base + (int64_t)n * qkv.cs + nh * KD + h * KD + c * 4);
    }
    for (int i = threadIdx.x; i < N * (HD / 4); i += 64) {
        const int n = i / (HD / 4), c = i % (HD / 4);
        *reinterpret_cast<float4 *>(sv + n * HD + c * 4) = *reinterpret_cast<const float4 *>(base + (int64_t)n * qkv.cs + 2 * nh * KD + h * HD + c * 4);
    }
    __syncthreads();
    const int per = (N + nsplit - 1) / nsplit;
    const int n = part * per + threadIdx.x;
    if (threadIdx.x >= per || n >= N) return;
    float q[KD];
#pragma unroll
    for (int c = 0; c < KD / 4; ++c) {
        const float4 v = *reinterpret_cast<const float4 *>(base + (int64_t)n * qkv.cs + h * KD + c * 4);
        q[4 * c] = v.x; q[4 * c + 1] = v.y; q[4 * c + 2] = v.z; q[4 * c + 3] = v.w;
    }
    float mx = -INFINITY;
    for (int m = 0; m < N; ++m) {
        float s = 0.f;
        const float4 *kr = reinterpret_cast<const float4 *>(sk + m * KD);
#pragma unroll
        for (int d = 0; d < KD / 4; ++d) { const float4 k4 = kr[d]; s = fmaf(q[4 * d], k4.x, s); s = fmaf(q[4 * d + 1], k4.y, s); s = fmaf(q[4 * d + 2], k4.z, s); s = fmaf(q[4 * d + 3], k4.w, s); }
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
        for (int d = 0; d < KD / 4; ++d) { const float4 k4 = kr[d]; s = fmaf(q[4 * d], k4.x, s); s = fmaf(q[4 * d + 1], k4.y, s); s = fmaf(q[4 * d + 2], k4.z, s); s = fmaf(q[4 * d + 3], k4.w, s); }
        const float p = expf(s * scale - mx);
        den += p;
        const float4 *vr = reinterpret_cast<const float4 *>(sv + m * HD);
#pragma unroll
        for (int d = 0; d < HD / 4; ++d) { const float4 v4 = vr[d]; acc[4 * d] = fmaf(p, v4.x, acc[4 * d]); acc[4 * d + 1] = fmaf(p, v4.y, acc[4 * d + 1]); acc[4 * d + 2] = fmaf(p, v4.z, acc[4 * d + 2]); acc[4 * d + 3] = fmaf(p, v4.w, acc[4 * d + 3]); }
    }
    float *op = (float *)out.p + (int64_t)b * out.bs + (int64_t)n * out.cs + out.co + h * HD;
#pragma unroll
    for (int c = 0; c < HD / 4; ++c)
        *reinterpret_cast<float4 *>(op + c * 4) = make_float4(acc[4 * c] / den, acc[4 * c + 1] / den, acc[4 * c + 2] / den, acc[4 * c + 3] / den);
}

static inline unsigned blocks_for32(int64_t n) { return (unsigned)((n + 255) / 256); }
static bool plain4(const TensorRef &t) { return t.cpb == 0 && (t.cs & 3) == 0 && (t.co & 3) == 0; }

hipError_t launch_dwconv3_f32(const TensorRef &in, const TensorRef &out, const TensorRef &res, const float *w, const float *bias, int B, int H, int W, int C,
                              int act, hipStream_t st) {
    if (C % 4 || !plain4(in) || !plain4(out) || (res.p && !plain4(res))) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_dwconv3_f32, dim3(blocks_for32((int64_t)B * H * W * (C / 4))), dim3(256), 0, st, in, out, res, w, bias, B, H, W, C, act);
    return hipGetLastError();
}

hipError_t launch_maxpool5_f32(const TensorRef &in, const TensorRef &out, int B, int H, int W, int C, hipStream_t st) {
    if (C % 4 || !plain4(in) || !plain4(out)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_maxpool5_f32, dim3(blocks_for32((int64_t)B * H * W * (C / 4))), dim3(256), 0, st, in, out, B, H, W, C);
    return hipGetLastError();
}

hipError_t launch_upsample2_f32(const TensorRef &in, const TensorRef &out, int B, int H, int W, int C, hipStream_t st) {
    if (C % 4 || !plain4(in) || !plain4(out)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_upsample2_f32, dim3(blocks_for32((int64_t)B * H * W * 4 * (C / 4))), dim3(256), 0, st, in, out, B, H, W, C);
    return hipGetLastError();
}

hipError_t launch_attention_f32(const TensorRef &qkv, const TensorRef &out, int B, int N, int nh, int kd, int hd, hipStream_t st) {
    if (kd != 32 || hd != 64 || !plain4(qkv) || !plain4(out) || N < 1) return hipErrorInvalidValue;
    const size_t lds = sizeof(float) * ((size_t)N * 32 + (size_t)N * 64);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)k_attention_f32<32, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int nsplit = (N + 63) / 64;  // one wave of queries per workgroup
    const float scale = (float)(1.0 / sqrt((double)kd));
    hipLaunchKernelGGL((k_attention_f32<32, 64>), dim3((unsigned)(B * nh * nsplit)), dim3(64), lds, st, qkv, out, N, nh, nsplit, scale);
    return hipGetLastError();
}

}  // namespace obb
