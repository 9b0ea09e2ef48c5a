// LDS-resident layer chains (see fused.h): one workgroup (8 waves) walks spatial tiles of one pyramid level; per tile it
//   1. commits the prefetched input block (tile + halo, all channels) from registers to LDS and issues the next tile's loads,
//   2. runs the chain's layers back to back: MFMA convs (1x1 / 3x3, `v_mfma_f32_16x16x32_{f16,bf16}`, same operand mapping and
//      weight-fragment order as k_conv_igemm with a single channel stage) and depthwise 3x3 convs, every intermediate rounded to
//      the 16-bit storage type exactly where the unfused path rounds it, but written to LDS instead of HBM,
//   3. writes the final tensor: 16-bit rows staged in LDS and stored as whole 16-B pieces, or fp32 rows of the head tensor.
// Positions outside the image are forced to zero in every intermediate that feeds a 3x3 layer (that layer zero-pads its input).
// Replaces the same ultralytics modules as conv.hip / nnops.hip (C3k2, the cv3 / cv4 branches of the OBB head).
#include "fused.h"

#include <algorithm>
#include <cstdlib>

namespace obb {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct FusedParams {
    const void *in; int64_t in_bs; int in_cs, in_co;
    void *out; int64_t out_bs; int out_cs, out_co;
    int B, H, W, TH, TW, tiles_x, tiles_y, ntiles, tpw;
    int in_halo, in_sh, in_off, in_pst, in_rw, in_nchunk;
    float inv_in_rw;
    unsigned in_span_bytes;
    int nsteps;
    FusedStep steps[kFusedMaxSteps];
    const bf16_t *wts; int w_bytes, w_lds_off;
    int dbg;  // timing experiments only (OBB_FUSED_DBG): 1 skip k-loops, 2 skip conv epilogues, 4 skip conv steps, 8 skip dw steps, 16 skip stores, 32 skip input loads
};

constexpr int NT = kFusedThreads, NW = NT / 64;

template <int KS, int NF, bool F16>
__device__ __forceinline__ void fused_conv(const FusedStep &S, const FusedParams &P, char *smem, int b, int oy0, int ox0) {
    typedef typename HX<F16>::vec8 hx8;
    constexpr int MFB = 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, pl = lane & 15;
    const int npx = S.oh * S.ow;
    const int nfrag = (npx + 15) >> 4;
    const int cpk = S.cin >> 3;
    const int nq = KS * KS * cpk;
    for (int cb = 0; cb < S.ncb; ++cb) {
        const char *wl = smem + S.w_off + cb * S.kst * NF * 1024;
        const int cbase = cb * 16 * NF + g * 4 * NF;
        float bias[NF * 4];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            float4 bv = *reinterpret_cast<const float4 *>(S.bias + cbase + f * 4);
            bias[f * 4 + 0] = bv.x; bias[f * 4 + 1] = bv.y; bias[f * 4 + 2] = bv.z; bias[f * 4 + 3] = bv.w;
        }
        for (int f0 = wave; f0 < nfrag; f0 += NW * MFB) {
            int pixbase[MFB], pyx[MFB];
#pragma unroll
            for (int i = 0; i < MFB; ++i) {
                int p = (f0 + i * NW) * 16 + pl;
                bool ok = p < npx;
                int ty = (int)(((float)p + 0.5f) * S.inv_ow);
                int tx = p - ty * S.ow;
                if (!ok) { ty = 0; tx = 0; }
                pixbase[i] = S.in_off + ((ty * S.stride + S.in_y0) * S.in_w + tx * S.stride + S.in_x0) * S.in_pst + S.in_cb;
                pyx[i] = ok ? ((ty << 16) | tx) : -1;
            }
            f32x4 acc[MFB][NF];
#pragma unroll
            for (int i = 0; i < MFB; ++i)
#pragma unroll
                for (int f = 0; f < NF; ++f) acc[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int ks = 0; ks < ((P.dbg & 1) ? 0 : S.kst); ++ks) {
                hx8 wcur[NF];
#pragma unroll
                for (int f = 0; f < NF; ++f) wcur[f] = *reinterpret_cast<const hx8 *>(wl + ((ks * NF + f) * 64 + lane) * 16);
                int q = ks * 4 + g;
                q = q < nq ? q : nq - 1;  // padding k-steps: any valid address, their weights are zero
                int off;
                if constexpr (KS == 3) {
                    int tap = q >> S.sh, c0 = q & (cpk - 1);
                    int dy = (tap * 11) >> 5, dx = tap - dy * 3;
                    off = (dy * S.in_w + dx) * S.in_pst + c0 * 16;
                } else {
                    off = q * 16;
                }
                hx8 a[MFB];
#pragma unroll
                for (int i = 0; i < MFB; ++i) a[i] = *reinterpret_cast<const hx8 *>(smem + pixbase[i] + off);
#pragma unroll
                for (int i = 0; i < MFB; ++i)
#pragma unroll
                    for (int f = 0; f < NF; ++f) acc[i][f] = HX<F16>::mfma(wcur[f], a[i], acc[i][f]);
            }
            // ---- epilogue: the lane owns couts [cbase, cbase + 4*NF) of its pixels
#pragma unroll
            for (int i = 0; i < MFB; ++i) {
                if (pyx[i] < 0 || (P.dbg & 2)) continue;
                const int ty = pyx[i] >> 16, tx = pyx[i] & 0xffff;
                const int gy = oy0 - S.halo + ty, gx = ox0 - S.halo + tx;
                const bool inside = (unsigned)gy < (unsigned)P.H && (unsigned)gx < (unsigned)P.W;
                float v[NF * 4];
#pragma unroll
                for (int f = 0; f < NF; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x = acc[i][f][r] + bias[f * 4 + r];
                        if (S.act) x = silu_f(x);
                        v[f * 4 + r] = x;
                    }
                if (S.res_off >= 0) {
                    const char *rp = smem + S.res_off + ((ty + S.res_y0) * S.res_w + tx + S.res_x0) * S.res_pst + S.res_cb + cbase * 2;
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        uint2 rv = *reinterpret_cast<const uint2 *>(rp + f * 8);
                        v[f * 4 + 0] += HX<F16>::lo(rv.x); v[f * 4 + 1] += HX<F16>::hi(rv.x);
                        v[f * 4 + 2] += HX<F16>::lo(rv.y); v[f * 4 + 3] += HX<F16>::hi(rv.y);
                    }
                }
                if (S.to_global) {  // fp32 rows of the head tensor
                    if (!inside) continue;
                    float *op = (float *)P.out + (int64_t)b * P.out_bs + ((int64_t)gy * P.W + gx) * P.out_cs + P.out_co + cbase;
                    const bool al = ((P.out_cs | P.out_co) & 3) == 0;
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        if (al && cbase + f * 4 + 4 <= S.cout) {
                            *reinterpret_cast<float4 *>(op + f * 4) = make_float4(v[f * 4], v[f * 4 + 1], v[f * 4 + 2], v[f * 4 + 3]);
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (cbase + f * 4 + r < S.cout) op[f * 4 + r] = v[f * 4 + r];
                        }
                    }
                } else {
                    const bool zero = S.mask && !inside;
                    char *orow = smem + S.out_off + ((ty + S.out_y0) * S.out_w + tx + S.out_x0) * S.out_pst + S.out_cb + cbase * 2;
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        if (cbase + f * 4 + 4 > S.cout) continue;
                        uint2 o;
                        o.x = zero ? 0u : HX<F16>::pack2(v[f * 4 + 0], v[f * 4 + 1]);
                        o.y = zero ? 0u : HX<F16>::pack2(v[f * 4 + 2], v[f * 4 + 3]);
                        *reinterpret_cast<uint2 *>(orow + f * 8) = o;
                    }
                }
            }
        }
    }
}

// depthwise 3x3 (+bias, SiLU): one work item = one pixel x 8 channels; same tap order and arithmetic as k_dwconv3
template <bool F16>
__device__ __forceinline__ void fused_dw(const FusedStep &S, const FusedParams &P, char *smem, int oy0, int ox0) {
    const int c8n = S.cin >> 3;
    const int items = (S.oh * S.ow) << S.sh;
    for (int it = threadIdx.x; it < items; it += NT) {
        const int pix = it >> S.sh, c8 = it & (c8n - 1);
        const int ty = (int)(((float)pix + 0.5f) * S.inv_ow);
        const int tx = pix - ty * S.ow;
        const char *ip = smem + S.in_off + ((ty + S.in_y0) * S.in_w + tx + S.in_x0) * S.in_pst + S.in_cb + c8 * 16;
        const char *wp = smem + S.w_off + c8 * 16;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
            uint4 xv = *reinterpret_cast<const uint4 *>(ip + (dy * S.in_w + dx) * S.in_pst);
            uint4 wv = *reinterpret_cast<const uint4 *>(wp + tap * S.cin * 2);
            float xf[8], wf[8];
            unpack8<F16>(xv, xf);
            unpack8<F16>(wv, wf);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += xf[j] * wf[j];
        }
        const float4 *bp = reinterpret_cast<const float4 *>(S.bias + c8 * 8);
        float4 b0 = bp[0], b1 = bp[1];
        float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        const int gy = oy0 - S.halo + ty, gx = ox0 - S.halo + tx;
        const bool zero = S.mask && !((unsigned)gy < (unsigned)P.H && (unsigned)gx < (unsigned)P.W);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = acc[j] + bb[j];
            if (S.act) v = silu_f(v);
            acc[j] = zero ? 0.f : v;
        }
        *reinterpret_cast<uint4 *>(smem + S.out_off + ((ty + S.out_y0) * S.out_w + tx + S.out_x0) * S.out_pst + S.out_cb + c8 * 16) = pack8<F16>(acc);
    }
}

// LDS block -> 16-bit NHWC slice in global memory: consecutive lanes store consecutive 16-B pieces of a pixel's row
__device__ __forceinline__ void fused_store(const FusedStep &S, const FusedParams &P, char *smem, int b, int oy0, int ox0) {
    const int cpp = S.cin >> 3;  // 16-B chunks per pixel
    const int n = S.oh * S.ow * cpp;
    bf16_t *obase = (bf16_t *)P.out + (int64_t)b * P.out_bs + P.out_co;
    for (int i = threadIdx.x; i < n; i += NT) {
        int p = i / cpp, ch = i - p * cpp;
        int ty = (int)(((float)p + 0.5f) * S.inv_ow);
        int tx = p - ty * S.ow;
        int gy = oy0 - S.halo + ty, gx = ox0 - S.halo + tx;
        if ((unsigned)gy >= (unsigned)P.H || (unsigned)gx >= (unsigned)P.W) continue;
        uint4 o = *reinterpret_cast<const uint4 *>(smem + S.in_off + ((ty + S.in_y0) * S.in_w + tx + S.in_x0) * S.in_pst + S.in_cb + ch * 16);
        *reinterpret_cast<uint4 *>(obase + ((int64_t)gy * P.W + gx) * P.out_cs + ch * 8) = o;
    }
}

// WPE = waves per SIMD the register budget is sized for: 4 = two 8-wave groups per CU (chains whose LDS footprint allows two)
template <bool F16, int MAXPF, int WPE>
__global__ __launch_bounds__(NT, WPE) void k_fused_chain(const FusedParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * P.tpw;
    const int t1 = min(t0 + P.tpw, P.ntiles);
    if (t0 >= t1) return;
    // every layer's weights stay resident in LDS for all tiles of this group
    for (int i = tid; i < (P.w_bytes >> 4); i += NT)
        *reinterpret_cast<u32x4 *>(smem + P.w_lds_off + i * 16) = reinterpret_cast<const u32x4 *>(P.wts)[i];

    // input block: this thread moves the 16-B chunks idx = tid + k*NT of the [pixels][in_C] block (buffer loads: offsets beyond
    // the descriptor's range return zeros, which is how zero padding is expressed)
    constexpr unsigned NOPIX = 0xffffffffu;
    const int c8m = (1 << P.in_sh) - 1;
    int ipos[MAXPF];
#pragma unroll
    for (int k = 0; k < MAXPF; ++k) {
        int idx = tid + k * NT;
        int pix = idx >> P.in_sh;
        int iy = (int)(((float)pix + 0.5f) * P.inv_in_rw);
        int ix = pix - iy * P.in_rw;
        ipos[k] = idx < P.in_nchunk ? ((iy << 16) | ix) : -1;
    }
    auto tile_origin = [&](int t, int &b, int &oy0, int &ox0) {
        int tx_i = t % P.tiles_x;
        int r = t / P.tiles_x;
        int ty_i = r % P.tiles_y;
        b = r / P.tiles_y;
        oy0 = ty_i * P.TH; ox0 = tx_i * P.TW;
    };
    u32x4 pre[MAXPF];
    auto issue = [&](int t) {
        int b, oy0, ox0;
        tile_origin(t, b, oy0, ox0);
        const bf16_t *base = (const bf16_t *)P.in + (int64_t)b * P.in_bs + P.in_co;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)P.in_span_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < MAXPF; ++k) {
            int gy = oy0 - P.in_halo + (ipos[k] >> 16), gx = ox0 - P.in_halo + (ipos[k] & 0xffff);
            bool ok = ipos[k] >= 0 && gy >= 0 && gy < P.H && gx >= 0 && gx < P.W && !(P.dbg & 32);
            unsigned off = ok ? (unsigned)((((int64_t)gy * P.W + gx) * P.in_cs + ((tid + k * NT) & c8m) * 8) * 2) : NOPIX;
            pre[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        }
    };
    issue(t0);
    __syncthreads();
    for (int t = t0; t < t1; ++t) {
        int b, oy0, ox0;
        tile_origin(t, b, oy0, ox0);
#pragma unroll
        for (int k = 0; k < MAXPF; ++k) {
            int idx = tid + k * NT;
            if (idx < P.in_nchunk) *reinterpret_cast<u32x4 *>(smem + P.in_off + (idx >> P.in_sh) * P.in_pst + (idx & c8m) * 16) = pre[k];
        }
        __syncthreads();
        if (t + 1 < t1) issue(t + 1);  // the next tile's HBM/L2 latency hides under this tile's layers
        for (int s = 0; s < P.nsteps; ++s) {
            const FusedStep &S = P.steps[s];
            if (S.type == FS_CONV) {
                if (P.dbg & 4) {
                } else if (S.ks == 3) {
                    if (S.NF == 1) fused_conv<3, 1, F16>(S, P, smem, b, oy0, ox0);
                    else if (S.NF == 2) fused_conv<3, 2, F16>(S, P, smem, b, oy0, ox0);
                    else fused_conv<3, 4, F16>(S, P, smem, b, oy0, ox0);
                } else {
                    if (S.NF == 1) fused_conv<1, 1, F16>(S, P, smem, b, oy0, ox0);
                    else if (S.NF == 2) fused_conv<1, 2, F16>(S, P, smem, b, oy0, ox0);
                    else fused_conv<1, 4, F16>(S, P, smem, b, oy0, ox0);
                }
            } else if (S.type == FS_DW) {
                if (!(P.dbg & 8)) fused_dw<F16>(S, P, smem, oy0, ox0);
            } else {
                if (!(P.dbg & 16)) fused_store(S, P, smem, b, oy0, ox0);
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side

template <bool F16, int MAXPF, int WPE>
static hipError_t launch_v(const FusedParams &P, dim3 grid, size_t lds, hipStream_t st) {
    static size_t lds_set = 0;  // opt-in for > 64 KiB of dynamic LDS, raised once per kernel variant
    if (lds > 64 * 1024 && lds > lds_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fused_chain<F16, MAXPF, WPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        lds_set = 160 * 1024;
    }
    hipLaunchKernelGGL((k_fused_chain<F16, MAXPF, WPE>), grid, dim3(NT), lds, st, P);
    return hipGetLastError();
}

hipError_t launch_fused(const FusedLaunch &L, hipStream_t st) {
    FusedParams P;
    P.in = L.in.p; P.in_bs = L.in.bs; P.in_cs = L.in.cs; P.in_co = L.in.co;
    P.out = L.out.p; P.out_bs = L.out.bs; P.out_cs = L.out.cs; P.out_co = L.out.co;
    P.B = L.B; P.H = L.H; P.W = L.W; P.TH = L.TH; P.TW = L.TW;
    P.tiles_x = (L.W + L.TW - 1) / L.TW; P.tiles_y = (L.H + L.TH - 1) / L.TH;
    int64_t ntiles = (int64_t)L.B * P.tiles_x * P.tiles_y;
    if (ntiles <= 0 || ntiles >= (1ll << 31)) return hipErrorInvalidValue;
    P.ntiles = (int)ntiles;
    const int c8n = L.in_C / 8;
    int sh = 0;
    while ((1 << sh) < c8n) ++sh;
    if ((1 << sh) != c8n || L.in_C % 8) return hipErrorInvalidValue;  // power-of-two channel chunks
    const int rh = L.TH + 2 * L.in_halo, rw = L.TW + 2 * L.in_halo;
    P.in_halo = L.in_halo; P.in_sh = sh; P.in_off = L.in_off; P.in_pst = L.in_pst; P.in_rw = rw;
    P.in_nchunk = rh * rw * c8n;
    P.inv_in_rw = 1.0f / (float)rw;
    int64_t span = ((int64_t)L.H * L.W * L.in.cs - L.in.co) * 2;
    if (span <= 0 || span >= (1ll << 32) - 65536) return hipErrorInvalidValue;
    P.in_span_bytes = (unsigned)span;
    if (L.nsteps < 1 || L.nsteps > kFusedMaxSteps) return hipErrorInvalidValue;
    P.nsteps = L.nsteps;
    for (int i = 0; i < L.nsteps; ++i) {
        P.steps[i] = L.steps[i];
        const FusedStep &S = L.steps[i];
        if (S.type == FS_CONV && S.ks == 3 && (1 << S.sh) != S.cin / 8) return hipErrorInvalidValue;
        if (S.type == FS_DW && (1 << S.sh) != S.cin / 8) return hipErrorInvalidValue;
    }
    P.wts = L.wts; P.w_bytes = L.w_bytes; P.w_lds_off = L.w_lds_off;
    static const int dbg = getenv("OBB_FUSED_DBG") ? atoi(getenv("OBB_FUSED_DBG")) : 0;
    P.dbg = dbg;
    if (L.w_bytes % 16 || L.lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    // a group walks up to 8 consecutive tiles once there are enough groups to fill the chip
    static const int tpw_max = getenv("OBB_FUSED_TPW") ? std::max(1, atoi(getenv("OBB_FUSED_TPW"))) : 8;
    const int groups_per_cu = L.lds_bytes <= 80 * 1024 ? 2 : 1;
    int64_t tpw = ntiles / (256 * groups_per_cu * 2);
    P.tpw = (int)std::max<int64_t>(1, std::min<int64_t>(tpw, tpw_max));
    dim3 grid((unsigned)((ntiles + P.tpw - 1) / P.tpw));
    const int npf = (P.in_nchunk + NT - 1) / NT;
    static const bool wpe4 = !(getenv("OBB_FUSED_WPE") && atoi(getenv("OBB_FUSED_WPE")) == 2);
    if (groups_per_cu == 1 || !wpe4) {
        if (npf <= 3) return L.f16 ? launch_v<true, 3, 2>(P, grid, L.lds_bytes, st) : launch_v<false, 3, 2>(P, grid, L.lds_bytes, st);
        if (npf <= 6) return L.f16 ? launch_v<true, 6, 2>(P, grid, L.lds_bytes, st) : launch_v<false, 6, 2>(P, grid, L.lds_bytes, st);
    } else if (groups_per_cu == 2) {
        if (npf <= 3) return L.f16 ? launch_v<true, 3, 4>(P, grid, L.lds_bytes, st) : launch_v<false, 3, 4>(P, grid, L.lds_bytes, st);
        if (npf <= 6) return L.f16 ? launch_v<true, 6, 4>(P, grid, L.lds_bytes, st) : launch_v<false, 6, 4>(P, grid, L.lds_bytes, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace obb
