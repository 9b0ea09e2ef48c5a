// fp32 C3k2 block with one Bottleneck -- the 104 x 104 block model.2 of the n / s scales (ultralytics C3k2(c3k=False, n=1): cv1 -> [y0 | y1],
// Bottleneck(y1) = y1 + cv2(cv1(y1)) = y2, closing 1x1 over [y0 | y1 | y2]) -- behind its cv1 as ONE launch.
//
// Unfused, the three layers are the most HBM-bound stretch of the fp32 forward (16 -> 8 -> 16 channels at 104 x 104 and a 48 -> 64 1x1:
// 27 FLOP per byte against a machine balance of ~25; 1.39 ms per 512 tiles at 37-48 TFLOP/s, 9.8 MB of traffic per tile).  Here a
// workgroup (8 waves, two per CU) owns a TH x TW tile of the block's output:
//   * X = y1 on (TH + 4) x (TW + 4) pixels (zero outside the image = the first conv's padding) and all weights go to LDS once;
//   * conv 1 (3x3, C -> C/2, SiLU) on the (TH + 2) x (TW + 2) halo region -> T in LDS (zero outside the image: the second conv pads ITS input);
//   * conv 2 (3x3, C/2 -> C, SiLU) + shortcut: a lane's four results of a 16-pixel fragment ARE the B operand (chunk g of its pixel) of the
//     closing 1x1's k piece over y2, so y2 never leaves the registers; y1 comes from X, y0 straight from global memory in operand layout;
//   * closing 1x1 (3 C -> CO, SiLU): 3 k pieces x CO / 16 cout fragments per pixel fragment; with the row permutation of c3k2f32_cout_perm a
//     lane ends with 16 consecutive output channels of its pixel (64 contiguous bytes = two 8-channel blocks).
// Every sum runs in the k order of the separate launches (one 16-channel stage, one 8-channel stage = 4 pieces + 2 single chunks, one
// 48-channel stage), so the block's output is BIT-IDENTICAL to the unfused plan (test_fp32_fused_forms_are_bit_identical).
// Traffic: 1.5 MB in + 2.8 MB out per tile.  MFMA floor (halo recompute and the 8-cout layer padded to a 16-row fragment included):
// 1640 instructions per 208-pixel tile.
#include "c3k2f32.h"

#include <algorithm>
#include <mutex>

namespace obb {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct C3k2Params {
    const float *cat; int64_t cat_bs; int cat_cs, cat_co; unsigned cat_span;
    float *out; int64_t out_bs; int out_cs, out_co, out_blk, out_ps;
    const float *wall, *b1, *b2;  // wall: [W1 | W2 | WC | bc] exactly as laid out in LDS
    int H, W, TH, TW, tiles_x, tiles_y;
    float inv_xw, inv_tw2, inv_tw;
};

__device__ __forceinline__ float silu32c(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }  // (= silu32 of f32path.hip)

constexpr int kC3MaxX = 448, kC3MaxT = 324;  // pixels of the X / T tiles the LDS layout holds

template <int C, int CO>
__global__ __launch_bounds__(512, 4) void k_c3k2_f32(const C3k2Params P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = 512, CH = C / 2, PSTX = C * 4 + 16, PSTT = CH * 4 + 16;
    constexpr int CPK1 = C / 4, KP1 = 9 * CPK1 / 4, CPK2 = CH / 4, KP2 = (9 * CPK2) / 4, KR2 = (9 * CPK2) % 4, KPC = 3 * C / 16, NFC = CO / 16;
    static_assert((9 * CPK1) % 4 == 0, "conv 1: whole pieces only");
    constexpr int XOFF = 0, TOFF = kC3MaxX * PSTX, W1OFF = TOFF + kC3MaxT * PSTT, W2OFF = W1OFF + KP1 * 1024, WCOFF = W2OFF + KP2 * 1024 + KR2 * 256,
                  BCOFF = WCOFF + NFC * KPC * 1024, NWCH = (BCOFF + CO * 4 - W1OFF) / 16;
    constexpr int F1 = 3, F2 = 2;  // fragments per wave: conv 1 (<= 24 fragments of the halo region), conv 2 + closing 1x1 (<= 16)
    constexpr unsigned NOPIX = 0xffffffffu;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, pl = lane & 15;
    const int t = blockIdx.x, tx_i = t % P.tiles_x, r_ = t / P.tiles_x, ty_i = r_ % P.tiles_y, b = r_ / P.tiles_y;
    const int oy0 = ty_i * P.TH, ox0 = tx_i * P.TW;
    const int XW = P.TW + 4, XH = P.TH + 4, TW2 = P.TW + 2, TH2 = P.TH + 2;
    const int npix1 = TH2 * TW2, npix2 = P.TH * P.TW;

    // ---- loads: y1 halo tile, all weights, y0 of this wave's output fragments (buffer loads: an offset past the range reads zeros)
    const __amdgpu_buffer_rsrc_t cat_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(P.cat + (int64_t)b * P.cat_bs + P.cat_co), 0, (int)P.cat_span, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)P.wall, 0, NWCH * 16, 0x00020000);
    constexpr int MAXLD = (kC3MaxX * CPK1 + NT - 1) / NT, MAXW = (NWCH + NT - 1) / NT;
    u32x4 xr[MAXLD], wr[MAXW];
    const int nchunkX = XH * XW * CPK1;
#pragma unroll
    for (int k = 0; k < MAXLD; ++k) {
        const int idx = tid + k * NT, pix = idx / CPK1, c = idx - pix * CPK1;
        const int iy = (int)(((float)pix + 0.5f) * P.inv_xw), ix = pix - iy * XW;
        const int gy = oy0 - 2 + iy, gx = ox0 - 2 + ix;
        const bool ok = idx < nchunkX && gy >= 0 && gy < P.H && gx >= 0 && gx < P.W;
        xr[k] = __builtin_amdgcn_raw_buffer_load_b128(cat_rsrc, ok ? (unsigned)((((int64_t)gy * P.W + gx) * P.cat_cs + C + c * 4) * 4) : NOPIX, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < MAXW; ++k) wr[k] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (unsigned)min(tid + k * NT, NWCH - 1) * 16u, 0, 0);
    const int nfrag1 = (npix1 + 15) >> 4, nfrag2 = (npix2 + 15) >> 4;
    f32x4 y0v[F2];
    int opix[F2];  // output pixel of fragment i (row-major in the image), -1: none
    int xc[F2], tb[F2];
#pragma unroll
    for (int i = 0; i < F2; ++i) {
        const int p0 = (wave + 8 * i) * 16 + pl, p = min(p0, npix2 - 1);
        const int r = (int)(((float)p + 0.5f) * P.inv_tw), x = p - r * P.TW;
        const int oy = oy0 + r, ox = ox0 + x;
        const bool ok = p0 < npix2 && oy < P.H && ox < P.W;
        opix[i] = ok ? oy * P.W + ox : -1;
        y0v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(cat_rsrc, ok ? (unsigned)(((int64_t)opix[i] * P.cat_cs + g * 4) * 4) : NOPIX, 0, 0));
        xc[i] = ((r + 2) * XW + x + 2) * PSTX;  // y1 at the output pixel
        tb[i] = (r * TW2 + x) * PSTT;           // top-left tap of conv 2
    }
#pragma unroll
    for (int k = 0; k < MAXLD; ++k) {
        const int idx = tid + k * NT, pix = idx / CPK1, c = idx - pix * CPK1;
        if (idx < nchunkX) *reinterpret_cast<u32x4 *>(smem + XOFF + pix * PSTX + c * 16) = xr[k];
    }
#pragma unroll
    for (int k = 0; k < MAXW; ++k) {
        const int idx = tid + k * NT;
        if (idx < NWCH) *reinterpret_cast<u32x4 *>(smem + W1OFF + idx * 16) = wr[k];
    }
    __syncthreads();

    // ---- conv 1: 3x3, C -> C/2 (16-row fragment, rows >= C/2 are zero weights), SiLU -> T
    {
        int xb[F1];
        bool val[F1];
#pragma unroll
        for (int i = 0; i < F1; ++i) {
            const int f = wave + 8 * i;
            val[i] = f < nfrag1;  // (wave-uniform)
            const int p = min(f * 16 + pl, npix1 - 1);
            const int r = (int)(((float)p + 0.5f) * P.inv_tw2), x = p - r * TW2;
            xb[i] = (r * XW + x) * PSTX;
        }
        f32x4 acc[F1];
#pragma unroll
        for (int i = 0; i < F1; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int piece = 0; piece < KP1; ++piece) {
            const f32x4 w = *reinterpret_cast<const f32x4 *>(smem + W1OFF + piece * 1024 + lane * 16);
            const int q = piece * 4 + g, tap = q / CPK1, c0 = q - tap * CPK1, dy = (tap * 11) >> 5, dx = tap - dy * 3;
            const int off = (dy * XW + dx) * PSTX + c0 * 16;
            f32x4 a[F1];
#pragma unroll
            for (int i = 0; i < F1; ++i) a[i] = *reinterpret_cast<const f32x4 *>(smem + XOFF + xb[i] + off);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < F1; ++i)
                    if (val[i]) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], a[i][s], acc[i], 0, 0, 0);
        }
        const float4 bv = *reinterpret_cast<const float4 *>(P.b1 + 4 * g);  // (padded to 16 floats)
#pragma unroll
        for (int i = 0; i < F1; ++i) {
            const int p = (wave + 8 * i) * 16 + pl;
            if (!val[i] || p >= npix1 || 4 * g >= CH) continue;
            const int r = (int)(((float)p + 0.5f) * P.inv_tw2), x = p - r * TW2;
            const int gy = oy0 - 1 + r, gx = ox0 - 1 + x;
            const bool inside = gy >= 0 && gy < P.H && gx >= 0 && gx < P.W;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (inside) o = make_float4(silu32c(acc[i][0] + bv.x), silu32c(acc[i][1] + bv.y), silu32c(acc[i][2] + bv.z), silu32c(acc[i][3] + bv.w));
            *reinterpret_cast<float4 *>(smem + TOFF + (r * TW2 + x) * PSTT + g * 16) = o;
        }
    }
    __syncthreads();

    // ---- conv 2: 3x3, C/2 -> C (4 pieces + 2 single chunks: the k order of the separate launch), SiLU, + y1 -> y2 in registers
    bool val2[F2];
#pragma unroll
    for (int i = 0; i < F2; ++i) val2[i] = wave + 8 * i < nfrag2;  // (wave-uniform)
    f32x4 acc2[F2];
#pragma unroll
    for (int i = 0; i < F2; ++i) acc2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int piece = 0; piece < KP2; ++piece) {
        const f32x4 w = *reinterpret_cast<const f32x4 *>(smem + W2OFF + piece * 1024 + lane * 16);
        const int q = piece * 4 + g, tap = q / CPK2, c0 = q - tap * CPK2, dy = (tap * 11) >> 5, dx = tap - dy * 3;
        const int off = (dy * TW2 + dx) * PSTT + c0 * 16;
        f32x4 a[F2];
#pragma unroll
        for (int i = 0; i < F2; ++i) a[i] = *reinterpret_cast<const f32x4 *>(smem + TOFF + tb[i] + off);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < F2; ++i)
                if (val2[i]) acc2[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], a[i][s], acc2[i], 0, 0, 0);
    }
#pragma unroll
    for (int rr = 0; rr < KR2; ++rr) {  // the stage's last chunks, one MFMA each: its four k slots are the chunk's four channels
        const float w1 = *reinterpret_cast<const float *>(smem + W2OFF + KP2 * 1024 + rr * 256 + lane * 4);
        const int q = KP2 * 4 + rr, tap = q / CPK2, c0 = q - tap * CPK2, dy = (tap * 11) >> 5, dx = tap - dy * 3;
        const int off = (dy * TW2 + dx) * PSTT + c0 * 16 + g * 4;
#pragma unroll
        for (int i = 0; i < F2; ++i) {
            const float a1 = *reinterpret_cast<const float *>(smem + TOFF + tb[i] + off);
            if (val2[i]) acc2[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1, a1, acc2[i], 0, 0, 0);
        }
    }
    f32x4 y1v[F2], y2v[F2];
    {
        const float4 bv = *reinterpret_cast<const float4 *>(P.b2 + 4 * g);
#pragma unroll
        for (int i = 0; i < F2; ++i) {
            y1v[i] = *reinterpret_cast<const f32x4 *>(smem + XOFF + xc[i] + g * 16);
            y2v[i][0] = y1v[i][0] + silu32c(acc2[i][0] + bv.x); y2v[i][1] = y1v[i][1] + silu32c(acc2[i][1] + bv.y);
            y2v[i][2] = y1v[i][2] + silu32c(acc2[i][2] + bv.z); y2v[i][3] = y1v[i][3] + silu32c(acc2[i][3] + bv.w);
        }
    }

    // ---- closing 1x1 over [y0 | y1 | y2] (k pieces 0, 1, 2), SiLU, store
    f32x4 acc3[F2][NFC];
#pragma unroll
    for (int i = 0; i < F2; ++i)
#pragma unroll
        for (int nf = 0; nf < NFC; ++nf) acc3[i][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int piece = 0; piece < KPC; ++piece) {
        f32x4 bop[F2];
#pragma unroll
        for (int i = 0; i < F2; ++i) bop[i] = piece == 0 ? y0v[i] : (piece == 1 ? y1v[i] : y2v[i]);
#pragma unroll
        for (int nf = 0; nf < NFC; ++nf) {
            const f32x4 w = *reinterpret_cast<const f32x4 *>(smem + WCOFF + (nf * KPC + piece) * 1024 + lane * 16);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < F2; ++i)
                    if (val2[i]) acc3[i][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], bop[i][s], acc3[i][nf], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < F2; ++i) {
        if (!val2[i] || opix[i] < 0) continue;
        const int ca = P.out_co + 16 * g;  // this lane's 16 consecutive output channels start here
        float *op = P.out + (int64_t)b * P.out_bs + (int64_t)opix[i] * P.out_cs + (P.out_blk ? (int64_t)(ca >> 3) * P.out_ps + (ca & 7) : (int64_t)ca);
#pragma unroll
        for (int nf = 0; nf < NFC; ++nf) {
            const float4 bv = *reinterpret_cast<const float4 *>(smem + BCOFF + (nf * 16 + 4 * g) * 4);
            const float4 o = make_float4(silu32c(acc3[i][nf][0] + bv.x), silu32c(acc3[i][nf][1] + bv.y), silu32c(acc3[i][nf][2] + bv.z), silu32c(acc3[i][nf][3] + bv.w));
            // channels 16 g + 4 nf ..: plain NHWC -> + 4 nf floats; 8-channel blocks -> block (nf >> 1) further on, + 4 (nf & 1) inside it
            float *o4 = P.out_blk ? op + (int64_t)(nf >> 1) * P.out_ps + 4 * (nf & 1) : op + 4 * nf;
            *reinterpret_cast<float4 *>(o4) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
void c3k2f32_tile(int H, int W, int &TH, int &TW) {
    TW = std::min(W, 52);
    if (W > 52 && W % 52 && W % 2 == 0 && W / 2 <= 52) TW = W / 2;  // two equal column tiles where that fits
    int th = 1;
    while (th + 1 <= H && (th + 5) * (TW + 4) <= kC3MaxX && (th + 3) * (TW + 2) <= kC3MaxT && (th + 3) * (TW + 2) <= 24 * 16 && (th + 1) * TW <= 16 * 16) ++th;
    TH = th;
}

bool c3k2f32_supported(int C, int CO, int H, int W) {
    if (C != 16 || CO != 64 || H < 1 || W < 1) return false;
    int TH, TW;
    c3k2f32_tile(H, W, TH, TW);
    return (TH + 4) * (TW + 4) <= kC3MaxX && (TH + 2) * (TW + 2) <= kC3MaxT && (TH + 2) * (TW + 2) <= 24 * 16 && TH * TW <= 16 * 16;
}

std::vector<int> c3k2f32_cout_perm(int CO) {
    std::vector<int> p(CO);
    for (int f = 0; f < CO / 16; ++f)
        for (int r = 0; r < 16; ++r) p[16 * f + r] = 16 * (r >> 2) + 4 * f + (r & 3);
    return p;
}

hipError_t launch_c3k2f32(const C3k2F32Launch &L, hipStream_t st) {
    if (!c3k2f32_supported(L.C, L.CO, L.H, L.W) || L.cat.cpb || (L.cat.cs & 3) || (L.cat.co & 3) || !L.w1 || !L.b1 || !L.b2 || L.B < 1) return hipErrorInvalidValue;
    if (L.out.cpb && !(L.out.cpb == 2 && L.out.cs == 8 && L.out.co % 8 == 0 && L.out.ps > 0)) return hipErrorInvalidValue;
    if (!L.out.cpb && ((L.out.cs | L.out.co) & 3)) return hipErrorInvalidValue;
    C3k2Params P;
    P.cat = (const float *)L.cat.p; P.cat_bs = L.cat.bs; P.cat_cs = L.cat.cs; P.cat_co = L.cat.co;
    const int64_t span = ((int64_t)L.H * L.W * L.cat.cs - L.cat.co) * 4;
    if (span <= 0 || span >= (1ll << 32) - 65536) return hipErrorInvalidValue;
    P.cat_span = (unsigned)span;
    P.out = (float *)L.out.p; P.out_bs = L.out.bs; P.out_cs = L.out.cs; P.out_co = L.out.co; P.out_blk = L.out.cpb ? 1 : 0; P.out_ps = (int)L.out.ps;
    P.wall = L.w1; P.b1 = L.b1; P.b2 = L.b2;
    P.H = L.H; P.W = L.W;
    c3k2f32_tile(L.H, L.W, P.TH, P.TW);
    P.tiles_x = (L.W + P.TW - 1) / P.TW; P.tiles_y = (L.H + P.TH - 1) / P.TH;
    P.inv_xw = 1.0f / (float)(P.TW + 4); P.inv_tw2 = 1.0f / (float)(P.TW + 2); P.inv_tw = 1.0f / (float)P.TW;
    const int64_t ntiles = (int64_t)L.B * P.tiles_y * P.tiles_x;
    if (ntiles >= (1ll << 31)) return hipErrorInvalidValue;
    constexpr int C = 16, CO = 64;
    constexpr size_t lds = (size_t)kC3MaxX * (C * 4 + 16) + (size_t)kC3MaxT * (C / 2 * 4 + 16) + 9 * 1024 + (4 * 1024 + 2 * 256) + (CO / 16) * 3 * 1024 + CO * 4;
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] { attr_err = hipFuncSetAttribute((const void *)k_c3k2_f32<16, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); });
    if (attr_err != hipSuccess) return attr_err;
    hipLaunchKernelGGL((k_c3k2_f32<16, 64>), dim3((unsigned)ntiles), dim3(512), lds, st, P);
    return hipGetLastError();
}

}  // namespace obb
