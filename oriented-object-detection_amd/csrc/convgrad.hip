// Third slice of the training step (`model.train(...)`, Train_OBB.py:796-841: bf16 autocast, fp32 master
// weights): the backward of a stride-1 convolution (the 3x3 64 -> 64 and 1x1 shapes of YOLO11's C3k2 / head blocks), bf16 tensors in,
// fp32 accumulation.
//
//   dgrad  dX[b, y, x, ci] = sum_{ky, kx, co} dY[b, y + p - ky, x + p - kx, co] * W[co][ci][ky][kx]          (p = k / 2)
//          = the forward convolution of dY with the spatially flipped, channel-transposed weights: it runs on the forward's bf16
//          MFMA implicit-GEMM kernel (conv.hip, v_mfma_f32_16x16x32_bf16) with weights repacked per call; output rounded to bf16 once.
//   wgrad  dW[co][ci][ky][kx] = sum_{b, y, x} dY[b, y, x, co] * X[b, y + ky - p, x + kx - p, ci]
//          a GEMM whose reduction runs over PIXELS, which are the strided dimension of both NHWC operands.  First form (this file):
//          bf16 values widened to fp32 in registers (exact) and multiplied on the exact-f32 matrix instruction
//          v_mfma_f32_16x16x4_f32, whose operands are ONE k value per lane -- no transposition of the tiles is needed; fp32
//          accumulation in registers over all the tiles a persistent workgroup walks, one fp32 slab per workgroup, summed by a second
//          kernel in a fixed order (deterministic; no float atomics).  The 16x-faster bf16 form (v_mfma_f32_16x16x32_bf16 fed by
//          ds_read_b64_tr_b16 transposed LDS reads) is the next step: it needs the same tiles and the same walk.
#include <algorithm>
#include <vector>

#include "conv.h"
#include "ctx.h"

namespace obb {

typedef __attribute__((ext_vector_type(4))) float f32x4g;

__device__ __forceinline__ float bf16_to_f32(unsigned short v) { return __uint_as_float((unsigned)v << 16); }

// Workgroup = 4 waves = one (64-cout block, 64-cin block) pair of dW for all KS*KS taps; wave w owns the cin fragment w (16 channels)
// and all four cout fragments: acc[4][TAPS] tiles of 16 x 16.  A tile of the input = R rows x W pixels of one image: dY rows padded
// with zeros to a multiple of 4 pixels (the k step), X rows with the zero halo of the convolution's padding.
template <int KS>
__global__ __launch_bounds__(256) void k_conv_wgrad(const unsigned short *__restrict__ x, const unsigned short *__restrict__ dy, int B, int H, int W, int cin, int cout,
                                                   int R, float *__restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) unsigned short swg[];
    constexpr int TAPS = KS * KS, PAD = KS / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, r16 = lane & 15;
    const int cob = blockIdx.y, cib = blockIdx.z;
    const int Wp = (W + 3) & ~3, Wx = Wp + 2 * PAD;  // padded widths of the dY / X tiles
    unsigned short *sx = swg, *sdy = swg + (size_t)(R + 2 * PAD) * Wx * 64;
    const int tiles_y = (H + R - 1) / R, ntiles = B * tiles_y;
    f32x4g acc[4][TAPS];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < TAPS; ++t) acc[c][t] = f32x4g{0.f, 0.f, 0.f, 0.f};
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_y, y0 = (tile % tiles_y) * R;
        __syncthreads();
        // stage X rows y0 - PAD .. y0 + R - 1 + PAD, pixels -PAD .. Wp - 1 + PAD, 64 channels of block cib (16-byte chunks, zeros outside)
        for (int i = tid; i < (R + 2 * PAD) * Wx * 8; i += 256) {
            const int c8 = i & 7, px = (i >> 3) % Wx, ry = (i >> 3) / Wx;
            const int yy = y0 + ry - PAD, xx = px - PAD;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = *reinterpret_cast<const uint4 *>(x + (((int64_t)b * H + yy) * W + xx) * cin + cib * 64 + c8 * 8);
            *reinterpret_cast<uint4 *>(sx + ((size_t)ry * Wx + px) * 64 + c8 * 8) = v;
        }
        for (int i = tid; i < R * Wp * 8; i += 256) {
            const int c8 = i & 7, px = (i >> 3) % Wp, ry = (i >> 3) / Wp;
            const int yy = y0 + ry;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (yy < H && px < W) v = *reinterpret_cast<const uint4 *>(dy + (((int64_t)b * H + yy) * W + px) * cout + cob * 64 + c8 * 8);
            *reinterpret_cast<uint4 *>(sdy + ((size_t)ry * Wp + px) * 64 + c8 * 8) = v;
        }
        __syncthreads();
        for (int ry = 0; ry < R; ++ry)
            for (int x0 = 0; x0 < Wp; x0 += 4) {  // one k step = 4 consecutive pixels of a row; lane group g takes pixel x0 + g
                float a[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) a[c] = bf16_to_f32(sdy[((size_t)ry * Wp + x0 + g) * 64 + c * 16 + r16]);
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const int ky = t / KS, kx = t % KS;
                    const float bv = bf16_to_f32(sx[((size_t)(ry + ky) * Wx + x0 + g + kx) * 64 + wave * 16 + r16]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c], bv, acc[c][t], 0, 0, 0);
                }
            }
    }
    // slab of this workgroup: [tap][co 64][ci 64] fp32; D layout: lane holds ci = r16 (column), couts 4 g .. 4 g + 3 (rows) of each tile
    float *slab = slabs + ((((size_t)blockIdx.x * gridDim.y + cob) * gridDim.z + cib) * TAPS) * 4096;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) slab[(size_t)t * 4096 + (c * 16 + 4 * g + q) * 64 + wave * 16 + r16] = acc[c][t][q];
}

// dW[co][ci][tap] (OIHW, fp32) = sum over the walkers' slabs, in walker order
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float *__restrict__ slabs, int nwalk, int ncob, int ncib, int taps, int cin, int cout, float *__restrict__ dw) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t per = (int64_t)ncob * ncib * taps * 4096;
    if (i >= per) return;
    float s = 0.f;
    for (int w = 0; w < nwalk; ++w) s += slabs[(size_t)w * per + i];
    const int ci_l = (int)(i & 63), co_l = (int)((i >> 6) & 63), t = (int)((i >> 12) % taps), blk = (int)(i / ((int64_t)taps * 4096));
    const int cib = blk % ncib, cob = blk / ncib;
    const int co = cob * 64 + co_l, ci = cib * 64 + ci_l;
    if (co < cout && ci < cin) dw[((size_t)co * cin + ci) * taps + t] = s;
}

// ---- assembled-chain pieces (round 4): device-side weight packing (no per-call host repack / synchronisation), the training forward that keeps
//      the pre-activation, SiLU forward / backward, bias gradient
// bf16 round-to-nearest-even of an fp32 value (host_to_bf16's bit arithmetic)
__device__ __forceinline__ unsigned short dev_to_bf16(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float dev_from_bf16(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// fp32 OIHW master weights (device) -> the forward kernel's bf16 A-operand fragment order [cout block][stage][k step][fragment][lane][8]
// (pack_conv_weights' index arithmetic, element for element).  tflip = 1: the dgrad form -- the logical conv has the channel roles swapped and
// the taps flipped: W'[ci][co][t] = W[co][ci][taps - 1 - t].
__global__ __launch_bounds__(256) void k_pack_conv_bf16(const float *__restrict__ w, int coutL, int cinL, int ks, int CK, int NF, int tflip, unsigned short *__restrict__ out,
                                                       int64_t total) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= total) return;
    const int cpk = CK / 8, taps = ks * ks, kst = ((ks == 3 ? 9 : 1) * cpk + 3) / 4, nstage = (cinL + CK - 1) / CK;
    const int j = (int)(o & 7), lane = (int)((o >> 3) & 63);
    int64_t rest = o >> 9;
    const int f = (int)(rest % NF); rest /= NF;
    const int k = (int)(rest % kst); rest /= kst;
    const int st = (int)(rest % nstage);
    const int cb = (int)(rest / nstage);
    const int r = lane & 15, gq = lane >> 4;
    const int co = cb * 16 * NF + (r >> 2) * 4 * NF + f * 4 + (r & 3);
    const int q = k * 4 + gq;
    const int tap = ks == 3 ? q / cpk : 0, c0 = ks == 3 ? (q % cpk) * 8 : q * 8;
    const int c = st * CK + c0 + j;
    float v = 0.f;
    if (co < coutL && c < cinL && tap < taps && (ks == 3 || c0 < CK))
        v = tflip ? w[((size_t)c * coutL + co) * taps + (taps - 1 - tap)] : w[((size_t)co * cinL + c) * taps + tap];
    out[o] = dev_to_bf16(v);
}

__global__ __launch_bounds__(256) void k_silu_bf16(const unsigned short *__restrict__ z, unsigned short *__restrict__ a, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = dev_from_bf16(z[i]);
    a[i] = dev_to_bf16(x / (1.0f + expf(-x)));
}

// dz = da * silu'(z), silu'(z) = s (1 + z (1 - s)), s = sigmoid(z): what autograd computes for x * sigmoid(x); bf16 in / out, fp32 inside
__global__ __launch_bounds__(256) void k_silu_bwd_bf16(const unsigned short *__restrict__ z, const unsigned short *__restrict__ da, unsigned short *__restrict__ dz, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = dev_from_bf16(z[i]), g = dev_from_bf16(da[i]);
    const float sg = 1.0f / (1.0f + expf(-x));
    dz[i] = dev_to_bf16(g * (sg * (1.0f + x * (1.0f - sg))));
}

// db[c] = sum over the pixels of dy[pixel][c]: a workgroup per 64-channel column block, fp32 partial sums per thread row, fixed-order tree
__global__ __launch_bounds__(256) void k_bias_grad_bf16(const unsigned short *__restrict__ dy, int64_t npix, int cout, float *__restrict__ db) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rowg = threadIdx.x >> 6;
    float s = 0.f;
    if (c < cout)
        for (int64_t p = rowg; p < npix; p += 4) s += dev_from_bf16(dy[p * cout + c]);
    part[rowg][threadIdx.x & 63] = s;
    __syncthreads();
    if (rowg == 0 && c < cout) db[c] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

}  // namespace obb

using namespace obb;

extern "C" {

int obb_conv_dgrad_bf16(obb_ctx *ctx, const uint16_t *dy, const float *w_oihw_host, int32_t B, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t ks,
                        uint16_t *dx, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && B >= 0 && H > 0 && W > 0 && (ks == 1 || ks == 3), "obb_conv_dgrad_bf16: bad arguments");
    OBB_REQUIRE(ctx, cin % 8 == 0 && cout % 8 == 0 && cin >= 8 && cout >= 8, "obb_conv_dgrad_bf16: channel counts must be multiples of 8");
    if (B == 0) return OBB_OK;
    OBB_REQUIRE(ctx, dy && w_oihw_host && dx, "obb_conv_dgrad_bf16: NULL buffer");
    hipStream_t st = (hipStream_t)s;
    const int taps = ks * ks;
    // W'[ci][co][ky][kx] = W[co][ci][k - 1 - ky][k - 1 - kx]: the forward conv of dY with W' is dX
    std::vector<float> wt((size_t)cin * cout * taps);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < taps; ++t) wt[((size_t)ci * cout + co) * taps + (taps - 1 - t)] = w_oihw_host[((size_t)co * cin + ci) * taps + t];
    const ConvTiling t = plan_conv(ks, 1, cout, cin, H, W, false);
    const std::vector<bf16_t> pk = pack_conv_weights(wt.data(), cin, cout, ks, t, nullptr, 0, false);
    const size_t wbytes = pk.size() * sizeof(bf16_t), bbytes = ((size_t)cin + 63) / 64 * 64 * 4 + 256;
    // [packed weights][zero bias][2 KiB: the forward kernel's `lut` argument -- unused as a table here (no uint8 input), but lanes without an
    //  output pixel store into lut + 512 B .. + 1.5 KiB (conv.hip EXACT: every store unconditional), so it must be real memory]
    const size_t woff = (wbytes + 255) & ~(size_t)255, boff = (bbytes + 255) & ~(size_t)255;
    char *ws = (char *)ctx->workspace(WS_TRAIN_B, woff + boff + 2048);
    if (!ws) return set_error(ctx, OBB_ERR_HIP, "obb_conv_dgrad_bf16: workspace allocation failed");
    float *bias = (float *)(ws + woff);
    const bf16_t *sink = (const bf16_t *)(ws + woff + boff);
    OBB_HIP(ctx, hipMemcpyAsync(ws, pk.data(), wbytes, hipMemcpyHostToDevice, st));
    OBB_HIP(ctx, hipMemsetAsync(bias, 0, bbytes, st));
    OBB_HIP(ctx, hipStreamSynchronize(st));  // (the packed weights live in host memory that goes out of scope)
    ConvLaunch L;
    L.in.p = (void *)dy; L.in.bs = (int64_t)H * W * cout; L.in.cs = cout; L.in.co = 0;
    L.out.p = (void *)dx; L.out.bs = (int64_t)H * W * cin; L.out.cs = cin; L.out.co = 0;
    L.wpk = (const bf16_t *)ws; L.bias = bias; L.lut = sink;
    L.B = B; L.Hin = L.Hout = H; L.Win = L.Wout = W; L.cin = cout; L.cout = cin; L.ks = ks; L.stride = 1; L.act = 0; L.f16 = 0;
    L.TH = t.TH; L.TW = t.TW; L.MF = t.MF; L.NF = t.NF; L.CK = t.CK;
    L.tiles_y = (H + t.TH - 1) / t.TH; L.tiles_x = (W + t.TW - 1) / t.TW;
    if (ks == 1) {  // 1x1: batch x pixels is one dense pixel row (as the forward engine issues it)
        const int64_t npx = (int64_t)B * H * W;
        OBB_REQUIRE(ctx, npx < (1ll << 31) / 4, "obb_conv_dgrad_bf16: too many pixels for one launch");
        L.B = 1; L.Hin = L.Hout = 1; L.Win = L.Wout = (int)npx;
        L.tiles_y = 1; L.tiles_x = (int)((npx + L.TW - 1) / L.TW);
    }
    hipError_t e = launch_conv(L, st);
    if (e != hipSuccess) return set_error(ctx, OBB_ERR_HIP, "obb_conv_dgrad_bf16: launch failed: %s", hipGetErrorString(e));
    return OBB_OK;
}

int obb_conv_wgrad_bf16(obb_ctx *ctx, const uint16_t *x, const uint16_t *dy, int32_t B, int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t ks, float *dw,
                        obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && B >= 1 && H > 0 && W > 0 && (ks == 1 || ks == 3), "obb_conv_wgrad_bf16: bad arguments");
    OBB_REQUIRE(ctx, cin % 64 == 0 && cout % 64 == 0, "obb_conv_wgrad_bf16: channel counts must be multiples of 64 (one workgroup = a 64 x 64 block of dW)");
    OBB_REQUIRE(ctx, x && dy && dw, "obb_conv_wgrad_bf16: NULL buffer");
    hipStream_t st = (hipStream_t)s;
    const int taps = ks * ks, pad = ks / 2;
    const int Wp = (W + 3) & ~3, Wx = Wp + 2 * pad;
    int R = 1;  // rows per tile: as many as 64 KiB of LDS hold
    while (R < H && ((size_t)(R + 1 + 2 * pad) * Wx + (size_t)(R + 1) * Wp) * 128 <= 64 * 1024) ++R;
    const size_t lds = ((size_t)(R + 2 * pad) * Wx + (size_t)R * Wp) * 128;
    OBB_REQUIRE(ctx, lds <= 64 * 1024, "obb_conv_wgrad_bf16: a row of %d pixels does not fit the LDS tile", W);
    const int ncob = cout / 64, ncib = cin / 64;
    const int ntiles = B * ((H + R - 1) / R);
    int nwalk = std::max(1, std::min(ntiles, 512 / (ncob * ncib)));
    const size_t per = (size_t)ncob * ncib * taps * 4096;
    float *slabs = (float *)ctx->workspace(WS_TRAIN_C, (size_t)nwalk * per * 4);
    if (!slabs) return set_error(ctx, OBB_ERR_HIP, "obb_conv_wgrad_bf16: workspace allocation failed");
    const dim3 grid((unsigned)nwalk, (unsigned)ncob, (unsigned)ncib);
    if (ks == 3) hipLaunchKernelGGL((k_conv_wgrad<3>), grid, dim3(256), lds, st, x, dy, (int)B, (int)H, (int)W, (int)cin, (int)cout, R, slabs);
    else hipLaunchKernelGGL((k_conv_wgrad<1>), grid, dim3(256), lds, st, x, dy, (int)B, (int)H, (int)W, (int)cin, (int)cout, R, slabs);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)cdiv((int64_t)per, 256)), dim3(256), 0, st, slabs, nwalk, ncob, ncib, taps, (int)cin, (int)cout, dw);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

int obb_conv_packed_elems(obb_ctx *ctx, int32_t cout, int32_t cin, int32_t ks, int32_t H, int32_t W, int32_t dgrad_form, int64_t *n_elems) {
    OBB_REQUIRE(ctx, ctx && n_elems && cout > 0 && cin > 0 && (ks == 1 || ks == 3) && H > 0 && W > 0, "obb_conv_packed_elems: bad arguments");
    const int coutL = dgrad_form ? cin : cout, cinL = dgrad_form ? cout : cin;
    const ConvTiling t = plan_conv(ks, 1, cinL, coutL, H, W, false);
    const int nstage = (cinL + t.CK - 1) / t.CK, ncb = (coutL + 16 * t.NF - 1) / (16 * t.NF);
    *n_elems = (int64_t)ncb * nstage * conv_ksteps(ks, t.CK) * t.NF * 64 * 8;
    return OBB_OK;
}

int obb_conv_pack_bf16(obb_ctx *ctx, const float *w_oihw, int32_t cout, int32_t cin, int32_t ks, int32_t H, int32_t W, int32_t dgrad_form, uint16_t *packed,
                       obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && w_oihw && packed, "obb_conv_pack_bf16: NULL buffer");
    int64_t n = 0;
    int rc = obb_conv_packed_elems(ctx, cout, cin, ks, H, W, dgrad_form, &n);
    if (rc) return rc;
    const int coutL = dgrad_form ? cin : cout, cinL = dgrad_form ? cout : cin;
    const ConvTiling t = plan_conv(ks, 1, cinL, coutL, H, W, false);
    hipLaunchKernelGGL(k_pack_conv_bf16, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, w_oihw, coutL, cinL, (int)ks, t.CK, t.NF, dgrad_form ? 1 : 0, packed, n);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

int obb_conv_fwd_bf16(obb_ctx *ctx, const uint16_t *x, const uint16_t *packed_w, const float *bias, int32_t B, int32_t H, int32_t W, int32_t cin, int32_t cout,
                      int32_t ks, uint16_t *y, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && B >= 0 && H > 0 && W > 0 && (ks == 1 || ks == 3), "obb_conv_fwd_bf16: bad arguments");
    OBB_REQUIRE(ctx, cin % 8 == 0 && cout % 8 == 0 && cin >= 8 && cout >= 8, "obb_conv_fwd_bf16: channel counts must be multiples of 8");
    if (B == 0) return OBB_OK;
    OBB_REQUIRE(ctx, x && packed_w && y, "obb_conv_fwd_bf16: NULL buffer");
    hipStream_t st = (hipStream_t)s;
    const ConvTiling t = plan_conv(ks, 1, cin, cout, H, W, false);
    // [bias padded to the kernel's 64-float granule (zeros without one)][2 KiB: the `lut` argument = the sink of the kernel's unconditional stores]
    const size_t bbytes = (((size_t)cout + 63) / 64 * 64 * 4 + 256 + 255) & ~(size_t)255;
    char *ws = (char *)ctx->workspace(WS_TRAIN_A, bbytes + 2048);
    if (!ws) return set_error(ctx, OBB_ERR_HIP, "obb_conv_fwd_bf16: workspace allocation failed");
    OBB_HIP(ctx, hipMemsetAsync(ws, 0, bbytes, st));
    if (bias) OBB_HIP(ctx, hipMemcpyAsync(ws, bias, (size_t)cout * 4, hipMemcpyDeviceToDevice, st));
    ConvLaunch L;
    L.in.p = (void *)x; L.in.bs = (int64_t)H * W * cin; L.in.cs = cin; L.in.co = 0;
    L.out.p = (void *)y; L.out.bs = (int64_t)H * W * cout; L.out.cs = cout; L.out.co = 0;
    L.wpk = (const bf16_t *)packed_w; L.bias = (const float *)ws; L.lut = (const bf16_t *)(ws + bbytes);
    L.B = B; L.Hin = L.Hout = H; L.Win = L.Wout = W; L.cin = cin; L.cout = cout; L.ks = ks; L.stride = 1; L.act = 0; L.f16 = 0;
    L.TH = t.TH; L.TW = t.TW; L.MF = t.MF; L.NF = t.NF; L.CK = t.CK;
    L.tiles_y = (H + t.TH - 1) / t.TH; L.tiles_x = (W + t.TW - 1) / t.TW;
    if (ks == 1) {
        const int64_t npx = (int64_t)B * H * W;
        OBB_REQUIRE(ctx, npx < (1ll << 31) / 4, "obb_conv_fwd_bf16: too many pixels for one launch");
        L.B = 1; L.Hin = L.Hout = 1; L.Win = L.Wout = (int)npx;
        L.tiles_y = 1; L.tiles_x = (int)((npx + L.TW - 1) / L.TW);
    }
    hipError_t e = launch_conv(L, st);
    if (e != hipSuccess) return set_error(ctx, OBB_ERR_HIP, "obb_conv_fwd_bf16: launch failed: %s", hipGetErrorString(e));
    return OBB_OK;
}

int obb_silu_bf16(obb_ctx *ctx, const uint16_t *z, uint16_t *a, int64_t n, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && n >= 0, "obb_silu_bf16: bad arguments");
    if (n == 0) return OBB_OK;
    OBB_REQUIRE(ctx, z && a, "obb_silu_bf16: NULL buffer");
    hipLaunchKernelGGL(k_silu_bf16, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, z, a, n);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

int obb_silu_bwd_bf16(obb_ctx *ctx, const uint16_t *z, const uint16_t *da, uint16_t *dz, int64_t n, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && n >= 0, "obb_silu_bwd_bf16: bad arguments");
    if (n == 0) return OBB_OK;
    OBB_REQUIRE(ctx, z && da && dz, "obb_silu_bwd_bf16: NULL buffer");
    hipLaunchKernelGGL(k_silu_bwd_bf16, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, z, da, dz, n);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

int obb_bias_grad_bf16(obb_ctx *ctx, const uint16_t *dy, int64_t npix, int32_t cout, float *db, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && npix >= 1 && cout >= 1, "obb_bias_grad_bf16: bad arguments");
    OBB_REQUIRE(ctx, dy && db, "obb_bias_grad_bf16: NULL buffer");
    hipLaunchKernelGGL(k_bias_grad_bf16, dim3((unsigned)((cout + 63) / 64)), dim3(256), 0, (hipStream_t)s, dy, npix, (int)cout, db);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

}  // extern "C"
