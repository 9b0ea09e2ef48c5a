// fp32 1x1 convolution (Conv 1x1 + folded BN + SiLU [+ residual] of the OBBModel forward: the C3k2 / C3k / SPPF / C2PSA entry and exit
// convs, Detect_OBB.py:79-83 -> ultralytics layers 4-22) with the ACTIVATIONS READ STRAIGHT FROM GLOBAL MEMORY into the MFMA B operand.
//
// k_conv_f32 stages both operands of a 1x1 layer through LDS: every activation byte is written to LDS and read back for 64 output channels
// only -- 1/9 of a 3x3 layer's reuse -- with a barrier / commit / barrier bubble per 64-channel stage; the 1x1 layers with >= 128 input
// channels sat at 83 TFLOP/s where the 3x3 layers reach 100-114.  Here:
//   * a workgroup owns ONE 64-cout block: its weights (K x 64 floats, <= 80 KB) go to LDS once, in A-fragment order, and stay there while the
//     workgroup walks pixel tiles -- the only barrier of the kernel is behind that load;
//   * a lane's B operand of a 16-channel piece is one 16-byte load of ITS pixel (lane = pixel pl, chunk g): exactly the operand layout, so
//     nothing is staged, committed or synchronised; a wave's tile is 3 fragments of 16 pixels x 4 cout fragments, i.e. 4 LDS reads (weights)
//     + 3 global loads per 48 MFMAs, and it runs free of the other waves: the fragments are dealt out per wave in contiguous ranges
//     (a 2- / 1-fragment tail tile at the end) and the next tile's first two pieces are loaded under this tile's epilogue;
//   * the loads of piece p + 2 are issued behind the MFMAs of piece p (two register sets, the piece loop unrolled by two so that no register
//     is copied: a rotation `cur = next` would wait for the youngest load, see profiles/r03_summary.md);
//   * k order = (piece, element s) exactly as k_conv_f32's 64-channel stages: BIT-IDENTICAL results (test).
// Measured (profiles/r04_summary.md): the 25 layers 3.65 -> 3.3 ms per 512 tiles, forward 32.13 -> 31.45 ms per 1024 tiles.  Timing-only
// ablations of the 26 x 26 layers (177 us): without the in-loop loads 163, without the stores 161 -- neither the loads nor the stores are
// the bound; the 52 x 52 layers move 1.24 GB for 34 GFLOP (HBM and MFMA time about equal).  The two virtual upsample-concat layers were tried
// on this kernel and stay on k_conv_f32 (no gain: the coarse-pixel arithmetic costs the registers of the cross-tile prefetch).
// Row order of the weights (pack_pw32_weights): a lane ends with 16 consecutive output channels of its pixel (64 contiguous bytes / two
// 8-channel blocks).
#include "pw32.h"

#include <algorithm>
#include <map>
#include <mutex>

namespace obb {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct Pw32Params {
    const float *in; int in_cs, in_co; unsigned in_span;
    float *out; int64_t out_bs; int out_cs, out_co, out_blk, out_ps, hw;
    const float *res; int res_cs, res_co;
    const float *wpk, *bias;
    int npix, npiece, cout, act, ncb, nfrag, nwalk;
};

__device__ __forceinline__ float silu32p(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }  // (= silu32 of f32path.hip)

// One wave's tile: M fragments of 16 pixels (from pixel p0) x the workgroup's 64 output channels.  b0 / b1 arrive LOADED with pieces 0 / 1 of
// this tile and leave loaded with pieces 0 / 1 of the next one (its first `nxt` fragments, 48 pixels further on: a wave's range is contiguous),
// so that the epilogue of a tile runs under the first loads of the next and a wave never waits for a cold load between tiles.
// Pixels past the end of the tensor need no lane test on the way in: their offsets are past the buffer range and read as zero.
template <int M>
__device__ __forceinline__ void pw_tile(const Pw32Params &P, const __amdgpu_buffer_rsrc_t in_rsrc, const char *wl, const int p0, const int pl, const int g,
                                        const int cbase, u32x4 (&b0)[3], u32x4 (&b1)[3], const int nxt) {
    constexpr int NCF = 4;
    constexpr unsigned NOPIX = 0xffffffffu;
    unsigned goff[M];
#pragma unroll
    for (int f = 0; f < M; ++f) goff[f] = (unsigned)(((int64_t)(p0 + f * 16 + pl) * P.in_cs + g * 4) * 4);
    f32x4 acc[NCF][M];
#pragma unroll
    for (int nf = 0; nf < NCF; ++nf)
#pragma unroll
        for (int f = 0; f < M; ++f) acc[nf][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned delta = (unsigned)P.in_cs * (48u * 4u);  // the same fragment of the next tile
    for (int pc = 0; pc < P.npiece; pc += 2) {  // (npiece is even)
        const bool last = pc + 2 >= P.npiece;
        {
            f32x4 w[NCF];
#pragma unroll
            for (int nf = 0; nf < NCF; ++nf) w[nf] = *reinterpret_cast<const f32x4 *>(wl + (size_t)(pc * NCF + nf) * 1024);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int f = 0; f < M; ++f)
#pragma unroll
                    for (int nf = 0; nf < NCF; ++nf) acc[nf][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nf][s], __builtin_bit_cast(f32x4, b0[f])[s], acc[nf][f], 0, 0, 0);
            const unsigned add = last ? delta : (unsigned)(pc + 2) * 64u;  // piece pc + 2 of this tile, or piece 0 of the next
#pragma unroll
            for (int f = 0; f < M; ++f) b0[f] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (!last || f < nxt) ? goff[f] + add : NOPIX, 0, 0);
        }
        {
            f32x4 w[NCF];
#pragma unroll
            for (int nf = 0; nf < NCF; ++nf) w[nf] = *reinterpret_cast<const f32x4 *>(wl + (size_t)((pc + 1) * NCF + nf) * 1024);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int f = 0; f < M; ++f)
#pragma unroll
                    for (int nf = 0; nf < NCF; ++nf) acc[nf][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nf][s], __builtin_bit_cast(f32x4, b1[f])[s], acc[nf][f], 0, 0, 0);
            const unsigned add = last ? delta + 64u : (unsigned)(pc + 3) * 64u;
#pragma unroll
            for (int f = 0; f < M; ++f) b1[f] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (!last || f < nxt) ? goff[f] + add : NOPIX, 0, 0);
        }
    }
    // ---- epilogue: + bias, SiLU, + residual, store (a lane: 16 consecutive channels of its pixel)
#pragma unroll
    for (int f = 0; f < M; ++f) {
        const int p = p0 + f * 16 + pl;
        if (p >= P.npix) continue;
        float *op;
        if (P.out_blk) {
            const int ob = p / P.hw, opx = p - ob * P.hw, ca = P.out_co + cbase;
            op = P.out + (int64_t)ob * P.out_bs + (int64_t)(ca >> 3) * P.out_ps + (int64_t)opx * 8 + (ca & 7);
        } else op = P.out + (int64_t)p * P.out_cs + P.out_co + cbase;
        const float *rp = P.res ? P.res + (int64_t)p * P.res_cs + P.res_co + cbase : nullptr;
#pragma unroll
        for (int nf = 0; nf < NCF; ++nf) {
            const float4 bv = *reinterpret_cast<const float4 *>(P.bias + cbase + 4 * nf);
            float v[4] = {acc[nf][f][0] + bv.x, acc[nf][f][1] + bv.y, acc[nf][f][2] + bv.z, acc[nf][f][3] + bv.w};
            if (P.act) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = silu32p(v[j]);
            }
            if (rp) {
                const float4 rv = *reinterpret_cast<const float4 *>(rp + 4 * nf);
                v[0] = rv.x + v[0]; v[1] = rv.y + v[1]; v[2] = rv.z + v[2]; v[3] = rv.w + v[3];
            }
            float *o4 = P.out_blk ? op + (int64_t)(nf >> 1) * P.out_ps + 4 * (nf & 1) : op + 4 * nf;
            *reinterpret_cast<float4 *>(o4) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// The pixel fragments (16 pixels each) are dealt out per WAVE: wave w of walker t (the workgroups of one walker index, one per cout block, sit on
// one XCD and read the same lines through its L2) owns the contiguous range [ww F / NWV, (ww + 1) F / NWV), ww = t NW + w, and walks it three
// fragments at a time, the rest as a 2- or 1-fragment tile.  (Whole 384-pixel tiles dealt out per workgroup left 3.5 tiles per workgroup on
// the 26 x 26 levels -- a quarter of the chip idle during the fourth; per wave it is 10.6 fragments against a maximum of 11.)
template <int NW>
__global__ __launch_bounds__(NW * 64, 4) void k_pw_f32(const Pw32Params P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [piece][cout fragment 4][lane][16 B]
    constexpr int NT = NW * 64, NCF = 4;
    constexpr unsigned NOPIX = 0xffffffffu;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, pl = lane & 15;
    // XCD-aware order (as k_conv_f32): the cout blocks of one pixel range are consecutive on one XCD
    const int xcd = blockIdx.x & 7, lin = blockIdx.x >> 3;
    const int cb = lin % P.ncb;
    const int t = (lin / P.ncb) * 8 + xcd;  // walker index, < nwalk (the grid is exactly nwalk x ncb)
    const int64_t ww = (int64_t)t * NW + wave, nwv = (int64_t)P.nwalk * NW;
    int fr = (int)(ww * P.nfrag / nwv);
    const int fr_end = (int)((ww + 1) * P.nfrag / nwv);
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(P.in + P.in_co), 0, (int)P.in_span, 0x00020000);
    u32x4 b0[3], b1[3];  // the first tile's pieces 0 / 1: in flight under the weight load
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        const unsigned off = fr + f < fr_end ? (unsigned)(((int64_t)((fr + f) * 16 + pl) * P.in_cs + g * 4) * 4) : NOPIX;
        b0[f] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0);
        b1[f] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off == NOPIX ? NOPIX : off + 64u, 0, 0);
    }
    {   // this cout block's weights -> LDS, once
        const int nchunk = P.npiece * NCF * 64;
        const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(P.wpk + (size_t)cb * nchunk * 4), 0, nchunk * 16, 0x00020000);
        for (int i0 = 0; i0 < nchunk; i0 += 4 * NT) {
            u32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (unsigned)min(i0 + k * NT + tid, nchunk - 1) * 16u, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i0 + k * NT + tid < nchunk) *reinterpret_cast<u32x4 *>(smem + (size_t)(i0 + k * NT + tid) * 16) = v[k];
        }
    }
    __syncthreads();
    const char *wl = smem + lane * 16;
    const int cbase = cb * 64 + 16 * g;  // this lane's 16 consecutive output channels
    for (; fr_end - fr >= 3; fr += 3) pw_tile<3>(P, in_rsrc, wl, fr * 16, pl, g, cbase, b0, b1, min(3, fr_end - fr - 3));
    if (fr_end - fr == 2) pw_tile<2>(P, in_rsrc, wl, fr * 16, pl, g, cbase, b0, b1, 0);
    else if (fr_end - fr == 1) pw_tile<1>(P, in_rsrc, wl, fr * 16, pl, g, cbase, b0, b1, 0);
}

// ------------------------------------------------------------------------------------------------ host side
bool pw32_supported(int cin, int cout) { return cin >= 64 && cin % 32 == 0 && cin <= 512 && cout >= 64 && cout % 64 == 0; }

std::vector<float> pack_pw32_weights(const float *w, int cout, int cin, const int *perm) {
    const int npiece = cin / 16, ncb = cout / 64;
    std::vector<float> out((size_t)ncb * npiece * 4 * 64 * 4);
    size_t o = 0;
    for (int cb = 0; cb < ncb; ++cb)
        for (int pc = 0; pc < npiece; ++pc)
            for (int f = 0; f < 4; ++f)
                for (int lane = 0; lane < 64; ++lane) {
                    const int r = lane & 15, g = lane >> 4;
                    const int co = cb * 64 + 16 * (r >> 2) + 4 * f + (r & 3);
                    const int src = perm ? perm[co] : co;
                    for (int s = 0; s < 4; ++s) out[o++] = w[(size_t)src * cin + 16 * pc + 4 * g + s];
                }
    return out;
}

template <int NW>
static hipError_t launch_pw(const Pw32Params &P0, size_t lds, hipStream_t st) {
    const void *fn = (const void *)k_pw_f32<NW>;
    static std::mutex mu;
    static std::map<std::pair<int, size_t>, int> occ;  // (device, LDS bytes) -> resident workgroups on the chip
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    int resident = 0;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = occ.find({dev, lds});
        if (it == occ.end()) {
            int n = 0, ncu = 0;
            if ((e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024)) != hipSuccess) return e;
            if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, NW * 64, lds)) != hipSuccess) return e;
            if ((e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
            it = occ.emplace(std::make_pair(dev, lds), std::max(1, n) * std::max(1, ncu)).first;
        }
        resident = it->second;
    }
    Pw32Params P = P0;
    P.nfrag = (int)(((int64_t)P.npix + 15) / 16);
    // walkers (workgroups per cout block): at most the resident workgroups, at least three fragments per wave, in units of 8 (the XCD order)
    const int64_t want8 = ((int64_t)P.nfrag + 3 * NW * 8 - 1) / (3 * NW * 8);
    const int64_t slots = std::max<int64_t>(1, std::min<int64_t>(want8, resident / (8 * P.ncb)));
    P.nwalk = (int)(slots * 8);
    hipLaunchKernelGGL((k_pw_f32<NW>), dim3((unsigned)(P.nwalk * P.ncb)), dim3(NW * 64), lds, st, P);
    return hipGetLastError();
}

hipError_t launch_pw32(const Pw32Launch &L, hipStream_t st) {
    if (!pw32_supported(L.cin, L.cout) || L.in.cpb || L.res.cpb || !L.in.p || !L.out.p || !L.wpk || !L.bias || L.npix < 1 || L.npix >= (1ll << 31)) return hipErrorInvalidValue;
    if ((L.in.cs | L.in.co) & 3 || (L.res.p && ((L.res.cs | L.res.co) & 3))) return hipErrorInvalidValue;
    if (L.out.cpb && !(L.out.cpb == 2 && L.out.cs == 8 && L.out.co % 8 == 0 && L.out.ps > 0 && L.hw > 0)) return hipErrorInvalidValue;
    if (!L.out.cpb && ((L.out.cs | L.out.co) & 3)) return hipErrorInvalidValue;
    Pw32Params P;
    P.in = (const float *)L.in.p; P.in_cs = L.in.cs; P.in_co = L.in.co;
    const int64_t span = (L.npix * L.in.cs - L.in.co) * 4;
    if (span <= 0 || span >= (1ll << 32) - (1ll << 20)) return hipErrorInvalidValue;  // (a wave addresses up to 63 pixels past the end)
    P.in_span = (unsigned)span;
    P.out = (float *)L.out.p; P.out_bs = L.out.bs; P.out_cs = L.out.cs; P.out_co = L.out.co; P.out_blk = L.out.cpb ? 1 : 0; P.out_ps = (int)L.out.ps; P.hw = L.hw;
    P.res = (const float *)L.res.p; P.res_cs = L.res.cs; P.res_co = L.res.co;
    P.wpk = L.wpk; P.bias = L.bias;
    P.npix = (int)L.npix; P.npiece = L.cin / 16; P.cout = L.cout; P.act = L.act; P.ncb = L.cout / 64;
    const size_t lds = (size_t)P.npiece * 4096;
    // K <= 320: 8-wave workgroups, two per CU; above: one 16-wave workgroup per CU (its weights alone are up to 128 KB)
    if (lds <= 80 * 1024) return launch_pw<8>(P, lds, st);
    return launch_pw<16>(P, lds, st);
}

}  // namespace obb
