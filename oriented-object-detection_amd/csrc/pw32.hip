// fp32 1x1 convolution (Conv 1x1 + folded BN + SiLU [+ residual] of the OBBModel forward: the C3k2 / C3k / SPPF / C2PSA entry and exit
// convs, Detect_OBB.py:79-83 -> ultralytics layers 4-22) with the ACTIVATIONS READ STRAIGHT FROM GLOBAL MEMORY into the MFMA B operand.
//
// k_conv_f32 stages both operands of a 1x1 layer through LDS: every activation byte is written to LDS and read back for 64 output channels
// only -- 1/9 of a 3x3 layer's reuse -- with a barrier / commit / barrier bubble per 64-channel stage; the 1x1 layers with >= 128 input
// channels sat at 83 TFLOP/s where the 3x3 layers reach 100-114.  Here:
//   * a workgroup owns ONE 64-cout block: its weights (K x 64 floats, <= 80 KB) go to LDS once, in A-fragment order, and stay there while the
//     workgroup walks pixel tiles -- the only barrier of the kernel is behind that load;
//   * a lane's B operand of a 16-channel piece is one 16-byte load of ITS pixel (lane = pixel pl, chunk g): exactly the operand layout, so
//     nothing is staged, committed or synchronised; a wave owns MFM fragments of 16 pixels x 4 cout fragments, i.e. 4 LDS reads (weights)
//     + MFM global loads per 16 MFM MFMAs, and runs free of the other waves;
//   * the loads of piece p + 2 are issued behind the MFMAs of piece p (two register sets, the piece loop unrolled by two so that no register
//     is copied: a rotation `cur = next` would wait for the youngest load, see profiles/r03_summary.md);
//   * k order = (piece, element s) exactly as k_conv_f32's 64-channel stages: BIT-IDENTICAL results (test).
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
    int npix, npiece, cout, act, ncb, ntiles, tstep;
};

__device__ __forceinline__ float silu32p(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }  // (= silu32 of f32path.hip)

template <int NW, int MFM>
__global__ __launch_bounds__(NW * 64, 4) void k_pw_f32(const Pw32Params P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [piece][cout fragment 4][lane][16 B]
    constexpr int NT = NW * 64, NCF = 4, TPX = NW * MFM * 16;
    constexpr unsigned NOPIX = 0xffffffffu;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, pl = lane & 15;
    // XCD-aware order (as k_conv_f32): the cout blocks of one pixel range are consecutive on one XCD and share its lines through that L2
    const int xcd = blockIdx.x & 7, lin = blockIdx.x >> 3;
    const int cb = lin % P.ncb;
    int t = (lin / P.ncb) * 8 + xcd;
    if (t >= P.ntiles) return;
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
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(P.in + P.in_co), 0, (int)P.in_span, 0x00020000);
    const char *wl = smem + lane * 16;
    const int cbase = cb * 64 + 16 * g;  // this lane's 16 consecutive output channels
    for (; t < P.ntiles; t += P.tstep) {
        const int p0 = t * TPX + wave * (MFM * 16);
        if (p0 >= P.npix) continue;  // (wave-uniform: no barrier below)
        unsigned goff[MFM];
#pragma unroll
        for (int f = 0; f < MFM; ++f) {
            const int p = p0 + f * 16 + pl;
            goff[f] = p < P.npix ? (unsigned)(((int64_t)p * P.in_cs + g * 4) * 4) : NOPIX;
        }
        f32x4 acc[NCF][MFM];
#pragma unroll
        for (int nf = 0; nf < NCF; ++nf)
#pragma unroll
            for (int f = 0; f < MFM; ++f) acc[nf][f] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 b0[MFM], b1[MFM];
#pragma unroll
        for (int f = 0; f < MFM; ++f) b0[f] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, goff[f], 0, 0);
#pragma unroll
        for (int f = 0; f < MFM; ++f) b1[f] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, goff[f] == NOPIX ? NOPIX : goff[f] + 64u, 0, 0);
        for (int pc = 0; pc < P.npiece; pc += 2) {  // (npiece is even)
            {
                f32x4 w[NCF];
#pragma unroll
                for (int nf = 0; nf < NCF; ++nf) w[nf] = *reinterpret_cast<const f32x4 *>(wl + (size_t)(pc * NCF + nf) * 1024);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int f = 0; f < MFM; ++f)
#pragma unroll
                        for (int nf = 0; nf < NCF; ++nf) acc[nf][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nf][s], __builtin_bit_cast(f32x4, b0[f])[s], acc[nf][f], 0, 0, 0);
                const unsigned add = (unsigned)(pc + 2) * 64u;
                const bool more = pc + 2 < P.npiece;  // (unconditional loads: past the last piece they read nothing -- offset out of range)
#pragma unroll
                for (int f = 0; f < MFM; ++f) b0[f] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (more && goff[f] != NOPIX) ? goff[f] + add : NOPIX, 0, 0);
            }
            {
                f32x4 w[NCF];
#pragma unroll
                for (int nf = 0; nf < NCF; ++nf) w[nf] = *reinterpret_cast<const f32x4 *>(wl + (size_t)((pc + 1) * NCF + nf) * 1024);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int f = 0; f < MFM; ++f)
#pragma unroll
                        for (int nf = 0; nf < NCF; ++nf) acc[nf][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nf][s], __builtin_bit_cast(f32x4, b1[f])[s], acc[nf][f], 0, 0, 0);
                const unsigned add = (unsigned)(pc + 3) * 64u;
                const bool more = pc + 3 < P.npiece;
#pragma unroll
                for (int f = 0; f < MFM; ++f) b1[f] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (more && goff[f] != NOPIX) ? goff[f] + add : NOPIX, 0, 0);
            }
        }
        // ---- epilogue: + bias, SiLU, + residual, store (a lane: 16 consecutive channels of its pixel)
#pragma unroll
        for (int f = 0; f < MFM; ++f) {
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

template <int NW, int MFM>
static hipError_t launch_pw(const Pw32Params &P0, size_t lds, hipStream_t st) {
    const void *fn = (const void *)k_pw_f32<NW, MFM>;
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
    constexpr int TPX = NW * MFM * 16;
    P.ntiles = (int)(((int64_t)P.npix + TPX - 1) / TPX);
    // a grid of at most the resident workgroups, in units of 8 pixel tiles x ncb cout blocks (the XCD-aware order); each walks t, t + tstep, ...
    const int64_t tiles8 = ((int64_t)P.ntiles + 7) / 8;
    const int64_t slots = std::max<int64_t>(1, std::min<int64_t>(tiles8, resident / (8 * P.ncb)));
    P.tstep = (int)(slots * 8);
    hipLaunchKernelGGL((k_pw_f32<NW, MFM>), dim3((unsigned)(slots * 8 * P.ncb)), dim3(NW * 64), lds, st, P);
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
    if (span <= 0 || span >= (1ll << 32) - 65536) return hipErrorInvalidValue;
    P.in_span = (unsigned)span;
    P.out = (float *)L.out.p; P.out_bs = L.out.bs; P.out_cs = L.out.cs; P.out_co = L.out.co; P.out_blk = L.out.cpb ? 1 : 0; P.out_ps = (int)L.out.ps; P.hw = L.hw;
    P.res = (const float *)L.res.p; P.res_cs = L.res.cs; P.res_co = L.res.co;
    P.wpk = L.wpk; P.bias = L.bias;
    P.npix = (int)L.npix; P.npiece = L.cin / 16; P.cout = L.cout; P.act = L.act; P.ncb = L.cout / 64;
    const size_t lds = (size_t)P.npiece * 4096;
    // K <= 320: 8-wave workgroups, two per CU; above: one 16-wave workgroup per CU (its weights alone are up to 128 KB)
    if (lds <= 80 * 1024) return launch_pw<8, 3>(P, lds, st);
    return launch_pw<16, 3>(P, lds, st);
}

}  // namespace obb
