// The inner C3k block of the stride-32 level (13x13 feature map, 64 hidden channels: model.8.m.0 / model.22.m.0 of YOLO11n) as ONE
// persistent workgroup per image.
//
// At this level a layer is a single 13x13 tile per image, i.e. one workgroup per CU: every separate launch costs >= 8-14 us of
// dispatch + dependent memory round trips for ~1 us of MFMA work, and the block is seven such launches.  Here the whole block
//     [cv1 | cv2] 1x1 128 -> 64 + 64   (a, b)
//     m.0: 3x3 64 -> 64, 3x3 64 -> 64 + shortcut ;  m.1: the same          (a -> a')
//     cv3: 1x1 [a' | b] -> 128
// runs with every activation resident in LDS (15x15 zero-bordered images for the tensors that feed 3x3 convs) and the weights streamed
// from L2 in 16 / 24-KiB chunks (4 or 6 k-steps x 4 fragments) through a 2-deep register queue and a double-buffered LDS slot: one
// barrier per chunk.  The chunk sequence is instantiated at compile time (16 chunks, all addresses constant), which also makes every
// wait for a prefetched chunk an exact `vmcnt`.  The Bottleneck shortcut is added in place (a lane re-writes the element it read).  MFMA mapping, weight-fragment order, k order (one channel stage) and rounding points are those of
// k_conv_igemm; 8 waves = 4 (three pixel fragments each) x 2 (halves of the 64 couts of a block).
// Replaces ultralytics C3k (nn/modules/block.py) inside C3k2(c3k=True) for this shape; SURVEY Appendix A3.
#include "c3kimg.h"

#include <cstdlib>
#include <type_traits>
#include <utility>

namespace obb {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct C3kImgParams {
    const bf16_t *in; int64_t in_bs; int in_cs, in_co;
    bf16_t *out; int64_t out_bs; int out_cs, out_co;
    const bf16_t *wts;  // c3k::kPieces 16-byte pieces: the six layers' MFMA fragments back to back
    const float *bias;  // [6][128]
};

namespace c3k {
constexpr int HW = 13, NPX = 169, BW = 15;            // image, pixels, bordered width
constexpr int P128 = 272, P64 = 144;                  // LDS bytes per pixel: 128 / 64 channels + 16 B pad
constexpr int IN_B = NPX * P128, BORD_B = BW * BW * P64, BB_B = NPX * P64, WBUF_B = 6 * 4096;
constexpr int OFF_IN = 0, OFF_A = IN_B, OFF_BB = OFF_A + BORD_B, OFF_W = OFF_BB + BB_B;
constexpr int LDS_B = OFF_W + 2 * WBUF_B;             // 151 856 B
constexpr int OFF_T = OFF_IN;                         // T (bordered, 64 ch) and the output staging re-use the input image's space
// chunk table.  L0 = [cv1|cv2] and L5 = cv3: one chunk per cout block (4 k-steps, 16 KiB); L1..L4 = 3x3: three chunks of 6 k-steps (24 KiB)
constexpr int kChunks = 2 + 12 + 2;
constexpr int DEPTH = 2;                              // register queue depth (chunks in flight from L2)
__host__ __device__ constexpr int layer_of(int ci) { return ci < 2 ? 0 : (ci < 14 ? 1 + (ci - 2) / 3 : 5); }
__host__ __device__ constexpr int first_of(int L) { return L == 0 ? 0 : (L == 5 ? 14 : 2 + (L - 1) * 3); }
__host__ __device__ constexpr int ksteps_of(int ci) { return (layer_of(ci) == 0 || layer_of(ci) == 5) ? 4 : 6; }
__host__ __device__ constexpr int piece_of(int ci) {  // first 16-B piece of the chunk in the weight stream (256 pieces per k-step)
    int p = 0;
    for (int c = 0; c < ci; ++c) p += ksteps_of(c) * 256;
    return p;
}
constexpr int kPieces = (2 * 4 + 12 * 6 + 2 * 4) * 256;
}  // namespace c3k

template <typename F, int... I>
__device__ __forceinline__ void for_each_chunk(F &step, std::integer_sequence<int, I...>) {
    (step(std::integral_constant<int, I>{}), ...);
}

template <bool F16>
__global__ __launch_bounds__(512) void k_c3k_image(const C3kImgParams P) {
    using namespace c3k;
    typedef typename HX<F16>::vec8 hx8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ __attribute__((aligned(16))) float s_bias[6 * 128];
    // 8 waves: wave & 3 owns three 16-pixel fragments, wave >> 2 owns one half (2 of 4 fragments = 32) of the 64 couts of a block
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, half = tid >> 8, g = lane >> 4, pl = lane & 15;
    const int b = blockIdx.x;
    const u32x4 *wsrc = reinterpret_cast<const u32x4 *>(P.wts);

    // weight queue: a chunk = 4 or 6 k-steps x 256 16-B pieces, one piece per thread and k-step
    u32x4 wq[DEPTH][3];
    auto fetch = [&](auto ci_const) {
        constexpr int ci = decltype(ci_const)::value;
#pragma unroll
        for (int k = 0; k < ksteps_of(ci) / 2; ++k) wq[ci % DEPTH][k] = wsrc[piece_of(ci) + k * 512 + tid];
    };
    fetch(std::integral_constant<int, 0>{});
    fetch(std::integral_constant<int, 1>{});
    // input image, bias, zero border of A
    {
        const bf16_t *src = P.in + (int64_t)b * P.in_bs + P.in_co;
        u32x4 img[6];  // 169 x 16 pieces / 512 threads: all loads in flight before the first use
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            int i = min(tid + k * 512, NPX * 16 - 1);
            img[k] = *reinterpret_cast<const u32x4 *>(src + (int64_t)(i >> 4) * P.in_cs + (i & 15) * 8);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            int i = tid + k * 512;
            if (i < NPX * 16) *reinterpret_cast<u32x4 *>(smem + OFF_IN + (i >> 4) * P128 + (i & 15) * 16) = img[k];
        }
        for (int i = tid; i < 6 * 128; i += 512) s_bias[i] = P.bias[i];
        for (int i = tid; i < (BW * BW - NPX) * 9; i += 512) {
            int cell = i / 9, c = i - cell * 9;
            // border cells of a 15x15 grid in raster order: row 0 (15), rows 1..13 (2 each), row 14 (15)
            int by, bx;
            if (cell < 15) { by = 0; bx = cell; }
            else if (cell < 15 + 26) { by = 1 + (cell - 15) / 2; bx = ((cell - 15) & 1) ? 14 : 0; }
            else { by = 14; bx = cell - 41; }
            *reinterpret_cast<u32x4 *>(smem + OFF_A + (by * BW + bx) * P64 + c * 16) = u32x4{0u, 0u, 0u, 0u};
        }
    }
    // per-lane pixel offsets of the wave's three fragments
    int o_in[3], o_bord[3], o_int[3], o_bb[3];
    bool pval[3];
#pragma unroll
    for (int mf = 0; mf < 3; ++mf) {
        int p = (wave * 3 + mf) * 16 + pl;
        pval[mf] = p < NPX;
        p = pval[mf] ? p : NPX - 1;
        const int ty = p / HW, tx = p - ty * HW;
        o_in[mf] = p * P128;
        o_bord[mf] = (ty * BW + tx) * P64;              // tap (0,0) of the 3x3 window in a bordered image
        o_int[mf] = ((ty + 1) * BW + tx + 1) * P64;     // the pixel itself in a bordered image
        o_bb[mf] = p * P64;
    }
    f32x4 acc[3][2];

    auto zero_acc = [&]() {
#pragma unroll
        for (int mf = 0; mf < 3; ++mf)
#pragma unroll
            for (int f = 0; f < 2; ++f) acc[mf][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // epilogue: + bias, SiLU, (+ residual from a bordered image), the 8 couts of this lane -> dst (16 B at dst_off[mf] + g*32 + half*16)
    auto epilogue = [&](int L, int cb, const int (&dst_off)[3], int dst_base, int res_base) {
#pragma unroll
        for (int mf = 0; mf < 3; ++mf) {
            if (!pval[mf]) continue;
            float v[8];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                float4 bv = *reinterpret_cast<const float4 *>(s_bias + L * 128 + cb * 64 + g * 16 + half * 8 + f * 4);
                v[f * 4 + 0] = silu_f(acc[mf][f][0] + bv.x); v[f * 4 + 1] = silu_f(acc[mf][f][1] + bv.y);
                v[f * 4 + 2] = silu_f(acc[mf][f][2] + bv.z); v[f * 4 + 3] = silu_f(acc[mf][f][3] + bv.w);
            }
            if (res_base >= 0) {
                uint4 rv = *reinterpret_cast<const uint4 *>(smem + res_base + o_int[mf] + g * 32 + half * 16);
                v[0] += HX<F16>::lo(rv.x); v[1] += HX<F16>::hi(rv.x); v[2] += HX<F16>::lo(rv.y); v[3] += HX<F16>::hi(rv.y);
                v[4] += HX<F16>::lo(rv.z); v[5] += HX<F16>::hi(rv.z); v[6] += HX<F16>::lo(rv.w); v[7] += HX<F16>::hi(rv.w);
            }
            uint4 o;
            o.x = HX<F16>::pack2(v[0], v[1]); o.y = HX<F16>::pack2(v[2], v[3]);
            o.z = HX<F16>::pack2(v[4], v[5]); o.w = HX<F16>::pack2(v[6], v[7]);
            *reinterpret_cast<uint4 *>(smem + dst_base + dst_off[mf] + g * 32 + half * 16) = o;
        }
    };

    // one chunk; instantiated for every chunk index (compile-time layer / addresses / queue slot)
    auto step = [&](auto ci_const) {
        constexpr int ci = decltype(ci_const)::value;
        constexpr int L = layer_of(ci), lc = ci - first_of(L);  // layer, chunk inside the layer
        // ---- this chunk's weights: registers -> LDS slot, then refill the register slot with chunk ci + DEPTH
        {
            char *wl = smem + OFF_W + (ci & 1) * WBUF_B;
#pragma unroll
            for (int k = 0; k < ksteps_of(ci) / 2; ++k) *reinterpret_cast<u32x4 *>(wl + (k * 512 + tid) * 16) = wq[ci % DEPTH][k];
        }
        __syncthreads();
        if constexpr (L == 1 && lc == 0) {
            // T re-uses the input image's space.  Every wave is past layer 0 now (this chunk's barrier), so the input is dead: zero T's
            // border here; its interior is written by this layer's epilogue and the whole image is first read by layer 2.
            for (int i = tid; i < (BW * BW - NPX) * 9; i += 512) {
                int cell = i / 9, c = i - cell * 9, by, bx;
                if (cell < 15) { by = 0; bx = cell; }
                else if (cell < 15 + 26) { by = 1 + (cell - 15) / 2; bx = ((cell - 15) & 1) ? 14 : 0; }
                else { by = 14; bx = cell - 41; }
                *reinterpret_cast<u32x4 *>(smem + OFF_T + (by * BW + bx) * P64 + c * 16) = u32x4{0u, 0u, 0u, 0u};
            }
        }
        if constexpr (ci + DEPTH < kChunks) fetch(std::integral_constant<int, ci + DEPTH>{});
        if constexpr (lc == 0 || L == 0 || L == 5) zero_acc();  // layer start / every cout block of the 1x1 layers
        // ---- the chunk's k-steps
        const char *wl = smem + OFF_W + (ci & 1) * WBUF_B;
        // software-pipelined: the operands of k-step ks+1 are requested from LDS before the MFMAs of k-step ks are issued
        hx8 wc[2][2], ac[2][3];
        auto fetch_ops = [&](auto ks_const) {
            constexpr int ks = decltype(ks_const)::value;
            constexpr int sl = ks & 1;
#pragma unroll
            for (int f = 0; f < 2; ++f) wc[sl][f] = *reinterpret_cast<const hx8 *>(wl + ((ks * 4 + half * 2 + f) * 64 + lane) * 16);
            if constexpr (L == 0) {  // 1x1 over the 128-channel input image: k-step kk of 4 -> chunk q = kk*4 + g
#pragma unroll
                for (int mf = 0; mf < 3; ++mf) ac[sl][mf] = *reinterpret_cast<const hx8 *>(smem + OFF_IN + o_in[mf] + (ks * 4 + g) * 16);
            } else if constexpr (L == 5) {  // 1x1 over [a' | b]: chunks 0..7 = a' (bordered image A), 8..15 = b
                const int q = ks * 4 + g;
#pragma unroll
                for (int mf = 0; mf < 3; ++mf)
                    ac[sl][mf] = ks < 2 ? *reinterpret_cast<const hx8 *>(smem + OFF_A + o_int[mf] + q * 16)
                                        : *reinterpret_cast<const hx8 *>(smem + OFF_BB + o_bb[mf] + (q - 8) * 16);
            } else {  // 3x3 over a bordered 64-channel image: k-step kk of 18 -> chunk q = kk*4 + g -> (tap, 8-channel chunk)
                const int kk = lc * 6 + ks;
                const int q = kk * 4 + g, tap = q >> 3, c8 = q & 7, dy = tap / 3, dx = tap - dy * 3;
                const int src = (L == 1 || L == 3) ? OFF_A : OFF_T;
#pragma unroll
                for (int mf = 0; mf < 3; ++mf) ac[sl][mf] = *reinterpret_cast<const hx8 *>(smem + src + o_bord[mf] + (dy * BW + dx) * P64 + c8 * 16);
            }
        };
        auto kstep = [&](auto ks_const) {
            constexpr int ks = decltype(ks_const)::value;
            if constexpr (ks + 1 < ksteps_of(ci)) fetch_ops(std::integral_constant<int, ks + 1>{});
#pragma unroll
            for (int mf = 0; mf < 3; ++mf)
#pragma unroll
                for (int f = 0; f < 2; ++f) acc[mf][f] = HX<F16>::mfma(wc[ks & 1][f], ac[ks & 1][mf], acc[mf][f]);
        };
        fetch_ops(std::integral_constant<int, 0>{});
        for_each_chunk(kstep, std::make_integer_sequence<int, ksteps_of(ci)>{});
        // ---- layer / cout-block boundaries
        if constexpr (L == 0 && lc == 0) epilogue(0, 0, o_int, OFF_A, -1);       // a  = SiLU(cv1 x)
        if constexpr (L == 0 && lc == 1) epilogue(0, 1, o_bb, OFF_BB, -1);       // b  = SiLU(cv2 x)
        if constexpr (L == 1 && lc == 2) epilogue(1, 0, o_int, OFF_T, -1);       // t  = SiLU(m0.cv1 a)
        if constexpr (L == 2 && lc == 2) epilogue(2, 0, o_int, OFF_A, OFF_A);    // a  = a + SiLU(m0.cv2 t)   (in place: a lane re-writes what it read)
        if constexpr (L == 3 && lc == 2) epilogue(3, 0, o_int, OFF_T, -1);       // t  = SiLU(m1.cv1 a)
        if constexpr (L == 4 && lc == 2) epilogue(4, 0, o_int, OFF_A, OFF_A);    // a' = a + SiLU(m1.cv2 t)
        if constexpr (L == 5) {                                                  // out = SiLU(cv3 [a' | b]) -> staging rows [pixel][128 ch] (input image's space;
            int o_out[3];                                                        //   T, which shares it, was last read by layer 4: two chunk barriers ago)
#pragma unroll
            for (int mf = 0; mf < 3; ++mf) o_out[mf] = o_in[mf] + (lc == 1 ? 128 : 0);
            epilogue(5, lc, o_out, OFF_IN, -1);
        }
    };
    for_each_chunk(step, std::make_integer_sequence<int, kChunks>{});
    __syncthreads();
    {
        bf16_t *dst = P.out + (int64_t)b * P.out_bs + P.out_co;
        for (int i = tid; i < NPX * 16; i += 512) {
            int px = i >> 4, c = i & 15;
            *reinterpret_cast<u32x4 *>(dst + (int64_t)px * P.out_cs + c * 8) = *reinterpret_cast<const u32x4 *>(smem + OFF_IN + px * P128 + c * 16);
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side

bool c3kimg_supported(int H, int W, int c_in, int c_hidden, int c_out, int n) { return H == 13 && W == 13 && c_in == 128 && c_hidden == 64 && c_out == 128 && n == 2; }

int c3kimg_pieces() { return c3k::kPieces; }

hipError_t launch_c3kimg(const C3kImgLaunch &L, hipStream_t st) {
    if (!L.wts || !L.bias || L.in.cpb || L.out.cpb || L.B <= 0) return hipErrorInvalidValue;
    C3kImgParams P;
    P.in = (const bf16_t *)L.in.p; P.in_bs = L.in.bs; P.in_cs = L.in.cs; P.in_co = L.in.co;
    P.out = (bf16_t *)L.out.p; P.out_bs = L.out.bs; P.out_cs = L.out.cs; P.out_co = L.out.co;
    P.wts = L.wts; P.bias = L.bias;
    static bool attr[2] = {false, false};
    const int k = L.f16 ? 0 : 1;
    if (!attr[k]) {
        hipError_t e = L.f16 ? hipFuncSetAttribute(reinterpret_cast<const void *>(&k_c3k_image<true>), hipFuncAttributeMaxDynamicSharedMemorySize, c3k::LDS_B)
                             : hipFuncSetAttribute(reinterpret_cast<const void *>(&k_c3k_image<false>), hipFuncAttributeMaxDynamicSharedMemorySize, c3k::LDS_B);
        if (e != hipSuccess) return e;
        attr[k] = true;
    }
    if (L.f16) hipLaunchKernelGGL(k_c3k_image<true>, dim3((unsigned)L.B), dim3(512), c3k::LDS_B, st, P);
    else hipLaunchKernelGGL(k_c3k_image<false>, dim3((unsigned)L.B), dim3(512), c3k::LDS_B, st, P);
    return hipGetLastError();
}

}  // namespace obb
