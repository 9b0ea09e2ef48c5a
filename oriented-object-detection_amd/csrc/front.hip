// Network front in ONE launch: model.0 (Conv 3x3 s2 on the uint8 tile; the predictor's BGR->RGB, /255 preprocess fused)
// -> model.1 (Conv 3x3 s2) -> model.2.cv1 (Conv 1x1), each + folded BN + SiLU (Detect_OBB.py:81-83 -> OBBModel layers 0, 1 and the cv1 of
// layer 2).
//
// As separate launches these are the two most memory-bound kernels of the forward: the 16-channel half-resolution tensor between them
// (1.4 MB per 416-px tile) is written once and read back once, 2.8 MB of the forward's 28 MB per tile, for 19 M + 50 M MACs.  Here a
// workgroup owns one 13 x 13 tile of the QUARTER-resolution output: it fetches the 55 x 55 uint8 pixels under it, computes the 27 x 27 x 16
// stem outputs it needs straight into the LDS tile the second conv reads (8 % halo recompute), runs the second conv and the 1x1 on it and
// stores 32 channels per pixel.  HBM traffic of the three layers: 0.57 MB in + 0.69 MB out per tile.
//
// Per tile (4 waves, 4 barriers):
//   raw      the tile's byte rows as 16-B chunks (prefetched a tile ahead into registers) -> LDS
//   convert  bytes -> 16-bit v/255 into an image [55 rows][56 px][4 ch] (zeros outside the picture = the stem's zero padding)
//   stem     46 fragments of 16 stem pixels: k = (dy, dx-pair, channel) as in stem.hip (two 16x16x32 steps), + bias, SiLU, 16 bit,
//            forced to ZERO where the stem pixel lies outside the stem's output (that is model.1's zero padding) -> LDS [729 px][16 ch]
//   conv     3x3 stride 2 over that tile: 5 k-steps x (3 pixel fragments x 2 cout fragments) per wave, weights in registers
//   tail     a lane's 8 outputs of a pixel are one B-operand fragment of the 1x1 (k = channel g*8 + j): straight from registers;
//            its 8 results are 16 contiguous bytes, the four lanes of a pixel 64: stored from registers
#include "front.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "stem.h"

namespace obb {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct FrontParams {
    const uint8_t *in; int64_t in_bs; unsigned in_bytes;
    bf16_t *out; int64_t out_bs; int out_cs, out_co, out_bsh, out_bmask; int64_t out_ps;
    const bf16_t *w0, *w1, *w2; const float *b0, *b1, *b2;
    int Hin, Win, Hs, Ws, Ho, Wo;
    int tiles_x, tiles_y, ntiles, tpw;
    unsigned long long *stamps;  // diagnostic build (-DOBB_STAMPS) only
};

#ifdef OBB_STAMPS  // s_memtime phase sums per wave (tools/stamp_conv.sh): 0 barrier + raw copy, 1 convert, 2 stem, 3 conv + tail + stores, 4 fetch issue
#define STAMP_INIT unsigned long long st_acc[5] = {0, 0, 0, 0, 0}, st_tiles = 0, st_prev = __builtin_amdgcn_s_memtime();
#define STAMP(i) { __builtin_amdgcn_sched_barrier(0); unsigned long long st_now = __builtin_amdgcn_s_memtime(); st_acc[i] += st_now - st_prev; st_prev = st_now; __builtin_amdgcn_sched_barrier(0); }
#define STAMP_TILE ++st_tiles;
#define STAMP_FLUSH if (P.stamps && lane == 0) { for (int i_ = 0; i_ < 5; ++i_) atomicAdd(P.stamps + i_, st_acc[i_]); atomicAdd(P.stamps + 5, st_tiles); }
#else
#define STAMP_INIT
#define STAMP(i)
#define STAMP_TILE
#define STAMP_FLUSH
#endif

constexpr int kT = 13, kSW = 2 * kT + 1, kIW = 2 * kSW + 1, kSP = kSW * kSW, kPST = 48, kCvtPx = kIW + 1, kCvtPitch = kCvtPx * 8 + 16;
constexpr int kFrontActBytes = (kSP * kPST + 15) / 16 * 16;

template <int CH, bool F16>
__global__ __launch_bounds__(256, 2) void k_front(const FrontParams P) {
    typedef typename HX<F16>::vec8 hx8;
    constexpr int MAXC = CH == 3 ? 12 : 15, RAWP = MAXC * 16, NITEM = kIW * MAXC, MAXI = (NITEM + 255) / 256;
    static_assert(MAXI * 256 * 16 <= kFrontActBytes, "the raw bytes share the activation tile's LDS");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *raw = smem, *act = smem;  // raw is dead once converted; the stem writes the tile over it
    char *cvt = smem + kFrontActBytes;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, pl = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t0 = blockIdx.x * P.tpw, t1 = min(t0 + P.tpw, P.ntiles);
    if (t0 >= t1) return;

    // ---- weights and biases in registers: stem 2 fragments, conv 5 k-steps x 2, tail 2
    hx8 w0[2], w1[5][2], w2[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) w0[ks] = *reinterpret_cast<const hx8 *>(P.w0 + (ks * 64 + lane) * 8);
#pragma unroll
    for (int ks = 0; ks < 5; ++ks)
#pragma unroll
        for (int f = 0; f < 2; ++f) w1[ks][f] = *reinterpret_cast<const hx8 *>(P.w1 + ((ks * 2 + f) * 64 + lane) * 8);
#pragma unroll
    for (int f = 0; f < 2; ++f) w2[f] = *reinterpret_cast<const hx8 *>(P.w2 + (f * 64 + lane) * 8);
    float b0[4], b1[8], b2[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) b0[i] = P.b0[g * 4 + i];
#pragma unroll
    for (int i = 0; i < 8; ++i) { b1[i] = P.b1[g * 8 + i]; b2[i] = P.b2[g * 8 + i]; }

    // ---- tile-independent lane state
    int koff[5];  // conv k step ks, lane group g: chunk q = ks*4 + g -> tap q/2, channels (q&1)*8 ..; padding chunks re-read the last one (zero weights)
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        int q = ks * 4 + g;
        q = q < 18 ? q : 17;
        const int tap = q >> 1, dy = tap / 3, dx = tap - dy * 3;
        koff[ks] = (dy * kSW + dx) * kPST + (q & 1) * 16;
    }
    int pixb[3], ptyx[3];
#pragma unroll
    for (int mf = 0; mf < 3; ++mf) {
        // (slots past the 169 pixels repeat pixel 168: same inputs, same value, same address -- the stores stay unconditional, so the
        //  compiler can count them and the wait for the next tile's prefetched bytes does not drain this tile's stores)
        const int p = min((wave * 3 + mf) * 16 + pl, kT * kT - 1);
        const int ty = p / kT, tx = p - ty * kT;
        pixb[mf] = (ty * 2 * kSW + tx * 2) * kPST;
        ptyx[mf] = (ty << 16) | tx;
    }
    int scvt[2];  // stem k step ks, lane group g: chunk q = ks*4 + g -> row dy = q/2, pixel pair (q&1)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        int q = ks * 4 + g;
        q = q < 6 ? q : 5;
        scvt[ks] = (q >> 1) * kCvtPitch + (q & 1) * 16;
    }
    constexpr int kStemSlots = 12;
    int sa_off[kStemSlots], sw_off[kStemSlots];  // per stem slot: byte offset of the pixel's window in the converted image / of its 8 output bytes in the tile
    unsigned zrow = 0, zcol = 0;                 // bit k: slot k lies in stem row / column -1 of a top / left tile
#pragma unroll
    for (int k = 0; k < kStemSlots; ++k) {
        const int p = min((wave + 4 * k) * 16 + pl, kSP - 1);
        const int sy = p / kSW, sx = p - sy * kSW;
        sa_off[k] = (2 * sy) * kCvtPitch + (2 * sx) * 8;
        sw_off[k] = p * kPST + g * 8;
        zrow |= (sy == 0 ? 1u : 0u) << k; zcol |= (sx == 0 ? 1u : 0u) << k;
    }
    const f32x4 bias0 = f32x4{b0[0], b0[1], b0[2], b0[3]};  // accumulators start at the bias
    int64_t lane_out[3];  // per pixel fragment: element offset of the lane's 16 output bytes relative to the tile's first pixel
#pragma unroll
    for (int mf = 0; mf < 3; ++mf)
        lane_out[mf] = ((int64_t)(ptyx[mf] >> 16) * P.Wo + (ptyx[mf] & 0xffff)) * P.out_cs + (int64_t)(g >> P.out_bsh) * P.out_ps + ((g & P.out_bmask) << 3);
    auto tile_origin = [&](int t, int &b, int &oy0, int &ox0) {
        const int tx_i = t % P.tiles_x, r = t / P.tiles_x;
        b = r / P.tiles_y;
        oy0 = (r - b * P.tiles_y) * kT; ox0 = tx_i * kT;
    };
    u32x4 pre[MAXI];
    auto fetch = [&](int t) {  // the tile's 55 byte rows as aligned 16-B chunks (rows above / below the picture: zeros by the range check)
        int b, oy0, ox0;
        tile_origin(t, b, oy0, ox0);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(P.in + (int64_t)b * P.in_bs), 0, (int)P.in_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < MAXI; ++k) {
            const int idx = tid + k * 256;
            const int r = idx / MAXC, c = idx - r * MAXC;
            const int rowstart = ((4 * oy0 - 3 + r) * P.Win + 4 * ox0 - 3) * CH;
            const unsigned off = idx < NITEM ? (unsigned)((rowstart & ~15) + c * 16) : 0xffffffffu;
            pre[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        }
    };
    fetch(t0);
    // first tile: settled before the loop, so that inside the loop the prefetched bytes are always exactly "three stores old" on the in-order
    // memory counter and the wait for them leaves the previous tile's stores in flight (the compiler merges the loop entry state with the back edge)
    __builtin_amdgcn_s_waitcnt((0 & 15) | (7 << 4) | (15 << 8) | ((0 >> 4) << 14));
    STAMP_INIT

    for (int t = t0; t < t1; ++t) {
        int b, oy0, ox0;
        tile_origin(t, b, oy0, ox0);
        const int giy0 = 4 * oy0 - 3, gix0 = 4 * ox0 - 3;
        const bool border = oy0 == 0 || ox0 == 0;  // only the top / left tiles see padding
        __syncthreads();  // the previous tile's conv is done with the activation tile
#pragma unroll
        for (int k = 0; k < MAXI; ++k) {
            *reinterpret_cast<u32x4 *>(raw + (tid + k * 256) * 16) = pre[k];  // [row][MAXC chunks] (slots past the last row: zeros into unused LDS)
        }
        __syncthreads();
        STAMP(0)
        // ---- bytes -> 16-bit v/255 (v * (1/255) rounds like v / 255 for every byte: stem_scale_is_exact, checked before this kernel is chosen)
        if constexpr (CH == 3) {
            // four pixels = 12 bytes at any byte phase: four aligned dwords, three v_alignbyte, twelve byte -> float conversions
            for (int i = tid; i < kIW * (kCvtPx / 4); i += 256) {
                const int r = i / (kCvtPx / 4), k4 = i - r * (kCvtPx / 4);
                const int gy = giy0 + r, gx = gix0 + 4 * k4;
                const bool rowok = gy >= 0;  // (a tile never reaches below or right of the picture: 4 * 13 * tiles = side)
                const int sb = (((gy * P.Win + gix0) * 3) & 15) + 12 * k4;
                const unsigned *wp = reinterpret_cast<const unsigned *>(raw + r * RAWP + (sb & ~3));
                const unsigned w0_ = wp[0], w1_ = wp[1], w2_ = wp[2], w3_ = wp[3];
                const unsigned sh = (unsigned)(sb & 3);
                const unsigned a0 = __builtin_amdgcn_alignbyte(w1_, w0_, sh), a1 = __builtin_amdgcn_alignbyte(w2_, w1_, sh), a2 = __builtin_amdgcn_alignbyte(w3_, w2_, sh);
                const unsigned by[12] = {a0 & 0xffu, (a0 >> 8) & 0xffu, (a0 >> 16) & 0xffu, a0 >> 24, a1 & 0xffu, (a1 >> 8) & 0xffu, (a1 >> 16) & 0xffu, a1 >> 24,
                                         a2 & 0xffu, (a2 >> 8) & 0xffu, (a2 >> 16) & 0xffu, a2 >> 24};
                unsigned o[8];
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    const float c0 = (float)by[px * 3 + 0] * (1.0f / 255.0f), c1 = (float)by[px * 3 + 1] * (1.0f / 255.0f), c2 = (float)by[px * 3 + 2] * (1.0f / 255.0f);
                    o[px * 2 + 0] = HX<F16>::pack2(c0, c1);
                    o[px * 2 + 1] = HX<F16>::pack2(c2, 0.f);
                }
                if (border) {  // (tile-uniform) pixels left of / above the picture are the stem's zero padding; pixel 55 of a row only meets zero weights
#pragma unroll
                    for (int px = 0; px < 4; ++px) {
                        const unsigned m = (unsigned)-(int)(rowok && gx + px >= 0);
                        o[px * 2 + 0] &= m; o[px * 2 + 1] &= m;
                    }
                }
                u32x4 *dst = reinterpret_cast<u32x4 *>(cvt + r * kCvtPitch + k4 * 32);
                dst[0] = u32x4{o[0], o[1], o[2], o[3]};
                dst[1] = u32x4{o[4], o[5], o[6], o[7]};
            }
        } else {
            for (int i = tid; i < kIW * kCvtPx; i += 256) {
                const int r = i / kCvtPx, x = i - r * kCvtPx;
                const int gy = giy0 + r, gx = gix0 + x;
                uint2 o = make_uint2(0u, 0u);
                if (x < kIW && gy >= 0 && gy < P.Hin && gx >= 0 && gx < P.Win) {
                    const int phase = ((gy * P.Win + gix0) * CH) & 15;
                    const uint8_t *sp = reinterpret_cast<const uint8_t *>(raw) + r * RAWP + phase + x * CH;
                    const float c0 = (float)sp[0] * (1.0f / 255.0f), c1 = (float)sp[1] * (1.0f / 255.0f), c2 = (float)sp[2] * (1.0f / 255.0f);
                    const float c3 = (float)sp[CH - 1] * (1.0f / 255.0f);
                    o.x = HX<F16>::pack2(c0, c1);
                    o.y = HX<F16>::pack2(c2, c3);
                }
                *reinterpret_cast<uint2 *>(cvt + r * kCvtPitch + x * 8) = o;
            }
        }
        __syncthreads();
        STAMP(1)
        fetch(min(t + 1, t1 - 1));  // next tile's bytes, under this tile's arithmetic (unconditional: the last tile re-reads its own; keeps the count exact)
        STAMP(4)
        // ---- stem: 46 fragments of 16 stem pixels, dealt round-robin to the waves (12 slots per wave, addresses precomputed; the two spare
        //      slots of waves 2 and 3 repeat pixel 728: same value to the same address, so nothing in the loop is conditional)
        const unsigned zmask = border ? ((oy0 == 0 ? zrow : 0u) | (ox0 == 0 ? zcol : 0u)) : 0u;  // bit k: slot k is model.1's zero padding
#pragma unroll
        for (int k = 0; k < kStemSlots; ++k) {
            const char *ap = cvt + sa_off[k];
            f32x4 acc = bias0;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) acc = HX<F16>::mfma(w0[ks], *reinterpret_cast<const hx8 *>(ap + scvt[ks]), acc);
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = silu_f(acc[i]);
            uint2 o;
            o.x = HX<F16>::pack2(v[0], v[1]); o.y = HX<F16>::pack2(v[2], v[3]);
            if (border) {  // (tile-uniform)
                const unsigned m = ((zmask >> k) & 1u) - 1u;  // 0 where padded
                o.x &= m; o.y &= m;
            }
            *reinterpret_cast<uint2 *>(act + sw_off[k]) = o;
        }
        __syncthreads();
        STAMP(2)
        // ---- conv 3x3 s2 (16 -> 32) on the LDS tile
        f32x4 acc[3][2];
#pragma unroll
        for (int mf = 0; mf < 3; ++mf)
#pragma unroll
            for (int f = 0; f < 2; ++f) acc[mf][f] = f32x4{b1[f * 4 + 0], b1[f * 4 + 1], b1[f * 4 + 2], b1[f * 4 + 3]};
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
            hx8 a[3];
#pragma unroll
            for (int mf = 0; mf < 3; ++mf) a[mf] = *reinterpret_cast<const hx8 *>(act + pixb[mf] + koff[ks]);
#pragma unroll
            for (int mf = 0; mf < 3; ++mf)
#pragma unroll
                for (int f = 0; f < 2; ++f) acc[mf][f] = HX<F16>::mfma(w1[ks][f], a[mf], acc[mf][f]);
        }
        bf16_t *const otile = P.out + (int64_t)b * P.out_bs + P.out_co + ((int64_t)oy0 * P.Wo + ox0) * P.out_cs;
        // ---- SiLU, 16 bit -> the 1x1 from registers -> + bias, SiLU, 16 bit -> store
#pragma unroll
        for (int mf = 0; mf < 3; ++mf) {
            float v[8];
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[f * 4 + i] = silu_f(acc[mf][f][i]);
            uint4 y;
            y.x = HX<F16>::pack2(v[0], v[1]); y.y = HX<F16>::pack2(v[2], v[3]); y.z = HX<F16>::pack2(v[4], v[5]); y.w = HX<F16>::pack2(v[6], v[7]);
            hx8 a2;
            __builtin_memcpy(&a2, &y, 16);
            f32x4 acc2[2];
#pragma unroll
            for (int f = 0; f < 2; ++f) acc2[f] = HX<F16>::mfma(w2[f], a2, f32x4{b2[f * 4 + 0], b2[f * 4 + 1], b2[f * 4 + 2], b2[f * 4 + 3]});
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[f * 4 + i] = silu_f(acc2[f][i]);
            uint4 o;
            o.x = HX<F16>::pack2(v[0], v[1]); o.y = HX<F16>::pack2(v[2], v[3]); o.z = HX<F16>::pack2(v[4], v[5]); o.w = HX<F16>::pack2(v[6], v[7]);
            *reinterpret_cast<uint4 *>(otile + lane_out[mf]) = o;
        }
        STAMP(3)
        STAMP_TILE
    }
    STAMP_FLUSH
}

bool front_supported(int cin, int c0, int c1, int c2, int Hin, int Win) {
    return (cin == 3 || cin == 4) && c0 == 16 && c1 == 32 && c2 == 32 && Hin >= 52 && Win >= 52 && Hin % 52 == 0 && Win % 52 == 0 && Hin <= 1664 && Win <= 1664;
}

hipError_t launch_front(const FrontLaunch &L, hipStream_t st) {
    if (!front_supported(L.cin, 16, 32, 32, L.Hin, L.Win) || !L.in || !L.out.p || !L.w0 || !L.w1 || !L.w2 || !L.b0 || !L.b1 || !L.b2 || L.B <= 0) return hipErrorInvalidValue;
    FrontParams P;
    P.in = L.in; P.in_bs = (int64_t)L.Hin * L.Win * L.cin; P.in_bytes = (unsigned)P.in_bs;
    if (P.in_bs % 16 || ((uintptr_t)L.in & 15)) return hipErrorInvalidValue;  // 16-B chunks of whole images
    P.out = (bf16_t *)L.out.p; P.out_bs = L.out.bs; P.out_cs = L.out.cs; P.out_co = L.out.co;
    P.out_bsh = 31; P.out_bmask = 0x7fffffff; P.out_ps = 0;
    if (L.out.cpb > 0) {  // channel-blocked output (conv.h TensorRef): chunk c of a pixel lives in block c >> bsh
        const int blk = 8 * L.out.cpb;
        if ((L.out.cpb & (L.out.cpb - 1)) || L.out.co % blk || L.out.cs != blk) return hipErrorInvalidValue;
        P.out_bsh = 0;
        while ((1 << P.out_bsh) < L.out.cpb) ++P.out_bsh;
        P.out_bmask = L.out.cpb - 1; P.out_ps = L.out.ps;
        P.out = (bf16_t *)L.out.p + (int64_t)(L.out.co / blk) * L.out.ps; P.out_co = 0;
    } else if (L.out.co % 8 || L.out.cs % 8) return hipErrorInvalidValue;
    P.w0 = L.w0; P.w1 = L.w1; P.w2 = L.w2; P.b0 = L.b0; P.b1 = L.b1; P.b2 = L.b2;
    P.Hin = L.Hin; P.Win = L.Win; P.Hs = L.Hin / 2; P.Ws = L.Win / 2; P.Ho = L.Hin / 4; P.Wo = L.Win / 4;
    P.tiles_y = P.Ho / kT; P.tiles_x = P.Wo / kT;
    const int64_t nt = (int64_t)L.B * P.tiles_y * P.tiles_x;
    if (nt >= (1ll << 31)) return hipErrorInvalidValue;
    P.ntiles = (int)nt;
    P.tpw = (int)std::max<int64_t>(1, std::min<int64_t>(nt / 2048, 32));  // >= 8 groups per CU in flight, up to 32 tiles per group beyond that
    const dim3 grid((unsigned)((nt + P.tpw - 1) / P.tpw));
    const size_t lds = (size_t)kFrontActBytes + (size_t)kIW * kCvtPitch;
    auto go = [&](auto kernel) -> hipError_t {
        static std::vector<const void *> attr_set;  // (the instantiations share one function-pointer type: keyed by address)
        if (std::find(attr_set.begin(), attr_set.end(), (const void *)kernel) == attr_set.end()) {
            hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            if (e != hipSuccess) return e;
            attr_set.push_back((const void *)kernel);
        }
        hipLaunchKernelGGL(kernel, grid, dim3(256), lds, st, P);
        return hipGetLastError();
    };
    P.stamps = nullptr;
#ifdef OBB_STAMPS
    static unsigned long long *stamp_dev = nullptr;
    if (!stamp_dev) (void)hipMalloc((void **)&stamp_dev, 64);
    (void)hipMemsetAsync(stamp_dev, 0, 64, st);
    P.stamps = stamp_dev;
#endif
    hipError_t e = L.cin == 3 ? (L.f16 ? go(k_front<3, true>) : go(k_front<3, false>)) : (L.f16 ? go(k_front<4, true>) : go(k_front<4, false>));
#ifdef OBB_STAMPS
    if (e == hipSuccess) {
        unsigned long long h[8];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, stamp_dev, 64, hipMemcpyDeviceToHost);
        const double n = h[5] ? (double)h[5] : 1.0;
        fprintf(stderr, "STAMPS front %dx%d | per wave-tile cycles: barrier+raw %.0f  convert %.0f  stem %.0f  conv+tail %.0f  fetch_issue %.0f  (wave-tiles %.0f)\n", L.Hin, L.Win,
                h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, n);
    }
#endif
    return e;
}

}  // namespace obb
