// Implicit-GEMM convolution on the gfx950 matrix cores: the OBBModel forward's Conv(+folded BN)+SiLU layers
// (ultralytics nn/modules Conv, reached from Detect_OBB.py:81-83).
//
// Mapping (per workgroup = 4 waves, 256 threads):
//   * GEMM is computed TRANSPOSED, D^T[cout][pixel] = W[cout][k] * X[k][pixel], with v_mfma_f32_16x16x32_bf16:
//       A operand = packed weights (16 couts x 32 k), streamed straight from global/L2 in fragment order (1 KiB
//                   contiguous per wave-instruction; the same lines are shared by the 4 waves of the group),
//       B operand = activations (32 k x 16 pixels) read with ds_read_b128 from an LDS-staged input tile (halo included),
//                   so the 9 taps of a 3x3 re-use one staged copy.
//     D then has pixel = lane&15 and cout = (lane>>4)*4 + reg; the host permutes couts inside a 16*NF block so that a
//     lane owns 4*NF CONTIGUOUS output channels of one pixel -> the NHWC store is 8*NF bytes per lane, 128 B per pixel.
//   * k index = (tap, cin) with cin fastest, in chunks of 8 channels (one 16 B LDS read); chunk q of a stage maps to
//     tap = q / (CK/8), c0 = 8 * (q % (CK/8)), which works for every power-of-two CK >= 8 (3x3 with 8 or 16 input
//     channels packs several taps into one 32-deep MFMA step).  Padding k-steps carry zero weights.
//   * waves split the pixel dimension (MF fragments of 16 pixels each); all waves share the same 16*NF couts.
//   * epilogue fused: + bias, SiLU, + residual, bf16 (or fp32) store into a channel slice of the output buffer, so
//     Concat / chunk never exist as kernels.
//   * the network input layer reads the uint8 NHWC tile directly (BGR->RGB + /255 via a 256-entry LUT = exactly
//     bf16(v/255)), i.e. the predictor's preprocess is fused into conv0.
#include "conv.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace obb {

typedef __attribute__((ext_vector_type(4))) float f32x4;

// Diagnostic build only (-DOBB_STAMPS, tools/stamp_conv.sh): s_memtime stamps around the phases of a tile, summed per wave and added
// to ConvParams::stamps[0..5] = {barrier wait, LDS staging, prefetch issue, k loop, epilogue, tiles}.  Never compiled into the product.
#ifdef OBB_STAMPS
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
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;  // native vector: stays in registers where HIP's uint4 struct may not

struct ConvParams {
    const void *in; int64_t in_bs; int in_cs, in_co;
    void *out; int64_t out_bs; int out_cs, out_co;
    const bf16_t *res; int64_t res_bs; int res_cs, res_co;
    // channel-blocked addressing (TensorRef::cpb): chunk cc of a pixel lives at (cc >> bsh) * ps + pixel * pp + (cc & bmask) * 8 elements;
    // plain NHWC is the degenerate case bsh = 31, bmask = ~0, pp = cs
    int in_bsh, in_bmask; unsigned in_ps2 /*bytes*/;
    int out_bsh, out_bmask; int64_t out_ps;
    int res_bsh, res_bmask; int64_t res_ps;
    const bf16_t *wpk; const float *bias; const bf16_t *lut;
    int Hin, Win, Hout, Wout, cin, cout, stride, act, flip_bgr;
    int TH, TW, CK, sh /*log2(CK/8)*/, tiles_x, tiles_y, nstage, kst, out_hw, act_bytes;
    int tpw, ntiles;  // consecutive pixel tiles per workgroup; total pixel tiles (batch included)
    int NI, B;        // NI > 1: a tile = NI whole images of a small map (TH x TW = the map: the 4 x 4 level of the 128-px scale), B images in all
    int gx, ncb;      // workgroups along the tile axis; cout blocks
    unsigned in_span_bytes, w_bytes;  // buffer-descriptor ranges: bytes of one image's input slice span; bytes of the packed weights
    // 1x1 over a virtual concat [nearest-x2 upsample of a low-res tensor | full-res tensor] (1-D launches only): channels [0, up_c) come
    // from `in` (low-res NHWC slice, pixel (b, y/2, x/2)), the rest from `in2`; neither the upsampled tensor nor the concat is ever stored
    const void *in2; int in2_cs, in2_co; unsigned in2_span_bytes; int up_c, up_W, up_HW;
    // fused trailing 1x1 conv (TAIL): out2[pixel][cout2] = W2 . y[pixel][0 .. 16*NF) + bias2, fp32 rows of the head tensor
    const bf16_t *w2pk; const float *bias2; float *out2; int64_t out2_bs; int out2_cs, out2_co, out2_hw, cout2, kst2, w2_off;
    unsigned long long *stamps;  // diagnostic build (-DOBB_STAMPS) only
    int dbg;  // timing experiments only (OBB_CONV_DBG): 1 skip MFMA loop, 2 skip activation loads, 4 skip SiLU, 8 skip stores, 16 skip weight loads
    float inv_twin, inv_tw;
};

// TAIL > 0: the layer is followed by a plain 1x1 conv (no activation) whose output goes to the head tensor: that second GEMM runs
// on the staged 16-bit output tile while it is still in LDS (TAIL = its NF), and the intermediate tensor is never written.
// VCAT: 1x1 over a virtual [upsample | skip] concat (ConvParams::up_c); a separate instantiation so that the plain kernels stay branch-free
// T16: the trailing 1x1 has an activation and a 16-bit (possibly channel-blocked) output of its own -- the cv1 of the C3k2 block behind a
// stride-2 backbone conv: its result replaces the staged tile in LDS and leaves through the same coalesced write-out.
// WRES: "weights resident": a 3x3 stride-1 layer with <= 64 input channels staged as ONE channel stage (CK = cin).  The whole weight
// block of the group's couts (<= 72 KiB) is copied into LDS once and stays there for every tile the group walks, and a tile's input
// (13 x 13 + halo, all channels) is ONE set of loads, prefetched into registers a whole tile ahead.  The multi-stage form pays a
// global-memory latency per 16-channel stage (4 per tile at cin = 64) with 60 MFMAs of cover each; here a tile is 216 MFMAs back to
// back behind a single, fully covered latency.  One group per CU (104 KiB of LDS), one wave per SIMD: no register limit to respect.
// NIT: multi-image tiles (ConvParams::NI > 1) -- a separate instantiation of the one shape that has them (3x3, one fragment per wave, 64-cout
// groups, register-direct stores), so that the index arithmetic of the form costs the other variants, all at their register caps, nothing.
template <int KS, int MF, int NF, bool IN_U8, bool OUT_F32, bool F16, int TAIL = 0, bool VCAT = false, bool T16 = false, bool WRES = false, bool NIT = false>
// Register budget: the 64-cout 3x3 variants need ~210 VGPRs (two waves per SIMD); everything else fits 168 without spills, which is the
// difference between two and three resident waves per SIMD (allocation granule 8: 170 registers already drop to two).
__global__ __launch_bounds__(256, WRES ? 1 : ((NF == 4 && (KS == 3 ? MF >= 2 : MF == 3)) ? 2 : 3)) void k_conv_igemm(const ConvParams P) {
    typedef typename HX<F16>::vec8 hx8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ __attribute__((aligned(16))) bf16_t s_lut[IN_U8 ? 256 : 8];  // u8 -> half(v/255); sized in multiples of 16 B (statics precede the dynamic region)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, pl = lane & 15;
    constexpr int PAD = KS / 2;
    const int S = P.stride;
    const int THin = (P.TH - 1) * S + KS, TWin = (P.TW - 1) * S + KS;
    const int PST = IN_U8 ? 16 : P.CK * 2 + 16;  // bytes per staged pixel (+16 B pad spreads consecutive pixels over LDS banks; the
                                                 // 8-channel input layer packs pixels densely: twice the groups per CU)
    const int cpk = P.CK >> 3;
    const int nq = (KS == 3 ? 9 : 1) * cpk;
    const int in_px1 = THin * TWin;      // staged pixels of one image
    const int in_px = NIT ? in_px1 * P.NI : in_px1;
    const int nchunk = in_px << P.sh;
    // XCD-aware launch order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  The groups that compute
    // different cout blocks of the SAME pixel tiles read the same input: they are made consecutive on one XCD, so the second and
    // later readers hit that XCD's L2 instead of fetching the tile from HBM once per cout block.
    const int xcd = blockIdx.x & 7, lin = blockIdx.x >> 3;
    const int cb = lin % P.ncb;
    const int bx = (lin / P.ncb) * 8 + xcd;
    if (bx >= P.gx) return;
    char *wlds = smem + P.act_bytes;
    if constexpr (IN_U8) s_lut[tid] = P.lut[tid];
    // bias of this group's 16*NF couts, kept in LDS: reading it in the epilogue must not touch vmcnt (loads and stores share that one
    // in-order counter: a global load there would also wait for the next tile's prefetch issued before it)
    __shared__ __attribute__((aligned(16))) float s_bias[16 * NF];
    if (tid < 16 * NF) s_bias[tid] = P.bias[cb * 16 * NF + tid];
    __shared__ __attribute__((aligned(16))) float s_bias2[TAIL > 0 ? 16 * TAIL : 4];
    if constexpr (TAIL > 0) {
        if (tid < 16 * TAIL) s_bias2[tid] = P.bias2[tid];
        for (int i = tid; i < P.kst2 * TAIL * 64; i += 256)  // the tail's weight fragments stay in LDS for the whole kernel
            *reinterpret_cast<u32x4 *>(smem + P.w2_off + i * 16) = reinterpret_cast<const u32x4 *>(P.w2pk)[i];
    }

    // ---- tile-independent per-lane state
    int pixbase[MF], ptyx[MF];  // LDS byte offset of the lane's pixel (one per M fragment); (image-in-tile << 24 | ty << 16 | tx) or -1
    const int tpi = P.TH * P.TW;   // output pixels of one image's part of the tile
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
        int p = (wave * MF + mf) * 16 + pl;
        const int il = NIT ? (int)(((float)p + 0.5f) / (float)tpi) : 0;
        const int q = p - il * tpi;
        int ty = (int)(((float)q + 0.5f) * P.inv_tw);
        int tx = q - ty * P.TW;
        bool ok = p < (NIT ? tpi * P.NI : tpi);
        pixbase[mf] = ok ? (il * in_px1 + (ty * S) * TWin + tx * S) * PST : 0;
        ptyx[mf] = ok ? ((il << 24) | (ty << 16) | tx) : -1;
    }
    // staging plan: this thread moves the 16-B chunks idx = tid + k*256 of the [in_px][CK] tile
    constexpr int MAXLD = WRES ? 8 : ((KS == 1) ? 4 : 6);
    constexpr unsigned NOPIX = 0xffffffffu;
    int ipos[MAXLD];  // (iy << 16 | ix) inside the input tile, or -1
    if constexpr (!IN_U8) {
#pragma unroll
        for (int k = 0; k < MAXLD; ++k) {
            int idx = tid + k * 256;
            int pix = idx >> P.sh;
            const int il = NIT ? (int)(((float)pix + 0.5f) / (float)in_px1) : 0;
            pix -= il * in_px1;
            int iy = (int)(((float)pix + 0.5f) * P.inv_twin);
            int ix = pix - iy * TWin;
            ipos[k] = idx < nchunk ? ((il << 24) | (iy << 16) | ix) : -1;
        }
    }
    const int cbase = cb * 16 * NF + g * 4 * NF;
    const bool do_act = P.act && !(P.dbg & 4);
    constexpr bool DBUF = NF == 4 && (KS == 3 ? MF >= 2 : MF == 3);  // the two-waves-per-SIMD register class (see __launch_bounds__)
    // 64-cout groups store their 16-bit outputs straight from registers: a lane's 16 couts are 32 contiguous bytes and the four lanes of a
    // pixel a whole 128-B row piece (or one 32-B row of four channel-blocked planes) -- nothing left for an LDS transpose to coalesce
    constexpr bool DIRECT = NF == 4 && TAIL == 0 && !OUT_F32 && !IN_U8;

    // ---- tile bookkeeping: a group walks `tpw` consecutive pixel tiles so that prologue, weight staging (single-stage layers)
    //      and the first activation fetch of the next tile are amortised / overlapped
    const int t0 = bx * P.tpw;
    const int t1 = min(t0 + P.tpw, P.ntiles);
    auto tile_origin = [&](int t, int &b, int &oy0, int &ox0) {
        int tx_i = t % P.tiles_x;
        int r = t / P.tiles_x;
        int ty_i = r % P.tiles_y;
        b = NIT ? t * P.NI : r / P.tiles_y;  // (first) image of the tile
        oy0 = ty_i * P.TH; ox0 = tx_i * P.TW;
    };
    // Activations are fetched with buffer loads (one 32-bit byte offset per chunk, descriptor in SGPRs): the descriptor's range
    // check returns zeros for any offset >= num_records, which is how zero padding and partial channel stages are expressed.
    unsigned goff[MAXLD];  // BYTE offset of each chunk at stage 0 of the tile being fetched, or NOPIX (-> zeros)
    unsigned goff2[VCAT ? MAXLD : 1];  // same for the second source of a virtual concat
    __amdgpu_buffer_rsrc_t in_rsrc, in2_rsrc;
    if constexpr (VCAT) in2_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((const bf16_t *)P.in2 + P.in2_co), 0, (int)P.in2_span_bytes, 0x00020000);
    auto plan_tile = [&](int t) {
        int b, oy0, ox0;
        tile_origin(t, b, oy0, ox0);
        const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;
        const bf16_t *base = (const bf16_t *)P.in + (int64_t)b * P.in_bs + P.in_co;
        in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)P.in_span_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < MAXLD; ++k) {
            const int il = NIT ? ipos[k] >> 24 : 0;
            int gy = iy0 + (NIT ? (ipos[k] >> 16) & 0xff : ipos[k] >> 16), gx = ix0 + (ipos[k] & 0xffff);
            bool ok = ipos[k] >= 0 && gy >= 0 && gy < P.Hin && gx >= 0 && gx < P.Win && (!NIT || b + il < P.B);
            if constexpr (VCAT) {  // gx = pixel index over the whole batch at full resolution
                const int bb = gx / P.up_HW, r = gx - bb * P.up_HW;
                const int yy = r / P.up_W, xx = r - yy * P.up_W;
                const int64_t sp = (int64_t)bb * (P.up_HW >> 2) + (int64_t)(yy >> 1) * (P.up_W >> 1) + (xx >> 1);
                goff[k] = ok ? (unsigned)(sp * P.in_cs * 2) : NOPIX;
                goff2[k] = ok ? (unsigned)((int64_t)gx * P.in2_cs * 2) : NOPIX;
            } else {
                goff[k] = ok ? (unsigned)(((NIT ? (int64_t)il * P.in_bs : 0) + ((int64_t)gy * P.Win + gx) * P.in_cs) * 2) : NOPIX;  // (image-in-tile +) pixel part; the chunk part is added per stage
            }
        }
    };
    u32x4 pre[MAXLD];
    auto load_stage = [&](int stage) {  // global -> registers (asynchronous until the values are used)
        const int crem = P.cin - stage * P.CK;  // channels left (multiple of 8): the last stage may be partial
#pragma unroll
        for (int k = 0; k < MAXLD; ++k) {
            int c8 = (tid + k * 256) & (cpk - 1);
            unsigned off = (c8 * 8 < crem && !(P.dbg & 2)) ? goff[k] : NOPIX;
            const int cc = stage * cpk + c8;  // chunk index inside the input slice
            if constexpr (VCAT) {  // stage-uniform choice of the source (up_c is a multiple of CK); offsets selected, ONE load per chunk
                const bool second = stage * P.CK >= P.up_c;
                unsigned o2 = (c8 * 8 < crem) ? goff2[k] : NOPIX;
                o2 = o2 == NOPIX ? NOPIX : o2 + (unsigned)(cc - (P.up_c >> 3)) * 16u;
                off = off == NOPIX ? NOPIX : off + (unsigned)cc * 16u;
                pre[k] = __builtin_amdgcn_raw_buffer_load_b128(second ? in2_rsrc : in_rsrc, second ? o2 : off, 0, 0);
            } else {
                off = off == NOPIX ? NOPIX : off + (unsigned)(cc >> P.in_bsh) * P.in_ps2 + (unsigned)(cc & P.in_bmask) * 16u;
                pre[k] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0);
            }
        }
    };
    auto store_stage = [&]() {  // registers -> LDS
#pragma unroll
        for (int k = 0; k < MAXLD; ++k) {
            int idx = tid + k * 256;
            if (idx < nchunk) *reinterpret_cast<u32x4 *>(smem + (idx >> P.sh) * PST + (idx & (cpk - 1)) * 16) = pre[k];
        }
    };
    // Weights of one channel stage (kst * NF fragments of 1 KiB, already in MFMA A-operand lane order) go through LDS too: the
    // group fetches them ONCE (not once per wave), a whole stage ahead, so neither the L1/TA path nor L2 latency sits in the k-loop.
    constexpr int MAXW = WRES ? 1 : ((KS == 1) ? (NF + 1) / 2 : (NF * 5 * 64 + 255) / 256);  // 1x1: CK <= 64 (kst <= 2); 3x3: CK <= 16 (kst <= 5)
    const int nwchunk = P.kst * NF * 64;  // 16-B chunks of weights per stage
    u32x4 prew[MAXW];
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)P.wpk, 0, (int)P.w_bytes, 0x00020000);
    auto load_w = [&](int stage) {
        const unsigned wbase = (unsigned)((cb * P.nstage + stage) * nwchunk) * 16u;
#pragma unroll
        for (int k = 0; k < MAXW; ++k) {
            int idx = tid + k * 256;
            prew[k] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, (P.dbg & 16) ? 0xffffffffu : wbase + (unsigned)(idx < nwchunk ? idx : nwchunk - 1) * 16u, 0, 0);
        }
    };
    auto store_w = [&]() {
#pragma clang loop unroll(full)
        for (int k = 0; k < MAXW; ++k) {
            int idx = tid + k * 256;
            idx = idx < nwchunk ? idx : nwchunk - 1;  // clamped duplicates rewrite the last chunk with its own value (branch-free)
            *reinterpret_cast<u32x4 *>(wlds + idx * 16) = prew[k];
        }
    };

    // network input layer: raw uint8 pixels (3 or 4 bytes) are prefetched a tile ahead as packed words, converted by the LUT on the
    // way into LDS (channels 3..7 of the 8-channel padded pixel stay zero)
    constexpr int MAXU8 = IN_U8 ? 3 : 1;  // ceil(27*27 / 256) pixels per thread
    unsigned u8pre[MAXU8];
    unsigned u8ok = 0;  // bit k: pixel k of this thread lies inside the image
    auto load_u8 = [&](int t) {
        u8ok = 0;
        int b, oy0, ox0;
        tile_origin(t, b, oy0, ox0);
        const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;
        const uint8_t *src = (const uint8_t *)P.in + (int64_t)b * P.in_bs;
#pragma unroll
        for (int k = 0; k < MAXU8; ++k) {
            int pix = tid + k * 256;
            int iy = (int)(((float)pix + 0.5f) * P.inv_twin);
            int ix = pix - iy * TWin;
            int gy = iy0 + iy, gx = ix0 + ix;
            unsigned v = 0;
            if (pix < in_px && gy >= 0 && gy < P.Hin && gx >= 0 && gx < P.Win) {
                const uint8_t *sp = src + ((int64_t)gy * P.Win + gx) * P.in_cs;
                v = (unsigned)sp[0] | ((unsigned)sp[1] << 8) | ((unsigned)sp[2] << 16);
                if (P.cin == 4) v |= (unsigned)sp[3] << 24;
                u8ok |= 1u << k;
            }
            u8pre[k] = v;
        }
    };
    auto store_u8 = [&]() {
#pragma unroll
        for (int k = 0; k < MAXU8; ++k) {
            int pix = tid + k * 256;
            if (pix >= in_px) continue;
            unsigned v = u8pre[k];
            uint4 o = make_uint4(0, 0, 0, 0);
            if ((u8ok >> k) & 1u) {
                unsigned b0 = v & 0xff, b1 = (v >> 8) & 0xff, b2 = (v >> 16) & 0xff, b3 = v >> 24;
                uint32_t c0 = s_lut[P.flip_bgr ? b2 : b0], c1 = s_lut[b1], c2 = s_lut[P.flip_bgr ? b0 : b2];
                uint32_t c3 = (P.cin == 4) ? (uint32_t)s_lut[b3] : 0u;
                o.x = c0 | (c1 << 16);
                o.y = c2 | (c3 << 16);
            }
            *reinterpret_cast<uint4 *>(smem + pix * PST) = o;
        }
    };

    if (t0 >= t1) return;
    if constexpr (WRES) {  // the whole (single-stage) weight block of this cout group, once
        const u32x4 *wsrc = reinterpret_cast<const u32x4 *>(P.wpk) + (size_t)cb * nwchunk;
        for (int i = tid; i < nwchunk; i += 256) *reinterpret_cast<u32x4 *>(wlds + i * 16) = wsrc[i];
    } else load_w(0);
    if constexpr (!IN_U8) { plan_tile(t0); load_stage(0); }
    else load_u8(t0);
    bool w_resident = WRES;
    // EXACT: register-direct 16-bit stores with a compile-time count per tile.  Loads and stores retire through ONE in-order counter
    // (vmcnt): the wait for the next tile's prefetched operands at the top of stage 0 must leave this tile's stores (issued after them) in
    // flight, which the compiler can only do if their number is the same on every path into that wait.  Hence: stage 0 is a separate copy
    // of the stage body (inside the stage loop the prefetch is the youngest operation, at the top of a tile 2*MF stores are younger), the
    // prefetch and the stores are unconditional (the last tile re-reads itself; lanes without an output pixel store into a sink), and the
    // prologue's loads are settled before the loop so that the loop-entry state is "nothing outstanding".  Without this every tile
    // drained its predecessor's stores before touching LDS (ISA: s_waitcnt vmcnt(0) at the top of every stage).
    constexpr bool EXACT = DIRECT && !VCAT && !WRES;
    static_assert(!NIT || (EXACT && KS == 3 && MF == 1), "multi-image tiles: the register-direct store path only");
    if constexpr (EXACT) __builtin_amdgcn_s_waitcnt((0 & 15) | (7 << 4) | (15 << 8) | ((0 >> 4) << 14));  // vmcnt(0)
    STAMP_INIT

    for (int t = t0; t < t1; ++t) {
        int b, oy0, ox0;
        tile_origin(t, b, oy0, ox0);
        f32x4 acc[MF][NF];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int f = 0; f < NF; ++f) acc[mf][f] = f32x4{0.f, 0.f, 0.f, 0.f};

        auto stage_body = [&](int stage) {
            __syncthreads();
            STAMP(0)
            // ---- stage this channel chunk of the input tile (and, unless resident, the stage's weights) into LDS
            if (!w_resident) store_w();
            if constexpr (IN_U8) {
                store_u8();
                __syncthreads();
                if (t + 1 < t1) load_u8(t + 1);  // next tile's pixels: the byte loads overlap this tile's MFMAs and epilogue
            } else {
                store_stage();
                __syncthreads();
                STAMP(1)
                // prefetch into registers: the next channel stage of this tile, or stage 0 of the next tile; the HBM/L2 latency
                // hides under this stage's MFMAs (and under the epilogue)
                if constexpr (WRES) { if (t + 1 < t1) { plan_tile(t + 1); load_stage(0); } }
                else if constexpr (EXACT) {
                    if (stage + 1 < P.nstage) { load_w(stage + 1); load_stage(stage + 1); }
                    else { if (P.nstage > 1) load_w(0); plan_tile(min(t + 1, t1 - 1)); load_stage(0); }
                }
                else if (stage + 1 < P.nstage) { load_w(stage + 1); load_stage(stage + 1); }
                else if (t + 1 < t1) { plan_tile(t + 1); load_stage(0); if (P.nstage > 1) load_w(0); }
                STAMP(2)
            }

            // ---- K loop over this stage: weight and activation fragments both come from LDS (ds_read_b128, lane-linear / padded rows)
            auto ld_operands = [&](int ks, hx8 (&w)[NF], hx8 (&a)[MF]) {
#pragma unroll
                for (int f = 0; f < NF; ++f) w[f] = *reinterpret_cast<const hx8 *>(wlds + ((ks * NF + f) * 64 + lane) * 16);
                int q = ks * 4 + g;
                q = q < nq ? q : nq - 1;  // padding k-steps: any valid address, their weights are zero
                int off;
                if constexpr (KS == 3) {
                    int tap = q >> P.sh, c0 = q & (cpk - 1);
                    int dy = (tap * 11) >> 5, dx = tap - dy * 3;
                    off = (dy * TWin + dx) * PST + c0 * 16;
                } else {
                    off = q * 16;
                }
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) a[mf] = *reinterpret_cast<const hx8 *>(smem + pixbase[mf] + off);
            };
            auto mfma_step = [&](const hx8 (&w)[NF], const hx8 (&a)[MF]) {
#pragma unroll
                for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                    for (int f = 0; f < NF; ++f) acc[mf][f] = HX<F16>::mfma(w[f], a[mf], acc[mf][f]);
            };
            const int kst = (P.dbg & 1) ? 0 : P.kst;
            if constexpr (DBUF) {
                // two operand sets: the NF + MF LDS reads of step k+1 are issued before the MFMAs of step k (these variants have the
                // registers: they run two waves per SIMD either way), so the matrix pipe only waits for LDS at the first step of a stage
                hx8 wA[NF], aA[MF], wB[NF], aB[MF];
                if (kst > 0) ld_operands(0, wA, aA);
                for (int ks = 0; ks < kst; ks += 2) {
                    ld_operands(min(ks + 1, kst - 1), wB, aB);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_step(wA, aA);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks + 1 < kst) {
                        ld_operands(min(ks + 2, kst - 1), wA, aA);
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_step(wB, aB);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
                for (int ks = 0; ks < kst; ++ks) {
                    hx8 wcur[NF], a[MF];
                    ld_operands(ks, wcur, a);
                    // all NF + MF operand reads of the k step are in flight before its first MFMA (left alone, the scheduler re-used one
                    // register quad for the MF activation fragments: read, wait, 4 MFMAs, three times per step)
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_step(wcur, a);
                }
            }
        };
        if constexpr (EXACT) {
            stage_body(0);
            for (int stage = 1; stage < P.nstage; ++stage) stage_body(stage);
        } else {
            for (int stage = 0; stage < P.nstage; ++stage) stage_body(stage);
        }
        STAMP(3)
        w_resident = (P.nstage == 1);  // single-stage layers keep their weights in LDS for every following tile

        // ---- epilogue: lane owns couts [cbase, cbase + 4*NF) of its pixels: + bias, SiLU, + residual
        const bool full = (cbase + 4 * NF <= P.cout);
        constexpr int ROWB = 32 * NF + 16;  // staged output row: 16*NF halves + 16 B pad
        // coalesced write-out of the staged 16-bit tile: consecutive lanes store consecutive 16-B pieces of a pixel's output row
        // (CPP = 16-B chunks per pixel, occ0 = first 8-channel chunk of this group inside the output slice)
        auto write_out = [&](auto cpp_c, int cout_o, int occ0) {
            constexpr int CPP = decltype(cpp_c)::value;
            const int npx = P.TH * P.TW;
            bf16_t *obase = (bf16_t *)P.out + (int64_t)b * P.out_bs + P.out_co;
            for (int i = tid; i < npx * CPP && !(P.dbg & 8); i += 256) {
                int p = i / CPP, ch = i - p * CPP;
                int ty = (int)(((float)p + 0.5f) * P.inv_tw);
                int tx = p - ty * P.TW;
                if (oy0 + ty >= P.Hout || ox0 + tx >= P.Wout) continue;
                const int occ = occ0 + ch;                 // 8-channel chunk index inside the output slice
                if (occ * 8 + 8 > cout_o) continue;        // cout tail of the last block (cout is a multiple of 8)
                uint4 o = *reinterpret_cast<const uint4 *>(smem + p * ROWB + ch * 16);
                *reinterpret_cast<uint4 *>(obase + (int64_t)(occ >> P.out_bsh) * P.out_ps + ((int64_t)(oy0 + ty) * P.Wout + ox0 + tx) * P.out_cs +
                                           ((occ & P.out_bmask) << 3)) = o;
            }
        };
        float bias[NF * 4];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            float4 bv = *reinterpret_cast<const float4 *>(s_bias + g * 4 * NF + f * 4);
            bias[f * 4 + 0] = bv.x; bias[f * 4 + 1] = bv.y; bias[f * 4 + 2] = bv.z; bias[f * 4 + 3] = bv.w;
        }
        if constexpr (!OUT_F32 && !DIRECT) __syncthreads();  // every wave is done reading the input tile: its LDS becomes the output staging area
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            int ty = NIT ? (ptyx[mf] >> 16) & 0xff : ptyx[mf] >> 16, tx = ptyx[mf] & 0xffff, il = NIT ? ptyx[mf] >> 24 : 0;
            bool ok = ptyx[mf] >= 0 && (oy0 + ty < P.Hout) && (ox0 + tx < P.Wout) && (!NIT || b + il < P.B) && !(P.dbg & 8);
            if constexpr (!EXACT) { if (!ok) continue; }
            else if (!ok) { ty = 0; tx = 0; il = 0; }  // (addresses stay in range; the stores below go to the sink)
            const int64_t opix = (int64_t)(oy0 + ty) * P.Wout + ox0 + tx;
            float v[NF * 4];
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[f * 4 + r] = acc[mf][f][r] + bias[f * 4 + r];
            if (do_act) {  // one uniform branch around 4*NF independent SiLUs (a test per element serialises the transcendental chains)
#pragma unroll
                for (int c = 0; c < NF * 4; ++c) v[c] = silu_f(v[c]);
            }
            if (P.res) {
                const bf16_t *rp = P.res + (int64_t)(b + il) * P.res_bs + opix * P.res_cs + P.res_co + (int64_t)((cbase >> 3) >> P.res_bsh) * P.res_ps +
                                   (((cbase >> 3) & P.res_bmask) << 3) + (cbase & 7);
                if (full) {
#pragma unroll
                    for (int h = 0; h < NF / 2 + (NF == 1); ++h) {
                        if constexpr (NF == 1) {
                            uint2 rv = *reinterpret_cast<const uint2 *>(rp);
                            v[0] += HX<F16>::lo(rv.x); v[1] += HX<F16>::hi(rv.x);
                            v[2] += HX<F16>::lo(rv.y); v[3] += HX<F16>::hi(rv.y);
                        } else {
                            uint4 rv = *reinterpret_cast<const uint4 *>(rp + h * 8);
                            v[h * 8 + 0] += HX<F16>::lo(rv.x); v[h * 8 + 1] += HX<F16>::hi(rv.x);
                            v[h * 8 + 2] += HX<F16>::lo(rv.y); v[h * 8 + 3] += HX<F16>::hi(rv.y);
                            v[h * 8 + 4] += HX<F16>::lo(rv.z); v[h * 8 + 5] += HX<F16>::hi(rv.z);
                            v[h * 8 + 6] += HX<F16>::lo(rv.w); v[h * 8 + 7] += HX<F16>::hi(rv.w);
                        }
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < NF * 4; ++c)
                        if (cbase + c < P.cout) v[c] += HX<F16>::one(rp[c]);
                }
            }
            if constexpr (OUT_F32) {  // network head: fp32 rows of the caller's tensor, float4 per lane
                int64_t ob = b, opx = opix;
                if (P.out_hw > 0) { ob = opx / P.out_hw; opx -= ob * P.out_hw; }  // 1-D launch, per-image output rows
                float *op = (float *)P.out + ob * P.out_bs + opx * P.out_cs + P.out_co + cbase;
                if (full && ((P.out_cs | P.out_co) & 3) == 0) {
#pragma unroll
                    for (int f = 0; f < NF; ++f) *reinterpret_cast<float4 *>(op + f * 4) = make_float4(v[f * 4], v[f * 4 + 1], v[f * 4 + 2], v[f * 4 + 3]);
                } else {
#pragma unroll
                    for (int c = 0; c < NF * 4; ++c)
                        if (cbase + c < P.cout) op[c] = v[c];
                }
            } else if constexpr (DIRECT) {
                bf16_t *obase = (bf16_t *)P.out + (int64_t)(b + il) * P.out_bs + P.out_co + opix * P.out_cs;
#pragma unroll
                for (int h = 0; h < NF / 2; ++h) {
                    uint4 o;
                    o.x = HX<F16>::pack2(v[h * 8 + 0], v[h * 8 + 1]); o.y = HX<F16>::pack2(v[h * 8 + 2], v[h * 8 + 3]);
                    o.z = HX<F16>::pack2(v[h * 8 + 4], v[h * 8 + 5]); o.w = HX<F16>::pack2(v[h * 8 + 6], v[h * 8 + 7]);
                    const int occ = cb * 2 * NF + g * (NF / 2) + h;  // 8-channel chunk index inside the output slice
                    if constexpr (EXACT) {  // one store per piece on every path: lanes without a pixel / beyond the last cout write the sink
                        bf16_t *dst = obase + (int64_t)(occ >> P.out_bsh) * P.out_ps + ((occ & P.out_bmask) << 3);
                        if (!ok || occ * 8 + 8 > P.cout) dst = const_cast<bf16_t *>(P.lut) + 256 + lane * 8;
                        *reinterpret_cast<uint4 *>(dst) = o;
                    } else if (occ * 8 + 8 <= P.cout)
                        *reinterpret_cast<uint4 *>(obase + (int64_t)(occ >> P.out_bsh) * P.out_ps + ((occ & P.out_bmask) << 3)) = o;
                }
            } else {  // 16-bit outputs are staged through LDS so that the global stores below are whole 16-B-per-lane row pieces
                char *orow = smem + ((wave * MF + mf) * 16 + pl) * ROWB + g * 8 * NF;
                if constexpr (NF == 1) {
                    uint2 o;
                    o.x = HX<F16>::pack2(v[0], v[1]); o.y = HX<F16>::pack2(v[2], v[3]);
                    *reinterpret_cast<uint2 *>(orow) = o;
                } else {
#pragma unroll
                    for (int h = 0; h < NF / 2; ++h) {
                        uint4 o;
                        o.x = HX<F16>::pack2(v[h * 8 + 0], v[h * 8 + 1]); o.y = HX<F16>::pack2(v[h * 8 + 2], v[h * 8 + 3]);
                        o.z = HX<F16>::pack2(v[h * 8 + 4], v[h * 8 + 5]); o.w = HX<F16>::pack2(v[h * 8 + 6], v[h * 8 + 7]);
                        *reinterpret_cast<uint4 *>(orow + h * 16) = o;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // one fragment at a time: keeps the epilogue's live range at 4*NF values
        }
        if constexpr (TAIL > 0) {
            __syncthreads();
            f32x4 acc2[MF][TAIL];
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                for (int f = 0; f < TAIL; ++f) acc2[mf][f] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int ks = 0; ks < P.kst2; ++ks) {
                hx8 w2[TAIL];
#pragma unroll
                for (int f = 0; f < TAIL; ++f) w2[f] = *reinterpret_cast<const hx8 *>(smem + P.w2_off + ((ks * TAIL + f) * 64 + lane) * 16);
                int q = ks * 4 + g;
                q = q < 2 * NF ? q : 2 * NF - 1;  // k = the 16*NF staged channels; padding chunks carry zero weights
                hx8 a2[MF];
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) a2[mf] = *reinterpret_cast<const hx8 *>(smem + ((wave * MF + mf) * 16 + pl) * ROWB + q * 16);
                __builtin_amdgcn_sched_barrier(0);  // operand reads of the step ahead of its MFMAs (as in the main k loop)
#pragma unroll
                for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                    for (int f = 0; f < TAIL; ++f) acc2[mf][f] = HX<F16>::mfma(w2[f], a2[mf], acc2[mf][f]);
            }
            const int c2base = g * 4 * TAIL;
            if constexpr (T16) {
                static_assert(TAIL <= NF, "the tail's 16-bit rows reuse the staging rows of the main layer");
                // + bias, SiLU, 16 bit, back into this wave's OWN staging rows (its reads of them are complete: LDS ops are in order)
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) {
                    float v[TAIL * 4];
#pragma unroll
                    for (int f = 0; f < TAIL; ++f) {
                        float4 bv = *reinterpret_cast<const float4 *>(s_bias2 + c2base + f * 4);
                        v[f * 4 + 0] = silu_f(acc2[mf][f][0] + bv.x); v[f * 4 + 1] = silu_f(acc2[mf][f][1] + bv.y);
                        v[f * 4 + 2] = silu_f(acc2[mf][f][2] + bv.z); v[f * 4 + 3] = silu_f(acc2[mf][f][3] + bv.w);
                    }
                    char *orow = smem + ((wave * MF + mf) * 16 + pl) * ROWB + g * 8 * TAIL;
                    if constexpr (TAIL == 1) {
                        uint2 o;
                        o.x = HX<F16>::pack2(v[0], v[1]); o.y = HX<F16>::pack2(v[2], v[3]);
                        *reinterpret_cast<uint2 *>(orow) = o;
                    } else {
#pragma unroll
                        for (int h = 0; h < TAIL / 2; ++h) {
                            uint4 o;
                            o.x = HX<F16>::pack2(v[h * 8 + 0], v[h * 8 + 1]); o.y = HX<F16>::pack2(v[h * 8 + 2], v[h * 8 + 3]);
                            o.z = HX<F16>::pack2(v[h * 8 + 4], v[h * 8 + 5]); o.w = HX<F16>::pack2(v[h * 8 + 6], v[h * 8 + 7]);
                            *reinterpret_cast<uint4 *>(orow + h * 16) = o;
                        }
                    }
                }
                __syncthreads();
                write_out(std::integral_constant<int, 2 * TAIL>(), P.cout2, 0);
            } else {
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                int ty = ptyx[mf] >> 16, tx = ptyx[mf] & 0xffff;
                if (ptyx[mf] < 0 || oy0 + ty >= P.Hout || ox0 + tx >= P.Wout) continue;
                int64_t ob = b, opx = (int64_t)(oy0 + ty) * P.Wout + ox0 + tx;
                if (P.out2_hw > 0) { ob = opx / P.out2_hw; opx -= ob * P.out2_hw; }  // 1-D launch, per-image output rows
                float *op = P.out2 + ob * P.out2_bs + opx * P.out2_cs + P.out2_co + c2base;
#pragma unroll
                for (int f = 0; f < TAIL; ++f) {
                    float4 bv = *reinterpret_cast<const float4 *>(s_bias2 + c2base + f * 4);
                    float v0 = acc2[mf][f][0] + bv.x, v1 = acc2[mf][f][1] + bv.y, v2 = acc2[mf][f][2] + bv.z, v3 = acc2[mf][f][3] + bv.w;
                    if (c2base + f * 4 + 4 <= P.cout2 && ((P.out2_cs | P.out2_co) & 3) == 0) {
                        *reinterpret_cast<float4 *>(op + f * 4) = make_float4(v0, v1, v2, v3);
                    } else {
                        if (c2base + f * 4 + 0 < P.cout2) op[f * 4 + 0] = v0;
                        if (c2base + f * 4 + 1 < P.cout2) op[f * 4 + 1] = v1;
                        if (c2base + f * 4 + 2 < P.cout2) op[f * 4 + 2] = v2;
                        if (c2base + f * 4 + 3 < P.cout2) op[f * 4 + 3] = v3;
                    }
                }
            }
            }
        } else if constexpr (!OUT_F32 && !DIRECT) {
            __syncthreads();
            write_out(std::integral_constant<int, 2 * NF>(), P.cout, cb * 2 * NF);
        }
        STAMP(4)
        STAMP_TILE
    }
    STAMP_FLUSH
}

// ------------------------------------------------------------------------------------------------ weights-resident 3x3, two half-groups
//
// k_conv3_pair: 3x3 stride-1 layers with 64 input channels and 64-cout groups on 13 x 13 tiles (the P3/P4/P5 box-branch convs of the head).
// What bounds k_conv_igemm on these layers is the CU's load path, not HBM and not the matrix pipe: every 169-pixel tile re-fetches its
// group's 72 KiB of weights from L2 (in 16-channel stages) next to 28 KiB of activations, and a CU sustains only ~10-30 B/clk from L2.
// Keeping the weights resident needs 72 KiB + one 32 KiB input tile per group, i.e. one 4-wave group per CU, whose staging, barriers and
// epilogue then run with the matrix pipe idle (measured: no gain).  Here ONE 512-thread workgroup per CU shares the resident weights
// between two 4-wave halves that walk alternate tiles one phase apart: while one half runs its 216-MFMA k loop the other stores its
// previous tile, copies the next input tile from registers to LDS and issues the loads of the tile after (two phases = a whole tile
// ahead).  One barrier per phase; loads and stores stay in flight across it (raw s_barrier, LDS counter only).
//   TAIL = 4: the layer is followed by the head's plain 1x1 (64 -> <= 64 fp32 rows).  With the cout order chosen below a lane's two packed
//   8-channel output pieces of a pixel are exactly its two B-operand fragments of that second GEMM (natural k order), so the tail runs
//   straight from registers and the intermediate never touches LDS.
//   S = 2 (the stride-2 backbone convs with 64 input channels): the 27 x 27 x 64 input tile of a 13 x 13 output tile does not fit twice next
//   to the weights, so a half-group takes a 4-row stripe of 13 outputs (9 x 27 input pixels, 35 KiB): one 16-pixel fragment per wave.  The
//   k loop is then bound by LDS operand reads (4 weight + 1 activation fragment per 4 MFMAs) at about twice the MFMA time.
//   TAIL = 16: the cv1 of the following C3k2 block (1x1 + SiLU, 16-bit output) behind the conv, from registers like the fp32 tail;
//   the conv's own output is never written.
//   NF = 5 (no tail, stride 1): two sibling convs on the same input merged along cout, 64 + 16 channels (the first convs of the head's box
//   and angle branches at P3): the fifth fragment holds couts 64 + g*4 + i, stored as one 8-byte piece per lane (32 contiguous bytes per
//   pixel); its weights come from the second 64-cout block of the standard packing.
template <bool F16, int TAIL, int S, int NF = 4>
__global__ __launch_bounds__(512, 1) void k_conv3_pair(const ConvParams P) {
    typedef typename HX<F16>::vec8 hx8;
    static_assert(NF == 4 || (NF == 5 && TAIL == 0 && S == 1), "merged 80-cout form: plain stride-1 layers only");
    constexpr int MF = S == 1 ? 3 : 1, PST = 144, KST = 18, TH = S == 1 ? 13 : 4, T = 13, THIN = (TH - 1) * S + 3, TWIN = (T - 1) * S + 3;
    constexpr int IN_PX = THIN * TWIN, NCHUNK = IN_PX * 8, MAXLD = 8, TF = TAIL > 0 ? 4 : 0;  // TF: cout fragments of the tail
    static_assert(NCHUNK <= MAXLD * 256, "staging plan");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ __attribute__((aligned(16))) float s_bias[16 * NF];
    __shared__ __attribute__((aligned(16))) float s_bias2[TAIL > 0 ? 64 : 4];
    const int tid = threadIdx.x, lane = tid & 63, htid = tid & 255;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = wave8 >> 2, hw = wave8 & 3;  // half-group, wave inside it
    const int g = lane >> 4, pl = lane & 15;
    const int xcd = blockIdx.x & 7, lin = blockIdx.x >> 3;
    const int cb = lin % P.ncb;
    const int bx = (lin / P.ncb) * 8 + xcd;
    if (bx >= P.gx) return;
    char *abuf = smem + h * P.act_bytes;
    char *wlds = smem + 2 * P.act_bytes;
    char *w2lds = wlds + KST * NF * 1024;
    if (tid < 16 * NF) s_bias[tid] = P.bias[cb * 16 * NF + tid];
    if constexpr (TAIL > 0) {
        if (tid < 64) s_bias2[tid] = P.bias2[tid];
        // tail weights (2 k-steps x 4 fragments).  fp32 head rows (TAIL = 4): rows re-ordered to the natural cout order (fragment f, row r ->
        // cout f*16 + r), so that a store instruction covers 64 contiguous bytes per pixel (4 lanes x float4) instead of four 16-B pieces
        // 64 B apart.  16-bit output (TAIL = 16): the order of the main weights below (two 16-B pieces per lane, 64 contiguous bytes per
        // pixel and instruction).
        for (int i = tid; i < 2 * TF * 64; i += 512) {
            const int l = i & 63, f = (i >> 6) % TF, ks = i / (64 * TF);
            const int r = l & 15, gq = l >> 4, gg = r >> 2, ii = r & 3;
            int src;
            if constexpr (TAIL == 4) src = ((ks * TF + gg) * 64) + (f * 4 + ii) + 16 * gq;
            else src = ((ks * TF + (gg & 1) * 2 + (f & 1)) * 64) + ((f >> 1) * 2 + (gg >> 1)) * 4 + ii + 16 * gq;
            *reinterpret_cast<u32x4 *>(w2lds + i * 16) = reinterpret_cast<const u32x4 *>(P.w2pk)[src];
        }
    }
    const int t0 = bx * P.tpw, t1 = min(t0 + P.tpw, P.ntiles);
    if (t0 >= t1) return;
    const int n_h = (t1 - t0 - h + 1) >> 1;            // tiles of this half: t0 + h, t0 + h + 2, ...
    const int np = max(1 + 2 * ((t1 - t0 + 1) >> 1), 2 + 2 * ((t1 - t0) >> 1));  // phases until both halves are done

    int pixb[MF], ptyx[MF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
        const int p = (hw * MF + mf) * 16 + pl;
        const int ty = p / T, tx = p - ty * T;
        const bool ok = p < TH * T;
        pixb[mf] = (ok ? (ty * S * TWIN + tx * S) * PST : 0) + g * 16;
        ptyx[mf] = ok ? ((ty << 16) | tx) : -1;
    }
    int ipos[MAXLD];
#pragma unroll
    for (int k = 0; k < MAXLD; ++k) {
        const int idx = htid + k * 256, pix = idx >> 3;
        const int iy = pix / TWIN, ix = pix - iy * TWIN;
        ipos[k] = idx < NCHUNK ? ((iy << 16) | ix) : -1;
    }
    auto tile_origin = [&](int t, int &b, int &oy0, int &ox0) {
        const int tx_i = t % P.tiles_x, r = t / P.tiles_x;
        const int ty_i = r % P.tiles_y;
        b = r / P.tiles_y;
        oy0 = ty_i * TH; ox0 = tx_i * T;
    };
    constexpr unsigned NOPIX = 0xffffffffu;
    u32x4 pre[MAXLD];
    unsigned rel[MAXLD];  // byte offset of the thread's chunks relative to the tile's first input pixel (NOPIX for the surplus slots)
#pragma unroll
    for (int k = 0; k < MAXLD; ++k) {
        const int cc = (htid + k * 256) & 7;
        rel[k] = ipos[k] >= 0 ? (unsigned)(((ipos[k] >> 16) * P.Win + (ipos[k] & 0xffff)) * P.in_cs * 2) + (unsigned)(cc >> P.in_bsh) * P.in_ps2 + (unsigned)(cc & P.in_bmask) * 16u : NOPIX;
    }
    auto fetch_tile = [&](int t) {  // global -> registers, one whole input tile (zero padding through the descriptor's range check)
        int b, oy0, ox0;
        tile_origin(t, b, oy0, ox0);
        const bf16_t *base = (const bf16_t *)P.in + (int64_t)b * P.in_bs + P.in_co;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)P.in_span_bytes, 0x00020000);
        const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
        if (iy0 >= 0 && ix0 >= 0 && iy0 + THIN <= P.Hin && ix0 + TWIN <= P.Win) {  // (tile-uniform) no padding: one add per chunk
            const unsigned tb = (unsigned)((iy0 * P.Win + ix0) * P.in_cs * 2);
#pragma unroll
            for (int k = 0; k < MAXLD; ++k) pre[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, rel[k] == NOPIX ? NOPIX : tb + rel[k], 0, 0);
            return;
        }
#pragma unroll
        for (int k = 0; k < MAXLD; ++k) {
            const int gy = iy0 + (ipos[k] >> 16), gx = ix0 + (ipos[k] & 0xffff);
            const bool ok = ipos[k] >= 0 && gy >= 0 && gy < P.Hin && gx >= 0 && gx < P.Win;
            const int cc = (htid + k * 256) & 7;
            const unsigned off = ok ? (unsigned)(((int64_t)gy * P.Win + gx) * P.in_cs * 2) + (unsigned)(cc >> P.in_bsh) * P.in_ps2 + (unsigned)(cc & P.in_bmask) * 16u : NOPIX;
            pre[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        }
    };
    auto store_tile = [&]() {  // registers -> this half's LDS tile
#pragma unroll
        for (int k = 0; k < MAXLD; ++k) {
            const int idx = htid + k * 256;
            if (idx < NCHUNK) *reinterpret_cast<u32x4 *>(abuf + (idx >> 3) * PST + (idx & 7) * 16) = pre[k];
        }
    };
    if (n_h > 0) fetch_tile(t0 + h);
    {   // the resident weight block of this cout group, rows re-ordered so that value (f, i) of lane group g is cout
        // (f >> 1) * 32 + g * 8 + (f & 1) * 4 + i: a lane's two 16-B output pieces then sit 64 B apart and the four lanes of a pixel write
        // 64 contiguous bytes per store instruction; the same order is the natural k order of the tail's B operand
        const u32x4 *wsrc = reinterpret_cast<const u32x4 *>(P.wpk) + (size_t)cb * (KST * 4 * 64);  // (the packing has 4 fragments per 64-cout block)
        for (int i = tid; i < KST * NF * 64; i += 512) {
            const int l = i & 63, f = (i >> 6) % NF, ks = i / (64 * NF);
            const int r = l & 15, gq = l >> 4, gg = r >> 2, ii = r & 3;
            int src;
            if (f < 4) src = ((ks * 4 + (gg & 1) * 2 + (f & 1)) * 64) + ((f >> 1) * 2 + (gg >> 1)) * 4 + ii + 16 * gq;
            else src = KST * 4 * 64 + ((ks * 4 + gg) * 64) + ii + 16 * gq;  // cout 64 + gg*4 + ii = fragment gg, row ii of the second block
            *reinterpret_cast<u32x4 *>(wlds + i * 16) = wsrc[src];
        }
    }
    const bool do_act = P.act != 0;
    f32x4 acc[MF][NF];

    auto kloop = [&]() {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int f = 0; f < NF; ++f) acc[mf][f] = f32x4{0.f, 0.f, 0.f, 0.f};
        // k step ks = tap ks/2, channels (ks & 1) * 32 + g * 8 .. + 7: the tap offset is a constant of the unrolled step
        auto ld = [&](int ks, hx8 (&w)[NF], hx8 (&a)[MF]) {
#pragma unroll
            for (int f = 0; f < NF; ++f) w[f] = *reinterpret_cast<const hx8 *>(wlds + ((ks * NF + f) * 64 + lane) * 16);
            const int tap = ks >> 1, dy = tap / 3, dx = tap - dy * 3;
            const int off = (dy * TWIN + dx) * PST + (ks & 1) * 64;
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) a[mf] = *reinterpret_cast<const hx8 *>(abuf + pixb[mf] + off);
        };
        auto step = [&](const hx8 (&w)[NF], const hx8 (&a)[MF]) {
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                for (int f = 0; f < NF; ++f) acc[mf][f] = HX<F16>::mfma(w[f], a[mf], acc[mf][f]);
        };
        hx8 wA[NF], aA[MF], wB[NF], aB[MF];
        ld(0, wA, aA);
#pragma unroll
        for (int ks = 0; ks < KST; ks += 2) {
            ld(ks + 1, wB, aB);
            __builtin_amdgcn_sched_barrier(0);
            step(wA, aA);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 2 < KST) ld(ks + 2, wA, aA);
            __builtin_amdgcn_sched_barrier(0);
            step(wB, aB);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // What crosses the barrier between a tile's k loop and its stores: the packed 16-bit outputs (no tail) or the tail's accumulators.
    // finish() runs right behind the k loop, in the same phase: bias, SiLU and packing of the main conv (and the tail's MFMAs) -- the k
    // loop alone is shorter than the other half's store + stage + fetch phase, so this moves vector work from the longer phase to the
    // shorter one; store_out() (tail bias / SiLU, address arithmetic, stores) stays in the next phase.
    uint4 oc[TAIL == 0 ? MF : 1][2];
    uint2 oc5[NF == 5 ? MF : 1];  // fifth fragment of the merged form
    f32x4 acc2c[TAIL > 0 ? MF : 1][TAIL > 0 ? TF : 1];
    auto finish = [&]() {
        float bias[NF * 4];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const float4 bv = *reinterpret_cast<const float4 *>(s_bias + (f < 4 ? (f >> 1) * 32 + g * 8 + (f & 1) * 4 : 64 + g * 4));
            bias[f * 4 + 0] = bv.x; bias[f * 4 + 1] = bv.y; bias[f * 4 + 2] = bv.z; bias[f * 4 + 3] = bv.w;
        }
        hx8 w2[TAIL > 0 ? 2 * TF : 1];
        if constexpr (TAIL > 0) {
#pragma unroll
            for (int i = 0; i < 2 * TF; ++i) w2[i] = *reinterpret_cast<const hx8 *>(w2lds + (i * 64 + lane) * 16);
        }
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            float v[NF * 4];
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[f * 4 + r] = acc[mf][f][r] + bias[f * 4 + r];
            if (do_act) {
#pragma unroll
                for (int c = 0; c < NF * 4; ++c) v[c] = silu_f(v[c]);
            }
            uint4 o[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                o[hh].x = HX<F16>::pack2(v[hh * 8 + 0], v[hh * 8 + 1]); o[hh].y = HX<F16>::pack2(v[hh * 8 + 2], v[hh * 8 + 3]);
                o[hh].z = HX<F16>::pack2(v[hh * 8 + 4], v[hh * 8 + 5]); o[hh].w = HX<F16>::pack2(v[hh * 8 + 6], v[hh * 8 + 7]);
            }
            if constexpr (TAIL == 0) {
                oc[mf][0] = o[0]; oc[mf][1] = o[1];
                if constexpr (NF == 5) { oc5[mf].x = HX<F16>::pack2(v[16], v[17]); oc5[mf].y = HX<F16>::pack2(v[18], v[19]); }
            }
            else {
#pragma unroll
                for (int f = 0; f < TF; ++f) acc2c[mf][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    hx8 a2;
                    __builtin_memcpy(&a2, &o[ks], 16);
#pragma unroll
                    for (int f = 0; f < TF; ++f) acc2c[mf][f] = HX<F16>::mfma(w2[ks * TF + f], a2, acc2c[mf][f]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto store_out = [&](int t) {
        int b, oy0, ox0;
        tile_origin(t, b, oy0, ox0);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            const int ty = ptyx[mf] >> 16, tx = ptyx[mf] & 0xffff;
            const bool ok = ptyx[mf] >= 0 && (oy0 + ty < P.Hout) && (ox0 + tx < P.Wout);
            const int64_t opix = (int64_t)(oy0 + ty) * P.Wout + ox0 + tx;
            if constexpr (TAIL == 0) {
                if (ok) {
                    bf16_t *obase = (bf16_t *)P.out + (int64_t)b * P.out_bs + P.out_co + opix * P.out_cs;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int occ = cb * 8 + hh * 4 + g;  // 8-channel chunk index inside the output slice
                        if (occ * 8 + 8 <= P.cout) *reinterpret_cast<uint4 *>(obase + (int64_t)(occ >> P.out_bsh) * P.out_ps + ((occ & P.out_bmask) << 3)) = oc[mf][hh];
                    }
                    if constexpr (NF == 5) {
                        const int occ = 8 + (g >> 1);  // channels 64 + g*4 .. + 3: half of chunk 8 or 9
                        if (occ * 8 + 8 <= P.cout) *reinterpret_cast<uint2 *>(obase + (int64_t)(occ >> P.out_bsh) * P.out_ps + ((occ & P.out_bmask) << 3) + (g & 1) * 4) = oc5[mf];
                    }
                }
            } else if constexpr (TAIL == 4) {
                if (ok) {  // (2-D launch, 16-B aligned fp32 rows, cout2 a multiple of 4: checked by the host)
                    float *op = P.out2 + (int64_t)b * P.out2_bs + opix * P.out2_cs + P.out2_co + g * 4;
#pragma unroll
                    for (int f = 0; f < TF; ++f) {
                        const float4 bv = *reinterpret_cast<const float4 *>(s_bias2 + f * 16 + g * 4);
                        if (f * 16 + g * 4 + 4 <= P.cout2)
                            *reinterpret_cast<float4 *>(op + f * 16) = make_float4(acc2c[mf][f][0] + bv.x, acc2c[mf][f][1] + bv.y, acc2c[mf][f][2] + bv.z, acc2c[mf][f][3] + bv.w);
                    }
                }
            } else {  // + bias, SiLU, 16 bit -> the tail's own output slice
                float v2[16];
#pragma unroll
                for (int f = 0; f < TF; ++f) {
                    const float4 bv = *reinterpret_cast<const float4 *>(s_bias2 + (f >> 1) * 32 + g * 8 + (f & 1) * 4);
                    v2[f * 4 + 0] = silu_f(acc2c[mf][f][0] + bv.x); v2[f * 4 + 1] = silu_f(acc2c[mf][f][1] + bv.y);
                    v2[f * 4 + 2] = silu_f(acc2c[mf][f][2] + bv.z); v2[f * 4 + 3] = silu_f(acc2c[mf][f][3] + bv.w);
                }
                if (ok) {
                    bf16_t *obase = (bf16_t *)P.out + (int64_t)b * P.out_bs + P.out_co + opix * P.out_cs;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        uint4 o2;
                        o2.x = HX<F16>::pack2(v2[hh * 8 + 0], v2[hh * 8 + 1]); o2.y = HX<F16>::pack2(v2[hh * 8 + 2], v2[hh * 8 + 3]);
                        o2.z = HX<F16>::pack2(v2[hh * 8 + 4], v2[hh * 8 + 5]); o2.w = HX<F16>::pack2(v2[hh * 8 + 6], v2[hh * 8 + 7]);
                        const int occ = hh * 4 + g;
                        if (occ * 8 + 8 <= P.cout2) *reinterpret_cast<uint4 *>(obase + (int64_t)(occ >> P.out_bsh) * P.out_ps + ((occ & P.out_bmask) << 3)) = o2;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // phase q of a half (q = p - h): 0 = stage tile 0; odd = k loop + finish of tile q/2; even = store tile q/2 - 1, stage tile q/2, fetch tile q/2 + 1
    STAMP_INIT  // (diagnostic build: 0 barrier wait, 1 epilogue, 2 tile copy to LDS incl. the wait for its loads, 3 k loop, 4 fetch issue)
    for (int p = 0; p < np; ++p) {
        const int q = p - h;
        if (q >= 0) {
            const int i = q >> 1;
            if (q & 1) {
                if (i < n_h) { kloop(); finish(); STAMP(3) STAMP_TILE }
            } else {
                if (i >= 1 && i - 1 < n_h) { store_out(t0 + h + 2 * (i - 1)); STAMP(1) }
                if (i < n_h) {
                    store_tile();
                    STAMP(2)
                    if (i + 1 < n_h) { fetch_tile(t0 + h + 2 * (i + 1)); STAMP(4) }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        STAMP(0)
    }
    STAMP_FLUSH
}

// ------------------------------------------------------------------------------------------------ host side

static int ilog2(int v) { int s = 0; while ((1 << s) < v) ++s; return s; }

int conv_ksteps(int ks, int CK) { return ((ks == 3 ? 9 : 1) * (CK / 8) + 3) / 4; }

ConvTiling plan_conv(int ks, int stride, int cin, int cout, int Hout, int Wout, bool pair) {
    ConvTiling t;
    t.NF = cout <= 16 ? 1 : (cout <= 32 ? 2 : 4);
    if (ks == 1) {
        // 1x1: the whole batch is one long pixel row (caller passes Hout = 1, Wout = B*H*W)
        static const int ck1 = getenv("OBB_CK1") ? atoi(getenv("OBB_CK1")) : 64;
        t.TH = 1; t.MF = 2; t.TW = 64 * t.MF;
        t.CK = cin >= 128 ? std::min(ck1, 64) : (cin >= 64 ? 64 : 32);
        return t;
    }
    int cin8 = (cin + 7) / 8 * 8;
    if (Hout % 13 == 0 && Wout % 13 == 0) { t.TH = 13; t.TW = 13; t.MF = 3; }
    else if (Hout >= 16 && Wout >= 16) { t.TH = 8; t.TW = 16; t.MF = 2; }
    else { t.TH = 8; t.TW = 8; t.MF = 1; }
    static const int ck3 = getenv("OBB_CK3") ? std::min(atoi(getenv("OBB_CK3")), 16) : 16;  // weights stage = 5 * NF KiB
    int ck = 1;
    while (ck * 2 <= cin8 && ck * 2 <= ck3) ck *= 2;  // power of two <= min(cin8, ck3)
    if (ck < 8) ck = 8;
    if (stride == 2 && ck > 16) ck = 16;             // keep the (2T+1)^2 halo tile small enough for several groups per CU
    t.CK = ck;
    // weights-resident single-stage form (WRES): 3x3 stride 1 on 13 x 13 tiles with 32 or 64 input channels
    // (measured per 256 tiles: 64 -> 16 couts at 52^2 60.9 -> 50.4 us, 64 -> 32 at 26^2 24.6 -> 23.6 us; the 64-cout groups (104 KiB of LDS, one
    //  group per CU) gained nothing -- 112.6 -> 115.7 us -- so they keep the 16-channel stages: their k loop is bound by LDS operand
    //  reads, 7 KiB per 12 MFMAs and wave, not by the per-stage global-memory latency)
    if (stride == 1 && t.MF == 3 && (cin == 32 || cin == 64) && (t.NF == 2 || t.NF == 1)) t.CK = cin;
    // 64 -> 64-cout groups: the same single-stage packing, run by k_conv3_pair (two half-groups per CU sharing the resident weights)
    if (pair && stride == 1 && t.MF == 3 && cin == 64 && t.NF == 4) t.CK = 64;
    // stride 2, 64 input channels: 4-row stripes of 13 outputs per half-group (one fragment per wave), same packing
    if (pair && stride == 2 && cin == 64 && t.NF == 4 && Wout % 13 == 0 && Hout >= 4) { t.TH = 4; t.TW = 13; t.MF = 1; t.CK = 64; }
    return t;
}

std::vector<bf16_t> pack_conv_weights(const float *w, int cout, int cin, int ks, const ConvTiling &t, const int *perm, int in_u8, bool f16) {
    const int CK = t.CK, NF = t.NF, cpk = CK / 8;
    const int cin_eff = in_u8 ? 8 : cin;
    const int nstage = (cin_eff + CK - 1) / CK;
    const int kst = conv_ksteps(ks, CK);
    const int ncb = (cout + 16 * NF - 1) / (16 * NF);
    const int taps = ks * ks;
    std::vector<bf16_t> out((size_t)ncb * nstage * kst * NF * 64 * 8, 0);
    size_t o = 0;
    for (int cb = 0; cb < ncb; ++cb)
        for (int st = 0; st < nstage; ++st)
            for (int k = 0; k < kst; ++k)
                for (int f = 0; f < NF; ++f)
                    for (int lane = 0; lane < 64; ++lane) {
                        int r = lane & 15, gq = lane >> 4;
                        int co = cb * 16 * NF + (r >> 2) * 4 * NF + f * 4 + (r & 3);
                        int q = k * 4 + gq;
                        int tap = (ks == 3) ? q / cpk : 0;
                        int c0 = (ks == 3) ? (q % cpk) * 8 : q * 8;
                        for (int j = 0; j < 8; ++j, ++o) {
                            int c = st * CK + c0 + j;
                            bool ok = co < cout && c < cin && tap < taps && (ks == 3 || c0 < CK);
                            if (!ok) continue;
                            int src = perm ? perm[co] : co;
                            out[o] = host_to_half(w[((size_t)src * cin + c) * taps + tap], f16);
                        }
                    }
    return out;
}

static bool conv_multi_stage(const ConvLaunch &L) { return (L.in_u8 ? 8 : L.cin) > L.CK; }
static bool conv_wres(const ConvLaunch &L) { return L.ks == 3 && L.CK > 16; }
static int tail_nf(int cout2);
// k_conv3_pair's shapes (plan_conv gives them CK = 64): everything else with CK = 64 stays on the one-group WRES form of k_conv_igemm
static bool conv_pair(const ConvLaunch &L) {
    if (L.ks != 3 || L.NF != 4 || L.CK != 64 || L.cin != 64 || L.TW != 13 || L.in_u8 || L.out_f32 || L.res.p || L.up_c != 0) return false;
    if (L.cout == 80) return L.stride == 1 && L.MF == 3 && L.TH == 13 && L.tail_cout == 0;  // two merged siblings, 64 + 16 couts: one group of five fragments
    if (L.cout % 64) return false;
    if (L.stride == 1)
        return L.MF == 3 && L.TH == 13 && (L.tail_cout == 0 || (!L.tail_act16 && tail_nf(L.tail_cout) == 4 && L.cout == 64 && L.tail_cout % 4 == 0 && L.tail_out_hw == 0 &&
                                                                ((L.tail_out.cs | L.tail_out.co) & 3) == 0));
    return L.stride == 2 && L.MF == 1 && L.TH == 4 && (L.tail_cout == 0 || (L.tail_act16 && L.tail_cout == 64 && L.cout == 64));
}

// LDS layout: [input tile (one channel stage) | weights of the stage].  The epilogue's output staging starts at offset 0; for
// multi-stage layers it may run over the weights as well (they are re-staged at every stage anyway), single-stage layers keep
// their weights resident across tiles, so there the staging area must fit in front of them.
static size_t conv_act_bytes(const ConvLaunch &L) {
    int THin = (L.TH - 1) * L.stride + L.ks, TWin = (L.TW - 1) * L.stride + L.ks;
    size_t in_tile = (size_t)std::max(1, L.NI) * THin * TWin * (L.in_u8 ? 16 : L.CK * 2 + 16);
    size_t out_tile = L.out_f32 ? 0 : (size_t)64 * L.MF * (32 * L.NF + 16);
    size_t a = conv_multi_stage(L) ? in_tile : std::max(in_tile, out_tile);
    return (a + 15) / 16 * 16;
}

static size_t conv_main_lds(const ConvLaunch &L) {
    size_t out_tile = L.out_f32 ? 0 : (size_t)64 * L.MF * (32 * L.NF + 16);
    return (std::max(conv_act_bytes(L) + (size_t)conv_ksteps(L.ks, L.CK) * L.NF * 1024, out_tile) + 15) / 16 * 16;
}

static int tail_nf(int cout2) { return cout2 <= 16 ? 1 : (cout2 <= 32 ? 2 : 4); }

size_t conv_lds_bytes(const ConvLaunch &L) {
    size_t t = L.tail_cout > 0 ? (size_t)conv_ksteps(1, 16 * L.NF) * tail_nf(L.tail_cout) * 1024 : 0;  // tail weights behind everything else
    return conv_main_lds(L) + t;
}

int conv_ni_supported(const ConvLaunch &L) {
    // the register-direct store path of k_conv_igemm (64-cout groups, no tail, 16-bit plain-NHWC output) on the one-fragment-per-wave tile
    if (L.ks != 3 || L.NF != 4 || L.MF != 1 || L.in_u8 || L.out_f32 || L.tail_cout > 0 || L.up_c > 0 || L.CK > 16) return 1;  // (channel-blocked slices included: an image is one stride inside every block plane)
    if (L.Hout != L.Wout || (L.Hout != 4 && L.Hout != 2)) return 1;
    return 64 / (L.Hout * L.Wout);
}

bool conv_tail_supported(int ks, int MF, int NF, int cout1, int cout2, bool act16, int TH) {
    if (cout1 != 16 * NF || cout2 < 1 || cout2 > 64) return false;
    const int nf2 = tail_nf(cout2);
    if (act16 && ks == 3 && MF == 1 && NF == 4 && TH == 4) return cout2 == 64;  // k_conv3_pair, stride-2 stripes
    if (act16) return ks == 3 && MF == 3 && (NF == 2 || NF == 4) && nf2 == NF && cout2 == 16 * nf2;  // whole 16-bit rows of the staging area
    return (ks == 3 && MF == 3 && NF == 4 && nf2 == 4) || (ks == 3 && MF == 3 && NF == 1 && nf2 == 1) || (ks == 1 && MF == 2 && NF == 4 && nf2 == 1);
}

template <typename K>
static hipError_t launch_big_lds(K kernel, const ConvParams &P, dim3 grid, size_t lds, hipStream_t st) {
    static bool attr_set = false;  // one flag per kernel instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);  // + the static arrays (bias, LUT) <= 160 KiB
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kernel, grid, dim3(256), lds, st, P);
    return hipGetLastError();
}

template <int KS, int MF, int NF, bool F16>
static hipError_t launch_t2(const ConvLaunch &L, const ConvParams &P, dim3 grid, size_t lds, hipStream_t st) {
    if constexpr (KS == 3 && MF == 3) {
        if (conv_wres(L)) {  // weights-resident single-stage form: plain / 16-bit tail (T16) / fp32 head tail
            if (L.tail_cout > 0 && L.tail_act16) return hipErrorInvalidValue;  // (the fused cv1 sits behind stride-2 convs only)
            if (L.tail_cout > 0) {
                constexpr int T = NF == 4 ? 4 : 1;
                if constexpr (NF == 4 || NF == 1) {
                    if (tail_nf(L.tail_cout) != T) return hipErrorInvalidValue;
                    return launch_big_lds(k_conv_igemm<KS, MF, NF, false, false, F16, T, false, false, true>, P, grid, lds, st);
                } else return hipErrorInvalidValue;
            }
            return launch_big_lds(k_conv_igemm<KS, MF, NF, false, false, F16, 0, false, false, true>, P, grid, lds, st);
        }
    }
    if (L.tail_cout > 0 && L.tail_act16) {  // stride-2 backbone conv + the cv1 of the following C3k2 block
        if constexpr (KS == 3 && MF == 3 && (NF == 2 || NF == 4)) {
            if (L.in_u8 || L.out_f32 || tail_nf(L.tail_cout) != NF) return hipErrorInvalidValue;
            hipLaunchKernelGGL((k_conv_igemm<KS, MF, NF, false, false, F16, NF, false, true>), grid, dim3(256), lds, st, P);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    if (L.tail_cout > 0) {  // only the three shapes of the OBB head are instantiated (conv_tail_supported)
        constexpr int T = (KS == 3 && MF == 3 && NF == 4) ? 4 : 1;
        if constexpr ((KS == 3 && MF == 3 && (NF == 4 || NF == 1)) || (KS == 1 && MF == 2 && NF == 4)) {
            if (L.in_u8 || L.out_f32 || tail_nf(L.tail_cout) != T) return hipErrorInvalidValue;
            hipLaunchKernelGGL((k_conv_igemm<KS, MF, NF, false, false, F16, T>), grid, dim3(256), lds, st, P);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    if (L.up_c > 0) {  // virtual upsample-concat input: one shape (the 1x1 convs behind Upsample + Concat in the neck)
        if constexpr (KS == 1 && MF == 2 && NF == 4) {
            if (L.in_u8 || L.out_f32) return hipErrorInvalidValue;
            hipLaunchKernelGGL((k_conv_igemm<KS, MF, NF, false, false, F16, 0, true>), grid, dim3(256), lds, st, P);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    if (L.in_u8) {
        if constexpr (KS == 3) hipLaunchKernelGGL((k_conv_igemm<KS, MF, NF, true, false, F16>), grid, dim3(256), lds, st, P);
        else return hipErrorInvalidValue;
    } else if (L.out_f32) hipLaunchKernelGGL((k_conv_igemm<KS, MF, NF, false, true, F16>), grid, dim3(256), lds, st, P);
    else if (P.NI > 1) {
        if constexpr (KS == 3 && MF == 1 && NF == 4) hipLaunchKernelGGL((k_conv_igemm<KS, MF, NF, false, false, F16, 0, false, false, false, true>), grid, dim3(256), lds, st, P);
        else return hipErrorInvalidValue;
    } else hipLaunchKernelGGL((k_conv_igemm<KS, MF, NF, false, false, F16>), grid, dim3(256), lds, st, P);
    return hipGetLastError();
}

template <int KS, int MF, int NF>
static hipError_t launch_t(const ConvLaunch &L, const ConvParams &P, dim3 grid, size_t lds, hipStream_t st) {
    return L.f16 ? launch_t2<KS, MF, NF, true>(L, P, grid, lds, st) : launch_t2<KS, MF, NF, false>(L, P, grid, lds, st);
}

template <int KS, int MF>
static hipError_t launch_nf(const ConvLaunch &L, const ConvParams &P, dim3 grid, size_t lds, hipStream_t st) {
    switch (L.NF) {
        case 1: return launch_t<KS, MF, 1>(L, P, grid, lds, st);
        case 2: return launch_t<KS, MF, 2>(L, P, grid, lds, st);
        case 4: return launch_t<KS, MF, 4>(L, P, grid, lds, st);
    }
    return hipErrorInvalidValue;
}

static hipError_t launch_conv_impl(const ConvLaunch &L, hipStream_t st, unsigned long long **stamps_out);

hipError_t launch_conv(const ConvLaunch &L, hipStream_t st) {
    unsigned long long *sd = nullptr;
    hipError_t e = launch_conv_impl(L, st, &sd);
#ifdef OBB_STAMPS
    if (e == hipSuccess && sd) {  // diagnostic build: synchronise and print this launch's phase sums (cycles per wave and tile)
        unsigned long long h[8];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, sd, 64, hipMemcpyDeviceToHost);
        const double n = h[5] ? (double)h[5] : 1.0;
        if (conv_pair(L))
            fprintf(stderr, "STAMPS pair k%d s%d cin%d cout%d out%dx%d tail%d | per wave-tile cycles: barrier %.0f  epilogue %.0f  lds_copy %.0f  kloop %.0f  fetch_issue %.0f  (wave-tiles %.0f)\n",
                    L.ks, L.stride, L.cin, L.cout, L.Hout, L.Wout, L.tail_cout, h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, n);
        else
        fprintf(stderr, "STAMPS k%d s%d cin%d cout%d out%dx%d CK%d MF%d NF%d tail%d | per wave-tile cycles: barrier %.0f  lds_stage %.0f  prefetch %.0f  kloop %.0f  epilogue %.0f  (wave-tiles %.0f)\n",
                L.ks, L.stride, L.cin, L.cout, L.Hout, L.Wout, L.CK, L.MF, L.NF, L.tail_cout, h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, n);
    }
#endif
    return e;
}

static hipError_t launch_conv_impl(const ConvLaunch &L, hipStream_t st, unsigned long long **stamps_out) {
    ConvParams P;
    P.in = L.in.p; P.in_bs = L.in.bs; P.in_cs = L.in.cs; P.in_co = L.in.co;
    P.out = L.out.p; P.out_bs = L.out.bs; P.out_cs = L.out.cs; P.out_co = L.out.co;
    P.res = (const bf16_t *)L.res.p; P.res_bs = L.res.bs; P.res_cs = L.res.cs; P.res_co = L.res.co;
    P.in_bsh = P.out_bsh = P.res_bsh = 31;
    P.in_bmask = P.out_bmask = P.res_bmask = 0x7fffffff;
    P.in_ps2 = 0; P.out_ps = 0; P.res_ps = 0;
    int64_t in_block_span = 0;  // blocked input: elements from the slice's first block to the end of its last block (one image)
    {
        auto blocked = [](const TensorRef &T, int C, int &bsh, int &bmask, int64_t &ps, int &co, int &cs, int64_t *span) -> bool {
            if (T.cpb <= 0) return true;
            const int blk = 8 * T.cpb;
            if ((T.cpb & (T.cpb - 1)) || T.co % blk || T.cs != blk) return false;
            bsh = 0;
            while ((1 << bsh) < T.cpb) ++bsh;
            bmask = T.cpb - 1;
            ps = T.ps;
            if (span) *span = (int64_t)((C + blk - 1) / blk - 1) * T.ps;
            co = 0;  // the slice offset is a whole number of blocks: folded into the base pointer by the caller below
            return true;
        };
        int64_t ps_in = 0;
        int co_in = P.in_co, cs_in = P.in_cs;
        if (!L.in_u8) {
            if (!blocked(L.in, L.cin, P.in_bsh, P.in_bmask, ps_in, co_in, cs_in, &in_block_span)) return hipErrorInvalidValue;
            if (L.in.cpb > 0) {
                if (ps_in * 2 >= (1ll << 32)) return hipErrorInvalidValue;
                P.in = (const bf16_t *)L.in.p + (int64_t)(L.in.co / (8 * L.in.cpb)) * L.in.ps;
                P.in_co = 0;
                P.in_ps2 = (unsigned)(ps_in * 2);
            }
        } else if (L.in.cpb > 0) return hipErrorInvalidValue;
        int co_out = P.out_co, cs_out = P.out_cs;
        if (!blocked(L.out, L.cout, P.out_bsh, P.out_bmask, P.out_ps, co_out, cs_out, nullptr) || (L.out.cpb > 0 && L.out_f32)) return hipErrorInvalidValue;
        if (L.out.cpb > 0) { P.out = (bf16_t *)L.out.p + (int64_t)(L.out.co / (8 * L.out.cpb)) * L.out.ps; P.out_co = 0; }
        int co_res = P.res_co, cs_res = P.res_cs;
        if (L.res.p) {
            if (!blocked(L.res, L.cout, P.res_bsh, P.res_bmask, P.res_ps, co_res, cs_res, nullptr)) return hipErrorInvalidValue;
            if (L.res.cpb > 0) { P.res = (const bf16_t *)L.res.p + (int64_t)(L.res.co / (8 * L.res.cpb)) * L.res.ps; P.res_co = 0; }
        }
    }
    if (!L.lut) return hipErrorInvalidValue;  // also the sink of the unconditional stores (lut + 512 B .. + 1.5 KiB): never NULL
    P.wpk = L.wpk; P.bias = L.bias; P.lut = L.lut;
    P.Hin = L.Hin; P.Win = L.Win; P.Hout = L.Hout; P.Wout = L.Wout;
    P.cin = L.cin; P.cout = L.cout; P.stride = L.stride; P.act = L.act; P.flip_bgr = L.flip_bgr;
    P.TH = L.TH; P.TW = L.TW; P.CK = L.CK; P.sh = ilog2(L.CK / 8);
    P.tiles_x = L.tiles_x; P.tiles_y = L.tiles_y; P.out_hw = L.out_hw; P.act_bytes = (int)conv_act_bytes(L);
    const int NI = std::max(1, L.NI);
    P.NI = NI; P.B = L.B;
    if (NI > 1 && (NI != conv_ni_supported(L) || L.TH != L.Hout || L.TW != L.Wout || L.tiles_x != 1 || L.tiles_y != 1 || L.out_hw)) return hipErrorInvalidValue;
#if defined(OBB_STAMPS) || defined(OBB_DIAG)  // timing-only ablations exist in the diagnostic build alone (tools/stamp_conv.sh): the product library ignores the variable
    static const int dbg = getenv("OBB_CONV_DBG") ? atoi(getenv("OBB_CONV_DBG")) : 0;
    P.dbg = dbg;
#else
    P.dbg = 0;
#endif
    P.stamps = nullptr;
#ifdef OBB_STAMPS
    static unsigned long long *stamp_dev = nullptr;
    if (!stamp_dev) (void)hipMalloc((void **)&stamp_dev, 64);
    (void)hipMemsetAsync(stamp_dev, 0, 64, st);
    P.stamps = stamp_dev;
    *stamps_out = stamp_dev;
#endif
    P.in2 = L.in2.p; P.in2_cs = L.in2.cs; P.in2_co = L.in2.co; P.up_c = L.up_c; P.up_W = L.up_W; P.up_HW = L.up_HW; P.in2_span_bytes = 0;
    if (L.up_c > 0) {  // virtual [upsample | skip] concat: 1-D 1x1 launches over plain NHWC sources only
        if (L.ks != 1 || L.in_u8 || L.B != 1 || L.Hin != 1 || L.in.cpb || L.in2.cpb || !L.in2.p || L.up_c % L.CK || L.up_c >= L.cin || (L.up_W & 1) ||
            (L.up_HW % L.up_W) || ((L.up_HW / L.up_W) & 1) || L.Win % L.up_HW)
            return hipErrorInvalidValue;
        int64_t s2 = ((int64_t)L.Win * L.in2.cs - L.in2.co) * 2, s1 = ((int64_t)(L.Win / 4) * L.in.cs - L.in.co) * 2;
        if (s1 <= 0 || s2 <= 0 || s1 >= (1ll << 32) - 65536 || s2 >= (1ll << 32) - 65536) return hipErrorInvalidValue;
        P.in2_span_bytes = (unsigned)s2;
    }
    P.w2pk = L.tail_wpk; P.bias2 = L.tail_bias; P.out2 = (float *)L.tail_out.p; P.out2_bs = L.tail_out.bs; P.out2_cs = L.tail_out.cs;
    P.out2_co = L.tail_out.co; P.out2_hw = L.tail_out_hw; P.cout2 = L.tail_cout; P.kst2 = conv_ksteps(1, 16 * L.NF);
    P.w2_off = (int)conv_main_lds(L);
    if (L.tail_cout > 0 && (!conv_tail_supported(L.ks, L.MF, L.NF, L.cout, L.tail_cout, L.tail_act16, L.TH) || !L.tail_wpk || !L.tail_bias || (!L.tail_act16 && !L.tail_out.p))) return hipErrorInvalidValue;

    int cin_eff = L.in_u8 ? 8 : L.cin;
    P.nstage = (cin_eff + L.CK - 1) / L.CK;
    P.kst = conv_ksteps(L.ks, L.CK);
    {
        int64_t span = ((int64_t)L.Hin * L.Win * L.in.cs - L.in.co) * 2;  // from the slice's first element to the end of the image
        if (L.in.cpb > 0) span = (in_block_span + (int64_t)L.Hin * L.Win * L.in.cs) * 2;  // ... to the end of the slice's last block
        if (L.up_c > 0) span = ((int64_t)(L.Win / 4) * L.in.cs - L.in.co) * 2;  // the low-res source of a virtual concat
        span += (int64_t)(NI - 1) * L.in.bs * 2;  // a multi-image tile reads up to the end of its last image
        int64_t wb = (int64_t)((L.cout + 16 * L.NF - 1) / (16 * L.NF)) * P.nstage * P.kst * L.NF * 1024;
        if (span <= 0 || span >= (1ll << 32) - 65536 || wb >= (1ll << 31)) return hipErrorInvalidValue;  // 32-bit buffer offsets
        P.in_span_bytes = (unsigned)span;
        P.w_bytes = (unsigned)wb;
        if (conv_wres(L) && !conv_pair(L) && (L.stride != 1 || L.MF != 3 || L.CK != L.cin || (L.cin != 32 && L.cin != 64) || L.in_u8 || L.out_f32 || L.up_c > 0)) return hipErrorInvalidValue;
    }
    int TWin = (L.TW - 1) * L.stride + L.ks;
    P.inv_twin = 1.0f / (float)TWin;
    P.inv_tw = 1.0f / (float)L.TW;
    if ((1 << P.sh) != L.CK / 8 || NI * L.TH * L.TW > 64 * L.MF || L.MF < 1 || L.MF > 3) return hipErrorInvalidValue;
    {
        int THin = (L.TH - 1) * L.stride + L.ks;
        if (!L.in_u8 && (int64_t)NI * THin * TWin * (L.CK / 8) > (conv_wres(L) ? 8 : (L.ks == 1 ? 4 : 6)) * 256) return hipErrorInvalidValue;  // staging plan: chunks per thread
        if (L.ks == 1 && L.CK > 64) return hipErrorInvalidValue;
    }
    int ncb = (L.cout + 16 * L.NF - 1) / (16 * L.NF);
    int64_t ntiles = NI > 1 ? ((int64_t)L.B + NI - 1) / NI : (int64_t)L.B * L.tiles_y * L.tiles_x;
    if (ntiles >= (1ll << 31)) return hipErrorInvalidValue;
    P.ntiles = (int)ntiles;
    if (conv_pair(L)) {  // one 512-thread workgroup per CU, its two halves walking alternate tiles
        const int nf5 = L.cout == 80 ? 1 : 0;
        if (nf5) ncb = 1;
        const int64_t groups = std::max<int64_t>(1, 256 / ncb);
        P.tpw = (int)std::max<int64_t>(2, (ntiles + groups - 1) / groups);
        P.gx = (int)((ntiles + P.tpw - 1) / P.tpw); P.ncb = ncb;
        const size_t lds = 2 * (size_t)P.act_bytes + (size_t)P.kst * (L.NF + nf5) * 1024 + (L.tail_cout > 0 ? 2 * 4 * 1024 : 0);
        if (P.kst != 18 || P.nstage != 1 || lds > 158 * 1024) return hipErrorInvalidValue;
        const dim3 grid((unsigned)((P.gx + 7) / 8 * 8 * ncb));
        auto go = [&](auto kernel) -> hipError_t {
            static std::vector<const void *> attr_set;  // (the four instantiations share one function-pointer type: keyed by address)
            if (std::find(attr_set.begin(), attr_set.end(), (const void *)kernel) == attr_set.end()) {
                hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);  // + static bias arrays <= 160 KiB
                if (e != hipSuccess) return e;
                attr_set.push_back((const void *)kernel);
            }
            hipLaunchKernelGGL(kernel, grid, dim3(512), lds, st, P);
            return hipGetLastError();
        };
        if (L.stride == 2) {
            if (L.tail_cout > 0) return L.f16 ? go(k_conv3_pair<true, 16, 2>) : go(k_conv3_pair<false, 16, 2>);
            return L.f16 ? go(k_conv3_pair<true, 0, 2>) : go(k_conv3_pair<false, 0, 2>);
        }
        if (nf5) return L.f16 ? go(k_conv3_pair<true, 0, 1, 5>) : go(k_conv3_pair<false, 0, 1, 5>);
        if (L.tail_cout > 0) return L.f16 ? go(k_conv3_pair<true, 4, 1>) : go(k_conv3_pair<false, 4, 1>);
        return L.f16 ? go(k_conv3_pair<true, 0, 1>) : go(k_conv3_pair<false, 0, 1>);
    }
    // tiles per workgroup: keep >= ~8 groups per CU in flight, walk up to 8 consecutive tiles per group beyond that
    static const int tpw_max = getenv("OBB_TPW") ? atoi(getenv("OBB_TPW")) : 8;
    int64_t tpw = ntiles * ncb / (256 * 8);
    P.tpw = (int)std::max<int64_t>(1, std::min<int64_t>(tpw, tpw_max));
    if (conv_wres(L)) P.tpw = (int)std::max<int64_t>(1, std::min<int64_t>(ntiles * ncb / (256 * 2), 16));  // one group per CU: two rounds of groups, <= 16 tiles each
    P.gx = (int)((ntiles + P.tpw - 1) / P.tpw); P.ncb = ncb;
    dim3 grid((unsigned)((P.gx + 7) / 8 * 8 * ncb));  // 1-D: see the XCD-aware decoding at the top of the kernel
    size_t lds = conv_lds_bytes(L);
    if (lds > (conv_wres(L) ? 152 : 64) * 1024) return hipErrorInvalidValue;
    if (L.ks == 3) {
        switch (L.MF) {
            case 1: return launch_nf<3, 1>(L, P, grid, lds, st);
            case 2: return launch_nf<3, 2>(L, P, grid, lds, st);
            case 3: return launch_nf<3, 3>(L, P, grid, lds, st);
        }
    } else if (L.ks == 1) {
        switch (L.MF) {
            case 1: return launch_nf<1, 1>(L, P, grid, lds, st);
            case 2: return launch_nf<1, 2>(L, P, grid, lds, st);
            case 3: return launch_nf<1, 3>(L, P, grid, lds, st);
        }
    }
    return hipErrorInvalidValue;
}

}  // namespace obb
