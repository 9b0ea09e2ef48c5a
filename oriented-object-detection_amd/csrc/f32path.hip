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
//   * epilogue: + bias, SiLU (v_exp_f32 + v_rcp_f32, 1 ulp each: silu32 below), + residual, fp32 store into a channel slice of the consumer's buffer.
#include "f32path.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

namespace obb {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct C32Params {
    const void *in; int64_t in_bs; int in_cs, in_co;
    const void *in2; int in2_cs, in2_co; unsigned in2_span_bytes; int up_c, up_W, up_HW;  // VCAT: second source of a virtual [upsample | skip] concat
    float *out; int64_t out_bs; int out_cs, out_co;
    const float *res; int64_t res_bs; int res_cs, res_co;
    const float *wpk, *bias, *lut;
    const float *w2, *b2; float *out2; int64_t out2_bs; int out2_cs, out2_co, out2_hw, cout2, act2, kst2;  // TAIL: fused trailing 1x1
    float *cmax; int64_t cmax_bs;  // TAIL = 1 over the class logits: per-anchor maximum of the tail's outputs (nullptr: none)
    int Hin, Win, Hout, Wout, cin, cout, stride, act, flip_bgr;
    int TH, TW, CK, sh /*log2(CK/4)*/, tiles_x, tiles_y, ntiles, nstage, kst, out_hw, ncb;
    int krem, wcb;  // 4-channel chunks of a stage beyond the kst whole pieces (0..3: one MFMA each); bytes of one cout fragment's stage weights
    int dbg;    // diagnostic build (-DOBB_DIAG) only: timing ablations, see launch_conv32
    int tstep;  // > 0: resident workgroups, each walks the tiles t, t + tstep, ... (see XT in k_conv_f32)
    int dw_act;  // DW: SiLU behind the depthwise conv
    int NI, B;  // NI > 1: a tile = NI whole images of a small map (TH x TW = the map), B images in all
    float inv_twin, inv_tw;
    unsigned in_span_bytes;  // buffer-descriptor range of one image's input slice (its check returns zeros past the end)
    // channel-blocked tensors (TensorRef::cpb = 2: blocks of 8 channels, per image [C / 8][pixel][8], so that an 8- or 16-channel stage of a
    // 3x3 layer reads whole dense runs instead of 32- / 64-byte pieces of wide pixel rows -- measured x3.75 HBM re-read on model.3 with plain
    // NHWC): element (b, pixel, c) = b * bs + (c >> 3) * ps + pixel * 8 + (c & 7); cs = 8.  in: 2-D launches (3x3 / depthwise prologue);
    // in2: the skip source of a virtual concat; out: the layer's own output (not the TAIL's)
    int in_blk, in2_blk, out_blk, in_ps, in2_ps, out_ps, up_stages;
    int64_t in2_bs;
    unsigned in_sadd, in2_sadd;  // bytes from one channel stage to the next
};

// SiLU with the hardware exp2 / reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each): relative error <= ~1e-6 for |x| <= 10, below the
// summation-order noise of an fp32 convolution (~sqrt(K) * 6e-8); the library expf + IEEE division cost 4x the epilogue time
__device__ __forceinline__ float silu32(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// Workgroup = NW waves: WC of them along cout (16 couts each), WP = NW / WC along the pixel fragments; wave (wc, wp) owns the fragments
// f = wp + WP * i, i < MFM, of the tile (interleaved, so that a partly filled tile leaves every wave about the same work).
//   * BOTH operands of a channel stage live in LDS: the activation tile [pixels + halo][CK] and the stage's weights of the workgroup's WC
//     cout fragments (kst pieces of 1 KiB each, already in A-fragment order).  The k loop therefore contains no global load at all
//     (a per-wave weight stream from L2 cost a `s_waitcnt vmcnt(0)` per k step -- the register rotation reads the youngest load -- and
//     with it every overlap of the next stage's fetch), only ds_read_b128 + MFMA.
//   * pipeline per channel stage (one register set, one LDS buffer): the NEXT stage's activation chunks and weight pieces are fetched
//     into registers while this stage's k loop runs; barrier; LDS write; barrier; issue the fetch after next; k loop.
//   * TAIL (= WC2 > 0): the activated output tile goes to LDS instead of memory and a second GEMM (K = cout, 16 * WC2 >= cout2 couts)
//     runs from there: the producer's tensor is never written.  Used for the last 1x1 of every head branch and for the cv1 of a
//     C3k2 block behind its stride-2 conv.
//   * VCAT: the 1x1 behind [Upsample | skip] reads both sources in place (stage-uniform choice: up_c is a multiple of CK).
//   * DW (1x1 only): a depthwise 3x3 (+ bias, SiLU) in FRONT of the 1x1, per channel stage: the stage's input tile is staged with a
//     one-pixel halo, every thread computes depthwise outputs for its (pixel, 4-channel chunk) items straight from that LDS tile (taps in
//     (ky, kx) order like k_dwconv3_f32, weights from the stage's weight block) and writes them where the 1x1's B operand is read -- the
//     depthwise tensor never exists in memory (the class branch of the head: DWConv -> Conv 1x1 [-> Conv 1x1 to the head rows]).
//   * NC = 2 (plain and VCAT forms): a wave owns TWO cout fragments (32 couts) x MFM <= 4 pixel fragments instead of one x 7: WC = 2 waves along
//     cout, 4 along the pixels, tiles of <= 256 pixels x 64 couts.  Six operand reads (2 weight + 4 activation pieces) feed 32 MFMAs where
//     the 1 x 7 shape needs eight for 28 -- the k loop turned out to be sensitive to its LDS reads (doubling the activation reads in a
//     timing-only build cost 20-27 % on every MFMA-bound layer) -- and with the cout order below a lane's eight couts are 32 contiguous
//     bytes, the four lanes of a pixel a whole 128-byte line.  Fragment nc, accumulator row g * 4 + j <-> cout 32 * block + g * 8 + nc * 4 + j
//     (pack_conv32_weights permutes the rows accordingly).
//   * BLK: some operand lives in 8-channel blocks (C32Params::in_blk / in2_blk / out_blk); a template switch, so that the plain forms carry
//     none of its address arithmetic or parameters (with run-time flags only, every 1x1 form lost 5-10 % to the extra scalar registers)
template <int KS, int MFM, int WC, int NW, bool IN_U8, bool VCAT, int TAIL, bool DW = false, int NC = 1, bool BLK = false>
__global__ __launch_bounds__(NW * 64, NW / 2) void k_conv_f32(const C32Params P) {  // <= 128 VGPRs: two workgroups per CU
    static_assert(NC == 1 || (NC == 2 && TAIL == 0 && !DW && !IN_U8), "two cout fragments per wave: plain / VCAT forms only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SKS = DW ? 3 : KS;  // kernel size the STAGING sees (halo); the MFMA loop sees KS
    constexpr int NT = NW * 64, WP = NW / WC, PAD = SKS / 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: the fragment guards below become scalar branches
    const int g = lane >> 4, pl = lane & 15;
    const int wc = wave % WC, wp = wave / WC;
    // XCD-aware order (as k_conv_igemm): the cout blocks of one pixel tile are consecutive on one XCD and share the tile through its L2
    const int xcd = blockIdx.x & 7, lin = blockIdx.x >> 3;
    const int cb = lin % P.ncb;
    int t = (lin / P.ncb) * 8 + xcd;
    if (t >= P.ntiles) return;
    // XT: a workgroup walks the tiles t, t + tstep, ... (tstep > 0: a grid of resident workgroups) and the FIRST stage of the next tile
    // is fetched under the last k loop of this one and committed to LDS before this tile's epilogue: the fetch latency and the output
    // stores of a tile, exposed once per tile otherwise (a 1x1 layer has 1-3 stages per tile), overlap the neighbouring tiles' MFMAs.
    // The fetch registers are dead again before the epilogue starts (committed), so the form costs no registers.  (TAIL / DW / uint8
    // forms need the LDS tile or further registers in their epilogues and keep one tile per workgroup.)
    constexpr bool XT = TAIL == 0 && !DW && !IN_U8;
    const int S = P.stride;
    const int THin = (P.TH - 1) * S + SKS, TWin = (P.TW - 1) * S + SKS;
    const int PST = P.CK * 4 + 16;  // bytes per staged pixel (+16 B spreads consecutive pixels over the banks)
    const int cpk = P.CK >> 2;
    const int nq = (KS == 3 ? 9 : 1) * cpk;
    const int in_px1 = THin * TWin;          // staged pixels of one image
    const int in_px = in_px1 * P.NI;
    const int tpi = P.TH * P.TW;             // output pixels of one image's part of the tile
    const int npix = tpi * P.NI;
    const float inv_tpi = 1.0f / (float)tpi, inv_in1 = 1.0f / (float)in_px1;

    int pixbase[MFM];
#pragma unroll
    for (int mf = 0; mf < MFM; ++mf) {
        const int p = (wp + WP * mf) * 16 + pl;
        const int il = P.NI > 1 ? (int)(((float)p + 0.5f) * inv_tpi) : 0;
        const int q = p - il * tpi;
        const int ty = (int)(((float)q + 0.5f) * P.inv_tw);
        const int tx = q - ty * P.TW;
        if constexpr (DW) pixbase[mf] = (p < npix ? p : 0) * PST;  // (relative to the depthwise-output tile, added below)
        else pixbase[mf] = p < npix ? (il * in_px1 + (ty * S) * TWin + tx * S) * PST : 0;  // fragments past the tile compute on pixel 0 and are dropped
    }
    const int F = (cb * WC + wc) * NC;  // (first) cout fragment of this wave
    // weights of (cout block cb, stage): WC * kst pieces of 1 KiB, contiguous (pack_conv32_weights); LDS image behind the activation tile
    const int dwb_off = ((in_px * PST + 1023) >> 10) << 10;  // DW: the depthwise-output tile [npix][CK] behind the input tile
    const int act_bytes = DW ? dwb_off + (((npix * PST + 1023) >> 10) << 10) : dwb_off;
    const int nwchunk = WC * NC * (P.wcb >> 4) + (DW ? 10 * cpk : 0);  // 16-B chunks of one stage's weights (+ DW: 9 taps + bias of the stage's channels)
    if constexpr (DW) {
#pragma unroll
        for (int mf = 0; mf < MFM; ++mf) pixbase[mf] += dwb_off;
    }
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(P.wpk + (size_t)cb * P.nstage * nwchunk * 4), 0, P.nstage * nwchunk * 16, 0x00020000);
    char *const wlds = smem + act_bytes;

    f32x4 acc[NC][MFM];
#pragma unroll
    for (int nc = 0; nc < NC; ++nc)
#pragma unroll
        for (int mf = 0; mf < MFM; ++mf) acc[nc][mf] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging plan: this thread moves the 16-B chunks idx = tid + k * NT of the [in_px][CK] tile.  Activations come through buffer
    // loads (descriptor in SGPRs, one 32-bit byte offset per chunk): an offset past the descriptor's range reads zeros, which is how
    // the zero padding is written
    constexpr int MAXLD = IN_U8 ? 3 : (KS == 1 ? (DW ? 3 : (VCAT ? 4 : 7)) : 5);  // plan_conv32 keeps a stage within that many x NT chunks of 16 B
    constexpr int MAXW = (WC * NC * (KS == 3 ? 9 : 4) * 64 + (DW ? 10 * 8 : 0) + NT - 1) / NT;  // weight chunks per thread and stage (kst <= 9 / 4; DW: + 10 x CK floats, CK <= 32)
    constexpr unsigned NOPIX = 0xffffffffu;
    const int nchunk = in_px * cpk;
    unsigned goff[IN_U8 ? 1 : MAXLD], goff2[VCAT ? MAXLD : 1];
    __amdgpu_buffer_rsrc_t in_rsrc, in2_rsrc;
    if constexpr (VCAT) in2_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((const float *)P.in2 + P.in2_co), 0, (int)P.in2_span_bytes, 0x00020000);
    // per-tile (wave-uniform) state: what the k loop and the epilogue of a tile need; `plan` also lays out the tile's fetch offsets
    struct Tile { int b, nimg, oy0, ox0, nfrag1; bool lastv; };
    auto plan = [&](int tt, Tile &T) {
        const int tx_i = tt % P.tiles_x, r_ = tt / P.tiles_x, ty_i = r_ % P.tiles_y;
        T.b = P.NI > 1 ? tt * P.NI : r_ / P.tiles_y;  // (first) image of this tile
        T.nimg = P.NI > 1 ? min(P.NI, P.B - T.b) : 1;
        T.oy0 = ty_i * P.TH; T.ox0 = tx_i * P.TW;
        const int iy0 = T.oy0 * S - PAD, ix0 = T.ox0 * S - PAD;
        // valid pixels of this tile (edge tiles are clipped): fragments past them are not computed
        const int vh = min(P.TH, P.Hout - T.oy0), vw = min(P.TW, P.Wout - T.ox0);
        const int nfrag = ((P.NI > 1 ? T.nimg * tpi : (vh == P.TH ? tpi : vh * P.TW)) + 15) >> 4;  // whole rows: a clipped tile loses its trailing rows (1-D: TH = 1, all kept)
        T.nfrag1 = (P.TH == 1) ? ((vw + 15) >> 4) : nfrag;
        T.lastv = wp + WP * (MFM - 1) < T.nfrag1;  // (wave-uniform) the last fragment of this wave exists
        if constexpr (!IN_U8) {
            in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)((const float *)P.in + (int64_t)T.b * P.in_bs + P.in_co), 0, (int)P.in_span_bytes, 0x00020000);
            int tidv = tid;
            if constexpr (XT) asm volatile("" : "+v"(tidv));  // (opaque: the tile-independent part of the offsets would be hoisted out of the tile loop and held in registers)
#pragma unroll
            for (int k = 0; k < MAXLD; ++k) {
                const int idx = tidv + k * NT;
                int pix, c;
                if constexpr (KS == 1 && !DW) { pix = idx / cpk; c = idx - pix * cpk; }  // (1x1: CK may be 48 -- three 16-channel groups in one stage)
                else { pix = idx >> P.sh; c = idx & (cpk - 1); }
                const int il = P.NI > 1 ? (int)(((float)pix + 0.5f) * inv_in1) : 0;
                const int pq = pix - il * in_px1;
                const int iy = (int)(((float)pq + 0.5f) * P.inv_twin), ix = pq - iy * TWin;
                const int gy = iy0 + iy, gx = ix0 + ix;
                const bool ok = idx < nchunk && il < T.nimg && gy >= 0 && gy < P.Hin && gx >= 0 && gx < P.Win;
                const int cch = c * 4;  // first channel of this chunk inside its stage
                if constexpr (VCAT) {  // 1-D: gx = flattened (image, y, x) of the full-resolution level
                    const int bb = gx / P.up_HW, r = gx - bb * P.up_HW;
                    const int yy = r / P.up_W, xx = r - yy * P.up_W;
                    const int64_t sp = (int64_t)bb * (P.up_HW >> 2) + (int64_t)(yy >> 1) * (P.up_W >> 1) + (xx >> 1);
                    goff[k] = ok ? (unsigned)((sp * P.in_cs + cch) * 4) : NOPIX;
                    if constexpr (BLK) {
                        const int c2 = P.in2_blk ? (cch >> 3) * P.in2_ps + (cch & 7) : cch;
                        goff2[k] = ok ? (unsigned)(((int64_t)bb * P.in2_bs + (int64_t)r * P.in2_cs + c2) * 4) : NOPIX;
                    } else goff2[k] = ok ? (unsigned)(((int64_t)gx * P.in2_cs + cch) * 4) : NOPIX;
                } else {
                    const int c1 = (BLK && P.in_blk) ? (cch >> 3) * P.in_ps + (cch & 7) : cch;
                    goff[k] = ok ? (unsigned)(((int64_t)il * P.in_bs + ((int64_t)gy * P.Win + gx) * P.in_cs + c1) * 4) : NOPIX;
                }
            }
        }
    };

    u32x4 pre[IN_U8 ? 1 : MAXLD];
    f32x4 prew[MAXW];
    auto fetch = [&](int stage) {  // issue the loads of one channel stage (registers `pre`, `prew`)
#ifdef OBB_DIAG
        if (P.dbg & 1) return;  // timing only: no global fetch (the LDS tile holds whatever it held)
#endif
        // (one descriptor + scalar offsets: per-chunk 64-bit addresses would be hoisted out of the loops and held in 2 x MAXW registers)
        const int wsoff = stage * nwchunk * 16;
#pragma unroll
        for (int k = 0; k < MAXW; ++k) {
            // (every load unconditional: a load under a branch turns the register into a loop-carried copy, and the copy waits for the load)
            const int rem = nwchunk - k * NT;  // (wave-uniform) chunks left for this round; none: the round re-reads chunk 0 and is dropped at the commit
            prew[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, rem > 0 ? min(tid, rem - 1) * 16 : 0, rem > 0 ? wsoff + k * NT * 16 : wsoff, 0));
        }
        if constexpr (!IN_U8) {
            bool second = false;
            unsigned add = BLK ? (unsigned)stage * P.in_sadd : (unsigned)(stage * P.CK * 4);
            if constexpr (VCAT) {
                second = stage * P.CK >= P.up_c;
                if (second) add = BLK ? (unsigned)(stage - P.up_stages) * P.in2_sadd : add - (unsigned)(P.up_c * 4);
            }
#pragma unroll
            for (int k = 0; k < MAXLD; ++k) {
                if constexpr (VCAT) {
                    const unsigned o = second ? goff2[k] : goff[k];
                    pre[k] = __builtin_amdgcn_raw_buffer_load_b128(second ? in2_rsrc : in_rsrc, o == NOPIX ? NOPIX : o + add, 0, 0);
                } else {
                    pre[k] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, goff[k] == NOPIX ? NOPIX : goff[k] + add, 0, 0);
                }
            }
        }
    };
    auto commit = [&]() {  // registers -> LDS
#pragma unroll
        for (int k = 0; k < MAXW; ++k) {
            const int idx = tid + k * NT;
            if (idx < nwchunk) *reinterpret_cast<f32x4 *>(wlds + idx * 16) = prew[k];
        }
        if constexpr (!IN_U8) {
            int tidc = tid;
            // (KS 3 x 64 couts, the forms at the register limit: opaque, i.e. the LDS addresses are recomputed per commit -- a few VALU ops per
            // 250 MFMAs -- instead of living in MAXLD registers across the k loops; the 1x1 forms have the room and 4.5x fewer MFMAs per commit)
            if constexpr (KS == 3 && WC == 4) asm volatile("" : "+v"(tidc));
#pragma unroll
            for (int k = 0; k < MAXLD; ++k) {
                const int idx = tidc + k * NT;
                int pix, c;
                if constexpr (KS == 1 && !DW) { pix = idx / cpk; c = idx - pix * cpk; }
                else { pix = idx >> P.sh; c = idx & (cpk - 1); }
                if (idx < nchunk) *reinterpret_cast<u32x4 *>(smem + ((pix * PST + c * 16) & ~15)) = pre[k];  // (& ~15: a no-op that keeps the 16-byte alignment visible -> ds_write_b128)
            }
        }
    };

    Tile cur, nxt;
    plan(t, cur);
    fetch(0);
    if constexpr (IN_U8) {  // the uint8 network input (one stage): every load is issued before the first LDS store
        float4 pv[MAXLD];
        const uint8_t *src = (const uint8_t *)P.in + (int64_t)cur.b * P.in_bs;
        const int iy0 = cur.oy0 * S - PAD, ix0 = cur.ox0 * S - PAD;
#pragma unroll
        for (int k = 0; k < MAXLD; ++k) {
            const int pix = tid + k * NT;
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
            pv[k] = v;
        }
#pragma unroll
        for (int k = 0; k < MAXLD; ++k) {
            const int pix = tid + k * NT;
            if (pix < in_px) *reinterpret_cast<float4 *>(smem + pix * PST) = pv[k];
        }
    }
    commit();
    __syncthreads();
    const char *const wfrag = wlds + wc * NC * P.wcb + lane * 16;  // (NC = 2: the second fragment's block follows at + wcb)
    const char *const wrem = wlds + wc * NC * P.wcb + P.kst * 1024 + lane * 4;  // the remainder chunks' weights: [chunk][lane] floats
    const int cbase = F * 16 + g * 4 * NC;  // epilogue: lane owns couts [cbase, cbase + 4 NC) of its pixels (NC = 2: fragment nc holds [cbase + 4 nc, + 4))
    // (the bias is loaded ONCE, outside the tile loop: a load inside it whose uses sit behind the per-pixel guards stays "pending" for the
    // compiler on the skipping paths, and the first LDS read of the next k loop that reuses its register then waits vmcnt(0) -- i.e. for
    // the stage prefetch issued just before)
    float4 bvn[NC];  // bias is padded: always readable
#pragma unroll
    for (int nc = 0; nc < NC; ++nc) bvn[nc] = TAIL > 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4 *>(P.bias + cbase + 4 * nc);
    for (;;) {
    bool more = false;  // (XT) the next tile's first stage is in flight / in LDS
    for (int stage = 0; stage < P.nstage; ++stage) {
        const bool last = stage + 1 == P.nstage;
        if constexpr (DW) {  // depthwise 3x3 + bias + SiLU of this stage's channels: input tile (LDS) -> B-operand tile (LDS)
            const float *dwl = reinterpret_cast<const float *>(wlds + WC * P.wcb);  // [9 taps + bias][CK]
            // an item = TWO horizontally adjacent output pixels x one 4-channel chunk: per tap row 4 input vectors + 3 weight vectors feed 6
            // fma groups (12 + 9 LDS reads per pixel pair where one pixel per item took 18 + 18: the depthwise phase was LDS-bandwidth-bound,
            // about as long as the stage's MFMAs); the sums run in the same (ky, kx) order per output, so the results are unchanged
            const int pairs = (P.TW + 1) >> 1;
            const float inv_pairs = 1.0f / (float)pairs;
            for (int idx = tid; idx < P.TH * pairs * cpk; idx += NT) {
                const int c = idx & (cpk - 1), q = idx >> P.sh;
                const int ty = (int)(((float)q + 0.5f) * inv_pairs), tx = (q - ty * pairs) * 2;
                const bool two = tx + 1 < P.TW;  // (odd widths: the last item of a row holds one pixel)
                const char *src = smem + (ty * TWin + tx) * PST + c * 16;
                float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                for (int ky = 0; ky < 3; ++ky) {  // (a row of taps at a time: the fully unrolled form held 18 operand vectors and spilled)
                    const char *row = src + ky * TWin * PST;
                    const float4 v0 = *reinterpret_cast<const float4 *>(row), v1 = *reinterpret_cast<const float4 *>(row + PST),
                                 v2 = *reinterpret_cast<const float4 *>(row + 2 * PST), v3 = *reinterpret_cast<const float4 *>(row + (two ? 3 : 2) * PST);
                    const float4 vv[4] = {v0, v1, v2, v3};
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {  // (per output the taps still run kx = 0, 1, 2 inside ky = 0, 1, 2)
                        const float4 wv = *reinterpret_cast<const float4 *>(dwl + (ky * 3 + kx) * P.CK + c * 4);
                        a0[0] = fmaf(vv[kx].x, wv.x, a0[0]); a0[1] = fmaf(vv[kx].y, wv.y, a0[1]); a0[2] = fmaf(vv[kx].z, wv.z, a0[2]); a0[3] = fmaf(vv[kx].w, wv.w, a0[3]);
                        a1[0] = fmaf(vv[kx + 1].x, wv.x, a1[0]); a1[1] = fmaf(vv[kx + 1].y, wv.y, a1[1]); a1[2] = fmaf(vv[kx + 1].z, wv.z, a1[2]); a1[3] = fmaf(vv[kx + 1].w, wv.w, a1[3]);
                    }
                }
                const float4 bvd = *reinterpret_cast<const float4 *>(dwl + 9 * P.CK + c * 4);
                float4 o0 = make_float4(a0[0] + bvd.x, a0[1] + bvd.y, a0[2] + bvd.z, a0[3] + bvd.w), o1 = make_float4(a1[0] + bvd.x, a1[1] + bvd.y, a1[2] + bvd.z, a1[3] + bvd.w);
                if (P.dw_act) {
                    o0.x = silu32(o0.x); o0.y = silu32(o0.y); o0.z = silu32(o0.z); o0.w = silu32(o0.w);
                    o1.x = silu32(o1.x); o1.y = silu32(o1.y); o1.z = silu32(o1.z); o1.w = silu32(o1.w);
                }
                char *dst = smem + dwb_off + (ty * P.TW + tx) * PST + c * 16;
                *reinterpret_cast<float4 *>(dst) = o0;
                if (two) *reinterpret_cast<float4 *>(dst + PST) = o1;
            }
            __syncthreads();
        }
        // the next stage (XT: of the next tile) goes in flight under this stage's MFMAs (DW: behind the depthwise phase, whose registers
        // are then dead)
        // (ONE fetch site: two of them meet in a phi of the fetch registers, whose copies wait for the loads right behind their issue)
        int fstage = stage + 1;
        if constexpr (XT) {
            if (last && P.tstep > 0 && t + P.tstep < P.ntiles) { plan(t + P.tstep, nxt); fstage = 0; more = true; }
        }
        if (!last || more) fetch(fstage);
        for (int ks = 0; ks < P.kst; ++ks) {
            f32x4 w[NC];
#pragma unroll
            for (int nc = 0; nc < NC; ++nc) w[nc] = *reinterpret_cast<const f32x4 *>(wfrag + nc * P.wcb + ks * 1024);
            int q = ks * 4 + g;
            if constexpr (KS == 1) q = q < nq ? q : nq - 1;  // (1x1: the last piece may be padded -- any valid address, its weights are zero)
            int off;
            if constexpr (KS == 3) {
                const int tap = q >> P.sh, c0 = q & (cpk - 1);
                const int dy = (tap * 11) >> 5, dx = tap - dy * 3;
                off = (dy * TWin + dx) * PST + c0 * 16;
            } else {
                off = q * 16;
            }
            // the fragments go through in two halves: only H operand vectors are live at a time
            constexpr int H = MFM > 4 ? (MFM + 1) / 2 : MFM;
#pragma unroll
            for (int m0 = 0; m0 < MFM; m0 += H) {
                f32x4 a[H];
#pragma unroll
                for (int i = 0; i < H; ++i)
                    if (m0 + i < MFM) a[i] = *reinterpret_cast<const f32x4 *>(smem + pixbase[m0 + i] + off);
#ifdef OBB_DIAG
                if (P.dbg & 4) {  // timing only: every activation read issued twice (how sensitive is the k loop to its LDS read volume?)
#pragma unroll
                    for (int i = 0; i < H; ++i)
                        if (m0 + i < MFM) { f32x4 d2 = *reinterpret_cast<const volatile f32x4 *>(smem + pixbase[m0 + i] + off); asm volatile("" ::"v"(d2)); }
                }
                if (P.dbg & 8) {  // timing only: no MFMAs (LDS reads + loop only)
#pragma unroll
                    for (int i = 0; i < H; ++i)
                        if (m0 + i < MFM) asm volatile("" ::"v"(a[i]));
#pragma unroll
                    for (int nc = 0; nc < NC; ++nc) asm volatile("" ::"v"(w[nc]));
                    continue;
                }
#endif
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < H; ++i)
#pragma unroll
                        for (int nc = 0; nc < NC; ++nc) {
                            if (m0 + i < MFM - 1) acc[nc][m0 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nc][s], a[i][s], acc[nc][m0 + i], 0, 0, 0);
                            else if (m0 + i == MFM - 1) { if (cur.lastv) acc[nc][MFM - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nc][s], a[i][s], acc[nc][MFM - 1], 0, 0, 0); }
                        }
            }
        }
        // the stage's last nq % 4 chunks, ONE MFMA each: the four k slots of the instruction are the chunk's four channels (lane group g
        // reads channel g: ds_read_b32) -- a 3x3 stage of 8 channels is 18 chunks = 4 pieces + 2 of these = 18 MFMAs per fragment, where
        // padding the stage to 5 pieces cost 20
        if constexpr (KS == 3)
        for (int r = 0; r < P.krem; ++r) {
            float w1[NC];
#pragma unroll
            for (int nc = 0; nc < NC; ++nc) w1[nc] = *reinterpret_cast<const float *>(wrem + nc * P.wcb + r * 256);
            const int q = P.kst * 4 + r;
            const int tap = q >> P.sh, c0 = q & (cpk - 1);
            const int dy = (tap * 11) >> 5, dx = tap - dy * 3;
            const int off = (dy * TWin + dx) * PST + c0 * 16 + g * 4;
            float a1[MFM];
#pragma unroll
            for (int mf = 0; mf < MFM; ++mf) a1[mf] = *reinterpret_cast<const float *>(smem + pixbase[mf] + off);
#pragma unroll
            for (int mf = 0; mf < MFM; ++mf)
#pragma unroll
                for (int nc = 0; nc < NC; ++nc) {
                    if (mf < MFM - 1) acc[nc][mf] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[nc], a1[mf], acc[nc][mf], 0, 0, 0);
                    else if (cur.lastv) acc[nc][MFM - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[nc], a1[mf], acc[nc][MFM - 1], 0, 0, 0);
                }
        }
        if (!last || more) {
            __syncthreads();  // every wave is done reading this stage
            commit();
            if (!last) __syncthreads();
        }
    }

    // ---- epilogue of tile `cur`
    if constexpr (TAIL > 0) {
        // activated tile -> LDS [pixel][cout] (+16 B per row), then the second GEMM over it
        constexpr int WC2 = TAIL, WP2 = NW / WC2, NFR = WP * MFM, MFM2 = (NFR + WP2 - 1) / WP2;
        const int TPST = P.cout * 4 + 16;
        __syncthreads();  // the k loop's LDS reads are done
        {
            const float4 bv = *reinterpret_cast<const float4 *>(P.bias + cbase);
#pragma unroll
            for (int mf = 0; mf < MFM; ++mf) {
                const int p = (wp + WP * mf) * 16 + pl;
                float v[4] = {acc[0][mf][0] + bv.x, acc[0][mf][1] + bv.y, acc[0][mf][2] + bv.z, acc[0][mf][3] + bv.w};
                if (P.act) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = silu32(v[j]);
                }
                if (cbase < P.cout) *reinterpret_cast<float4 *>(smem + p * TPST + cbase * 4) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        __syncthreads();
        const int wc2 = wave % WC2, wp2 = wave / WC2;
        const float *w2base = P.w2 + (size_t)wc2 * P.kst2 * 256 + lane * 4;
        f32x4 acc2[MFM2];
#pragma unroll
        for (int i = 0; i < MFM2; ++i) acc2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 wn = *reinterpret_cast<const f32x4 *>(w2base);
        for (int u = 0; u < P.kst2; ++u) {
            const f32x4 w = wn;
            wn = *reinterpret_cast<const f32x4 *>(w2base + (size_t)min(u + 1, P.kst2 - 1) * 256);
            f32x4 a[MFM2];
#pragma unroll
            for (int i = 0; i < MFM2; ++i) {
                const int f2 = wp2 + WP2 * i;
                a[i] = *reinterpret_cast<const f32x4 *>(smem + ((f2 < NFR ? f2 : 0) * 16 + pl) * TPST + (u * 16 + g * 4) * 4);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < MFM2; ++i)
                    if (wp2 + WP2 * i < cur.nfrag1) acc2[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s], a[i][s], acc2[i], 0, 0, 0);
        }
        const int c2 = wc2 * 16 + g * 4;
        const bool lane_on = c2 < P.cout2;
        if (!lane_on && !(TAIL == 1 && P.cmax)) return;  // (with the class-logit maximum on, the four lanes of a pixel stay together for the shuffles)
        const float4 bv2 = *reinterpret_cast<const float4 *>(P.b2 + (lane_on ? c2 : 0));
        const bool full2 = c2 + 4 <= P.cout2;
#pragma unroll
        for (int i = 0; i < MFM2; ++i) {
            const int p = (wp2 + WP2 * i) * 16 + pl;
            if (p >= npix) continue;
            const int il = P.NI > 1 ? (int)(((float)p + 0.5f) * inv_tpi) : 0;
            const int pq = p - il * tpi;
            const int ty = (int)(((float)pq + 0.5f) * P.inv_tw), tx = pq - ty * P.TW;
            const int oy = cur.oy0 + ty, ox = cur.ox0 + tx;
            if (oy >= P.Hout || ox >= P.Wout || il >= cur.nimg) continue;
            const int64_t opix = (int64_t)oy * P.Wout + ox;
            float v[4] = {acc2[i][0] + bv2.x, acc2[i][1] + bv2.y, acc2[i][2] + bv2.z, acc2[i][3] + bv2.w};
            if (P.act2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = silu32(v[j]);
            }
            int64_t ob = cur.b + il, opx = opix;
            if (P.out2_hw > 0) { ob = opx / P.out2_hw; opx -= ob * P.out2_hw; }
            if constexpr (TAIL == 1) {
                if (P.cmax) {  // max over the tail's cout2 outputs of this pixel: own four, then the pixel's other lanes (lane ^ 16, lane ^ 32)
                    float m = -INFINITY;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (c2 + j < P.cout2) m = fmaxf(m, v[j]);
                    m = fmaxf(m, __shfl_xor(m, 16));
                    m = fmaxf(m, __shfl_xor(m, 32));
                    if (g == 0) P.cmax[ob * P.cmax_bs + opx] = m;
                    if (!lane_on) continue;
                }
            }
            float *op = P.out2 + ob * P.out2_bs + opx * P.out2_cs + P.out2_co + c2;
            if (full2 && ((P.out2_cs | P.out2_co) & 3) == 0) {
                *reinterpret_cast<float4 *>(op) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c2 + j < P.cout2) op[j] = v[j];
            }
        }
        return;
    }
#ifdef OBB_DIAG
    if (P.dbg & 2) { if (acc[0][0][0] != 123.456f) return; }  // timing only: no epilogue (the test keeps the accumulators alive)
#endif
    int plv = pl;
    if constexpr (XT) asm volatile("" : "+v"(plv));  // (opaque, as in `plan`: the pixel coordinates are recomputed per tile, not kept across the k loops)
    const bool full = cbase + 4 * NC <= P.cout;
#pragma unroll
    for (int mf = 0; mf < MFM; ++mf) {
        const int p = (wp + WP * mf) * 16 + plv;
        if (p >= npix || cbase >= P.cout) continue;
        const int il = P.NI > 1 ? (int)(((float)p + 0.5f) * inv_tpi) : 0;
        const int pq = p - il * tpi;
        const int ty = (int)(((float)pq + 0.5f) * P.inv_tw), tx = pq - ty * P.TW;
        const int oy = cur.oy0 + ty, ox = cur.ox0 + tx;
        if (oy >= P.Hout || ox >= P.Wout || il >= cur.nimg) continue;
        const int64_t opix = (int64_t)oy * P.Wout + ox;
        int64_t ob = cur.b + il, opx = opix;
        if (P.out_hw > 0) { ob = opx / P.out_hw; opx -= ob * P.out_hw; }
        const int ca = P.out_co + cbase;
        float *const op0 = P.out + ob * P.out_bs + opx * P.out_cs + ((BLK && P.out_blk) ? (int64_t)(ca >> 3) * P.out_ps + (ca & 7) : (int64_t)ca);
        const float *const rp0 = P.res ? P.res + (int64_t)(cur.b + il) * P.res_bs + opix * P.res_cs + P.res_co + cbase : nullptr;
#pragma unroll
        for (int nc = 0; nc < NC; ++nc) {  // (NC = 2: the two stores of a lane are 32 contiguous bytes)
            const float4 bv = bvn[nc];
            float v[4] = {acc[nc][mf][0] + bv.x, acc[nc][mf][1] + bv.y, acc[nc][mf][2] + bv.z, acc[nc][mf][3] + bv.w};
            if (P.act) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = silu32(v[j]);
            }
            if (rp0) {
                const float *rp = rp0 + 4 * nc;
                if (full) {
                    const float4 rv = *reinterpret_cast<const float4 *>(rp);
                    v[0] = rv.x + v[0]; v[1] = rv.y + v[1]; v[2] = rv.z + v[2]; v[3] = rv.w + v[3];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (cbase + 4 * nc + j < P.cout) v[j] = rp[j] + v[j];
                }
            }
            float *op = op0 + 4 * nc;
            if (full && ((P.out_cs | P.out_co) & 3) == 0) {
                *reinterpret_cast<float4 *>(op) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (cbase + 4 * nc + j < P.cout) op[j] = v[j];
            }
        }
    }
    if constexpr (!XT) return;
    if (!more) return;
    __syncthreads();  // the next tile's first stage is in LDS
    t += P.tstep;
    cur = nxt;
#pragma unroll
    for (int nc = 0; nc < NC; ++nc)
#pragma unroll
        for (int mf = 0; mf < MFM; ++mf) acc[nc][mf] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// ------------------------------------------------------------------------------------------------ host side
static constexpr int kNW = 8;  // waves per workgroup
static int ilog2_(int v) { int s = 0; while ((1 << s) < v) ++s; return s; }
// a stage's k = taps x CK channels in 4-channel chunks: kfull pieces of 4 chunks (four MFMAs behind one ds_read_b128 per operand) + krem
// single chunks (one MFMA each); wfloats = weights of one cout fragment and stage
// (1x1: whole pieces only, the last one zero-padded -- no 1x1 layer of the network has fewer than 16 input channels, and the remainder
// loop's registers are what the 1x1 forms at 7 fragments per wave do not have)
static int c32_kfull(int ks, int CK) { return ks == 3 ? (9 * (CK / 4)) / 4 : (CK / 4 + 3) / 4; }
static int c32_krem(int ks, int CK) { return ks == 3 ? (9 * (CK / 4)) % 4 : 0; }
static int c32_wfloats(int ks, int CK) { return c32_kfull(ks, CK) * 256 + c32_krem(ks, CK) * 64; }
static int c32_mfm_max(int WC) { return WC == 4 ? 7 : 4; }  // 224 / 256 / 512 pixels per tile
static int c32_mfm_min(int WC) { return WC == 4 ? 4 : (WC == 2 ? 2 : 1); }  // smallest instantiated fragment count (smaller tiles run it partly empty)
static int c32_maxld(int ks, bool in_u8) { return in_u8 ? 3 : (ks == 1 ? 7 : 5); }

static Conv32Tiling plan_conv32_nc(int ks, int stride, int cin, int cout, int Hout, int Wout, bool in_u8, bool vcat, int NC, int64_t *cost_out);

Conv32Tiling plan_conv32(int ks, int stride, int cin, int cout, int Hout, int Wout, bool in_u8, bool vcat, bool nc2) {
    int64_t c1 = 0, c2 = 0;
    const Conv32Tiling t1 = plan_conv32_nc(ks, stride, cin, cout, Hout, Wout, in_u8, vcat, 1, &c1);
    if (!nc2 || in_u8 || cout < 64 || cout % 32 || t1.NI > 1 || cin < 64) return t1;  // (48-channel concats: HBM-bound, measured 4 % worse)
    const Conv32Tiling t2 = plan_conv32_nc(ks, stride, cin, cout, Hout, Wout, in_u8, vcat, 2, &c2);
    // (costs in MFMA groups of the busiest wave per tile; the two-fragment form may cost a few percent more of them: it reads a third less LDS)
    // (measured: forcing the form onto the 52 x 52 layers -- 13 fragments in 16 slots -- costs cv2.0.0 903 -> 984 us, onto the 13 x 13 ones 4-8 %)
    return (ks == 1 || c2 * 100 <= c1 * 97) ? t2 : t1;
}

static Conv32Tiling plan_conv32_nc(int ks, int stride, int cin, int cout, int Hout, int Wout, bool in_u8, bool vcat, int NC, int64_t *cost_out) {
    Conv32Tiling t;
    t.NC = NC;
    t.WC = NC == 2 ? 2 : (cout >= 64 ? 4 : (cout >= 32 ? 2 : 1));
    if (in_u8) t.CK = 4;
    else {
        // (1x1: 64-channel stages = half the barriers and LDS commits of 32: 18.40 -> 18.27 ms per 512 tiles, the 13 x 13 layers 4-6 %)
        const int cap = ks == 1 ? (vcat ? 32 : 64) : (stride == 2 ? 8 : 16);  // (1x1 with 64-channel stages: measured, no gain -- 18.05 -> 18.22 ms per 512 tiles)
        int ck = 4;
        while (ck * 2 <= cap && cin % (ck * 2) == 0) ck *= 2;
        if (ks == 1 && !vcat && ck == 16 && cin % 48 == 0) ck = 48;  // 48- / 96-channel concats: one stage of three 16-channel groups instead of three stages of one k step
        t.CK = ck;
    }
    // (NC = 2, 1x1: 3 fragments per wave = 192-pixel tiles with 64-channel stages and no spill; 4 = 256 pixels forces 32-channel stages and
    // spills 44 bytes per lane -- same-box A/B over the forward: 17.84 ms with NC = 1, 17.59 with 4, 17.36 with 3)
    const int WP = kNW / t.WC, MFMX = NC == 2 ? (ks == 1 ? 3 : 4) : c32_mfm_max(t.WC), MFMN = NC == 2 ? 3 : c32_mfm_min(t.WC);
    const int maxpix = 16 * WP * MFMX;
    const int chunk_cap = (vcat ? 4 : c32_maxld(ks, in_u8)) * kNW * 64;  // 16-B chunks one stage may hold
    if (ks == 1) {  // 1-D: the caller flattens batch x pixels
        while (t.CK > 4 && (maxpix * (t.CK / 4) > chunk_cap || (int64_t)maxpix * (t.CK * 4 + 16) + t.WC * NC * c32_wfloats(1, t.CK) * 4 > 78 * 1024)) t.CK /= 2;
        t.TH = 1; t.TW = maxpix; t.MFM = MFMX; t.NI = 1;
        if (cost_out) *cost_out = 0;
        return t;
    }
    const int PST = t.CK * 4 + 16;
    t.TW = std::min(Wout, maxpix);
    const int tiles_x = (Wout + t.TW - 1) / t.TW;
    int best_th = 1;
    int64_t best_cost = -1;
    for (int th = 1; th <= std::min(Hout, maxpix / t.TW); ++th) {
        const int64_t in_px = (int64_t)((th - 1) * stride + ks) * ((t.TW - 1) * stride + ks);
        if (in_px * PST + t.WC * NC * c32_wfloats(ks, t.CK) * 4 > 78 * 1024 || in_px * (in_u8 ? 1 : t.CK / 4) > chunk_cap) break;
        // time ~ tiles x (fragments of the busiest wave + the fixed cost of a stage: barriers + LDS commit, about one fragment's MFMAs)
        const int mfm = std::max(((th * t.TW + 15) / 16 + WP - 1) / WP, MFMN);
        const int64_t cost = (int64_t)((Hout + th - 1) / th) * tiles_x * (mfm * NC + 1);
        if (best_cost < 0 || cost <= best_cost) { best_cost = cost; best_th = th; }
    }
    t.TH = best_th;
    t.NI = 1;
    if (cost_out) *cost_out = best_cost;
    if (NC == 1 && !in_u8 && t.TH == Hout && t.TW == Wout && Hout * Wout * 2 <= maxpix) {
        // a small map (the 128-px scale's 8 x 8 / 4 x 4 levels): one tile = NI whole images, as many as the fragment budget, the LDS
        // tile and the staging plan take -- a tile of ONE such map would leave most of the workgroup's fragments empty
        const int64_t in1 = (int64_t)((Hout - 1) * stride + ks) * ((Wout - 1) * stride + ks);
        int ni = maxpix / (Hout * Wout);
        while (ni > 1 && (ni * in1 * PST + t.WC * c32_wfloats(ks, t.CK) * 4 > 78 * 1024 || ni * in1 * (t.CK / 4) > chunk_cap)) --ni;
        t.NI = ni;
    }
    t.MFM = std::max(((t.NI * t.TH * t.TW + 15) / 16 + WP - 1) / WP, MFMN);
    return t;
}

bool conv32_tail_supported(const Conv32Tiling &t, int cout1, int cout2) {
    if (cout1 != 16 * t.WC || (cout1 != 16 && cout1 != 32 && cout1 != 64)) return false;  // the whole producer in one workgroup
    if (cout2 < 1 || cout2 > 64) return false;
    return (size_t)16 * (kNW / t.WC) * t.MFM * (cout1 * 4 + 16) <= 80 * 1024;
}

Conv32Tiling plan_dwpw32(int cin, int cout, int H, int W) {
    Conv32Tiling t{0, 0, 16, 4, 0, 1, 1};
    if (cin % 16 || cout % 64 || W > 224) return t;
    const int WP = kNW / t.WC, maxpix = 16 * WP * c32_mfm_max(t.WC), PST = t.CK * 4 + 16;
    t.TW = W;
    int best = 0;
    int64_t best_cost = -1;
    for (int th = 1; th <= std::min(H, maxpix / W); ++th) {
        const int64_t in_px = (int64_t)(th + 2) * (W + 2);
        const int64_t lds = ((in_px * PST + 1023) & ~1023ll) + (((int64_t)th * W * PST + 1023) & ~1023ll) + t.WC * 1024 + 10 * t.CK * 4;
        if (lds > 78 * 1024 || in_px * (t.CK / 4) > 3 * kNW * 64) break;
        const int mfm = std::max(((th * W + 15) / 16 + WP - 1) / WP, c32_mfm_min(t.WC));
        const int64_t cost = (int64_t)((H + th - 1) / th) * (mfm + 1);
        if (best_cost < 0 || cost <= best_cost) { best_cost = cost; best = th; }
    }
    // (a tile is whole rows of ONE image: the 8 x 8 / 4 x 4 maps of the 128-px scale would fill 64 / 16 of a workgroup's >= 128 pixel slots
    // -- measured 8192 tiles: 452 us at 10.8 TFLOP/s for the 4 x 4 pair -- and keep their separate depthwise + flattened 1x1 launches)
    if (!best || best * W < 112) return t;
    t.TH = best;
    t.MFM = std::max(((t.TH * W + 15) / 16 + WP - 1) / WP, c32_mfm_min(t.WC));
    return t;
}

std::vector<float> pack_dwpw32_weights(const float *pw, int cout, int cin, const float *dw_c9, const float *dw_bias, const Conv32Tiling &t) {
    const int CK = t.CK, nstage = cin / CK, ncb = cout / (16 * t.WC);
    const std::vector<float> base = pack_conv32_weights(pw, cout, cin, 1, t, nullptr, false);  // [cb][stage][wc][kst][lane][4]
    const size_t blk = (size_t)t.WC * c32_wfloats(1, CK), dwn = (size_t)10 * CK;
    std::vector<float> out((size_t)ncb * nstage * (blk + dwn), 0.f);
    for (int cb = 0; cb < ncb; ++cb)
        for (int st = 0; st < nstage; ++st) {
            float *dst = out.data() + ((size_t)cb * nstage + st) * (blk + dwn);
            std::copy(base.begin() + ((size_t)cb * nstage + st) * blk, base.begin() + ((size_t)cb * nstage + st + 1) * blk, dst);
            for (int c = 0; c < CK; ++c) {
                for (int tp = 0; tp < 9; ++tp) dst[blk + (size_t)tp * CK + c] = dw_c9[(size_t)(st * CK + c) * 9 + tp];
                dst[blk + (size_t)9 * CK + c] = dw_bias[st * CK + c];
            }
        }
    return out;
}

std::vector<float> pack_conv32_weights(const float *w, int cout, int cin, int ks, const Conv32Tiling &t, const int *perm, bool in_u8) {
    const int CK = t.CK, cpk = CK / 4;
    const int cin_eff = in_u8 ? 4 : cin;
    const int nstage = (cin_eff + CK - 1) / CK;
    const int kst = c32_kfull(ks, CK), krem = c32_krem(ks, CK);
    const int NC = std::max(1, t.NC), nfb = t.WC * NC;  // fragments per cout block
    const int nf = (cout + 15) / 16;
    const int nfp = (nf + nfb - 1) / nfb * nfb;  // whole cout blocks
    const int taps = ks * ks;
    std::vector<float> out((size_t)nfp * nstage * c32_wfloats(ks, CK), 0.f);
    size_t o = 0;
    auto wat = [&](int co, int st, int q, int s) -> float {  // weight of cout co for channel s of the stage's chunk q
        const int tap = q / cpk, c = st * CK + (q % cpk) * 4 + s;
        if (co >= cout || c >= cin || q >= taps * cpk) return 0.f;
        return w[((size_t)(perm ? perm[co] : co) * cin + c) * taps + tap];
    };
    // cout of accumulator row r (= lane & 15) of the block's fragment f: one fragment per wave -> 16 f + r; two (NC = 2) -> the wave's 32 couts
    // interleaved so that a lane's 4 + 4 rows (g * 4 + j of both fragments) are 8 consecutive couts: 32 (f / 2) + (r >> 2) * 8 + (f & 1) * 4 + (r & 3)
    auto cout_of = [&](int cb, int f, int r) { return cb * nfb * 16 + (NC == 2 ? 32 * (f >> 1) + (r >> 2) * 8 + (f & 1) * 4 + (r & 3) : 16 * f + r); };
    // [cout block][stage][fragment of the block]{[piece][lane][4], [remainder chunk][lane]}: a workgroup's stage is one contiguous run
    for (int cb = 0; cb < nfp / nfb; ++cb)
        for (int st = 0; st < nstage; ++st)
            for (int f = 0; f < nfb; ++f) {
                for (int k = 0; k < kst; ++k)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int s = 0; s < 4; ++s) out[o++] = wat(cout_of(cb, f, lane & 15), st, k * 4 + (lane >> 4), s);
                for (int r = 0; r < krem; ++r)
                    for (int lane = 0; lane < 64; ++lane) out[o++] = wat(cout_of(cb, f, lane & 15), st, kst * 4 + r, lane >> 4);
            }
    return out;
}

size_t conv32_lds_bytes(const Conv32Launch &L) {
    if (L.dw) {
        const size_t PSTd = (size_t)L.CK * 4 + 16;
        size_t l = (((size_t)(L.TH + 2) * (L.TW + 2) * PSTd + 1023) & ~(size_t)1023) + (((size_t)L.TH * L.TW * PSTd + 1023) & ~(size_t)1023) +
                   (size_t)L.WC * c32_wfloats(1, L.CK) * 4 + (size_t)10 * L.CK * 4;
        if (L.tail_cout > 0) l = std::max(l, (size_t)16 * (kNW / L.WC) * L.MFM * (L.cout * 4 + 16));
        return l;
    }
    const int THin = (L.TH - 1) * L.stride + L.ks, TWin = (L.TW - 1) * L.stride + L.ks;
    size_t lds = (((size_t)std::max(1, L.NI) * THin * TWin * (L.CK * 4 + 16) + 1023) & ~(size_t)1023) + (size_t)L.WC * std::max(1, L.NC) * c32_wfloats(L.ks, L.CK) * 4;  // activation tile + stage weights
    if (L.tail_cout > 0) lds = std::max(lds, (size_t)16 * (kNW / L.WC) * L.MFM * (L.cout * 4 + 16));
    return lds;
}

template <int KS, int MFM, int WC, bool IN_U8, bool VCAT, int TAIL, bool DW = false, int NC = 1, bool BLK = false>
static hipError_t launch32_k(const C32Params &P0, dim3 grid, size_t lds, hipStream_t st) {
    const void *fn = (const void *)k_conv_f32<KS, MFM, WC, kNW, IN_U8, VCAT, TAIL, DW, NC, BLK>;
    static bool attr_set = false;  // (per instantiation) up to 80 KiB of dynamic LDS: two workgroups per CU
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    C32Params P = P0;
    P.tstep = 0;
    // (measured per layer, 512 tiles: layers of 1-3 stages per tile gain 3-18 % -- model.2.cv2 867 -> 707 us -- longer tiles hide their
    // first fetch and their stores behind the co-resident workgroup anyway and only lose to the coarser tile split)
#ifndef OBB_XT_MAX_STAGES
#define OBB_XT_MAX_STAGES 3
#endif
    if (TAIL == 0 && !DW && !IN_U8 && P0.tstep != 0 && P0.nstage <= OBB_XT_MAX_STAGES) {  // (launch_conv32 passes Conv32Launch::xtile in tstep)
        // cross-tile pipeline: a grid of resident workgroups, each walking ~equally many tiles (see XT in the kernel)
        static std::map<size_t, int> occ;  // resident workgroups per CU of this instantiation, by dynamic LDS size
        static int ncu = 0;
        static std::mutex mu;  // (contexts of different host threads launch through the same cache)
        std::lock_guard<std::mutex> lock(mu);
        auto it = occ.find(lds);
        if (it == occ.end()) {
            int n = 0, dev = 0;
            hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, kNW * 64, lds);
            if (e != hipSuccess) return e;
            if (!ncu) {
                if ((e = hipGetDevice(&dev)) != hipSuccess || (e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
            }
            it = occ.emplace(lds, std::max(1, n)).first;
        }
        const int64_t tiles8 = ((int64_t)P.ntiles + 7) / 8;
        const int64_t slots_max = std::max<int64_t>(1, (int64_t)it->second * ncu / (8 * P.ncb));
        const int64_t rounds = (tiles8 + slots_max - 1) / slots_max;
        if (rounds > 1) {
            const int64_t slots = (tiles8 + rounds - 1) / rounds;
            P.tstep = (int)(slots * 8);
            grid = dim3((unsigned)(slots * 8 * P.ncb));
        }
    }
    hipLaunchKernelGGL((k_conv_f32<KS, MFM, WC, kNW, IN_U8, VCAT, TAIL, DW, NC, BLK>), grid, dim3(kNW * 64), lds, st, P);
    return hipGetLastError();
}

// the instantiated (KS, MFM, WC) x feature combinations; anything else is refused (the plan builder only asks for these)
template <int KS, int MFM, int WC, bool BLK>
static hipError_t launch32_f(const Conv32Launch &L, const C32Params &P, int tail_wc2, dim3 grid, size_t lds, hipStream_t st) {
    if (L.in_u8) {
        if constexpr (KS == 3 && WC == 1 && !BLK) { if (!tail_wc2 && !L.up_c) return launch32_k<KS, MFM, WC, true, false, 0>(P, grid, lds, st); }
        return hipErrorInvalidValue;
    }
    if (L.up_c > 0) {
        if constexpr (KS == 1 && WC == 4) { if (!tail_wc2) return launch32_k<KS, MFM, WC, false, true, 0, false, 1, BLK>(P, grid, lds, st); }
        return hipErrorInvalidValue;
    }
    if (L.dw) {
        if constexpr (KS == 1 && WC == 4) {
            if (tail_wc2 == 0) return launch32_k<KS, MFM, WC, false, false, 0, true, 1, BLK>(P, grid, lds, st);
            if (tail_wc2 == 1) return launch32_k<KS, MFM, WC, false, false, 1, true, 1, BLK>(P, grid, lds, st);
        }
        return hipErrorInvalidValue;
    }
    switch (tail_wc2) {
        case 0: return launch32_k<KS, MFM, WC, false, false, 0, false, 1, BLK>(P, grid, lds, st);
        case 1: if constexpr (WC == 1 || WC == 4) return launch32_k<KS, MFM, WC, false, false, 1, false, 1, BLK>(P, grid, lds, st); break;
        case 2: if constexpr (WC == 2) return launch32_k<KS, MFM, WC, false, false, 2, false, 1, BLK>(P, grid, lds, st); break;
        case 4: if constexpr (WC == 4) return launch32_k<KS, MFM, WC, false, false, 4, false, 1, BLK>(P, grid, lds, st); break;
    }
    return hipErrorInvalidValue;
}

template <int KS, bool BLK>
static hipError_t launch32_wc(const Conv32Launch &L, const C32Params &P, int tail_wc2, dim3 grid, size_t lds, hipStream_t st) {
    if (L.NC == 2) {  // two cout fragments per wave: 2 waves along cout, 3 or 4 pixel fragments per wave; plain and (1x1) VCAT forms
        if (L.WC != 2 || tail_wc2 || L.dw || L.in_u8) return hipErrorInvalidValue;
        if (L.up_c > 0) {
            if constexpr (KS == 1) {
                if (L.MFM == 4) return launch32_k<KS, 4, 2, false, true, 0, false, 2, BLK>(P, grid, lds, st);
                if (L.MFM == 3) return launch32_k<KS, 3, 2, false, true, 0, false, 2, BLK>(P, grid, lds, st);
            }
            return hipErrorInvalidValue;
        }
        if (L.MFM == 4) return launch32_k<KS, 4, 2, false, false, 0, false, 2, BLK>(P, grid, lds, st);
        if (L.MFM == 3) return launch32_k<KS, 3, 2, false, false, 0, false, 2, BLK>(P, grid, lds, st);
        return hipErrorInvalidValue;
    }
    switch (L.WC * 16 + L.MFM) {
        case 4 * 16 + 7: return launch32_f<KS, 7, 4, BLK>(L, P, tail_wc2, grid, lds, st);
        case 4 * 16 + 6: return launch32_f<KS, 6, 4, BLK>(L, P, tail_wc2, grid, lds, st);
        case 4 * 16 + 5: return launch32_f<KS, 5, 4, BLK>(L, P, tail_wc2, grid, lds, st);
        case 4 * 16 + 4: return launch32_f<KS, 4, 4, BLK>(L, P, tail_wc2, grid, lds, st);
        case 2 * 16 + 4: return launch32_f<KS, 4, 2, BLK>(L, P, tail_wc2, grid, lds, st);
        case 2 * 16 + 3: return launch32_f<KS, 3, 2, BLK>(L, P, tail_wc2, grid, lds, st);
        case 2 * 16 + 2: return launch32_f<KS, 2, 2, BLK>(L, P, tail_wc2, grid, lds, st);
        case 1 * 16 + 4: return launch32_f<KS, 4, 1, BLK>(L, P, tail_wc2, grid, lds, st);
        case 1 * 16 + 3: return launch32_f<KS, 3, 1, BLK>(L, P, tail_wc2, grid, lds, st);
        case 1 * 16 + 2: return launch32_f<KS, 2, 1, BLK>(L, P, tail_wc2, grid, lds, st);
        case 1 * 16 + 1: return launch32_f<KS, 1, 1, BLK>(L, P, tail_wc2, grid, lds, st);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_conv32(const Conv32Launch &L, hipStream_t st) {
    if (L.res.cpb || L.tail_out.cpb) return hipErrorInvalidValue;  // plain NHWC only
    {   // channel-blocked (by 8) tensors: see C32Params
        auto blk8 = [](const TensorRef &t) { return t.cpb == 2 && t.cs == 8 && t.co % 8 == 0 && t.ps > 0 && t.ps < (1ll << 28); };
        if (L.in.cpb && (!blk8(L.in) || L.in_u8 || L.up_c > 0 || (L.ks != 3 && !L.dw) || L.CK % 8 || L.cin % 8)) return hipErrorInvalidValue;
        if (L.in2.cpb && (!blk8(L.in2) || L.up_c <= 0 || L.CK % 8)) return hipErrorInvalidValue;
        // (1x1 layers are issued over the flattened batch: their blocked output needs the per-image split of out_hw)
        if (L.out.cpb && (!blk8(L.out) || L.tail_cout > 0 || L.cout % 8 || (L.ks == 1 && !L.dw && L.out_hw <= 0))) return hipErrorInvalidValue;
    }
    if ((L.ks != 1 && L.ks != 3) || (L.WC != 1 && L.WC != 2 && L.WC != 4) || (L.NC != 1 && L.NC != 2)) return hipErrorInvalidValue;
    if (L.NC == 2 && (L.cout % 32 || L.tail_cout > 0 || L.dw || L.in_u8 || L.NI > 1)) return hipErrorInvalidValue;
    C32Params P;
    memset(&P, 0, sizeof P);
    P.in = L.in.p; P.in_bs = L.in.bs; P.in_cs = L.in.cs; P.in_co = L.in.co;
    P.out = (float *)L.out.p; P.out_bs = L.out.bs; P.out_cs = L.out.cs; P.out_co = L.out.co;
    if (L.in.cpb) { P.in_blk = 1; P.in_ps = (int)L.in.ps; P.in = (const float *)L.in.p + (int64_t)(L.in.co >> 3) * L.in.ps; P.in_co = 0; }
    if (L.out.cpb) { P.out_blk = 1; P.out_ps = (int)L.out.ps; }
    P.in_sadd = L.in.cpb ? (unsigned)((int64_t)(L.CK / 8) * L.in.ps * 4) : (unsigned)(L.CK * 4);
    P.res = (const float *)L.res.p; P.res_bs = L.res.bs; P.res_cs = L.res.cs; P.res_co = L.res.co;
    P.wpk = L.wpk; P.bias = L.bias; P.lut = L.lut;
    P.Hin = L.Hin; P.Win = L.Win; P.Hout = L.Hout; P.Wout = L.Wout; P.cin = L.cin; P.cout = L.cout; P.stride = L.stride; P.act = L.act;
    P.flip_bgr = L.flip_bgr;
    P.TH = L.TH; P.TW = L.TW; P.CK = L.CK; P.sh = ilog2_(L.CK / 4);
    if ((4 << P.sh) != L.CK && !(L.ks == 1 && L.CK == 48 && !L.up_c && !L.in_u8)) return hipErrorInvalidValue;
    const int WP = kNW / L.WC;
    const int NI = std::max(1, L.NI);
    if (L.MFM < 1 || L.MFM > (L.NC == 2 ? 4 : c32_mfm_max(L.WC)) || NI * L.TH * L.TW > 16 * WP * L.MFM || L.TH < 1 || L.TW < 1) return hipErrorInvalidValue;
    if (NI > 1 && (L.ks != 3 || L.in_u8 || L.up_c || L.tiles_x != 1 || L.tiles_y != 1 || L.TH != L.Hout || L.TW != L.Wout || L.out_hw || L.tail_out_hw)) return hipErrorInvalidValue;
    P.NI = NI; P.B = L.B; P.dw_act = L.dw_act;
    if (L.dw && (L.ks != 1 || L.stride != 1 || L.in_u8 || L.up_c || NI != 1 || L.WC != 4 || L.CK != 16 || L.out_hw || L.tail_out_hw || L.Hin != L.Hout || L.Win != L.Wout || L.TW != L.Wout ||
                 L.tiles_x != 1)) return hipErrorInvalidValue;
    const int cin_eff = L.in_u8 ? 4 : L.cin;
    if (!L.in_u8 && (L.cin % L.CK || (L.in.cs & 3) || (L.in.co & 3))) return hipErrorInvalidValue;
    if (L.in_u8 && (L.CK != 4 || (L.cin != 3 && L.cin != 4) || !L.lut)) return hipErrorInvalidValue;
    if (L.res.p && ((L.res.cs | L.res.co) & 3)) return hipErrorInvalidValue;
    P.nstage = (cin_eff + L.CK - 1) / L.CK;
    P.kst = c32_kfull(L.ks, L.CK); P.krem = c32_krem(L.ks, L.CK); P.wcb = c32_wfloats(L.ks, L.CK) * 4;
    P.tiles_x = L.tiles_x; P.tiles_y = L.tiles_y; P.out_hw = L.out_hw;
    const int64_t ntiles = NI > 1 ? ((int64_t)L.B + NI - 1) / NI : (int64_t)L.B * L.tiles_y * L.tiles_x;
    P.ncb = (L.cout + 16 * L.WC * L.NC - 1) / (16 * L.WC * L.NC);
    if (ntiles < 1 || (ntiles + 7) / 8 * 8 * P.ncb >= (1ll << 31)) return hipErrorInvalidValue;
    P.ntiles = (int)ntiles;
    const int sks = L.dw ? 3 : L.ks;  // staged halo
    const int TWin = (L.TW - 1) * L.stride + sks, THin = (L.TH - 1) * L.stride + sks;
    P.inv_twin = 1.0f / (float)TWin;
    P.inv_tw = 1.0f / (float)L.TW;
    {
        int64_t span = ((int64_t)L.Hin * L.Win * L.in.cs - L.in.co) * 4 + (int64_t)(NI - 1) * L.in.bs * 4;  // from the slice's first element to the end of the (last) image
        if (L.in.cpb) span = (L.in.bs - (int64_t)(L.in.co >> 3) * L.in.ps) * 4 + (int64_t)(NI - 1) * L.in.bs * 4;
        if (L.up_c > 0) {  // virtual [upsample | skip] concat: 1-D 1x1 launches over plain NHWC sources only
            if (L.ks != 1 || L.in_u8 || L.B != 1 || L.Hin != 1 || !L.in2.p || L.up_c % L.CK || L.up_c >= L.cin || (L.up_W & 1) || (L.up_HW % L.up_W) ||
                ((L.up_HW / L.up_W) & 1) || L.Win % L.up_HW || (L.in2.cs & 3) || (L.in2.co & 3))
                return hipErrorInvalidValue;
            span = ((int64_t)(L.Win / 4) * L.in.cs - L.in.co) * 4;  // the low-resolution source
            int64_t s2 = ((int64_t)L.Win * L.in2.cs - L.in2.co) * 4;
            P.in2 = L.in2.p; P.in2_cs = L.in2.cs; P.in2_co = L.in2.co;
            P.in2_bs = (int64_t)L.up_HW * L.in2.cs; P.in2_sadd = (unsigned)(L.CK * 4);
            if (L.in2.cpb) {  // the skip tensor in 8-channel blocks per image
                P.in2_blk = 1; P.in2_ps = (int)L.in2.ps; P.in2_bs = L.in2.bs;
                P.in2 = (const float *)L.in2.p + (int64_t)(L.in2.co >> 3) * L.in2.ps; P.in2_co = 0;
                P.in2_sadd = (unsigned)((int64_t)(L.CK / 8) * L.in2.ps * 4);
                s2 = ((int64_t)(L.Win / L.up_HW) * L.in2.bs - (int64_t)(L.in2.co >> 3) * L.in2.ps) * 4;
            }
            if (s2 <= 0 || s2 >= (1ll << 32) - 65536) return hipErrorInvalidValue;
            P.in2_span_bytes = (unsigned)s2;
            P.up_c = L.up_c; P.up_W = L.up_W; P.up_HW = L.up_HW; P.up_stages = L.up_c / L.CK;
        }
        if (!L.in_u8 && (span <= 0 || span >= (1ll << 32) - 65536)) return hipErrorInvalidValue;  // 32-bit buffer offsets
        P.in_span_bytes = L.in_u8 ? 0u : (unsigned)span;
    }
    if ((int64_t)NI * THin * TWin * (L.in_u8 ? 1 : L.CK / 4) > (int64_t)(L.dw ? 3 : (L.up_c > 0 ? 4 : c32_maxld(L.ks, L.in_u8))) * kNW * 64) return hipErrorInvalidValue;  // staging plan: chunks per thread
    int tail_wc2 = 0;
    if (L.tail_cout > 0) {
        const Conv32Tiling t{L.TH, L.TW, L.CK, L.WC, L.MFM, NI};
        if (!conv32_tail_supported(t, L.cout, L.tail_cout) || P.ncb != 1 || !L.tail_w || !L.tail_b || !L.tail_out.p || L.res.p) return hipErrorInvalidValue;
        tail_wc2 = L.tail_cout > 32 ? 4 : (L.tail_cout > 16 ? 2 : 1);
        if (L.WC == 1 && tail_wc2 != 1) return hipErrorInvalidValue;
        if (L.WC == 2 && tail_wc2 != 2) { if (tail_wc2 == 1) tail_wc2 = 2; else return hipErrorInvalidValue; }  // (a 16-cout tail behind a 32-cout layer: the second fragment is empty)
        P.w2 = L.tail_w; P.b2 = L.tail_b; P.out2 = (float *)L.tail_out.p; P.out2_bs = L.tail_out.bs; P.out2_cs = L.tail_out.cs; P.out2_co = L.tail_out.co;
        P.out2_hw = L.tail_out_hw; P.cout2 = L.tail_cout; P.act2 = L.tail_act; P.kst2 = L.cout / 16;
        if (L.cmax) {
            if (tail_wc2 != 1 || L.tail_act) return hipErrorInvalidValue;  // (one cout fragment holds all the logits; plain outputs)
            P.cmax = L.cmax; P.cmax_bs = L.cmax_bs;
        }
    }
    P.tstep = L.xtile ? 1 : 0;
#ifdef OBB_DIAG
    { static const int dbg = getenv("OBB_C32_DBG") ? atoi(getenv("OBB_C32_DBG")) : 0; P.dbg = dbg; }  // timing-only ablations: 1 no global fetch, 2 no epilogue, 4 every activation LDS read twice, 8 no MFMAs
#endif
    const size_t lds = conv32_lds_bytes(L);
    if (lds > 80 * 1024 || P.kst > (L.ks == 3 ? 9 : 4)) return hipErrorInvalidValue;  // (kst bound: the weight-fetch plan of the kernel, MAXW)
    dim3 grid((unsigned)((ntiles + 7) / 8 * 8 * P.ncb));
    if (L.in.cpb || L.in2.cpb || L.out.cpb) return L.ks == 3 ? launch32_wc<3, true>(L, P, tail_wc2, grid, lds, st) : launch32_wc<1, true>(L, P, tail_wc2, grid, lds, st);
    return L.ks == 3 ? launch32_wc<3, false>(L, P, tail_wc2, grid, lds, st) : launch32_wc<1, false>(L, P, tail_wc2, grid, lds, st);
}

// ------------------------------------------------------------------------------------------------ network input layer as row stripes
// model.0 (Conv 3x3 s2 on the uint8 tile, CIN = 3 or 4, 16 * NF couts) with the predictor's preprocess fused in (`im.float() / 255` = the
// 256-entry table, BGR -> RGB folded into the weight order).  The generic kernel pads the 3 input channels to 4-channel chunks per tap
// (K = 48 of which 27 are real) and converts each byte once per tap; here a workgroup owns 4 output rows x the full width:
//   * the 9 input rows are whole contiguous runs: 16-byte loads, table look-up, fp32 image in LDS ([row][x + 1][CIN], column 0 = the
//     zero padding, row -1 of the first stripe = zeros);
//   * k = (ky, kx, c) runs densely over the 9 * CIN taps (27 -> 28 or 36): one v_mfma_f32_16x16x4_f32 per 4 k values; the B operand
//     of lane (pixel j, k slot g) at step t is ONE ds_read_b32 at pixel base + koff[t] (per-lane constants), the A operands (weights)
//     stay in registers for the whole kernel;
//   * a wave owns one output row and walks its fragments two at a time (independent accumulators: no dependent-issue stall); a lane's
//     four couts are one float4 store and a wave instruction covers 16 pixels x 64 contiguous bytes.
struct Stem32Params {
    const uint8_t *in; int64_t in_bs;
    float *out; int64_t out_bs; int out_cs, out_co;
    const float *wA, *bias, *lut;
    int Hin, Win, Hout, Wout, act, RS;
};

template <int CIN, int NF>
__global__ __launch_bounds__(256) void k_stem_f32(const Stem32Params P) {
    extern __shared__ __attribute__((aligned(16))) float sst[];
    constexpr int K = 9 * CIN, KST = (K + 3) / 4, R = 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int stripes = P.Hout / R;
    const int b = blockIdx.x / stripes, oy0 = (blockIdx.x % stripes) * R;
    const int RS = P.RS, rowb = P.Win * CIN, dpr = rowb >> 2;  // floats per LDS row; bytes / dwords per input row
    float *slut = sst + (2 * R + 1) * RS;
    slut[tid] = P.lut[tid];
    // LDS row: floats [0, 4) = left margin (x = -1 lives in [4 - CIN, 4): the zero padding), pixel x at 4 + x * CIN -- the image part is
    // 16-byte aligned, so that a lane converts ONE dword of the tile (4 bytes) into ONE ds_write_b128 and neighbouring lanes write
    // neighbouring 16 bytes (a lane per 16-byte chunk wrote 64-byte-strided dwords: 16 lanes per bank)
    if (tid < (2 * R + 1) * 4) sst[(tid >> 2) * RS + (tid & 3)] = 0.f;
    __syncthreads();
    const int iy0 = 2 * oy0 - 1;
    const uint8_t *src = P.in + (int64_t)b * P.in_bs;
    // every load of the stripe is issued before the first conversion (a load -> convert -> store loop paid one memory latency per dword:
    // 11 of them per thread); rows above the image read as byte 0 = table entry 0.0f = the zero padding
    constexpr int MAXI = 16;  // dwords per thread: 9 rows x Win * CIN / 4 <= 16 * 256 (stem32_supported)
    const int total = (2 * R + 1) * dpr;
    unsigned wv[MAXI];
#pragma unroll
    for (int k = 0; k < MAXI; ++k) {
        const int i = tid + k * 256;
        const int lr = i / dpr, d = i - lr * dpr;
        const int iy = iy0 + lr;
        wv[k] = (i < total && iy >= 0 && iy < P.Hin) ? *reinterpret_cast<const unsigned *>(src + (int64_t)iy * rowb + d * 4) : 0u;
    }
#pragma unroll
    for (int k = 0; k < MAXI; ++k) {
        const int i = tid + k * 256;
        const int lr = i / dpr, d = i - lr * dpr;
        if (i < total)
            *reinterpret_cast<float4 *>(sst + lr * RS + 4 + d * 4) = make_float4(slut[wv[k] & 255u], slut[(wv[k] >> 8) & 255u], slut[(wv[k] >> 16) & 255u], slut[wv[k] >> 24]);
    }
    // per-lane constants: weights (A operand: cout = lane & 15 of fragment nf, k = 4 t + g) and the LDS offset of tap k
    float wA[NF][KST];
    int koff[KST];
#pragma unroll
    for (int t = 0; t < KST; ++t) {
        const int k = 4 * t + g;
        const int tap = k / CIN, c = k - tap * CIN;
        const int ky = tap / 3, kx = tap - ky * 3;
        koff[t] = (4 - CIN) + (k < K ? ky * RS + kx * CIN + c : 0);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) wA[nf][t] = P.wA[(nf * KST + t) * 64 + lane];
    }
    float4 bv[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) bv[nf] = *reinterpret_cast<const float4 *>(P.bias + nf * 16 + 4 * g);
    __syncthreads();
    const int r = wave, oy = oy0 + r;
    const int nfr = P.Wout >> 4;
    float *orow = P.out + (int64_t)b * P.out_bs + (int64_t)oy * P.Wout * P.out_cs + P.out_co + 4 * g;
    for (int f0 = 0; f0 < nfr; f0 += 2) {
        const bool two = f0 + 1 < nfr;  // (wave-uniform)
        const int pb0 = 2 * r * RS + 2 * (16 * f0 + j) * CIN, pb1 = two ? pb0 + 32 * CIN : pb0;
        f32x4 acc0[NF], acc1[NF];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) { acc0[nf] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[nf] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        float b0[KST], b1[KST];  // every operand read is issued before the first MFMA: one LDS latency per fragment pair, not one per k step
#pragma unroll
        for (int t = 0; t < KST; ++t) { b0[t] = sst[pb0 + koff[t]]; b1[t] = sst[pb1 + koff[t]]; }
#pragma unroll
        for (int t = 0; t < KST; ++t) {
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                acc0[nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[nf][t], b0[t], acc0[nf], 0, 0, 0);
                acc1[nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[nf][t], b1[t], acc1[nf], 0, 0, 0);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && !two) break;
            float *op = orow + (int64_t)(16 * (f0 + h) + j) * P.out_cs;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                const f32x4 a = h ? acc1[nf] : acc0[nf];
                float v[4] = {a[0] + bv[nf].x, a[1] + bv[nf].y, a[2] + bv[nf].z, a[3] + bv[nf].w};
                if (P.act) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = silu32(v[q]);
                }
                *reinterpret_cast<float4 *>(op + nf * 16) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

static int stem32_rs(int Win, int cin) { return (4 + Win * cin + 3) & ~3; }

bool stem32_supported(int cin, int cout, int ks, int stride, int Hin, int Win) {
    if (ks != 3 || stride != 2 || (cin != 3 && cin != 4) || (cout != 16 && cout != 32)) return false;
    if (Hin % 8 || Win % 32) return false;  // 4-row stripes of the output, 16-pixel fragments, 16-byte input chunks
    return (size_t)(9 * stem32_rs(Win, cin) + 256) * 4 <= 64 * 1024 && 9 * Win * cin / 4 <= 16 * 256;
}

// [fragment nf][k step t][lane]: weight of cout 16 nf + (lane & 15) for k = 4 t + (lane >> 4), k = (ky * 3 + kx) * cin + c over the
// channels AS STORED in the tile (flip_bgr: stored BGR, the conv's channel order is RGB)
std::vector<float> pack_stem32_weights(const float *w, int cout, int cin, bool flip_bgr) {
    const int K = 9 * cin, KST = (K + 3) / 4, NF = cout / 16;
    std::vector<float> out((size_t)NF * KST * 64, 0.f);
    for (int nf = 0; nf < NF; ++nf)
        for (int t = 0; t < KST; ++t)
            for (int lane = 0; lane < 64; ++lane) {
                const int k = 4 * t + (lane >> 4), co = nf * 16 + (lane & 15);
                if (k >= K) continue;
                const int tap = k / cin, c = k % cin;
                const int cs = (flip_bgr && c < 3) ? 2 - c : c;
                out[((size_t)nf * KST + t) * 64 + lane] = w[((size_t)co * cin + cs) * 9 + tap];
            }
    return out;
}

hipError_t launch_stem32(const Stem32Launch &L, hipStream_t st) {
    if (!stem32_supported(L.cin, L.cout, 3, 2, L.Hin, L.Win) || L.out.cpb || (L.out.cs & 3) || (L.out.co & 3) || !L.lut) return hipErrorInvalidValue;
    Stem32Params P;
    P.in = L.in; P.in_bs = (int64_t)L.Hin * L.Win * L.cin;
    P.out = (float *)L.out.p; P.out_bs = L.out.bs; P.out_cs = L.out.cs; P.out_co = L.out.co;
    P.wA = L.wpk; P.bias = L.bias; P.lut = L.lut;
    P.Hin = L.Hin; P.Win = L.Win; P.Hout = L.Hin / 2; P.Wout = L.Win / 2; P.act = L.act; P.RS = stem32_rs(L.Win, L.cin);
    const size_t lds = (size_t)(9 * P.RS + 256) * 4;
    const dim3 grid((unsigned)(L.B * (P.Hout / 4)));
    if (L.cin == 3 && L.cout == 16) hipLaunchKernelGGL((k_stem_f32<3, 1>), grid, dim3(256), lds, st, P);
    else if (L.cin == 4 && L.cout == 16) hipLaunchKernelGGL((k_stem_f32<4, 1>), grid, dim3(256), lds, st, P);
    else if (L.cin == 3 && L.cout == 32) hipLaunchKernelGGL((k_stem_f32<3, 2>), grid, dim3(256), lds, st, P);
    else hipLaunchKernelGGL((k_stem_f32<4, 2>), grid, dim3(256), lds, st, P);
    return hipGetLastError();
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

// Column-walking form of the same layer: a thread owns (x, 4 channels) and walks R consecutive output rows with a three-row window of
// the input in registers, so that every input chunk is loaded 3 times (x - 1, x, x + 1: neighbouring lanes, the same cache lines)
// instead of 9 and the nine tap weights are read once per thread.  Taps are summed in the same (ky, kx) order as above and the zero
// padding contributes exact zeros: results are bit-identical to k_dwconv3_f32.
template <int R>
__global__ __launch_bounds__(256) void k_dwconv3_col_f32(TensorRef in, TensorRef out, TensorRef res, const float *__restrict__ w, const float *__restrict__ bias,
                                                        int B, int H, int W, int C, int act) {
    const int c4n = C >> 2, segs = (H + R - 1) / R;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)B * segs * W * c4n) return;
    const int c4 = (int)(idx % c4n);
    int64_t r_ = idx / c4n;
    const int x = (int)(r_ % W); r_ /= W;
    const int seg = (int)(r_ % segs), b = (int)(r_ / segs);
    const int y0 = seg * R;
    const float *ip = (const float *)in.p + (int64_t)b * in.bs + in.co + c4 * 4;
    float4 wv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const float4 *>(w + t * C + c4 * 4);
    const float4 bv = *reinterpret_cast<const float4 *>(bias + c4 * 4);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    auto ld = [&](int yy, int xx) { return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? *reinterpret_cast<const float4 *>(ip + ((int64_t)yy * W + xx) * in.cs) : z; };
    float4 r0[3], r1[3], r2[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { r0[k] = ld(y0 - 1, x + k - 1); r1[k] = ld(y0, x + k - 1); }
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int y = y0 + i;
        if (y >= H) break;
#pragma unroll
        for (int k = 0; k < 3; ++k) r2[k] = ld(y + 1, x + k - 1);
        float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) { a[0] = fmaf(r0[k].x, wv[k].x, a[0]); a[1] = fmaf(r0[k].y, wv[k].y, a[1]); a[2] = fmaf(r0[k].z, wv[k].z, a[2]); a[3] = fmaf(r0[k].w, wv[k].w, a[3]); }
#pragma unroll
        for (int k = 0; k < 3; ++k) { a[0] = fmaf(r1[k].x, wv[3 + k].x, a[0]); a[1] = fmaf(r1[k].y, wv[3 + k].y, a[1]); a[2] = fmaf(r1[k].z, wv[3 + k].z, a[2]); a[3] = fmaf(r1[k].w, wv[3 + k].w, a[3]); }
#pragma unroll
        for (int k = 0; k < 3; ++k) { a[0] = fmaf(r2[k].x, wv[6 + k].x, a[0]); a[1] = fmaf(r2[k].y, wv[6 + k].y, a[1]); a[2] = fmaf(r2[k].z, wv[6 + k].z, a[2]); a[3] = fmaf(r2[k].w, wv[6 + k].w, a[3]); }
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
#pragma unroll
        for (int k = 0; k < 3; ++k) { r0[k] = r1[k]; r1[k] = r2[k]; }
    }
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

// The same attention core on the exact-f32 matrix instruction (N <= 192 tokens).  One workgroup (4 waves) per (tile, head); K [NP][36]
// and V [NP][64] of all tokens in LDS (NP = N rounded up to 16, rows past N are zeros); a wave owns 16-query fragments.
//   S^T = K Q^T:  A = 16 keys x 4 dims, B = 4 dims x 16 queries -> a lane holds, for ITS query (lane & 15), the scores of the keys
//                 16 f + 4 g + r (f = key fragment, g = lane >> 4, r = register): the softmax reductions are in-lane + two xor-shuffles
//   O^T = V^T P^T: the k step (f, r) takes the keys {16 f + 4 g + r}: the B operand IS register r of score fragment f (no shuffle, no LDS
//                 round trip), the A operand row i carries the value dims 4 i + df for the four output fragments df (ONE ds_read_b128 of
//                 V[key][4 i ..] feeds four MFMAs), so that a lane ends with the 16 contiguous output dims 16 g .. 16 g + 15 of its query.
// Two-pass softmax (max, then exp / sum) and an IEEE division at the end, like the scalar kernel above.
template <int NFMAX>
__global__ __launch_bounds__(256) void k_attention_mfma_f32(TensorRef qkv, TensorRef out, int N, int nh, float scale) {
    constexpr int KD = 32, HD = 64, KST = KD + 4;
    extern __shared__ __attribute__((aligned(16))) float sm32[];
    const int NP = (N + 15) & ~15, nf = NP >> 4;
    float *sk = sm32, *sv = sm32 + (size_t)NP * KST;
    const int b = blockIdx.x / nh, h = blockIdx.x % nh;
    const float *base = (const float *)qkv.p + (int64_t)b * qkv.bs + qkv.co;
    for (int i = threadIdx.x; i < NP * (KD / 4); i += 256) {
        const int n = i / (KD / 4), c = i % (KD / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < N) v = *reinterpret_cast<const float4 *>(base + (int64_t)n * qkv.cs + nh * KD + h * KD + c * 4);
        *reinterpret_cast<float4 *>(sk + n * KST + c * 4) = v;
    }
    for (int i = threadIdx.x; i < NP * (HD / 4); i += 256) {
        const int n = i / (HD / 4), c = i % (HD / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < N) v = *reinterpret_cast<const float4 *>(base + (int64_t)n * qkv.cs + 2 * nh * KD + h * HD + c * 4);
        *reinterpret_cast<float4 *>(sv + n * HD + c * 4) = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, j = lane & 15;
    for (int qf = wave; qf < nf; qf += 4) {
        const int qrow = min(16 * qf + j, N - 1);
        const float *qp = base + (int64_t)qrow * qkv.cs + h * KD + 4 * g;
        const f32x4 q0 = *reinterpret_cast<const f32x4 *>(qp), q1 = *reinterpret_cast<const f32x4 *>(qp + 16);
        f32x4 sc[NFMAX];
#pragma unroll
        for (int f = 0; f < NFMAX; ++f) sc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f0 = 0; f0 < NFMAX; f0 += 4) {  // four key fragments at a time: their MFMAs interleave (no back-to-back dependent issue)
            f32x4 k0[4], k1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = f0 + i < nf ? f0 + i : 0;
                const float *kp = sk + (16 * f + j) * KST + 4 * g;
                k0[i] = *reinterpret_cast<const f32x4 *>(kp);
                k1[i] = *reinterpret_cast<const f32x4 *>(kp + 16);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (f0 + i < NFMAX && f0 + i < nf) sc[f0 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(k0[i][s], q0[s], sc[f0 + i], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (f0 + i < NFMAX && f0 + i < nf) sc[f0 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(k1[i][s], q1[s], sc[f0 + i], 0, 0, 0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int f = 0; f < NFMAX; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = (f < nf && 16 * f + 4 * g + r < N) ? sc[f][r] * scale : -INFINITY;
                sc[f][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float den = 0.f;
#pragma unroll
        for (int f = 0; f < NFMAX; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = expf(sc[f][r] - mx);  // masked keys: exp(-inf) = 0
                sc[f][r] = pv;
                den += pv;
            }
        den += __shfl_xor(den, 16);
        den += __shfl_xor(den, 32);
        f32x4 o[4];
#pragma unroll
        for (int df = 0; df < 4; ++df) o[df] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < NFMAX; ++f) {
            if (f < nf) {  // (wave-uniform)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const f32x4 a = *reinterpret_cast<const f32x4 *>(sv + (16 * f + 4 * g + r) * HD + 4 * j);
#pragma unroll
                    for (int df = 0; df < 4; ++df) o[df] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[df], sc[f][r], o[df], 0, 0, 0);
                }
            }
        }
        const int n = 16 * qf + j;
        if (n < N) {
            float *op = (float *)out.p + (int64_t)b * out.bs + (int64_t)n * out.cs + out.co + h * HD + 16 * g;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<float4 *>(op + 4 * r) = make_float4(o[0][r] / den, o[1][r] / den, o[2][r] / den, o[3][r] / den);
        }
    }
}

// SPPF: the three chained 5x5 stride-1 max pools of `cat` = [x | m1 | m2 | m3] in one launch: a workgroup owns (image, 32 channels); the
// plane lives in LDS; each pass = 1x5 max of the 5x1 max (window clipped at the border = -inf padding); every pass's result is stored
// to its channel slice and is the next pass's input.
__global__ __launch_bounds__(256) void k_sppf_pools_f32(TensorRef cat, int H, int W, int C) {
    extern __shared__ __attribute__((aligned(16))) float4 sp32[];  // two planes [H*W][8 chunks of 4 channels]
    const int groups = C >> 5;
    const int b = blockIdx.x / groups, cg = blockIdx.x % groups;
    const int n = H * W * 8;
    float *base = (float *)cat.p + (int64_t)b * cat.bs + cat.co + cg * 32;
    float4 *cur = sp32, *tmp = sp32 + n;
    for (int i = threadIdx.x; i < n; i += 256) cur[i] = *reinterpret_cast<const float4 *>(base + (int64_t)(i >> 3) * cat.cs + (i & 7) * 4);
    __syncthreads();
    auto mx4 = [](float4 a, float4 c) { return make_float4(fmaxf(a.x, c.x), fmaxf(a.y, c.y), fmaxf(a.z, c.z), fmaxf(a.w, c.w)); };
    for (int pass = 1; pass <= 3; ++pass) {
        for (int i = threadIdx.x; i < n; i += 256) {
            const int pix = i >> 3, c = i & 7;
            const int y = pix / W, x = pix - y * W;
            float4 m = cur[i];
            for (int xx = max(0, x - 2); xx <= min(W - 1, x + 2); ++xx) m = mx4(m, cur[(y * W + xx) * 8 + c]);
            tmp[i] = m;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) {
            const int pix = i >> 3, c = i & 7;
            const int y = pix / W, x = pix - y * W;
            float4 m = tmp[i];
            for (int yy = max(0, y - 2); yy <= min(H - 1, y + 2); ++yy) m = mx4(m, tmp[(yy * W + x) * 8 + c]);
            cur[i] = m;  // (element i of `cur` is read only by this thread in this loop: in place)
            *reinterpret_cast<float4 *>(base + (int64_t)pass * C + (int64_t)pix * cat.cs + c * 4) = m;
        }
        __syncthreads();
    }
}

static inline unsigned blocks_for32(int64_t n) { return (unsigned)((n + 255) / 256); }
static bool plain4(const TensorRef &t) { return t.cpb == 0 && (t.cs & 3) == 0 && (t.co & 3) == 0; }

hipError_t launch_dwconv3_f32(const TensorRef &in, const TensorRef &out, const TensorRef &res, const float *w, const float *bias, int B, int H, int W, int C,
                              int act, hipStream_t st) {
    if (C % 4 || !plain4(in) || !plain4(out) || (res.p && !plain4(res))) return hipErrorInvalidValue;
    // the zero-padded taps of the simple kernel are skipped, here they are fma(0, w, a): the same value unless a weight is inf / nan
    if (H >= 8) {
        constexpr int R = 13;
        hipLaunchKernelGGL((k_dwconv3_col_f32<R>), dim3(blocks_for32((int64_t)B * ((H + R - 1) / R) * W * (C / 4))), dim3(256), 0, st, in, out, res, w, bias, B, H, W, C, act);
        return hipGetLastError();
    }
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

hipError_t launch_sppf_pools_f32(const TensorRef &cat, int B, int H, int W, int C, hipStream_t st) {
    const size_t lds = (size_t)2 * H * W * 128;
    if (C % 32 || !plain4(cat) || lds > 64 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_sppf_pools_f32, dim3((unsigned)(B * (C / 32))), dim3(256), lds, st, cat, H, W, C);
    return hipGetLastError();
}

hipError_t launch_attention_f32(const TensorRef &qkv, const TensorRef &out, int B, int N, int nh, int kd, int hd, bool use_mfma, hipStream_t st) {
    if (kd != 32 || hd != 64 || !plain4(qkv) || !plain4(out) || N < 1) return hipErrorInvalidValue;
    if (use_mfma && N <= 192) {
        const int NP = (N + 15) & ~15;
        const size_t lds_m = sizeof(float) * ((size_t)NP * 36 + (size_t)NP * 64);  // <= 75 KiB
        static bool attr_m = false;
        if (!attr_m) {
            hipError_t e = hipFuncSetAttribute((const void *)k_attention_mfma_f32<12>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_attention_mfma_f32<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            if (e != hipSuccess) return e;
            attr_m = true;
        }
        const float scale_m = (float)(1.0 / sqrt((double)kd));
        if (NP <= 64) hipLaunchKernelGGL((k_attention_mfma_f32<4>), dim3((unsigned)(B * nh)), dim3(256), lds_m, st, qkv, out, N, nh, scale_m);
        else hipLaunchKernelGGL((k_attention_mfma_f32<12>), dim3((unsigned)(B * nh)), dim3(256), lds_m, st, qkv, out, N, nh, scale_m);
        return hipGetLastError();
    }
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
