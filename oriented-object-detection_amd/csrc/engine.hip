// Model runtime: weight-blob loader, YOLO11-OBB graph builder (per input shape), buffer plan and forward executor.
//
// Replaces `YOLO("best416.pt")` + `model(net_input, ...)`'s OBBModel forward (Detect_OBB.py:26, 81-83; graph =
// ultralytics==8.3.196 yolo11-obb.yaml).  The graph is lowered to a flat list of fused kernel
// launches; Concat/chunk/split never materialise: every producer writes into the channel slice of the buffer its
// consumer reads (TensorRef = base, batch stride, pixel stride, channel offset).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <tuple>

#include "bneck.h"
#include "c3k2f32.h"
#include "c3kimg.h"
#include "ctx.h"
#include "dwpw.h"
#include "f32path.h"
#include "front.h"
#include "nnops.h"
#include "pw32.h"
#include "stem.h"

namespace obb {

static constexpr int kRegMax = 16;

struct ConvRecord {
    std::string name;
    int c1, c2, k, s, g, act;
    const float *w, *b;  // into the retained host blob copy
};

struct Buf {
    int H, W, C;
    bool f32;
    bool virt = false;  // virtual concat [nearest-x2 upsample of va | vb]: never materialised, read in place by a 1x1 conv (ConvLaunch::up_c)
    int va_buf = -1, va_co = 0, va_C = 0, vb_buf = -1, vb_co = 0, vb_C = 0;
    int blk = 0;  // > 0: channel-blocked layout [C / blk][image][pixel][blk] (TensorRef::cpb); 0 = plain NHWC
    int blk32 = 0;  // fp32 mode, 8: per-image channel blocks [image][C / 8][pixel][8] (f32path.hip C32Params): written by a conv, read by 3x3 / depthwise-prologue / virtual-concat-skip launches only
    std::string name;
    int64_t off = 0;  // byte offset into the slab per image-capacity unit (resolved at allocation)
    void *p = nullptr;
    int64_t per_img() const { return (int64_t)H * W * C; }
};

struct Slice { int buf = -1, co = 0, C = 0; };

enum OpType { OP_CONV32, OP_STEM32, OP_C3K2F32, OP_PW32, OP_CONV, OP_DW, OP_POOL, OP_UP, OP_ATTN, OP_STEM, OP_SPPF, OP_BNECK, OP_C3KIMG, OP_DWPW, OP_FRONT };

struct Op {
    OpType type;
    std::string name;
    Slice in, out, res;
    int H = 0, W = 0;        // input spatial dims
    int Ho = 0, Wo = 0;      // output spatial dims
    ConvLaunch conv;         // OP_CONV
    Conv32Launch c32;        // OP_CONV32 (fp32-arithmetic mode: f32path.hip)
    Stem32Launch stem32;     // OP_STEM32 (fp32 mode: network input layer as row stripes)
    Pw32Launch pw32;         // OP_PW32 (fp32 mode: 1x1 conv with the activations read straight from global memory into the MFMA operand)
    C3k2F32Launch c3k2f;     // OP_C3K2F32 (fp32 mode: Bottleneck + closing 1x1 of a C3k2 block in one launch)
    StemLaunch stem;         // OP_STEM (network input layer as row stripes)
    FrontLaunch front;       // OP_FRONT (model.0 + model.1 + model.2.cv1 in one launch)
    BneckLaunch bneck;       // OP_BNECK (fused Bottleneck over row stripes)
    C3kImgLaunch c3kimg;     // OP_C3KIMG (inner C3k of the stride-32 level, one persistent workgroup per image)
    DwPwLaunch dwpw;         // OP_DWPW (depthwise 3x3 -> 1x1 [-> plain 1x1 to the head] over row stripes)
    double macs = 0;         // MACs of all layers of a multi-layer op
    bool one_d = false;
    bool vin = false;        // OP_CONV: the input is a virtual upsample-concat buffer
    const bf16_t *dw_w = nullptr;  // OP_DW (device): 16-bit [9][C]
    const float *dw_b = nullptr;
    const float *dw_w32 = nullptr;  // OP_DW in fp32 mode: fp32 [9][C]
    int act = 0;
    int N = 0, nh = 0, kd = 0, hd = 0;  // OP_ATTN
    int head_level = -1;     // >= 0: output goes to the caller's head tensor at this level
    bool emit_cmax = false;  // this launch writes the class logits of its level through a fused tail: it can emit their per-anchor maximum too
    int lane = 0;            // 0 = caller's stream; 1..3 = side stream of a head branch (box / class / angle)
    int wait_feat = -1;      // >= 0: first op of a head branch at this pyramid level: wait until its input feature map exists
    int signal_feat = -1;    // >= 0: this op produces the feature map of that pyramid level
};

struct Plan {
    int h = 0, w = 0, A = 0, no = 0, no_pad = 0;
    std::vector<Buf> bufs;
    std::vector<Op> ops;
    std::map<std::string, Slice> named;
    std::vector<void *> dev_allocs;
    int lvl_off[3] = {0, 0, 0};
    int cmax_mask = 0;  // levels whose class-logit maximum is written by the forward itself (the others: k_class_max behind it)
    int cap = 0;
    void *slab = nullptr;
    int64_t bytes_per_img = 0;
    double macs_per_img = 0;
    // hipGraph cache: the ~110 launches of one forward are captured once per (sub-batch size, input pointer, output pointer) and
    // replayed; a key is captured the second time it is seen (the first run is eager: it also performs one-time attribute setup)
    static constexpr int kLanes = 4;
    hipStream_t lanes[kLanes] = {};
    hipEvent_t ev_feat[kLanes] = {}, ev_done[kLanes] = {};
    typedef std::tuple<int, const void *, void *, void *> GraphKey;  // (sub-batch, tiles, head, class-logit maxima or nullptr)
    std::map<GraphKey, hipGraphExec_t> graphs;
    std::map<GraphKey, int> seen;
    std::vector<GraphKey> graph_order;  // insertion order of `graphs` (eviction)
    void drop_graphs() {
        for (auto &kv : graphs) (void)hipGraphExecDestroy(kv.second);
        graphs.clear();
        seen.clear();
        graph_order.clear();
    }
    ~Plan() {
        drop_graphs();
        for (int i = 0; i < kLanes; ++i) {
            if (lanes[i]) (void)hipStreamDestroy(lanes[i]);
            if (ev_feat[i]) (void)hipEventDestroy(ev_feat[i]);
            if (ev_done[i]) (void)hipEventDestroy(ev_done[i]);
        }
        for (void *p : dev_allocs) (void)hipFree(p);
        if (slab) (void)hipFree(slab);
    }
};

struct Model {
    std::vector<char> blob;
    int nc = 0, ch = 0, max_ch = 0;
    float width = 0, depth = 0;
    std::string scale;
    std::map<std::string, ConvRecord> recs;
    int nrec_blob = 0;  // records of the weight blob itself (synthesised merged records are added to `recs` while plans are built)
    std::map<std::string, std::pair<std::vector<float>, std::vector<float>>> merged;  // weights / bias of synthesised (cout-concatenated) records
    bool hmerge = true;  // sibling convs on the same input run as one launch ("hmerge" / "tail" = 0: separate, every tap observable)
    std::map<std::pair<int, int>, std::unique_ptr<Plan>> plans;
    bf16_t *lut_dev = nullptr;
    float *lut32_dev = nullptr;  // fp32 mode: (float)v / 255.0f
    bool f32 = false;  // fp32 arithmetic end to end, one kernel per layer (obb_set_option "precision" = 32)
    bool f16 = true;  // storage precision of activations/weights (obb_set_option "precision")
    // fused forms, each with its obb_set_option switch (EngineOpts in ctx.h); "tail" = 0 turns every intermediate-swallowing one off
    bool tail = true;    // last 1x1 conv of each head branch fused behind its producer
    bool bneck = true;   // fused Bottleneck stripes at the 104 / 52 levels
    bool upfold = true;  // Upsample + Concat in front of a 1x1 conv read in place
    EngineOpts o;        // the remaining switches as they stood at obb_model_load
    ~Model() { if (lut_dev) (void)hipFree(lut_dev); if (lut32_dev) (void)hipFree(lut32_dev); }
};

// ---------------------------------------------------------------------------------------------- blob parsing ("OBBW" v1)
#pragma pack(push, 1)
struct BlobHeader { char magic[4]; uint32_t version, nrec; int32_t nc, ch; float width, depth; int32_t max_ch, reg_max; char scale[8]; };
struct BlobRec { char name[64]; int32_t c1, c2, k, s, g, act; uint64_t w_off, b_off; };
#pragma pack(pop)

static int parse_blob(obb_ctx *ctx, Model &M) {
    const size_t n = M.blob.size();
    if (n < sizeof(BlobHeader)) return set_error(ctx, OBB_ERR_FORMAT, "weight blob too small (%zu bytes)", n);
    BlobHeader H;
    memcpy(&H, M.blob.data(), sizeof H);
    if (memcmp(H.magic, "OBBW", 4) != 0 || H.version != 1) return set_error(ctx, OBB_ERR_FORMAT, "bad weight blob magic/version");
    if (H.reg_max != kRegMax) return set_error(ctx, OBB_ERR_FORMAT, "reg_max %d unsupported", H.reg_max);
    if (H.ch != 3 && H.ch != 4) return set_error(ctx, OBB_ERR_FORMAT, "input channels %d unsupported (3 or 4)", H.ch);
    if (H.nc < 1 || H.nc > 1024) return set_error(ctx, OBB_ERR_FORMAT, "nc %d out of range", H.nc);
    M.nc = H.nc; M.ch = H.ch; M.width = H.width; M.depth = H.depth; M.max_ch = H.max_ch;
    M.scale = std::string(H.scale, strnlen(H.scale, 8));
    size_t tbl = sizeof(BlobHeader);
    if ((size_t)H.nrec > (n - tbl) / sizeof(BlobRec)) return set_error(ctx, OBB_ERR_FORMAT, "record table truncated");
    for (uint32_t i = 0; i < H.nrec; ++i) {
        BlobRec R;
        memcpy(&R, M.blob.data() + tbl + i * sizeof(BlobRec), sizeof R);
        ConvRecord c;
        c.name = std::string(R.name, strnlen(R.name, 64));
        c.c1 = R.c1; c.c2 = R.c2; c.k = R.k; c.s = R.s; c.g = R.g; c.act = R.act;
        if (c.c1 <= 0 || c.c2 <= 0 || c.c1 > 8192 || c.c2 > 8192 || (c.k != 1 && c.k != 3) || (c.s != 1 && c.s != 2) || c.g <= 0 || c.c1 % c.g)
            return set_error(ctx, OBB_ERR_FORMAT, "record %s: unsupported conv shape", c.name.c_str());
        const size_t wn = (size_t)c.c2 * (c.c1 / c.g) * c.k * c.k * 4, bn = (size_t)c.c2 * 4;  // <= 8192 * 8192 * 36: no overflow
        // subtraction forms: an offset near 2^64 must not wrap past the end check
        if (R.w_off % 4 || R.b_off % 4 || R.w_off > n || wn > n - R.w_off || R.b_off > n || bn > n - R.b_off)
            return set_error(ctx, OBB_ERR_FORMAT, "record %s: data out of range", c.name.c_str());
        c.w = reinterpret_cast<const float *>(M.blob.data() + R.w_off);
        c.b = reinterpret_cast<const float *>(M.blob.data() + R.b_off);
        M.recs[c.name] = c;
    }
    M.nrec_blob = (int)M.recs.size();
    return OBB_OK;
}

// ---------------------------------------------------------------------------------------------- graph builder
static int make_divisible(double x, int d) { return (int)std::ceil(x / d) * d; }

struct Builder {
    obb_ctx *ctx;
    Model &M;
    Plan &P;
    int err = OBB_OK;
    bool use_front = false;  // model.0 + model.1 + model.2.cv1 run as one launch (front.hip)

    int ch(int c) const { return make_divisible(std::min(c, M.max_ch) * (double)M.width, 8); }
    int reps(int n) const { return n > 1 ? std::max((int)std::lround(n * (double)M.depth), 1) : n; }

    int buf(int H, int W, int C, const std::string &name, bool f32 = false, int blk = 0, bool blk32 = false) {
        Buf b;
        b.H = H; b.W = W; b.C = C; b.f32 = f32 || M.f32; b.name = name;
        if (!M.f32 && blk >= 16 && (blk & (blk - 1)) == 0 && C % blk == 0 && C > blk) b.blk = blk;
        if (M.f32 && blk32 && M.o.blk32 && M.tail && C % 8 == 0 && C > 8) b.blk32 = 8;
        P.bufs.push_back(b);
        return (int)P.bufs.size() - 1;
    }
    int vbuf(int H, int W, Slice up_src, Slice skip, const std::string &name) {
        Buf b;
        b.H = H; b.W = W; b.C = up_src.C + skip.C; b.f32 = false; b.name = name; b.virt = true;
        b.va_buf = up_src.buf; b.va_co = up_src.co; b.va_C = up_src.C; b.vb_buf = skip.buf; b.vb_co = skip.co; b.vb_C = skip.C;
        P.bufs.push_back(b);
        return (int)P.bufs.size() - 1;
    }
    Slice whole(int b) const { return Slice{b, 0, P.bufs[b].C}; }
    Slice sub(int b, int co, int C) const { return Slice{b, co, C}; }

    const ConvRecord *rec(const std::string &name) {
        auto it = M.recs.find(name);
        if (it == M.recs.end()) {
            if (!err) err = set_error(ctx, OBB_ERR_FORMAT, "weight blob has no record '%s'", name.c_str());
            return nullptr;
        }
        return &it->second;
    }

    template <typename T>
    T *upload(const std::vector<T> &v) {
        void *d = nullptr;
        if (hipMalloc(&d, v.size() * sizeof(T) + 256) != hipSuccess) {
            if (!err) err = set_error(ctx, OBB_ERR_HIP, "hipMalloc for weights failed");
            return nullptr;
        }
        P.dev_allocs.push_back(d);
        if (hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) {
            if (!err) err = set_error(ctx, OBB_ERR_HIP, "weight upload failed");
            return nullptr;
        }
        return (T *)d;
    }

    // generic dense conv (groups == 1).  in.buf == -1 -> the uint8 network input.
    // true if `name` (a conv whose only consumer is the plain 1x1 conv `tail_name` writing to the head tensor) can run with that
    // 1x1 fused behind it (conv.hip TAIL kernels)
    bool tail_ok(const std::string &name, const std::string &tail_name, int Hin, int Win) {
        if (!M.tail) return false;
        const ConvRecord *r = rec(name), *r2 = rec(tail_name);
        if (!r || !r2 || err) return false;
        if (r->g != 1 || r2->g != 1 || r2->k != 1 || r2->s != 1 || r2->act || r2->c1 != r->c2 || r->s != 1) return false;
        int Ho = (Hin + 2 * (r->k / 2) - r->k) / r->s + 1, Wo = (Win + 2 * (r->k / 2) - r->k) / r->s + 1;
        if (M.f32) return conv32_tail_supported(plan_conv32(r->k, r->s, r->c1, r->c2, Ho, Wo, false), r->c2, r2->c2);
        ConvTiling t = plan_conv(r->k, r->s, r->c1, r->c2, Ho, Wo, M.o.pair);
        return conv_tail_supported(r->k, t.MF, t.NF, r->c2, r2->c2);
    }

    // true if the 1x1 conv `tail_name` (with activation, 16-bit output) can run inside the launch of its only producer `name`
    // (a 3x3 conv holding all of its output channels in one workgroup): stride-2 backbone conv -> cv1 of the next C3k2 block
    bool tail16_ok(const std::string &name, const std::string &tail_name, int Hin, int Win) {
        if (!M.tail || !M.o.tail16) return false;
        const ConvRecord *r = rec(name), *r2 = rec(tail_name);
        if (!r || !r2 || err) return false;
        if (r->g != 1 || r2->g != 1 || r->k != 3 || r2->k != 1 || r2->s != 1 || !r2->act || !r->act || r2->c1 != r->c2) return false;
        int Ho = (Hin + 2 * (r->k / 2) - r->k) / r->s + 1, Wo = (Win + 2 * (r->k / 2) - r->k) / r->s + 1;
        if (M.f32) return conv32_tail_supported(plan_conv32(r->k, r->s, r->c1, r->c2, Ho, Wo, false), r->c2, r2->c2);
        ConvTiling t = plan_conv(r->k, r->s, r->c1, r->c2, Ho, Wo, M.o.pair);
        return conv_tail_supported(r->k, t.MF, t.NF, r->c2, r2->c2, true, t.TH);
    }

    // true if model.0 -> model.1 -> model.2.cv1 can run as ONE launch (front.hip): the n-scale widths on tiles whose sides are multiples of 52
    bool front_ok(int h, int w) {
        if (M.f32 || !M.o.front || !M.o.stem || !M.tail || !M.o.tail16) return false;
        const ConvRecord *r0 = rec("model.0"), *r1 = rec("model.1"), *r2 = rec("model.2.cv1");
        if (!r0 || !r1 || !r2 || err) return false;
        if (r0->g != 1 || r1->g != 1 || r2->g != 1 || r0->k != 3 || r1->k != 3 || r2->k != 1 || r0->s != 2 || r1->s != 2 || r2->s != 1 || !r0->act || !r1->act || !r2->act) return false;
        if (r0->c1 != M.ch || r1->c1 != r0->c2 || r2->c1 != r1->c2) return false;
        return front_supported(M.ch, r0->c2, r1->c2, r2->c2, h, w) && stem_scale_is_exact(M.f16);
    }
    void front(Slice out, int h, int w) {
        const ConvRecord *r0 = rec("model.0"), *r1 = rec("model.1"), *r2 = rec("model.2.cv1");
        if (!r0 || !r1 || !r2 || err) return;
        Op op;
        op.type = OP_FRONT; op.name = "model.0+model.1+model.2.cv1"; op.in = Slice{-1, 0, M.ch}; op.out = out;
        op.H = h; op.W = w; op.Ho = h / 4; op.Wo = w / 4;
        FrontLaunch &L = op.front;
        L.Hin = h; L.Win = w; L.cin = M.ch; L.f16 = M.f16;
        L.w0 = upload(pack_stem_weights(r0->w, r0->c2, M.ch, M.ch == 3, M.f16));
        const ConvTiling t1{13, 13, 3, 2, 16}, t2{1, 1, 1, 2, 32};
        L.w1 = upload(pack_conv_weights(r1->w, r1->c2, r1->c1, 3, t1, nullptr, 0, M.f16));
        L.w2 = upload(pack_conv_weights(r2->w, r2->c2, r2->c1, 1, t2, nullptr, 0, M.f16));
        auto up_bias = [&](const ConvRecord *r) {
            std::vector<float> bb(((size_t)r->c2 + 63) / 64 * 64 + 64, 0.f);
            for (int c = 0; c < r->c2; ++c) bb[c] = r->b[c];
            return upload(bb);
        };
        L.b0 = up_bias(r0); L.b1 = up_bias(r1); L.b2 = up_bias(r2);
        op.macs = (double)(h / 2) * (w / 2) * r0->c2 * M.ch * 9 + (double)(h / 4) * (w / 4) * ((double)r1->c2 * r1->c1 * 9 + (double)r2->c2 * r2->c1);
        P.macs_per_img += op.macs;
        P.ops.push_back(op);
        P.named["model.2.cv1"] = out;
    }

    void conv(const std::string &name, Slice in, int Hin, int Win, Slice out, Slice res = Slice(), int head_level = -1,
              const int *perm = nullptr, const char *tail_name = nullptr) {
        const ConvRecord *r = rec(name);
        if (!r || err) return;
        if (use_front && name == "model.1" && tail_name && std::string(tail_name) == "model.2.cv1") { front(out, Hin * 2, Win * 2); return; }
        bool in_u8 = in.buf < 0;
        int cin = in_u8 ? M.ch : in.C;
        if (r->g != 1 || r->c1 != cin || (!tail_name && r->c2 != out.C)) {
            err = set_error(ctx, OBB_ERR_FORMAT, "record %s: shape (%d->%d, g%d) does not match the graph (%d->%d)", name.c_str(), r->c1,
                            r->c2, r->g, cin, out.C);
            return;
        }
        Op op;
        op.type = OP_CONV; op.name = name; op.in = in; op.out = out; op.res = res; op.head_level = head_level;
        if (!in_u8 && P.bufs[in.buf].virt) {
            const Buf &vb = P.bufs[in.buf];
            if (r->k != 1 || in.co != 0 || in.C != vb.C || vb.va_C % 64 || cin < 128) {
                err = set_error(ctx, OBB_ERR_STATE, "layer %s cannot read the virtual concat '%s'", name.c_str(), vb.name.c_str());
                return;
            }
            op.vin = true;
        }
        op.H = Hin; op.W = Win;
        op.Ho = (Hin + 2 * (r->k / 2) - r->k) / r->s + 1;
        op.Wo = (Win + 2 * (r->k / 2) - r->k) / r->s + 1;
        op.one_d = (r->k == 1);
        if (M.f32 && in_u8 && M.o.stem && !perm && !tail_name && head_level < 0 && !res.C && stem32_supported(cin, r->c2, r->k, r->s, Hin, Win)) {
            op.type = OP_STEM32;
            Stem32Launch &S = op.stem32;
            S.Hin = Hin; S.Win = Win; S.cin = cin; S.cout = r->c2; S.act = r->act;
            S.wpk = upload(pack_stem32_weights(r->w, r->c2, cin, M.ch == 3));
            std::vector<float> sb(((size_t)r->c2 + 63) / 64 * 64 + 64, 0.f);
            for (int c = 0; c < r->c2; ++c) sb[c] = r->b[c];
            S.bias = upload(sb);
            S.lut = M.lut32_dev;
            op.macs = (double)op.Ho * op.Wo * r->c2 * cin * r->k * r->k;
            P.macs_per_img += op.macs;
            P.ops.push_back(op);
            P.named[name] = out;
            return;
        }
        if (M.f32 && !in_u8 && !op.vin && P.bufs[in.buf].blk32 && r->k != 3) {
            err = set_error(ctx, OBB_ERR_STATE, "layer %s cannot read the channel-blocked buffer '%s'", name.c_str(), P.bufs[in.buf].name.c_str());
            return;
        }
        if (M.f32 && ((res.C && P.bufs[res.buf].blk32) || (tail_name && out.buf >= 0 && P.bufs[out.buf].blk32))) {
            err = set_error(ctx, OBB_ERR_STATE, "layer %s: residual / fused-1x1 output in a channel-blocked buffer", name.c_str());
            return;
        }
        if (M.f32 && M.o.pw32 && r->k == 1 && !in_u8 && !op.vin && !tail_name && head_level < 0 && !P.bufs[in.buf].blk32 && pw32_supported(cin, r->c2) &&
            !(res.C && P.bufs[res.buf].blk32)) {
            // 1x1 with >= 64 input channels: weights resident in LDS, activations straight from global memory into the MFMA operand (pw32.hip)
            op.type = OP_PW32;
            Pw32Launch &L = op.pw32;
            L.cin = cin; L.cout = r->c2; L.act = r->act;
            std::vector<float> w2((size_t)r->c2 * cin);
            for (size_t i = 0; i < w2.size(); ++i) w2[i] = r->w[i];  // (k = 1: OIHW is [cout][cin])
            L.wpk = upload(pack_pw32_weights(w2.data(), r->c2, cin, perm));
            std::vector<float> bias32(((size_t)r->c2 + 63) / 64 * 64 + 64, 0.f);
            for (int c = 0; c < r->c2; ++c) bias32[c] = r->b[perm ? perm[c] : c];
            L.bias = upload(bias32);
            op.macs = (double)op.Ho * op.Wo * r->c2 * cin;
            P.macs_per_img += op.macs;
            P.ops.push_back(op);
            P.named[name] = out;
            return;
        }
        if (M.f32) {  // fp32-arithmetic mode: exact-f32 MFMA kernels (f32path.hip)
            op.type = OP_CONV32;
            Conv32Launch &L = op.c32;
            const Conv32Tiling t = plan_conv32(r->k, r->s, cin, r->c2, op.Ho, op.Wo, in_u8, op.vin, M.o.nc2 && !tail_name);
            L.ks = r->k; L.stride = r->s; L.cin = cin; L.cout = r->c2; L.act = r->act; L.in_u8 = in_u8; L.flip_bgr = (in_u8 && M.ch == 3);
            L.TH = t.TH; L.TW = t.TW; L.CK = t.CK; L.WC = t.WC; L.MFM = t.MFM; L.NI = t.NI; L.NC = std::max(1, t.NC);
            L.Hin = Hin; L.Win = Win; L.Hout = op.Ho; L.Wout = op.Wo;
            L.tiles_y = (op.Ho + t.TH - 1) / t.TH; L.tiles_x = (op.Wo + t.TW - 1) / t.TW;
            if (op.vin && (P.bufs[in.buf].va_C % t.CK || (t.WC != 4 && t.NC != 2))) { err = set_error(ctx, OBB_ERR_STATE, "layer %s cannot read the virtual concat in fp32 mode", name.c_str()); return; }
            L.wpk = upload(pack_conv32_weights(r->w, r->c2, cin, r->k, t, perm, in_u8));
            std::vector<float> bias32(((size_t)r->c2 + 63) / 64 * 64 + 64, 0.f);
            for (int c = 0; c < r->c2; ++c) bias32[c] = r->b[perm ? perm[c] : c];
            L.bias = upload(bias32);
            L.lut = M.lut32_dev;
            op.macs = (double)op.Ho * op.Wo * r->c2 * cin * r->k * r->k;
            if (tail_name) {  // fused trailing 1x1: `out` is the slice the TAIL writes (head rows, or the cv1 output of a C3k2 block)
                const ConvRecord *r2 = rec(tail_name);
                if (!r2 || err) return;
                if (!conv32_tail_supported(t, r->c2, r2->c2) || (r2->act && (r2->c2 != out.C || head_level >= 0))) {
                    err = set_error(ctx, OBB_ERR_STATE, "fused 1x1 %s: no fp32 kernel for this pair", tail_name);
                    return;
                }
                const Conv32Tiling t2{1, 1, r->c2, 1, 1, 1};
                L.tail_w = upload(pack_conv32_weights(r2->w, r2->c2, r->c2, 1, t2, nullptr, false));
                std::vector<float> b2(((size_t)r2->c2 + 63) / 64 * 64 + 64, 0.f);
                for (int c = 0; c < r2->c2; ++c) b2[c] = r2->b[c];
                L.tail_b = upload(b2);
                L.tail_cout = r2->c2; L.tail_act = r2->act;
                op.name = name + "+" + tail_name;
                if (head_level >= 0 && out.buf == -2 && out.co == 4 * kRegMax && r2->c2 <= 16 && !r2->act) { op.emit_cmax = true; P.cmax_mask |= 1 << head_level; }
                op.macs += (double)op.Ho * op.Wo * r2->c2 * r->c2;
                P.macs_per_img += op.macs;
                P.ops.push_back(op);
                if (r2->act) P.named[tail_name] = out;
                return;
            }
            P.macs_per_img += op.macs;
            P.ops.push_back(op);
            P.named[name] = out;
            return;
        }
        if (in_u8 && M.o.stem && !perm && head_level < 0 && !res.C && stem_supported(cin, r->c2, r->k, r->s, Hin, Win) && r->act && stem_scale_is_exact(M.f16)) {
            op.type = OP_STEM;
            StemLaunch &S = op.stem;
            S.Hin = Hin; S.Win = Win; S.cin = cin; S.cout = r->c2; S.act = r->act; S.f16 = M.f16;
            S.wpk = upload(pack_stem_weights(r->w, r->c2, cin, M.ch == 3, M.f16));
            std::vector<float> sb(((size_t)r->c2 + 63) / 64 * 64 + 64, 0.f);
            for (int c = 0; c < r->c2; ++c) sb[c] = r->b[c];
            S.bias = upload(sb);
            op.macs = (double)op.Ho * op.Wo * r->c2 * cin * r->k * r->k;
            P.macs_per_img += op.macs;
            P.ops.push_back(op);
            P.named[name] = out;
            return;
        }
        ConvTiling t = plan_conv(r->k, r->s, cin, r->c2, op.Ho, op.Wo, M.o.pair);
        if (in_u8) t.CK = 8;
        ConvLaunch &L = op.conv;
        L.ks = r->k; L.stride = r->s; L.cin = cin; L.cout = r->c2; L.act = r->act;
        L.in_u8 = in_u8; L.out_f32 = head_level >= 0 && !tail_name; L.flip_bgr = (in_u8 && M.ch == 3); L.f16 = M.f16;
        L.TH = t.TH; L.TW = t.TW; L.MF = t.MF; L.NF = t.NF; L.CK = t.CK;
        L.Hin = Hin; L.Win = Win; L.Hout = op.Ho; L.Wout = op.Wo;
        L.tiles_y = (op.Ho + t.TH - 1) / t.TH; L.tiles_x = (op.Wo + t.TW - 1) / t.TW;
        std::vector<bf16_t> pk = pack_conv_weights(r->w, r->c2, cin, r->k, t, perm, in_u8, M.f16);
        L.wpk = upload(pk);
        std::vector<float> bias(((size_t)r->c2 + 63) / 64 * 64 + 64, 0.f);
        for (int c = 0; c < r->c2; ++c) bias[c] = r->b[perm ? perm[c] : c];
        L.bias = upload(bias);
        L.lut = M.lut_dev;
        P.macs_per_img += (double)op.Ho * op.Wo * r->c2 * cin * r->k * r->k;
        if (tail_name) {  // fused trailing 1x1: `out` is the head slice of the TAIL's output; this layer's own output is never written
            const ConvRecord *r2 = rec(tail_name);
            if (!r2 || err) return;
            const bool act16 = r2->act != 0;  // cv1 of a C3k2 block: SiLU, 16-bit output of its own in `out`
            const int nf2 = r2->c2 <= 16 ? 1 : (act16 && r2->c2 <= 32 ? 2 : 4);
            if (act16 && (r2->c2 != out.C || head_level >= 0)) {
                err = set_error(ctx, OBB_ERR_STATE, "fused cv1 %s: output slice does not match", tail_name);
                return;
            }
            L.tail_act16 = act16;
            ConvTiling t2{1, 1, 1, nf2, r->c2};
            L.tail_wpk = upload(pack_conv_weights(r2->w, r2->c2, r->c2, 1, t2, nullptr, 0, M.f16));
            std::vector<float> b2(((size_t)r2->c2 + 63) / 64 * 64 + 64, 0.f);
            for (int c = 0; c < r2->c2; ++c) b2[c] = r2->b[c];
            L.tail_bias = upload(b2);
            L.tail_cout = r2->c2;
            op.name = name + "+" + tail_name;
            P.macs_per_img += (double)op.Ho * op.Wo * r2->c2 * r->c2;
            P.ops.push_back(op);
            if (act16) P.named[tail_name] = out;
            return;
        }
        P.ops.push_back(op);
        P.named[name] = out;
    }

    void dwconv(const std::string &name, Slice in, int H, int W, Slice out, Slice res = Slice()) {
        const ConvRecord *r = rec(name);
        if (!r || err) return;
        if (r->g != r->c1 || r->c1 != r->c2 || r->k != 3 || r->s != 1 || r->c1 != in.C || out.C != in.C) {
            err = set_error(ctx, OBB_ERR_FORMAT, "record %s: not a depthwise 3x3 matching the graph", name.c_str());
            return;
        }
        if (M.f32 && (P.bufs[in.buf].blk32 || P.bufs[out.buf].blk32)) { err = set_error(ctx, OBB_ERR_STATE, "depthwise layer %s on a channel-blocked buffer", name.c_str()); return; }
        int C = in.C;
        std::vector<bf16_t> w((size_t)9 * C + 8, 0);
        std::vector<float> b((size_t)C + 8, 0.f);
        for (int c = 0; c < C; ++c) {
            for (int t = 0; t < 9; ++t) w[(size_t)t * C + c] = host_to_half(r->w[(size_t)c * 9 + t], M.f16);
            b[c] = r->b[c];
        }
        Op op;
        op.type = OP_DW; op.name = name; op.in = in; op.out = out; op.res = res; op.H = H; op.W = W; op.Ho = H; op.Wo = W;
        op.act = r->act;
        if (M.f32) {
            std::vector<float> w32((size_t)9 * C + 8, 0.f);
            for (int c = 0; c < C; ++c)
                for (int t = 0; t < 9; ++t) w32[(size_t)t * C + c] = r->w[(size_t)c * 9 + t];
            op.dw_w32 = upload(w32);
        } else op.dw_w = upload(w);
        op.dw_b = upload(b);
        P.macs_per_img += (double)H * W * C * 9;
        P.ops.push_back(op);
        P.named[name] = out;
    }

    // fp32 mode: DWConv 3x3 -> Conv 1x1 [-> plain 1x1 into the head rows] as ONE launch of k_conv_f32 (DW variant: the depthwise output exists
    // only in LDS).  Returns false (nothing emitted) if the shapes have no kernel.
    bool dwpw32(const std::string &dwname, const std::string &pwname, Slice in, int H, int W, Slice out, const char *tailname, int head_level) {
        const ConvRecord *rd = rec(dwname), *rp = rec(pwname), *rt = tailname ? rec(tailname) : nullptr;
        if (!(M.tail && M.o.dwpw) || !rd || !rp || (tailname && !rt) || err) return false;
        if (in.buf < 0 || P.bufs[in.buf].virt || rd->g != rd->c1 || rd->c1 != rd->c2 || rd->k != 3 || rd->s != 1 || rd->c1 != in.C || rp->g != 1 || rp->k != 1 || rp->s != 1 ||
            rp->c1 != in.C)
            return false;
        if (rt && (rt->g != 1 || rt->k != 1 || rt->s != 1 || rt->act || rt->c1 != rp->c2 || head_level < 0)) return false;
        if (!rt && (out.buf < 0 || P.bufs[out.buf].virt || out.C != rp->c2)) return false;
        const Conv32Tiling t = plan_dwpw32(in.C, rp->c2, H, W);
        if (t.TH == 0 || (rt && !conv32_tail_supported(t, rp->c2, rt->c2)) || (rt && rt->c2 > 16)) return false;
        Op op;
        op.type = OP_CONV32; op.name = dwname + "+" + pwname + (rt ? std::string("+") + tailname : std::string()); op.in = in; op.out = out;
        op.H = H; op.W = W; op.Ho = H; op.Wo = W; op.head_level = rt ? head_level : -1; op.one_d = false;
        Conv32Launch &L = op.c32;
        L.ks = 1; L.stride = 1; L.cin = in.C; L.cout = rp->c2; L.act = rp->act; L.dw = 1; L.dw_act = rd->act;
        L.TH = t.TH; L.TW = t.TW; L.CK = t.CK; L.WC = t.WC; L.MFM = t.MFM; L.NI = 1;
        L.Hin = L.Hout = H; L.Win = L.Wout = W;
        L.tiles_y = (H + t.TH - 1) / t.TH; L.tiles_x = 1;
        L.wpk = upload(pack_dwpw32_weights(rp->w, rp->c2, in.C, rd->w, rd->b, t));
        std::vector<float> bias32(((size_t)rp->c2 + 63) / 64 * 64 + 64, 0.f);
        for (int c = 0; c < rp->c2; ++c) bias32[c] = rp->b[c];
        L.bias = upload(bias32);
        op.macs = (double)H * W * (9.0 * in.C + (double)in.C * rp->c2);
        if (rt) {
            const Conv32Tiling t2{1, 1, rp->c2, 1, 1, 1};
            L.tail_w = upload(pack_conv32_weights(rt->w, rt->c2, rp->c2, 1, t2, nullptr, false));
            std::vector<float> b2(((size_t)rt->c2 + 63) / 64 * 64 + 64, 0.f);
            for (int c = 0; c < rt->c2; ++c) b2[c] = rt->b[c];
            L.tail_b = upload(b2);
            L.tail_cout = rt->c2; L.tail_act = 0;
            if (out.buf == -2 && out.co == 4 * kRegMax) { op.emit_cmax = true; P.cmax_mask |= 1 << head_level; }
            op.macs += (double)H * W * rt->c1 * rt->c2;
        }
        P.macs_per_img += op.macs;
        P.ops.push_back(op);
        if (!rt) P.named[pwname] = out;
        return true;
    }

    // DWConv 3x3 `dwname` -> Conv 1x1 `pwname` [-> plain 1x1 `tailname` into the head tensor] as one stripe kernel (dwpw.hip).
    // Returns false (nothing emitted) if the shapes have no kernel.
    bool dwpw(const std::string &dwname, const std::string &pwname, Slice in, int H, int W, Slice out, const char *tailname = nullptr, int head_level = -1) {
        if (M.f32) return dwpw32(dwname, pwname, in, H, W, out, tailname, head_level);
        const bool on = M.tail && M.o.dwpw;
        const ConvRecord *rd = rec(dwname), *rp = rec(pwname), *rt = tailname ? rec(tailname) : nullptr;
        if (!on || !rd || !rp || (tailname && !rt) || err) return false;
        if (in.buf < 0 || P.bufs[in.buf].blk || P.bufs[in.buf].virt || rd->g != rd->c1 || rd->c1 != rd->c2 || rd->k != 3 || rd->s != 1 || rd->c1 != in.C || !rd->act ||
            rp->g != 1 || rp->k != 1 || rp->s != 1 || !rp->act || rp->c1 != in.C)
            return false;
        if (rt && (rt->g != 1 || rt->k != 1 || rt->s != 1 || rt->act || rt->c1 != rp->c2 || head_level < 0)) return false;
        if (!rt && (out.buf < 0 || P.bufs[out.buf].blk || P.bufs[out.buf].virt || out.C != rp->c2)) return false;
        if (!dwpw_supported(in.C, rp->c2, H, W, rt ? rt->c2 : 0)) return false;
        const int C = in.C;
        Op op;
        op.type = OP_DWPW; op.name = dwname + "+" + pwname + (rt ? std::string("+") + tailname : std::string()); op.in = in; op.out = out;
        op.H = H; op.W = W; op.Ho = H; op.Wo = W; op.head_level = rt ? head_level : -1;
        DwPwLaunch &L = op.dwpw;
        L.H = H; L.W = W; L.cin = C; L.f16 = M.f16;
        std::vector<bf16_t> w((size_t)9 * C + 8, 0);
        std::vector<float> b((size_t)C + 8, 0.f);
        for (int c = 0; c < C; ++c) {
            for (int t = 0; t < 9; ++t) w[(size_t)t * C + c] = host_to_half(rd->w[(size_t)c * 9 + t], M.f16);
            b[c] = rd->b[c];
        }
        L.dw_w = upload(w); L.dw_b = upload(b);
        ConvTiling tp{1, 1, 1, 4, C};
        L.pw_w = upload(pack_conv_weights(rp->w, rp->c2, C, 1, tp, nullptr, 0, M.f16));
        std::vector<float> pb(64 + 64, 0.f);
        for (int c = 0; c < rp->c2; ++c) pb[c] = rp->b[c];
        L.pw_b = upload(pb);
        op.macs = (double)H * W * (9.0 * C + (double)C * rp->c2);
        if (rt) {
            L.tail_cout = rt->c2;
            L.tail_w = upload(pack_dwpw_tail(rt->w, rt->c2, M.f16));
            std::vector<float> tb(64, 0.f);
            for (int c = 0; c < rt->c2; ++c) tb[c] = rt->b[c];
            L.tail_b = upload(tb);
            op.macs += (double)H * W * rt->c1 * rt->c2;
        }
        P.macs_per_img += op.macs;
        P.ops.push_back(op);
        if (!rt) P.named[pwname] = out;
        return true;
    }

    void pool(Slice in, int H, int W, Slice out) {
        Op op; op.type = OP_POOL; op.name = "maxpool5"; op.in = in; op.out = out; op.H = H; op.W = W; op.Ho = H; op.Wo = W;
        P.ops.push_back(op);
    }
    void upsample(Slice in, int H, int W, Slice out) {
        Op op; op.type = OP_UP; op.name = "upsample2"; op.in = in; op.out = out; op.H = H; op.W = W; op.Ho = 2 * H; op.Wo = 2 * W;
        P.ops.push_back(op);
    }

    // Two sibling convs of the same kind on the same input (same k, stride, activation) as ONE conv whose weights are concatenated along
    // cout: `out` = [outputs of a | outputs of b].  Returns the name of the synthesised record ("" if the pair does not qualify).
    std::string merged_record(const std::string &na, const std::string &nb) {
        if (!M.hmerge) return "";
        const ConvRecord *a = rec(na), *b = rec(nb);
        if (!a || !b || err) return "";
        if (a->g != 1 || b->g != 1 || a->k != b->k || a->s != b->s || a->act != b->act || a->c1 != b->c1) return "";
        const std::string nm = na + "|" + nb;
        if (!M.recs.count(nm)) {
            auto &st = M.merged[nm];
            const size_t wa = (size_t)a->c2 * a->c1 * a->k * a->k, wb = (size_t)b->c2 * b->c1 * b->k * b->k;
            st.first.assign(a->w, a->w + wa);
            st.first.insert(st.first.end(), b->w, b->w + wb);
            st.second.assign(a->b, a->b + a->c2);
            st.second.insert(st.second.end(), b->b, b->b + b->c2);
            ConvRecord r = *a;
            r.name = nm; r.c2 = a->c2 + b->c2; r.w = st.first.data(); r.b = st.second.data();
            M.recs[nm] = r;
        }
        return nm;
    }

    // cv2name (optional): the closing 1x1 of the surrounding C3k2 block (over [y0 | in | out] of the concat buffer -> cv2out); returns
    // true if that conv was fused behind the Bottleneck (the caller then must not emit it)
    bool bottleneck(const std::string &name, Slice in, int H, int W, Slice out, double e, const char *cv2name = nullptr, Slice y0 = Slice(),
                    Slice cv2out = Slice()) {
        int c_ = (int)(out.C * e);
        if (M.bneck && in.buf >= 0 && in.buf == out.buf && P.bufs[in.buf].blk == in.C && in.C == out.C && c_ * 2 == in.C &&
            bneck_supported(in.C, H, W)) {
            const ConvRecord *r1 = rec(name + ".cv1"), *r2 = rec(name + ".cv2");
            if (!r1 || !r2 || err) return false;
            if (r1->k == 3 && r2->k == 3 && r1->s == 1 && r2->s == 1 && r1->g == 1 && r2->g == 1 && r1->act && r2->act && r1->c1 == in.C &&
                r1->c2 == c_ && r2->c1 == c_ && r2->c2 == in.C) {
                Op op;
                op.type = OP_BNECK; op.name = name; op.in = in; op.out = out; op.H = H; op.W = W; op.Ho = H; op.Wo = W;
                BneckLaunch &L = op.bneck;
                L.H = H; L.W = W; L.C = in.C; L.f16 = M.f16;
                ConvTiling t1{1, 1, 1, 1, in.C}, t2{1, 1, 1, in.C / 16, c_};
                L.w1pk = upload(pack_conv_weights(r1->w, c_, in.C, 3, t1, nullptr, 0, M.f16));
                L.w2pk = upload(pack_conv_weights(r2->w, in.C, c_, 3, t2, nullptr, 0, M.f16));
                std::vector<float> b1(128, 0.f), b2(128, 0.f);
                for (int c = 0; c < c_; ++c) b1[c] = r1->b[c];
                for (int c = 0; c < in.C; ++c) b2[c] = r2->b[c];
                L.bias1 = upload(b1); L.bias2 = upload(b2);
                op.macs = (double)H * W * 9.0 * in.C * c_ * 2;
                bool fused_cv2 = false;
                const ConvRecord *rc = cv2name ? rec(cv2name) : nullptr;
                if (rc && M.o.bneck_cv2 && rc->k == 1 && rc->s == 1 && rc->g == 1 && rc->act && rc->c1 == 3 * in.C && rc->c2 == cv2out.C && cv2out.buf >= 0 &&
                    !P.bufs[cv2out.buf].virt && y0.buf == in.buf && y0.C == in.C && y0.co + in.C == in.co && in.co + in.C == out.co &&
                    bneck_cv2_supported(in.C, rc->c2)) {
                    ConvTiling tc{1, 1, 1, rc->c2 / 16, in.C == 32 ? 96 : 32};
                    L.CO = rc->c2;
                    L.wc32pk = upload(pack_conv_weights(rc->w, rc->c2, rc->c1, 1, tc, nullptr, 0, M.f16));
                    if (in.C == 16) L.wc16pk = upload(pack_bneck_k16(rc->w, rc->c2, rc->c1, 32, M.f16));
                    std::vector<float> bc(128 + 64, 0.f);
                    for (int c = 0; c < rc->c2; ++c) bc[c] = rc->b[c];
                    L.biasc = upload(bc);
                    op.name = name + "+" + cv2name;
                    op.out = cv2out; op.res = y0;  // res carries the y0 slice to the launch
                    op.macs += (double)H * W * rc->c1 * rc->c2;
                    fused_cv2 = true;
                }
                P.macs_per_img += op.macs;
                P.ops.push_back(op);
                if (fused_cv2) P.named[cv2name] = cv2out;
                else P.named[name + ".cv2"] = out;
                return fused_cv2;
            }
        }
        int t = buf(H, W, c_, name + ".t");
        conv(name + ".cv1", in, H, W, whole(t));
        conv(name + ".cv2", whole(t), H, W, out, in);  // shortcut add (c1 == c2)
        return false;
    }

    void c3k(const std::string &name, Slice in, int H, int W, Slice out, int n) {
        int c_ = out.C / 2;
        const bool img_on = M.hmerge && M.o.c3kimg && !M.f32;
        if (img_on && in.buf >= 0 && out.buf >= 0 && !P.bufs[in.buf].blk && !P.bufs[in.buf].virt && !P.bufs[out.buf].blk &&
            c3kimg_supported(H, W, in.C, c_, out.C, n)) {
            // the whole block in one launch: weight stream = the six layers' MFMA fragments back to back
            const std::string names[6] = {name + ".cv1", name + ".cv2", name + ".m.0.cv1", name + ".m.0.cv2", name + ".m.1.cv1", name + ".m.1.cv2"};
            const ConvRecord *r[7];
            bool ok = true;
            for (int i = 0; i < 6; ++i) { r[i] = rec(names[i]); ok = ok && r[i]; }
            r[6] = rec(name + ".cv3"); ok = ok && r[6];
            if (!ok || err) return;
            auto is = [](const ConvRecord *q, int k, int c1, int c2) { return q->k == k && q->s == 1 && q->g == 1 && q->act && q->c1 == c1 && q->c2 == c2; };
            if (is(r[0], 1, in.C, c_) && is(r[1], 1, in.C, c_) && is(r[2], 3, c_, c_) && is(r[3], 3, c_, c_) && is(r[4], 3, c_, c_) && is(r[5], 3, c_, c_) &&
                is(r[6], 1, 2 * c_, out.C)) {
                std::vector<bf16_t> stream;
                std::vector<float> bias(6 * 128, 0.f);
                auto add = [&](const float *w, int cout, int cin, int ks) {
                    ConvTiling t{1, 1, 1, 4, cin};
                    std::vector<bf16_t> pk = pack_conv_weights(w, cout, cin, ks, t, nullptr, 0, M.f16);
                    stream.insert(stream.end(), pk.begin(), pk.end());
                };
                std::vector<float> w01((size_t)2 * c_ * in.C);
                std::copy(r[0]->w, r[0]->w + (size_t)c_ * in.C, w01.begin());
                std::copy(r[1]->w, r[1]->w + (size_t)c_ * in.C, w01.begin() + (size_t)c_ * in.C);
                add(w01.data(), 2 * c_, in.C, 1);
                for (int c = 0; c < c_; ++c) { bias[c] = r[0]->b[c]; bias[c_ + c] = r[1]->b[c]; }
                for (int i = 2; i < 6; ++i) {
                    add(r[i]->w, c_, c_, 3);
                    for (int c = 0; c < c_; ++c) bias[(i - 1) * 128 + c] = r[i]->b[c];
                }
                add(r[6]->w, out.C, 2 * c_, 1);
                for (int c = 0; c < out.C; ++c) bias[5 * 128 + c] = r[6]->b[c];
                if ((int)(stream.size() / 8) != c3kimg_pieces()) { err = set_error(ctx, OBB_ERR_STATE, "c3k image kernel: weight stream has %zu pieces", stream.size() / 8); return; }
                Op op;
                op.type = OP_C3KIMG; op.name = name; op.in = in; op.out = out; op.H = H; op.W = W; op.Ho = H; op.Wo = W;
                op.c3kimg.wts = upload(stream); op.c3kimg.bias = upload(bias); op.c3kimg.f16 = M.f16;
                op.macs = (double)H * W * ((double)in.C * 2 * c_ + 4.0 * 9 * c_ * c_ + 2.0 * c_ * out.C);
                P.macs_per_img += op.macs;
                P.ops.push_back(op);
                P.named[name + ".cv3"] = out;
                return;
            }
        }
        const std::string mn = n >= 2 ? merged_record(name + ".cv1", name + ".cv2") : std::string();
        if (!mn.empty()) {
            // cv1 and cv2 read the same tensor: one launch writes [a | b]; the last Bottleneck later overwrites the (then dead) `a` member, so
            // the same buffer is cv3's concat input.  Members are dense blocks (channel-blocked buffer).
            int ab = buf(H, W, 2 * c_, name + ".cat", false, c_);
            conv(mn, in, H, W, whole(ab));
            P.named[name + ".cv1"] = sub(ab, 0, c_);
            Slice cur = sub(ab, 0, c_);
            for (int i = 0; i < n; ++i) {
                Slice dst = (i == n - 1) ? sub(ab, 0, c_) : whole(buf(H, W, c_, name + ".m" + std::to_string(i)));
                bottleneck(name + ".m." + std::to_string(i), cur, H, W, dst, 1.0);
                cur = dst;
            }
            P.named[name + ".cv2"] = sub(ab, c_, c_);
            conv(name + ".cv3", whole(ab), H, W, out);
            return;
        }
        if (err) return;
        int cat = buf(H, W, 2 * c_, name + ".cat");
        int a = buf(H, W, c_, name + ".a");
        conv(name + ".cv1", in, H, W, whole(a));
        Slice cur = whole(a);
        for (int i = 0; i < n; ++i) {
            Slice dst = (i == n - 1) ? sub(cat, 0, c_) : whole(buf(H, W, c_, name + ".m" + std::to_string(i)));
            bottleneck(name + ".m." + std::to_string(i), cur, H, W, dst, 1.0);
            cur = dst;
        }
        conv(name + ".cv2", in, H, W, sub(cat, c_, c_));
        conv(name + ".cv3", whole(cat), H, W, out);
    }

    // `prod` (optional): the conv whose only consumer is this block, not yet emitted, reading `pin` (pH x pW): if the pair has a
    // kernel (tail16_ok) this block's cv1 runs inside the producer's launch and the producer's output tensor never exists
    void c3k2(int li, Slice in, int H, int W, Slice out, int n, bool use_c3k, double e, const char *prod = nullptr, Slice pin = Slice(), int pH = 0,
              int pW = 0) {
        std::string name = "model." + std::to_string(li);
        int c = (int)(out.C * e);
        if (prod) {
            int cat = buf(H, W, (2 + n) * c, name + ".cat", false, use_c3k ? 0 : c);
            conv(prod, pin, pH, pW, sub(cat, 0, 2 * c), Slice(), -1, nullptr, (name + ".cv1").c_str());
            c3k2_rest(name, cat, c, H, W, out, n, use_c3k);
            return;
        }
        // [y0 | y1 | y2 ...]: the bottleneck reads / writes single members of this concat -> one dense block per member
        int cat = buf(H, W, (2 + n) * c, name + ".cat", false, use_c3k ? 0 : c);
        conv(name + ".cv1", in, H, W, sub(cat, 0, 2 * c));
        c3k2_rest(name, cat, c, H, W, out, n, use_c3k);
    }
    // fp32 mode: Bottleneck (3x3, 3x3, shortcut) + closing 1x1 of a C3k2 block with one Bottleneck as ONE launch (c3k2f32.hip); false: not emitted
    bool c3k2_f32(const std::string &name, int cat, int c, int H, int W, Slice out) {
        if (!M.f32 || !M.tail || !M.o.c3k2f || out.buf < 0 || P.bufs[out.buf].virt || P.bufs[cat].blk32 || !c3k2f32_supported(c, out.C, H, W)) return false;
        const ConvRecord *r1 = rec(name + ".m.0.cv1"), *r2 = rec(name + ".m.0.cv2"), *rc = rec(name + ".cv2");
        if (!r1 || !r2 || !rc || err) return false;
        auto is = [](const ConvRecord *q, int k, int c1, int c2) { return q->k == k && q->s == 1 && q->g == 1 && q->act && q->c1 == c1 && q->c2 == c2; };
        if (!is(r1, 3, c, c / 2) || !is(r2, 3, c / 2, c) || !is(rc, 1, 3 * c, out.C)) return false;
        const std::vector<int> perm = c3k2f32_cout_perm(out.C);
        const Conv32Tiling t1{1, 1, c, 1, 1, 1, 1}, t2{1, 1, c / 2, 1, 1, 1, 1}, tc{1, 1, 3 * c, out.C / 16, 1, 1, 1};
        std::vector<float> wall = pack_conv32_weights(r1->w, c / 2, c, 3, t1, nullptr, false);  // [W1 | W2 | WC | bc]: the kernel's LDS image
        const std::vector<float> w2 = pack_conv32_weights(r2->w, c, c / 2, 3, t2, nullptr, false), wc = pack_conv32_weights(rc->w, out.C, 3 * c, 1, tc, perm.data(), false);
        wall.insert(wall.end(), w2.begin(), w2.end());
        wall.insert(wall.end(), wc.begin(), wc.end());
        for (int i = 0; i < out.C; ++i) wall.push_back(rc->b[perm[i]]);
        if (wall.size() != (size_t)(9 * 256 + 4 * 256 + 2 * 64 + (out.C / 16) * 3 * 256 + out.C)) { err = set_error(ctx, OBB_ERR_STATE, "c3k2 fp32 kernel: weight image has %zu floats", wall.size()); return false; }
        std::vector<float> b1(64, 0.f), b2(64, 0.f);
        for (int i = 0; i < c / 2; ++i) b1[i] = r1->b[i];
        for (int i = 0; i < c; ++i) b2[i] = r2->b[i];
        Op op;
        op.type = OP_C3K2F32; op.name = name + ".m.0+" + name + ".cv2"; op.in = sub(cat, 0, 2 * c); op.out = out; op.H = H; op.W = W; op.Ho = H; op.Wo = W;
        C3k2F32Launch &L = op.c3k2f;
        L.H = H; L.W = W; L.C = c; L.CO = out.C;
        L.w1 = upload(wall); L.b1 = upload(b1); L.b2 = upload(b2);
        op.macs = (double)H * W * (9.0 * c * (c / 2) * 2 + 3.0 * c * out.C);
        P.macs_per_img += op.macs;
        P.ops.push_back(op);
        P.named[name + ".cv2"] = out;
        return true;
    }

    void c3k2_rest(const std::string &name, int cat, int c, int H, int W, Slice out, int n, bool use_c3k) {
        if (n == 1 && !use_c3k && c3k2_f32(name, cat, c, H, W, out)) return;
        if (n == 1 && !use_c3k && bottleneck(name + ".m.0", sub(cat, c, c), H, W, sub(cat, 2 * c, c), 0.5, (name + ".cv2").c_str(), sub(cat, 0, c), out)) return;
        if (n == 1 && !use_c3k) { conv(name + ".cv2", whole(cat), H, W, out); return; }
        for (int i = 0; i < n; ++i) {
            Slice src = sub(cat, (1 + i) * c, c), dst = sub(cat, (2 + i) * c, c);
            if (use_c3k) c3k(name + ".m." + std::to_string(i), src, H, W, dst, 2);
            else bottleneck(name + ".m." + std::to_string(i), src, H, W, dst, 0.5);
        }
        conv(name + ".cv2", whole(cat), H, W, out);
    }


    int build() {
        const int h = P.h, w = P.w;
        const bool big = M.scale == "m" || M.scale == "l" || M.scale == "x";
        const int n2 = reps(2);
        const int c64 = ch(64), c128 = ch(128), c256 = ch(256), c512 = ch(512), c1024 = ch(1024);
        const int H2 = h / 2, W2 = w / 2, H4 = h / 4, W4 = w / 4, H8 = h / 8, W8 = w / 8, H16 = h / 16, W16 = w / 16, H32 = h / 32, W32 = w / 32;
        // concat buffers that later layers read: producers write straight into their slices
        // [up(x10), x6] and [up(x13), x4] feed 1x1 convs only: with `fold` neither Upsample nor Concat is materialised, the 1x1 reads both
        // sources in place (4x fewer bytes for the upsampled half, no copy kernels)
        const bool fold = M.upfold && c1024 % 64 == 0 && c512 % 64 == 0 && c1024 + c512 >= 128 && !big;
        int cat13 = -1, cat16 = -1;
        if (!fold) {
            cat13 = buf(H16, W16, c1024 + c512, "cat13");  // [up(x10), x6]
            cat16 = buf(H8, W8, c512 + c512, "cat16");     // [up(x13), x4]
        }
        int cat19 = buf(H16, W16, c256 + c512, "cat19");   // [x17, x13]
        int cat22 = buf(H32, W32, c512 + c1024, "cat22");  // [x20, x10]
        // fp32 mode: tensors whose consumers are 3x3 convs in 8- / 16-channel stages (stride-2 backbone convs, first head convs), depthwise
        // prologues or the skip half of a virtual concat live in 8-channel blocks per image (Buf::blk32): a stage then reads dense runs
        Slice x4 = fold ? whole(buf(H8, W8, c512, "x4", false, 0, true)) : sub(cat16, c512, c512), x6 = fold ? whole(buf(H16, W16, c512, "x6", false, 0, true)) : sub(cat13, c1024, c512);
        Slice x10 = sub(cat22, c512, c1024), x13 = sub(cat19, c256, c512);

        const bool t1 = tail16_ok("model.1", "model.2.cv1", H2, W2), t3 = tail16_ok("model.3", "model.4.cv1", H4, W4);
        use_front = t1 && front_ok(h, w);  // model.0 + model.1 + model.2.cv1 as one launch: x0 never exists
        int b0 = use_front ? -1 : buf(H2, W2, c64, "x0");
        if (!use_front) conv("model.0", Slice{-1, 0, M.ch}, h, w, whole(b0));
        int b1 = t1 ? -1 : buf(H4, W4, c128, "x1");
        if (!t1) conv("model.1", whole(b0), H2, W2, whole(b1));
        int b2 = buf(H4, W4, c256, "x2", false, 16, true);  // consumed by a 3x3 stride-2 conv in 16-channel (fp32: 8-channel) stages
        if (t1) c3k2(2, Slice(), H4, W4, whole(b2), n2, big, 0.25, "model.1", use_front ? Slice{-1, 0, c64} : whole(b0), H2, W2);
        else c3k2(2, whole(b1), H4, W4, whole(b2), n2, big, 0.25);
        int b3 = t3 ? -1 : buf(H8, W8, c256, "x3");
        if (!t3) conv("model.3", whole(b2), H4, W4, whole(b3));
        if (t3) c3k2(4, Slice(), H8, W8, x4, n2, big, 0.25, "model.3", whole(b2), H4, W4);
        else c3k2(4, whole(b3), H8, W8, x4, n2, big, 0.25);
        int b5 = buf(H16, W16, c512, "x5");
        conv("model.5", x4, H8, W8, whole(b5));
        c3k2(6, whole(b5), H16, W16, x6, n2, true, 0.5);
        int b7 = buf(H32, W32, c1024, "x7");
        conv("model.7", x6, H16, W16, whole(b7));
        int b8 = buf(H32, W32, c1024, "x8");
        c3k2(8, whole(b7), H32, W32, whole(b8), n2, true, 0.5);
        // SPPF
        int c_ = c1024 / 2;
        int cat9 = buf(H32, W32, 4 * c_, "cat9");
        conv("model.9.cv1", whole(b8), H32, W32, sub(cat9, 0, c_));
        if (M.o.sppf_fuse && c_ % 32 == 0 && (size_t)H32 * W32 * (M.f32 ? 256 : 128) <= 64 * 1024) {  // the three pools in one launch, planes resident in LDS
            Op op; op.type = OP_SPPF; op.name = "sppf.pools"; op.in = sub(cat9, 0, c_); op.out = whole(cat9); op.H = H32; op.W = W32; op.Ho = H32; op.Wo = W32;
            P.ops.push_back(op);
        } else {
            for (int i = 0; i < 3; ++i) pool(sub(cat9, i * c_, c_), H32, W32, sub(cat9, (i + 1) * c_, c_));
        }
        int b9 = buf(H32, W32, c1024, "x9");
        conv("model.9.cv2", whole(cat9), H32, W32, whole(b9));
        // C2PSA
        int cp = c1024 / 2, nh = cp / 64, hd = cp / nh, kd = hd / 2;
        int t10 = buf(H32, W32, 2 * cp, "psa.ab");
        conv("model.10.cv1", whole(b9), H32, W32, whole(t10));
        Slice bsl = sub(t10, cp, cp);
        int qkvb = buf(H32, W32, cp + 2 * nh * kd, "psa.qkv"), ao = buf(H32, W32, cp, "psa.attn"), po = buf(H32, W32, cp, "psa.pe"),
            ff = buf(H32, W32, 2 * cp, "psa.ffn");
        // qkv output channels re-ordered [q_h0..q_h(nh-1) | k_h0.. | v_h0..] so that v is one contiguous slice
        std::vector<int> perm(cp + 2 * nh * kd);
        for (int hh = 0; hh < nh; ++hh) {
            int src0 = hh * (2 * kd + hd);
            for (int d = 0; d < kd; ++d) { perm[hh * kd + d] = src0 + d; perm[nh * kd + hh * kd + d] = src0 + kd + d; }
            for (int d = 0; d < hd; ++d) perm[2 * nh * kd + hh * hd + d] = src0 + 2 * kd + d;
        }
        for (int i = 0; i < n2; ++i) {
            std::string nm = "model.10.m." + std::to_string(i);
            conv(nm + ".attn.qkv", bsl, H32, W32, whole(qkvb), Slice(), -1, perm.data());
            Op at; at.type = OP_ATTN; at.name = nm + ".attn"; at.in = whole(qkvb); at.out = whole(ao); at.H = H32; at.W = W32; at.Ho = H32; at.Wo = W32;
            at.N = H32 * W32; at.nh = nh; at.kd = kd; at.hd = hd;
            P.macs_per_img += (double)nh * ((double)at.N * at.N * kd + (double)at.N * at.N * hd);
            P.ops.push_back(at);
            P.named[nm + ".attn"] = whole(ao);
            dwconv(nm + ".attn.pe", sub(qkvb, 2 * nh * kd, cp), H32, W32, whole(po), whole(ao));
            conv(nm + ".attn.proj", whole(po), H32, W32, bsl, bsl);  // x = x + attn(x), in place on the b half
            conv(nm + ".ffn.0", bsl, H32, W32, whole(ff));
            conv(nm + ".ffn.1", whole(ff), H32, W32, bsl, bsl);      // x = x + ffn(x)
        }
        conv("model.10.cv2", whole(t10), H32, W32, x10);
        if (fold) cat13 = vbuf(H16, W16, x10, x6, "cat13");
        else upsample(x10, H32, W32, sub(cat13, 0, c1024));
        c3k2(13, whole(cat13), H16, W16, x13, n2, big, 0.5);
        if (fold) cat16 = vbuf(H8, W8, x13, x4, "cat16");
        else upsample(x13, H16, W16, sub(cat16, 0, c512));
        // (the pyramid levels are also read by the class branch's depthwise conv: blocked only where that runs as a prologue (dwpw32))
        auto feat_blk = [&](int C, int H, int W) { return M.f32 && M.tail && M.o.dwpw && plan_dwpw32(C, std::max(c256, std::min(M.nc, 100)), H, W).TH > 0; };
        int b16 = buf(H8, W8, c256, "x16", false, 0, feat_blk(c256, H8, W8));
        c3k2(16, whole(cat16), H8, W8, whole(b16), n2, big, 0.5);
        P.ops.back().signal_feat = 0;
        conv("model.17", whole(b16), H8, W8, sub(cat19, 0, c256));
        int b19 = buf(H16, W16, c512, "x19", false, 0, feat_blk(c512, H16, W16));
        c3k2(19, whole(cat19), H16, W16, whole(b19), n2, big, 0.5);
        P.ops.back().signal_feat = 1;
        conv("model.20", whole(b19), H16, W16, sub(cat22, 0, c512));
        int b22 = buf(H32, W32, c1024, "x22");
        c3k2(22, whole(cat22), H32, W32, whole(b22), n2, true, 0.5);
        P.ops.back().signal_feat = 2;
        // OBB head
        const int chs[3] = {c256, c512, c1024};
        const int feats[3] = {b16, b19, b22};
        const int Hs[3] = {H8, H16, H32}, Ws[3] = {W8, W16, W32};
        int c2 = std::max(std::max(16, chs[0] / 4), kRegMax * 4), c3 = std::max(chs[0], std::min(M.nc, 100)), c4 = std::max(chs[0] / 4, 1);
        P.no = 4 * kRegMax + M.nc + 1;
        P.no_pad = (P.no + 3) / 4 * 4;  // head rows padded to 16 B so that every lane stores whole float4s
        int off = 0;
        for (int i = 0; i < 3; ++i) { P.lvl_off[i] = off; off += Hs[i] * Ws[i]; }
        P.A = off;
        auto mark_branch = [&](size_t first, int lane, int level) { (void)first; (void)lane; (void)level; };  // (branch lanes retired: see run_round)
        Slice u1s[3];  // first conv of the angle branch, when it ran merged with the box branch's first conv
        for (int i = 0; i < 3; ++i) {
            size_t first_op = P.ops.size();
            std::string p = "model.23.cv2." + std::to_string(i), p4 = "model.23.cv4." + std::to_string(i);
            // (only where a layer is one tile per image and therefore latency-bound: at the larger levels the padded second cout block costs
            //  more MFMA time than the saved launch and input read are worth -- measured)
            //  -- except where k_conv3_pair takes the merged 64 + 16 couts as ONE group of five fragments (64 input channels, 13 x 13 tiles):
            //  no padded block there, and the feature map is read once instead of twice)
            const bool pair80 = M.o.pair && !M.f32 && P.bufs[feats[i]].C == 64 && c2 == 64 && c4 == 16 && Hs[i] % 13 == 0 && Ws[i] % 13 == 0;
            // (fp32 mode: never -- the exact-f32 MFMA is the bound there and the merged 80 couts would pad to two 64-cout blocks)
            const std::string mn = (!M.f32 && c2 % 16 == 0 && c4 % 16 == 0 && (Hs[i] * Ws[i] <= 256 || pair80)) ? merged_record(p + ".0", p4 + ".0") : std::string();
            if (err) return err;
            int t1, t2 = buf(Hs[i], Ws[i], c2, p + ".t2");
            if (!mn.empty()) {  // box and angle branch start with a 3x3 conv on the same feature map: one launch, [t1 | u1] in 16-channel blocks
                int hb = buf(Hs[i], Ws[i], c2 + c4, p + ".t1u1", false, pair80 ? 0 : 16);  // (the pair kernel reads whole 128-B pixel rows: plain NHWC there)
                conv(mn, whole(feats[i]), Hs[i], Ws[i], whole(hb));
                P.named[p + ".0"] = sub(hb, 0, c2);
                P.named[p4 + ".0"] = sub(hb, c2, c4);
                u1s[i] = sub(hb, c2, c4);
                Slice t1s = sub(hb, 0, c2);
                if (tail_ok(p + ".1", p + ".2", Hs[i], Ws[i])) {
                    conv(p + ".1", t1s, Hs[i], Ws[i], Slice{-2, 0, 4 * kRegMax}, Slice(), i, nullptr, (p + ".2").c_str());
                } else {
                    conv(p + ".1", t1s, Hs[i], Ws[i], whole(t2));
                    conv(p + ".2", whole(t2), Hs[i], Ws[i], Slice{-2, 0, 4 * kRegMax}, Slice(), i);
                }
                mark_branch(first_op, 1, i);
                continue;
            }
            t1 = buf(Hs[i], Ws[i], c2, p + ".t1", false, 0, true);
            conv(p + ".0", whole(feats[i]), Hs[i], Ws[i], whole(t1));
            if (tail_ok(p + ".1", p + ".2", Hs[i], Ws[i])) {
                conv(p + ".1", whole(t1), Hs[i], Ws[i], Slice{-2, 0, 4 * kRegMax}, Slice(), i, nullptr, (p + ".2").c_str());
            } else {
                conv(p + ".1", whole(t1), Hs[i], Ws[i], whole(t2));
                conv(p + ".2", whole(t2), Hs[i], Ws[i], Slice{-2, 0, 4 * kRegMax}, Slice(), i);
            }
            mark_branch(first_op, 1, i);
        }
        for (int i = 0; i < 3; ++i) {
            size_t first_op = P.ops.size();
            std::string p = "model.23.cv3." + std::to_string(i);
            int e1 = buf(Hs[i], Ws[i], c3, p + ".e1");
            if (!dwpw(p + ".0.0", p + ".0.1", whole(feats[i]), Hs[i], Ws[i], whole(e1))) {
                int d1 = buf(Hs[i], Ws[i], chs[i], p + ".d1");
                dwconv(p + ".0.0", whole(feats[i]), Hs[i], Ws[i], whole(d1));
                conv(p + ".0.1", whole(d1), Hs[i], Ws[i], whole(e1));
            }
            if (err) return err;
            if (dwpw(p + ".1.0", p + ".1.1", whole(e1), Hs[i], Ws[i], Slice{-2, 4 * kRegMax, M.nc}, (p + ".2").c_str(), i)) {
                mark_branch(first_op, 2, i);
                continue;
            }
            if (err) return err;
            int d2 = buf(Hs[i], Ws[i], c3, p + ".d2"), e2 = buf(Hs[i], Ws[i], c3, p + ".e2");
            dwconv(p + ".1.0", whole(e1), Hs[i], Ws[i], whole(d2));
            if (tail_ok(p + ".1.1", p + ".2", Hs[i], Ws[i])) {
                conv(p + ".1.1", whole(d2), Hs[i], Ws[i], Slice{-2, 4 * kRegMax, M.nc}, Slice(), i, nullptr, (p + ".2").c_str());
            } else {
                conv(p + ".1.1", whole(d2), Hs[i], Ws[i], whole(e2));
                conv(p + ".2", whole(e2), Hs[i], Ws[i], Slice{-2, 4 * kRegMax, M.nc}, Slice(), i);
            }
            mark_branch(first_op, 2, i);
        }
        for (int i = 0; i < 3; ++i) {
            size_t first_op = P.ops.size();
            std::string p = "model.23.cv4." + std::to_string(i);
            int u2 = buf(Hs[i], Ws[i], c4, p + ".u2");
            Slice u1sl = u1s[i];
            if (u1sl.buf < 0) {
                int u1 = buf(Hs[i], Ws[i], c4, p + ".u1");
                conv(p + ".0", whole(feats[i]), Hs[i], Ws[i], whole(u1));
                u1sl = whole(u1);
            }
            if (tail_ok(p + ".1", p + ".2", Hs[i], Ws[i])) {
                conv(p + ".1", u1sl, Hs[i], Ws[i], Slice{-2, 4 * kRegMax + M.nc, 1}, Slice(), i, nullptr, (p + ".2").c_str());
            } else {
                conv(p + ".1", u1sl, Hs[i], Ws[i], whole(u2));
                conv(p + ".2", whole(u2), Hs[i], Ws[i], Slice{-2, 4 * kRegMax + M.nc, 1}, Slice(), i);
            }
            mark_branch(first_op, 3, i);
        }
        for (const char *nm : {"x0", "x1", "x2", "x3", "x5", "x7", "x8", "x9", "x16", "x19", "x22"})
            for (size_t b = 0; b < P.bufs.size(); ++b)
                if (P.bufs[b].name == nm) P.named[nm] = whole((int)b);
        P.named["x4"] = x4; P.named["x6"] = x6; P.named["x10"] = x10; P.named["x13"] = x13;
        return err;
    }
};

static int ensure_capacity(obb_ctx *ctx, Plan &P, int B) {
    if (B <= P.cap) return OBB_OK;
    int cap = std::max(B, 1);
    int64_t off = 0;
    for (Buf &b : P.bufs) {
        b.off = off;
        if (b.virt) continue;
        off += ((int64_t)cap * b.per_img() * (b.f32 ? 4 : 2) + 255) / 256 * 256;
    }
    void *slab = nullptr;
    OBB_HIP(ctx, hipDeviceSynchronize());
    P.drop_graphs();  // captured launches point into the old slab
    if (P.slab) { (void)hipFree(P.slab); P.slab = nullptr; P.cap = 0; }
    OBB_HIP(ctx, hipMalloc(&slab, (size_t)off + 256));
    P.slab = slab;
    P.cap = cap;
    P.bytes_per_img = off / cap;
    for (Buf &b : P.bufs) b.p = (char *)slab + b.off;
    return OBB_OK;
}

static TensorRef tref(const Plan &P, const Slice &s, int boff = 0) {
    TensorRef t;
    if (s.buf < 0) return t;
    const Buf &b = P.bufs[s.buf];
    if (b.blk > 0) {  // [C / blk][cap images][pixel][blk]
        t.p = (char *)b.p + (int64_t)boff * b.H * b.W * b.blk * 2;
        t.bs = (int64_t)b.H * b.W * b.blk; t.cs = b.blk; t.co = s.co;
        t.cpb = b.blk / 8; t.ps = (int64_t)P.cap * b.H * b.W * b.blk;
        return t;
    }
    t.p = (char *)b.p + (int64_t)boff * b.per_img() * (b.f32 ? 4 : 2);  // sub-batch `boff` owns its own image range of every buffer
    t.bs = b.per_img(); t.cs = b.C; t.co = s.co;
    if (b.blk32) { t.cs = b.blk32; t.cpb = b.blk32 / 4; t.ps = (int64_t)b.H * b.W * b.blk32; }  // [image][C / 8][pixel][8]
    return t;
}

static int get_plan(obb_ctx *ctx, int h, int w, Plan **out) {
    if (!ctx->model) return set_error(ctx, OBB_ERR_STATE, "no model loaded (call obb_model_load first)");
    OBB_REQUIRE(ctx, h > 0 && w > 0 && h % 32 == 0 && w % 32 == 0 && h <= 1280 && w <= 1280,
                "input %dx%d unsupported: both sides must be multiples of 32 in [32, 1280]", h, w);
    Model &M = *ctx->model;
    auto key = std::make_pair(h, w);
    auto it = M.plans.find(key);
    if (it == M.plans.end()) {
        std::unique_ptr<Plan> P(new Plan());
        P->h = h; P->w = w;
        Builder B{ctx, M, *P};
        int rc = B.build();
        if (rc) return rc;
        it = M.plans.emplace(key, std::move(P)).first;
    }
    *out = it->second.get();
    return OBB_OK;
}

// per-anchor maximum of the class logits (the gate of obb_decode_nms_gate) for plans whose class tails are not fused
__global__ __launch_bounds__(256) void k_class_max(const float *__restrict__ head, int64_t n, int no_pad, int c0, int nc, float *__restrict__ cmax) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *r = head + i * no_pad + c0;
    float m = -INFINITY;
    for (int j = 0; j < nc; ++j) m = fmaxf(m, r[j]);
    cmax[i] = m;
}

// One sub-batch: images [boff, boff + B) of every activation buffer, all launches on `st`.
// (Measured and dropped: walking the HBM-bound stride-2 .. stride-8 front of the network in slices of 32 .. 256 tiles so that a layer's
//  output is still in the 256 MiB Infinity Cache when the next layer reads it: 9.41 ms per 1024 tiles without, 9.96 / 9.55 / 9.39 / 9.37 ms
//  with slices of 32 / 64 / 128 / 256.)
static int run_forward(obb_ctx *ctx, Plan &P, const uint8_t *tiles, int B, float *head, float *cmax, hipStream_t st, int boff) {
    Model &M = *ctx->model;
    for (Op &op : P.ops) {
        hipError_t e = hipSuccess;
        switch (op.type) {
            case OP_CONV32: {
                Conv32Launch L = op.c32;
                L.B = B;
                L.xtile = M.o.xtile;
                if (op.in.buf == -1) {
                    L.in.p = (void *)tiles; L.in.bs = (int64_t)P.h * P.w * M.ch; L.in.cs = M.ch; L.in.co = 0;
                } else if (op.vin) {
                    const Buf &vb = P.bufs[op.in.buf];
                    L.in = tref(P, Slice{vb.va_buf, vb.va_co, vb.va_C}, boff);
                    L.in2 = tref(P, Slice{vb.vb_buf, vb.vb_co, vb.vb_C}, boff);
                    L.up_c = vb.va_C; L.up_W = op.W; L.up_HW = op.H * op.W;
                } else L.in = tref(P, op.in, boff);
                TensorRef o;
                int o_hw = 0;
                if (op.head_level >= 0) {
                    o.p = head + (int64_t)P.lvl_off[op.head_level] * P.no_pad;
                    o.bs = (int64_t)P.A * P.no_pad; o.cs = P.no_pad; o.co = op.out.co;
                    if (op.one_d) o_hw = op.Ho * op.Wo;
                } else {
                    o = tref(P, op.out, boff);
                    if (op.one_d && o.cpb) o_hw = op.Ho * op.Wo;  // channel-blocked per image: the flattened pixel row is split back into (image, pixel)
                }
                if (L.tail_cout > 0) { L.tail_out = o; L.tail_out_hw = o_hw; }
                if (cmax && op.emit_cmax) { L.cmax = cmax + P.lvl_off[op.head_level]; L.cmax_bs = P.A; }
                else { L.out = o; L.out_hw = o_hw; }
                L.res = tref(P, op.res, boff);
                if (op.one_d) {  // 1x1: batch x pixels is one dense pixel row
                    const int64_t npx = (int64_t)B * op.Ho * op.Wo;
                    L.B = 1; L.Hin = L.Hout = 1; L.Win = L.Wout = (int)npx;
                    L.tiles_y = 1; L.tiles_x = (int)((npx + L.TW - 1) / L.TW);
                }
                e = launch_conv32(L, st);
                break;
            }
            case OP_PW32: {
                Pw32Launch L = op.pw32;
                L.in = tref(P, op.in, boff); L.out = tref(P, op.out, boff); L.res = tref(P, op.res, boff);
                L.npix = (int64_t)B * op.Ho * op.Wo; L.hw = op.Ho * op.Wo;
                e = launch_pw32(L, st);
                break;
            }
            case OP_C3K2F32: {
                C3k2F32Launch L = op.c3k2f;
                L.B = B; L.cat = tref(P, op.in, boff); L.out = tref(P, op.out, boff);
                e = launch_c3k2f32(L, st);
                break;
            }
            case OP_STEM32: {
                Stem32Launch L = op.stem32;
                L.B = B; L.in = tiles; L.out = tref(P, op.out, boff);
                e = launch_stem32(L, st);
                break;
            }
            case OP_CONV: {
                ConvLaunch L = op.conv;
                L.B = B;
                if (op.in.buf == -1) {
                    L.in.p = (void *)tiles; L.in.bs = (int64_t)P.h * P.w * M.ch; L.in.cs = M.ch; L.in.co = 0;
                } else if (op.vin) {
                    const Buf &vb = P.bufs[op.in.buf];
                    L.in = tref(P, Slice{vb.va_buf, vb.va_co, vb.va_C}, boff);
                    L.in2 = tref(P, Slice{vb.vb_buf, vb.vb_co, vb.vb_C}, boff);
                    L.up_c = vb.va_C; L.up_W = op.W; L.up_HW = op.H * op.W;
                } else L.in = tref(P, op.in, boff);
                if (op.head_level >= 0) {
                    TensorRef hr;
                    hr.p = head + (int64_t)P.lvl_off[op.head_level] * P.no_pad;
                    hr.bs = (int64_t)P.A * P.no_pad; hr.cs = P.no_pad; hr.co = op.out.co;
                    if (L.tail_cout > 0) {
                        L.tail_out = hr;
                        if (op.one_d) L.tail_out_hw = op.Ho * op.Wo;
                    } else {
                        L.out = hr;
                        if (op.one_d) L.out_hw = op.Ho * op.Wo;
                    }
                } else L.out = tref(P, op.out, boff);
                L.res = tref(P, op.res, boff);
                if (op.one_d) {  // 1x1: batch x pixels is one dense pixel row
                    int64_t npx = (int64_t)B * op.Ho * op.Wo;
                    L.B = 1; L.Hin = L.Hout = 1; L.Win = L.Wout = (int)npx;
                    L.tiles_y = 1; L.tiles_x = (int)((npx + L.TW - 1) / L.TW);
                } else if (M.o.nitile) {  // the 4 x 4 / 2 x 2 maps of small tiles: several whole images per 64-pixel tile
                    const int ni = conv_ni_supported(L);
                    if (ni > 1) { L.NI = ni; L.TH = L.Hout; L.TW = L.Wout; L.tiles_x = L.tiles_y = 1; }
                }
                e = launch_conv(L, st);
                break;
            }
            case OP_DW:
                if (M.f32) { e = launch_dwconv3_f32(tref(P, op.in, boff), tref(P, op.out, boff), tref(P, op.res, boff), op.dw_w32, op.dw_b, B, op.H, op.W, op.in.C, op.act, st); break; }
                e = launch_dwconv3(tref(P, op.in, boff), tref(P, op.out, boff), tref(P, op.res, boff), op.dw_w, op.dw_b, B, op.H, op.W, op.in.C, op.act, M.f16, st); break;
            case OP_C3KIMG: {
                C3kImgLaunch L = op.c3kimg;
                L.B = B; L.in = tref(P, op.in, boff); L.out = tref(P, op.out, boff);
                e = launch_c3kimg(L, st);
                break;
            }
            case OP_DWPW: {
                DwPwLaunch L = op.dwpw;
                L.B = B; L.in = tref(P, op.in, boff);
                if (op.head_level >= 0) {
                    TensorRef hr;
                    hr.p = head + (int64_t)P.lvl_off[op.head_level] * P.no_pad;
                    hr.bs = (int64_t)P.A * P.no_pad; hr.cs = P.no_pad; hr.co = op.out.co;
                    L.tail_out = hr;
                    L.sink = (char *)M.lut_dev + 512;
                } else L.out = tref(P, op.out, boff);
                e = launch_dwpw(L, st);
                break;
            }
            case OP_BNECK: {
                BneckLaunch L = op.bneck;
                L.B = B; L.y1 = tref(P, op.in, boff);
                if (L.CO > 0) {  // closing 1x1 fused: y2 stays in registers; op.out is the block's output, op.res the y0 member
                    L.y2 = L.y1; L.y0 = tref(P, op.res, boff); L.out = tref(P, op.out, boff);
                } else L.y2 = tref(P, op.out, boff);
                e = launch_bneck(L, st);
                break;
            }
            case OP_SPPF:
                if (M.f32) { e = launch_sppf_pools_f32(tref(P, op.out, boff), B, op.H, op.W, op.in.C, st); break; }
                e = launch_sppf_pools(tref(P, op.out, boff), B, op.H, op.W, op.in.C, M.f16, st); break;
            case OP_POOL:
                if (M.f32) { e = launch_maxpool5_f32(tref(P, op.in, boff), tref(P, op.out, boff), B, op.H, op.W, op.in.C, st); break; }
                e = launch_maxpool5(tref(P, op.in, boff), tref(P, op.out, boff), B, op.H, op.W, op.in.C, M.f16, st); break;
            case OP_UP:
                if (M.f32) { e = launch_upsample2_f32(tref(P, op.in, boff), tref(P, op.out, boff), B, op.H, op.W, op.in.C, st); break; }
                e = launch_upsample2(tref(P, op.in, boff), tref(P, op.out, boff), B, op.H, op.W, op.in.C, st); break;
            case OP_STEM: {
                StemLaunch L = op.stem;
                L.B = B; L.in = tiles; L.out = tref(P, op.out, boff);
                e = launch_stem(L, st);
                break;
            }
            case OP_FRONT: {
                FrontLaunch L = op.front;
                L.B = B; L.in = tiles; L.out = tref(P, op.out, boff);
                e = launch_front(L, st);
                break;
            }
            case OP_ATTN:
                if (M.f32) { e = launch_attention_f32(tref(P, op.in, boff), tref(P, op.out, boff), B, op.N, op.nh, op.kd, op.hd, M.o.attn_mfma, st); break; }
                e = launch_attention(tref(P, op.in, boff), tref(P, op.out, boff), B, op.N, op.nh, op.kd, op.hd, M.f16, M.o.attn_mfma, st); break;
        }
        if (e != hipSuccess) return set_error(ctx, OBB_ERR_HIP, "forward: launch of '%s' failed: %s", op.name.c_str(), hipGetErrorString(e));
    }
    if (cmax && P.cmax_mask != 7) {  // a plan without the fused class tails (16-bit modes, "tail" = 0, small maps): one pass over the head rows
        const int64_t n = (int64_t)B * P.A;
        hipLaunchKernelGGL(k_class_max, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, (const float *)head, n, P.no_pad, 4 * kRegMax, M.nc, cmax);
        if (hipGetLastError() != hipSuccess) return set_error(ctx, OBB_ERR_HIP, "forward: launch of the class-maximum pass failed");
    }
    return OBB_OK;
}

// One round (<= max sub-batch) of the forward.  The round is split into `nsplit` independent sub-batches that run concurrently on
// side streams (fork/join around the caller's stream): the ~110 launches of a forward are a dependent chain of short kernels, and
// two chains in flight hide each other's ramp-up, tail and launch latency (measured: 4.64 -> 4.26 ms per 256 tiles with 2).
// "fwd_split" 0 (default) = 2 chains: caller stream + side streams + one more user stream still fit the 4 hardware queues.  (3 / 4
// chains, 1024-tile bench steps: fp16 8.82 -> 9.86 / 10.58 ms; fp32 within the run-to-run noise of 2 -- 34.99 vs 36.05 on one box,
// 35.52 vs 35.14 on the next.)
static int run_round(obb_ctx *ctx, Plan &P, const uint8_t *tiles, int B, float *head, float *cmax, hipStream_t main_st) {
    Model &M = *ctx->model;
    const int nsplit_cfg = ctx->opt.fwd_split > 0 ? std::min(Plan::kLanes, ctx->opt.fwd_split) : 2;
    int ns = (B >= 32 * nsplit_cfg) ? nsplit_cfg : 1;
    if (ns == 1) return run_forward(ctx, P, tiles, B, head, cmax, main_st, 0);
    if (!P.lanes[0]) {
        for (int i = 0; i < Plan::kLanes; ++i) {
            OBB_HIP(ctx, hipStreamCreateWithFlags(&P.lanes[i], hipStreamNonBlocking));
            OBB_HIP(ctx, hipEventCreateWithFlags(&P.ev_feat[i], hipEventDisableTiming));
            OBB_HIP(ctx, hipEventCreateWithFlags(&P.ev_done[i], hipEventDisableTiming));
        }
    }
    OBB_HIP(ctx, hipEventRecord(P.ev_feat[0], main_st));  // fork point
    for (int i = 0; i < ns; ++i) {
        int lo = (int)((int64_t)B * i / ns), hi = (int)((int64_t)B * (i + 1) / ns);
        OBB_HIP(ctx, hipStreamWaitEvent(P.lanes[i], P.ev_feat[0], 0));
        int rc = run_forward(ctx, P, tiles + (int64_t)lo * P.h * P.w * M.ch, hi - lo, head + (int64_t)lo * P.A * P.no_pad, cmax ? cmax + (int64_t)lo * P.A : nullptr, P.lanes[i], lo);
        if (rc) return rc;
        OBB_HIP(ctx, hipEventRecord(P.ev_done[i], P.lanes[i]));
    }
    for (int i = 0; i < ns; ++i) OBB_HIP(ctx, hipStreamWaitEvent(main_st, P.ev_done[i], 0));  // join
    return OBB_OK;
}

template <bool F16>
__global__ void k_half_slice_to_f32(const bf16_t *__restrict__ src, int64_t bs, int cs, int co, int C, int64_t npix_per_img, int B,
                                    float *__restrict__ dst, int blk, int64_t ps) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t total = (int64_t)B * npix_per_img * C;
    if (i >= total) return;
    int c = (int)(i % C) + co;
    int64_t pix = (i / C) % npix_per_img;
    int64_t b = i / ((int64_t)C * npix_per_img);
    dst[i] = HX<F16>::one(blk > 0 ? src[(int64_t)(c / blk) * ps + b * bs + pix * blk + c % blk] : src[b * bs + pix * cs + c]);
}

__global__ void k_f32_slice_copy(const float *__restrict__ src, int64_t bs, int cs, int co, int C, int64_t npix_per_img, int B, float *__restrict__ dst, int blk) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * npix_per_img * C) return;
    int c = (int)(i % C) + co;
    int64_t pix = (i / C) % npix_per_img;
    int64_t b = i / ((int64_t)C * npix_per_img);
    dst[i] = blk > 0 ? src[b * bs + (int64_t)(c / blk) * npix_per_img * blk + pix * blk + c % blk] : src[b * bs + pix * cs + c];
}

}  // namespace obb

using namespace obb;

extern "C" {

int obb_model_load(obb_ctx *ctx, const void *blob_host, size_t bytes) {
    OBB_REQUIRE(ctx, ctx && blob_host && bytes > 0, "obb_model_load: bad arguments");
    OBB_HIP(ctx, hipSetDevice(ctx->device));
    OBB_HIP(ctx, hipDeviceSynchronize());
    std::shared_ptr<Model> M(new Model());
    M->blob.assign((const char *)blob_host, (const char *)blob_host + bytes);
    int rc = parse_blob(ctx, *M);
    if (rc) return rc;
    M->f16 = ctx->opt_f16;
    M->f32 = ctx->opt_f32;
    M->o = ctx->opt;
    M->tail = ctx->opt.tail;
    M->upfold = ctx->opt.upfold;
    M->hmerge = M->tail && ctx->opt.hmerge;
    M->bneck = M->tail && ctx->opt.bneck;  // both swallow intermediate activations ("tail" = 0 keeps every layer observable)
    if (M->f32) {  // `im.float() / 255`: IEEE division, one table entry per byte value
        std::vector<float> lut32(256);
        for (int v = 0; v < 256; ++v) lut32[v] = (float)v / 255.0f;
        OBB_HIP(ctx, hipMalloc((void **)&M->lut32_dev, 1024));
        OBB_HIP(ctx, hipMemcpy(M->lut32_dev, lut32.data(), 1024, hipMemcpyHostToDevice));
    }
    // u8 -> half(v / 255): the predictor's `im.float() / 255` followed by the 16-bit storage rounding, exactly
    std::vector<bf16_t> lut(256);
    for (int v = 0; v < 256; ++v) lut[v] = host_to_half((float)v / 255.0f, M->f16);
    OBB_HIP(ctx, hipMalloc((void **)&M->lut_dev, 512 + 1024));  // + a 1-KiB sink for the stores of lanes without an output pixel (conv.hip EXACT)
    OBB_HIP(ctx, hipMemcpy(M->lut_dev, lut.data(), 512, hipMemcpyHostToDevice));
    ctx->model = M;
    return OBB_OK;
}

int obb_model_unload(obb_ctx *ctx, int32_t slot) {
    OBB_REQUIRE(ctx, ctx && slot >= 0 && slot < 64, "obb_model_unload: bad arguments");
    OBB_HIP(ctx, hipSetDevice(ctx->device));
    OBB_HIP(ctx, hipDeviceSynchronize());  // nothing of the model's plans (slabs, weights, graphs) may still be in flight
    if (slot == ctx->slot) ctx->model.reset();
    ctx->slots.erase(slot);
    return OBB_OK;
}

int obb_set_option(obb_ctx *ctx, const char *key, int64_t value) {
    OBB_REQUIRE(ctx, ctx && key, "obb_set_option: bad arguments");
    std::string k(key);
    if (k == "precision") {  // 16 = fp16 storage (default), 1016 = bf16 storage, 32 = fp32 arithmetic end to end; next obb_model_load
        OBB_REQUIRE(ctx, value == 16 || value == 1016 || value == 32, "obb_set_option: precision must be 16 (fp16), 1016 (bf16) or 32 (fp32)");
        ctx->opt_f16 = (value != 1016);
        ctx->opt_f32 = (value == 32);
        return OBB_OK;
    }
    {   // engine switches: 1 = the fused form (default), 0 = its separate launches; all apply to the next obb_model_load, except
        // "graph", "fwd_split" and "microbatch", which steer how obb_forward issues its launches from the next call on
        struct { const char *key; bool *flag; } sw[] = {
            {"tail", &ctx->opt.tail}, {"tail16", &ctx->opt.tail16}, {"bneck", &ctx->opt.bneck}, {"bneck_cv2", &ctx->opt.bneck_cv2},
            {"c3kimg", &ctx->opt.c3kimg}, {"dwpw", &ctx->opt.dwpw}, {"upfold", &ctx->opt.upfold}, {"stem", &ctx->opt.stem}, {"front", &ctx->opt.front}, {"pair", &ctx->opt.pair},
            {"hmerge", &ctx->opt.hmerge}, {"sppf_fuse", &ctx->opt.sppf_fuse}, {"attn_mfma", &ctx->opt.attn_mfma}, {"xtile", &ctx->opt.xtile}, {"nitile", &ctx->opt.nitile}, {"nc2", &ctx->opt.nc2}, {"blk32", &ctx->opt.blk32}, {"c3k2f", &ctx->opt.c3k2f}, {"pw32", &ctx->opt.pw32}, {"graph", &ctx->opt.graph}};
        for (auto &e : sw)
            if (k == e.key) { *e.flag = value != 0; return OBB_OK; }
        if (k == "fuse") return OBB_OK;  // (retired: the LDS-resident layer chains were slower than layer-by-layer on MI355X and are gone)
        if (k == "fwd_split") { OBB_REQUIRE(ctx, value >= 0 && value <= 4, "obb_set_option: fwd_split must be 0 (auto) .. 4"); ctx->opt.fwd_split = (int)value; return OBB_OK; }
        if (k == "microbatch") { OBB_REQUIRE(ctx, value >= 1 && value <= 1024, "obb_set_option: microbatch must be 1..1024"); ctx->opt.microbatch = (int)value; return OBB_OK; }
    }
    if (k == "model_slot") {  // several models per context (dual-scale 128 + 416): select which one load/forward address
        OBB_REQUIRE(ctx, value >= 0 && value < 64, "obb_set_option: model_slot must be in [0, 64)");
        if ((int)value != ctx->slot) {
            ctx->slots[ctx->slot] = ctx->model;
            ctx->model = ctx->slots[(int)value];
            ctx->slot = (int)value;
        }
        return OBB_OK;
    }
    return set_error(ctx, OBB_ERR_INVALID, "obb_set_option: unknown key '%s'", key);
}

int obb_model_info(const obb_ctx *cctx, int32_t h, int32_t w, int32_t *nc, int32_t *ch, int32_t *anchors, int32_t *nconv) {
    obb_ctx *ctx = const_cast<obb_ctx *>(cctx);
    OBB_REQUIRE(ctx, ctx, "obb_model_info: NULL context");
    if (!ctx->model) return set_error(ctx, OBB_ERR_STATE, "no model loaded");
    if (nc) *nc = ctx->model->nc;
    if (ch) *ch = ctx->model->ch;
    if (nconv) *nconv = (int32_t)ctx->model->nrec_blob;
    if (anchors) {
        OBB_REQUIRE(ctx, h > 0 && w > 0 && h % 32 == 0 && w % 32 == 0, "obb_model_info: h, w must be multiples of 32");
        *anchors = (h / 8) * (w / 8) + (h / 16) * (w / 16) + (h / 32) * (w / 32);
    }
    return OBB_OK;
}

int obb_forward(obb_ctx *ctx, const uint8_t *tiles, int32_t B, int32_t h, int32_t w, float *head, obb_stream_t s) {
    return obb_forward_gate(ctx, tiles, B, h, w, head, nullptr, s);
}

int obb_forward_gate(obb_ctx *ctx, const uint8_t *tiles, int32_t B, int32_t h, int32_t w, float *head, float *cmax, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && B >= 0, "obb_forward: bad arguments");
    if (B == 0) return OBB_OK;
    OBB_REQUIRE(ctx, tiles && head, "obb_forward: NULL buffer");
    Plan *P = nullptr;
    int rc = get_plan(ctx, h, w, &P);
    if (rc) return rc;
    // The batch is walked in sub-batches: bounds the activation slab (and keeps every 1-D launch inside 32-bit buffer offsets)
    // "microbatch" counts 416 x 416 tiles: smaller tiles get proportionally more per round (128 px: 11 264 -- the 10 764 tiles of the dual-scale map are ONE round of two 5382-tile chains --, at most 16 384) -- the same activation bytes and
    // launch sizes, instead of 1024-tile rounds whose 4 x 4 / 8 x 8 levels are 74-WG launches on a 256-CU chip.  (measured at 416 px,
    // B = 1024: rounds of 512 / 1024 -> 91.8 / 94.7 k tiles/s; 128 / 256: 67 / 79 k with the round-1 kernels)
    const int max_mb = (int)std::min<int64_t>(16384, (int64_t)ctx->opt.microbatch * std::max<int64_t>(1, (416 * 416 + (int64_t)h * w - 1) / ((int64_t)h * w)));
    rc = ensure_capacity(ctx, *P, std::min<int>(B, max_mb));
    if (rc) return rc;
    bool use_graph = ctx->opt.graph;
    {   // inside somebody else's capture (torch.cuda.graph around the registered op) the launches simply join that graph
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) use_graph = false;
    }
    hipStream_t st = (hipStream_t)s;
    // (equal rounds: 21 000 tiles of 128 px are 3 x 7000, not 10 240 + 10 240 + 520 -- a short round runs its small-map layers at a fraction of
    // the workgroups the chip holds)
    const int nrounds = (B + max_mb - 1) / max_mb, per_round = (B + nrounds - 1) / nrounds;
    for (int b0 = 0; b0 < B; b0 += per_round) {
        int nb = std::min<int>(per_round, B - b0);
        const uint8_t *tp = tiles + (int64_t)b0 * h * w * ctx->model->ch;
        float *hp = head + (int64_t)b0 * P->A * P->no_pad;
        float *cp = cmax ? cmax + (int64_t)b0 * P->A : nullptr;
        Plan::GraphKey key(nb, (const void *)tp, (void *)hp, (void *)cp);
        auto git = P->graphs.find(key);
        if (use_graph && git != P->graphs.end()) {
            OBB_HIP(ctx, hipGraphLaunch(git->second, st));
            continue;
        }
        if (P->seen.size() > 4096) P->seen.clear();  // keys are (batch, input pointer, output pointer): a caller that never repeats one must not grow this
        bool capture = use_graph && P->seen[key]++ >= 1;
        if (capture && P->graphs.size() >= 64) {  // cache full: drop the oldest half (insertion order), then capture the new key
            OBB_HIP(ctx, hipStreamSynchronize(st));
            while (P->graph_order.size() > 32) {
                auto it = P->graphs.find(P->graph_order.front());
                if (it != P->graphs.end()) { (void)hipGraphExecDestroy(it->second); P->graphs.erase(it); }
                P->graph_order.erase(P->graph_order.begin());
            }
        }
        if (capture && hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) != hipSuccess) { (void)hipGetLastError(); capture = false; }
        rc = run_round(ctx, *P, tp, nb, hp, cp, st);
        if (capture) {
            hipGraph_t g = nullptr;
            hipError_t e = hipStreamEndCapture(st, &g);
            if (rc == OBB_OK && e == hipSuccess && g) {
                hipGraphExec_t ge = nullptr;
                if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess) {
                    P->graphs[key] = ge;
                    P->graph_order.push_back(key);
                    (void)hipGraphDestroy(g);
                    OBB_HIP(ctx, hipGraphLaunch(ge, st));
                    continue;
                }
                (void)hipGraphDestroy(g);
            }
            (void)hipGetLastError();
            if (rc) return rc;
            rc = run_round(ctx, *P, tp, nb, hp, cp, st);  // capture unavailable: run eagerly
        }
        if (rc) return rc;
    }
    return OBB_OK;
}

int obb_debug_plan(obb_ctx *ctx, int32_t h, int32_t w, char *buf, int64_t buf_bytes, int64_t *needed) {
    OBB_REQUIRE(ctx, ctx && needed, "obb_debug_plan: bad arguments");
    Plan *P = nullptr;
    int rc = get_plan(ctx, h, w, &P);
    if (rc) return rc;
    std::string out;
    char line[512];
    for (const Op &op : P->ops) {
        double macs = 0;
        const char *ty = "?";
        int grid_x = 0, grid_y = 0, lds = 0;
        switch (op.type) {
            case OP_CONV32: {
                const Conv32Launch &L = op.c32;
                snprintf(line, sizeof line, "conv32 %s k%d s%d cin%d cout%d out%dx%d TH%d TW%d NI%d CK%d WC%d NC%d MFM%d dw%d tail%d vcat%d lds%d macs%.0f\n", op.name.c_str(), L.ks,
                         L.stride, L.cin, L.cout, op.Ho, op.Wo, L.TH, L.TW, L.NI, L.CK, L.WC, L.NC, L.MFM, L.dw, L.tail_cout, op.vin ? 1 : 0, (int)conv32_lds_bytes(L), op.macs);
                break;
            }
            case OP_CONV: {
                ty = "conv";
                const ConvLaunch &L = op.conv;
                macs = (double)op.Ho * op.Wo * L.cout * L.cin * L.ks * L.ks;
                grid_x = op.one_d ? -(op.Ho * op.Wo) : L.tiles_x * L.tiles_y;  // negative: pixels per image of a 1-D launch
                grid_y = (L.cout + 16 * L.NF - 1) / (16 * L.NF);
                lds = (int)conv_lds_bytes(L);
                if (L.tail_cout > 0) macs += (double)op.Ho * op.Wo * L.tail_cout * L.cout;
                const int ni = (!op.one_d && ctx->model->o.nitile) ? conv_ni_supported(L) : 1;  // (decided per launch: images per tile on the small maps)
                snprintf(line, sizeof line, "%s %s k%d s%d cin%d cout%d out%dx%d TH%d TW%d MF%d NF%d CK%d NI%d gx%d gy%d lds%d macs%.0f\n", ty,
                         op.name.c_str(), L.ks, L.stride, L.cin, L.cout, op.Ho, op.Wo, ni > 1 ? op.Ho : L.TH, ni > 1 ? op.Wo : L.TW, L.MF, L.NF, L.CK, ni, grid_x, grid_y, lds, macs);
                break;
            }
            case OP_DW: ty = "dwconv"; macs = (double)op.H * op.W * op.in.C * 9;
                snprintf(line, sizeof line, "%s %s c%d out%dx%d macs%.0f\n", ty, op.name.c_str(), op.in.C, op.Ho, op.Wo, macs); break;
            case OP_C3KIMG: snprintf(line, sizeof line, "c3kimg %s c%d out%dx%d macs%.0f\n", op.name.c_str(), op.in.C, op.Ho, op.Wo, op.macs); break;
            case OP_DWPW: snprintf(line, sizeof line, "dwpw %s c%d tail%d out%dx%d macs%.0f\n", op.name.c_str(), op.dwpw.cin, op.dwpw.tail_cout, op.Ho, op.Wo, op.macs); break;
            case OP_BNECK: snprintf(line, sizeof line, "bneck %s c%d co%d out%dx%d rows4 macs%.0f\n", op.name.c_str(), op.bneck.C, op.bneck.CO, op.Ho, op.Wo, op.macs); break;
            case OP_SPPF: snprintf(line, sizeof line, "pool %s c%d out%dx%d x3 macs0\n", op.name.c_str(), op.in.C, op.Ho, op.Wo); break;
            case OP_POOL: snprintf(line, sizeof line, "pool %s c%d out%dx%d macs0\n", op.name.c_str(), op.in.C, op.Ho, op.Wo); break;
            case OP_UP: snprintf(line, sizeof line, "upsample %s c%d out%dx%d macs0\n", op.name.c_str(), op.in.C, op.Ho, op.Wo); break;
            case OP_PW32:
                snprintf(line, sizeof line, "pw32 %s k1 s1 cin%d cout%d out%dx%d waves%d lds%d macs%.0f\n", op.name.c_str(), op.pw32.cin, op.pw32.cout, op.Ho, op.Wo,
                         op.pw32.cin * 256 <= 80 * 1024 ? 8 : 16, op.pw32.cin * 256, op.macs);
                break;
            case OP_C3K2F32: {
                int th, tw;
                c3k2f32_tile(op.H, op.W, th, tw);
                snprintf(line, sizeof line, "c3k2f32 %s c%d co%d out%dx%d TH%d TW%d macs%.0f\n", op.name.c_str(), op.c3k2f.C, op.c3k2f.CO, op.Ho, op.Wo, th, tw, op.macs);
                break;
            }
            case OP_STEM32:
                snprintf(line, sizeof line, "stem32 %s k3 s2 cin%d cout%d out%dx%d rows%d macs%.0f\n", op.name.c_str(), op.stem32.cin, op.stem32.cout, op.Ho, op.Wo, 4, op.macs);
                break;
            case OP_STEM:
                snprintf(line, sizeof line, "stem %s k3 s2 cin%d cout%d out%dx%d rows%d macs%.0f\n", op.name.c_str(), op.stem.cin, op.stem.cout, op.Ho, op.Wo, 4, op.macs);
                break;
            case OP_FRONT: snprintf(line, sizeof line, "front %s cin%d out%dx%d macs%.0f\n", op.name.c_str(), op.front.cin, op.Ho, op.Wo, op.macs); break;
            case OP_ATTN: macs = (double)op.nh * ((double)op.N * op.N * op.kd + (double)op.N * op.N * op.hd);
                snprintf(line, sizeof line, "attn %s N%d nh%d macs%.0f\n", op.name.c_str(), op.N, op.nh, macs); break;
        }
        out += line;
    }
    snprintf(line, sizeof line, "total_macs %.0f bytes_per_img %lld nbufs %zu nops %zu\n", P->macs_per_img, (long long)P->bytes_per_img, P->bufs.size(),
             P->ops.size());
    out += line;
    *needed = (int64_t)out.size() + 1;
    if (buf && buf_bytes >= *needed) memcpy(buf, out.c_str(), out.size() + 1);
    return OBB_OK;
}

int obb_debug_activation(obb_ctx *ctx, int32_t h, int32_t w, int32_t B, const char *name, float *out, int64_t max_elems,
                         int64_t *n_elems, int32_t *shape_hwc_host, obb_stream_t s) {
    OBB_REQUIRE(ctx, ctx && name && n_elems, "obb_debug_activation: bad arguments");
    Plan *P = nullptr;
    int rc = get_plan(ctx, h, w, &P);
    if (rc) return rc;
    auto it = P->named.find(name);
    if (it == P->named.end() || it->second.buf < 0) return set_error(ctx, OBB_ERR_INVALID, "obb_debug_activation: no activation named '%s'", name);
    OBB_REQUIRE(ctx, B <= P->cap, "obb_debug_activation: run obb_forward with B >= %d first", B);
    const Slice &sl = it->second;
    const Buf &b = P->bufs[sl.buf];
    int64_t n = (int64_t)B * b.H * b.W * sl.C;
    *n_elems = n;
    if (shape_hwc_host) { shape_hwc_host[0] = b.H; shape_hwc_host[1] = b.W; shape_hwc_host[2] = sl.C; }
    if (!out) return OBB_OK;
    OBB_REQUIRE(ctx, max_elems >= n, "obb_debug_activation: output too small (%lld < %lld)", (long long)max_elems, (long long)n);
    const int64_t hw = (int64_t)b.H * b.W;
    const int64_t d_bs = b.blk > 0 ? hw * b.blk : b.per_img(), d_ps = b.blk > 0 ? (int64_t)P->cap * hw * b.blk : 0;
    if (b.f32)
        hipLaunchKernelGGL(k_f32_slice_copy, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, (const float *)b.p, (int64_t)b.per_img(), b.C, sl.co, sl.C, hw,
                           B, out, b.blk32);
    else if (ctx->model->f16)
        hipLaunchKernelGGL(k_half_slice_to_f32<true>, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, (const bf16_t *)b.p, d_bs, b.C,
                           sl.co, sl.C, hw, B, out, b.blk, d_ps);
    else
        hipLaunchKernelGGL(k_half_slice_to_f32<false>, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, (const bf16_t *)b.p, d_bs, b.C,
                           sl.co, sl.C, hw, B, out, b.blk, d_ps);
    OBB_LAUNCH_CHECK(ctx);
    return OBB_OK;
}

}  // extern "C"
