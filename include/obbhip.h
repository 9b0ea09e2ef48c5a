/*
 * obbhip.h -- C-ABI of libobbhip.so: the MI355X (gfx950) drop-in for the Detect_OBB.py sliding-window
 * oriented-box inference path of Abolfazlmsl/Oriented-Object-Detection.
 *
 * The reference has no FFI: its seams are plain Python callables.  Each entry point below
 * names the reference interface it replaces (file:line in /root/reference).  Rules of the boundary:
 *   - plain pointers and sizes only; every DATA pointer is a DEVICE pointer unless the name ends in _host;
 *   - the caller allocates every buffer; the library owns only the context (weights, workspaces, hipGraphs);
 *   - every launch goes to the caller's stream (hipStream_t passed as void*); no hidden synchronisation except
 *     where a function is documented "(synchronises)";
 *   - int status return (0 = ok, <0 = error) + obb_last_error(); nothing throws across the boundary;
 *   - invalid / degenerate polygons yield IoU 0.0, not an error (Detect_OBB.py:150-151).
 *
 * Record layout shared by the detection-level calls ("det arrays"), all SoA:
 *   boxes  double[n*8]   x1,y1,...,x4,y4  global pixel coordinates   (Detect_OBB.py:256-260)
 *   cls    int32[n]                                                   (Detect_OBB.py:261)
 *   conf   double[n]     float32 confidences widened to double        (Detect_OBB.py:231)
 */
#ifndef OBBHIP_H
#define OBBHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OBB_OK 0
#define OBB_ERR_INVALID (-1) /* bad argument                          */
#define OBB_ERR_HIP (-2)     /* a HIP runtime call failed             */
#define OBB_ERR_STATE (-3)   /* e.g. forward before weights are loaded */
#define OBB_ERR_FORMAT (-4)  /* malformed weight blob                 */

typedef struct obb_ctx obb_ctx;
typedef void *obb_stream_t; /* hipStream_t */

int obb_version(void);
/* One context per device / rank.  (synchronises) */
int obb_ctx_create(int device, obb_ctx **out);
int obb_ctx_destroy(obb_ctx *ctx);
/* Last error text of this context (or of the calling thread when ctx == NULL). */
const char *obb_last_error(const obb_ctx *ctx);

/* Engine knobs (no reference counterpart).
 *   "precision"   16 = fp16 activation/weight storage (default; what Ultralytics' half=True inference uses), 1016 = bf16 storage; fp32
 *                 accumulation either way.  32 = fp32 arithmetic end to end (what the reference computes: Detect_OBB.py:79-83 calls the
 *                 model with half=False): fp32 weights and activations, convolutions on the exact-f32 matrix instruction (a k-ordered
 *                 fmaf chain), one kernel per layer, no fusion; error vs a double-precision evaluation of the same weights ~1e-5 on a
 *                 head logit, the same as torch's own fp32 forward (tests/test_gpu_fp32.py).  Applies to the next obb_model_load.
 *   "model_slot"  index (0..63) of the model that obb_model_load / obb_forward / obb_decode* address (several models may live in one
 *                 context, e.g. the 128 px and 416 px checkpoints of the dual-scale config).
 *   fused forms, 1 = on (default), 0 = the separate launches they replace; each applies to the next obb_model_load and exists so that
 *   parity tests can compare both forms on identical inputs:
 *     "tail"      last 1x1 conv of each head branch behind its producer; 0 also turns every other intermediate-swallowing fusion off
 *                 (bneck, hmerge, c3kimg, dwpw, tail16): every layer's output is then observable through obb_debug_activation
 *     "tail16"    cv1 of the C3k2 blocks 2 / 4 inside the preceding stride-2 conv      "bneck"     Bottleneck row stripes (104 / 52 levels)
 *     "bneck_cv2" closing 1x1 of those C3k2 blocks behind the Bottleneck               "c3kimg"    inner C3k of the stride-32 level per image
 *     "dwpw"      depthwise 3x3 -> 1x1 stripes of the class branch                     "upfold"    Upsample + Concat read in place by the 1x1
 *     "stem"      model.0 as row stripes on the uint8 tile                             "hmerge"    sibling convs on one input as one launch
 *     "sppf_fuse" the three SPPF pools in one launch                                   "attn_mfma" C2PSA attention on the matrix cores
 *     "front"     model.0 + model.1 + model.2.cv1 as one launch (tile sides % 52 == 0)  "pair"      64 -> 64-cout 3x3 convs on k_conv3_pair
 *     "nitile"    (16-bit modes) several whole images per tile on the 4 x 4 / 2 x 2 maps of small tiles (3x3 convs with 64-cout groups)
 *     "nc2"       (fp32) two cout fragments per wave (32 couts x <= 64 pixels) in the conv kernel where the plan allows: a third fewer LDS reads
 *     "xtile"     (fp32) conv workgroups stay resident and walk several tiles, the next tile's first stage fetched under this tile's last k loop
 *     "blk32"     (fp32) the tensors read by 3x3 convs in 8- / 16-channel stages stored as 8-channel blocks per image
 *     "c3k2f"     (fp32) Bottleneck + closing 1x1 of the 104 x 104 C3k2 block in one launch (k_c3k2_f32)
 *     "pw32"      (fp32) 1x1 layers with >= 64 input channels on k_pw_f32: weights resident in LDS, activations straight from global memory
 *   issue of a forward (take effect at the next obb_forward): "graph" 1 = capture / replay hipGraphs (default), "fwd_split" 0..4
 *   concurrent sub-batch chains (default 0 = 2), "microbatch" 416 x 416 tiles per round (default and maximum 1024; smaller tiles
 *   get proportionally more per round, at most 16 384: 128 px -> 11 264). */
int obb_set_option(obb_ctx *ctx, const char *key, int64_t value);

/* ------------------------------------------------------------------ S2: compute_polygon_iou  (Detect_OBB.py:144-154) */
/* out[i] = IoU(a[i], b[i]); a, b: double[m*8].  Replaces Shapely Polygon/is_valid/intersection/area per pair. */
int obb_poly_iou_pairs(obb_ctx *ctx, const double *a, const double *b, int64_t m, double *out, obb_stream_t s);
/* out[i*nb+j] = IoU(a[i], b[j]) if cls_a[i]==cls_b[j] (or either cls pointer is NULL), else 0.
 * The O(N1*N2) loops of Detect_OBB.py:387-403 and :541-547. */
int obb_poly_iou_matrix(obb_ctx *ctx, const double *a, const int32_t *cls_a, int64_t na, const double *b,
                        const int32_t *cls_b, int64_t nb, double *out, obb_stream_t s);

/* Center-Hit metric (Detect_OBB.py:609-648): out[i*nq + j] = 1 iff point i (a detection centre, :159-165) lies strictly inside the
 * valid quad j  (`poly.is_valid and poly.contains(Point(cx, cy))`, :631-634) and cls_p[i] == cls_q[j] (when both are given). */
int obb_points_in_quads(obb_ctx *ctx, const double *pts, const int32_t *cls_p, int64_t np, const double *quads, const int32_t *cls_q,
                        int64_t nq, uint8_t *out, obb_stream_t s);

/* 4-channel network input (RGB + distance-transform edge channel) for every crop of a batch: `build_multich(crop_bgr, out_channels=4)`,
 * Detect_OBB.py:87-133 (called per tile at :77).  bgr: uint8[B,h,w,3], out4: uint8[B,h,w,4] = [R, G, B, dt_edge].  The OpenCV steps
 * are restated from their documented algorithms (cv2 is absent offline: parity with cv2 itself is unpinned). */
int obb_build_multich(obb_ctx *ctx, const uint8_t *bgr, int32_t B, int32_t h, int32_t w, uint8_t *out4, obb_stream_t s);

/* ------------------------------------------------------------------ S3: merge_detections  (Detect_OBB.py:176-200) */
/* order[n] = stable descending argsort of key (Python list.sort(key=conf, reverse=True), Detect_OBB.py:183). */
int obb_sort_desc_stable(obb_ctx *ctx, const double *key, int64_t n, int32_t *order, obb_stream_t s);
/* Pair-suppression bit matrix over boxes already in sorted order, column-block-major: bit (j % 64) of word
 * mask[(j / 64) * n + i] is set iff j > i, cls equal and IoU(i,j) >= thr  (Detect_OBB.py:193).
 * mask: uint64[ceil(n/64) * n]. */
int obb_nms_mask(obb_ctx *ctx, const double *boxes_sorted, const int32_t *cls_sorted, int64_t n, double thr,
                 uint64_t *mask, obb_stream_t s);
/* Greedy scan of that matrix (Detect_OBB.py:186-198): keep[i] in sorted order, *n_keep = number kept. */
int obb_nms_reduce(obb_ctx *ctx, const uint64_t *mask, int64_t n, uint8_t *keep, int32_t *n_keep, obb_stream_t s);
/* The whole function: sort + suppression pairs + greedy scan.  order[n] (sorted position -> input index), keep[n] (sorted order).
 * No host read and no synchronisation at any n: the choice between the sparse pair list and the matrix-free dense scan (pair-list
 * overflow, thr <= 0) is taken on the device, so the call can be captured into a hipGraph once its workspaces exist (call it once
 * eagerly at the largest n first).  n <= 1.2 M rows. */
int obb_merge_detections(obb_ctx *ctx, const double *boxes, const int32_t *cls, const double *conf, int64_t n,
                         double thr, int32_t *order, uint8_t *keep, int32_t *n_keep, obb_stream_t s);
/* Batched form for the per-tile call at Detect_OBB.py:264: nseg independent segments, segment k owning rows
 * [seg_off[k], seg_off[k+1]) (device int32[nseg+1]); order holds GLOBAL row indices per sorted position.  Segments of up to 512 rows run
 * in one launch; longer ones are detected on the host and taken through the dense path of obb_merge_detections (synchronises then). */
int obb_merge_segments(obb_ctx *ctx, const double *boxes, const int32_t *cls, const double *conf,
                       const int32_t *seg_off, int32_t nseg, int64_t n, double thr, int32_t *order, uint8_t *keep,
                       obb_stream_t s);

/* ------------------------------------------------------------------ S4: cross_scale_consensus_filter  (Detect_OBB.py:347-423) */
/* Rows of all scales concatenated in ascending-scale order; scale k owns [off_host[k], off_host[k+1]).
 * out_idx[<= n] receives kept row indices in the reference's output order, *n_out their count.
 * Constants: CONS_IOU_PARTNER / CONS_LOW / CONS_HIGH (Detect_OBB.py:349-351). */
int obb_consensus(obb_ctx *ctx, const double *boxes, const int32_t *cls, const double *conf, const int64_t *off_host,
                  int32_t nscales, double iou_partner, double cons_low, double cons_high, int32_t *out_idx,
                  int32_t *n_out, obb_stream_t s);

/* ------------------------------------------------------------------ S5: detect_symbols  (Detect_OBB.py:202-266) */
/* Tile rectangles (x, y, x2, y2) in the reference's visiting order (Detect_OBB.py:211-223).  Host-only helper:
 * rects_host int32[max_tiles*4]; *n_tiles receives the full count even when it exceeds max_tiles. */
int obb_tile_grid(int32_t H, int32_t W, int32_t tile, int32_t overlap, int32_t *rects_host, int64_t max_tiles,
                  int64_t *n_tiles);
/* Per-detection work of the tile loop (Detect_OBB.py:229-262): local float32 corners + integer tile offset in
 * float64 (exact), inclusive border filter with `margin`, strike angle for class `strike_cls` (else 0.0).
 * local_pts float[n*8], det_tile int32[n] (index into rects), rects int32[ntiles*4] (device). */
int obb_tile_postprocess(obb_ctx *ctx, const float *local_pts, const int32_t *cls, const int32_t *det_tile, int64_t n,
                         const int32_t *rects, int32_t ntiles, int32_t margin, int32_t strike_cls, double *gboxes,
                         double *angle, uint8_t *inside, obb_stream_t s);
/* The whole stretch between the NMS output and the exchange records of a batch of tiles, on the device with no host-visible count in
 * between (Detect_OBB.py:228-264 per tile): Results.obb construction of each NMS row (regularize_rboxes, scale_boxes with the tile's
 * letterbox lb[t] = (gain, pad_x, pad_y) or NULL, xywhr2xyxyxyxy), the per-detection body (:229-262: global corners, border filter with
 * `margin`), the per-tile merge_detections (:264, threshold iou_thr) and the compaction of the survivors into 48-byte records
 * {tile id, class, conf (float32 bits), 0, 8 x local corner (float32 bits)} in tile order, merge (confidence) order inside a tile.
 * det float[B*max_det*7] + count int32[B]: the outputs of obb_decode_nms; tile_ids int32[B]: index of each tile into rects int32[*][4];
 * records int32[B*max_det*12] (capacity), tile_off int32[B+1] (first record of every tile), *n_records (device) = tile_off[B].
 * max_det <= 512.  Bit-identical to obb_results -> obb_tile_postprocess -> obb_merge_segments on the same rows (shared device code). */
int obb_tile_survivors(obb_ctx *ctx, const float *det, const int32_t *count, int32_t B, int32_t max_det, const float *lb,
                       const int32_t *tile_ids, const int32_t *rects, int32_t margin, int32_t strike_cls, double iou_thr,
                       int32_t *records, int32_t *tile_off, int32_t *n_records, obb_stream_t s);
/* Ordered compaction of a merge result (the list comprehension that closes merge_detections, Detect_OBB.py:198-200): out row r =
 * input row order[i] of the r-th sorted position i with keep[i] != 0; *n_out (device) = number of rows written.  angle / out_angle
 * may be NULL.  Output buffers hold n rows. */
int obb_select_kept(obb_ctx *ctx, const int32_t *order, const uint8_t *keep, int64_t n, const double *boxes, const int32_t *cls,
                    const double *conf, const double *angle, double *out_boxes, int32_t *out_cls, double *out_conf, double *out_angle,
                    int32_t *n_out, obb_stream_t s);
/* Consumer side of the 48-byte records: global float64 corners (float32 local corner + integer tile offset: exact), class, confidence
 * widened to float64 and the strike angle (Detect_OBB.py:229-234, 251-254), straight from the packed rows records int32[n][12]. */
int obb_records_to_dets(obb_ctx *ctx, const int32_t *records, int64_t n, const int32_t *rects, int32_t strike_cls, double *gboxes,
                        int32_t *cls, double *conf, double *angle, obb_stream_t s);
/* Multi-GPU exchange: the valid rows of a fixed-capacity all-gather buffer recv int32[world][capacity+1][12]
 * (row 0 of every rank's block carries its record count in column 0) -> dense out int32[<= world*capacity][12] in rank order;
 * counts int32[world+1] (device): every rank's count as sent (a count above `capacity` tells the caller to repeat the exchange with a
 * larger one) and, last, the number of rows written. */
int obb_gather_compact(obb_ctx *ctx, const int32_t *recv, int32_t world, int32_t capacity, int32_t *out, int32_t *counts, obb_stream_t s);
/* Gather all full-size tiles of one image into an NHWC uint8 batch (Detect_OBB.py:218-220 crop, vectorised).
 * image uint8[H*W*C]; rects device int32[ntiles*4] must all be tile x tile. */
int obb_gather_tiles(obb_ctx *ctx, const uint8_t *image, int32_t H, int32_t W, int32_t C, const int32_t *rects,
                     int32_t ntiles, int32_t tile, uint8_t *tiles_out, obb_stream_t s);
/* Letterbox one partial crop to (out_h, out_w) the way the Ultralytics predictor does for a single image
 * (LetterBox auto=True stride 32, pad value 114, bilinear resize only when the gain != 1; SURVEY.md Appendix A2;
 * call site Detect_OBB.py:81-83).  Returns gain and left/top pad through *_host pointers. */
int obb_letterbox(obb_ctx *ctx, const uint8_t *image, int32_t H, int32_t W, int32_t C, int32_t x, int32_t y, int32_t x2,
                  int32_t y2, int32_t imgsz, uint8_t *out, int32_t out_h, int32_t out_w, obb_stream_t s);

/* ------------------------------------------------------------------ training step, slice 2 */
/* RotatedTaskAlignedAssigner.forward of ultralytics==8.3.196 (utils/tal.py; reached from `model.train(...)`, Train_OBB.py:796-841, through
 * v8OBBLoss with topk = 10, alpha = 0.5, beta = 6.0): pd_scores float[bs][na][nc] (sigmoid scores), pd_bboxes float[bs][na][5] (x, y, w,
 * h, theta in pixels), anc_points float[na][2], gt_labels int32[bs][n_max], gt_bboxes float[bs][n_max][5], mask_gt uint8[bs][n_max] ->
 * target_labels int32[bs][na], target_bboxes float[bs][na][5], target_scores float[bs][na][nc], fg_mask uint8[bs][na],
 * target_gt_idx int32[bs][na].  na <= 10240.  All device pointers. */
int obb_rotated_tal_assign(obb_ctx *ctx, const float *pd_scores, const float *pd_bboxes, const float *anc_points, const int32_t *gt_labels,
                           const float *gt_bboxes, const uint8_t *mask_gt, int32_t bs, int32_t na, int32_t nc, int32_t n_max, int32_t topk,
                           float alpha, float beta, int32_t *target_labels, float *target_bboxes, float *target_scores, uint8_t *fg_mask,
                           int32_t *target_gt_idx, obb_stream_t s);

/* ------------------------------------------------------------------ training step, slice 3 */
/* Backward of a stride-1 convolution with `same` padding (k = 1 or 3), bf16 NHWC tensors, fp32 accumulation (Train_OBB.py:796-841: bf16
 * autocast over fp32 master weights).  dgrad: dy bf16[B][H][W][cout] (device), w fp32[cout][cin][k][k] (HOST: repacked per call into
 * the MFMA fragment order of the forward kernel; (synchronises)) -> dx bf16[B][H][W][cin].  wgrad: x bf16[B][H][W][cin], dy -> dw fp32[cout][cin][k][k]
 * (device), cin and cout multiples of 64; deterministic (fixed summation order). */
int obb_conv_dgrad_bf16(obb_ctx *ctx, const uint16_t *dy, const float *w_oihw_host, int32_t B, int32_t H, int32_t W, int32_t cin,
                        int32_t cout, int32_t ks, uint16_t *dx, obb_stream_t s);
int obb_conv_wgrad_bf16(obb_ctx *ctx, const uint16_t *x, const uint16_t *dy, int32_t B, int32_t H, int32_t W, int32_t cin, int32_t cout,
                        int32_t ks, float *dw, obb_stream_t s);
/* Assembled-chain pieces (round 4; Train_OBB.py:796-841 -> loss.backward() under bf16 autocast over fp32 master weights):
 * obb_conv_pack_bf16: fp32 OIHW master weights ON THE DEVICE -> the bf16 MFMA fragment order of the forward kernel (n_elems from
 *   obb_conv_packed_elems for the same shape); dgrad_form 1 packs the flipped / channel-transposed weights, so that
 *   obb_conv_fwd_bf16(dy, packed_dgrad, NULL, ..., cin = cout, cout = cin) IS the input gradient -- no host repack, no synchronisation.
 * obb_conv_fwd_bf16: y = conv(x) + bias, stride 1, `same` padding, NO activation (the pre-activation is kept for the backward), bf16
 *   NHWC in / out, fp32 accumulation, one bf16 rounding.
 * obb_silu_bf16 / obb_silu_bwd_bf16: a = z sigmoid(z);  dz = da * s (1 + z (1 - s)).   obb_bias_grad_bf16: db[c] = sum_pixels dy[pixel][c] (fp32). */
int obb_conv_packed_elems(obb_ctx *ctx, int32_t cout, int32_t cin, int32_t ks, int32_t H, int32_t W, int32_t dgrad_form, int64_t *n_elems);
int obb_conv_pack_bf16(obb_ctx *ctx, const float *w_oihw, int32_t cout, int32_t cin, int32_t ks, int32_t H, int32_t W, int32_t dgrad_form,
                       uint16_t *packed, obb_stream_t s);
int obb_conv_fwd_bf16(obb_ctx *ctx, const uint16_t *x, const uint16_t *packed_w, const float *bias, int32_t B, int32_t H, int32_t W, int32_t cin,
                      int32_t cout, int32_t ks, uint16_t *y, obb_stream_t s);
int obb_silu_bf16(obb_ctx *ctx, const uint16_t *z, uint16_t *a, int64_t n, obb_stream_t s);
int obb_silu_bwd_bf16(obb_ctx *ctx, const uint16_t *z, const uint16_t *da, uint16_t *dz, int64_t n, obb_stream_t s);
int obb_bias_grad_bf16(obb_ctx *ctx, const uint16_t *dy, int64_t npix, int32_t cout, float *db, obb_stream_t s);

/* ------------------------------------------------------------------ S1: model(...) -> results[0].obb  (Detect_OBB.py:26,81-83,228-231) */
/* Weight blob ("OBBW" format, produced by the Python side from BN-folded conv weights) for a
 * YOLO11-OBB graph (ultralytics==8.3.196 yolo11-obb.yaml).  Host pointer.  (synchronises) */
int obb_model_load(obb_ctx *ctx, const void *blob_host, size_t bytes);
/* Frees the model of `slot` (weights, activation slabs, captured graphs); the slot can be loaded again.  (synchronises) */
int obb_model_unload(obb_ctx *ctx, int32_t slot);
/* nc, input channels, number of anchors A for an (h, w) input, number of weight records. */
int obb_model_info(const obb_ctx *ctx, int32_t h, int32_t w, int32_t *nc, int32_t *ch, int32_t *anchors, int32_t *nconv);
/* OBBModel forward on B letterboxed inputs: tiles uint8[B*h*w*ch] (NHWC; BGR for ch==3 exactly as the reference
 * passes crops, Detect_OBB.py:93) -> raw head float[B*A*NO] with NO = 64+nc+1 rounded up to a multiple of 4 (per anchor:
 * 4x16 DFL logits, nc class logits, 1 angle logit, zero padding).  Preprocess (BGR->RGB, /255; Appendix A2) is fused
 * into the first convolution. */
int obb_forward(obb_ctx *ctx, const uint8_t *tiles, int32_t B, int32_t h, int32_t w, float *head, obb_stream_t s);
/* The same forward, which also leaves cmax float[B][A] = the largest class logit of every anchor (a dense tensor: the confidence gate of
 * obb_decode_nms_gate then reads 4 bytes per anchor instead of a 48-byte piece of every 320-byte head row).  Written by the fused class
 * tails of the head where the plan has them, by one extra pass over the head rows otherwise.  cmax NULL = obb_forward. */
int obb_forward_gate(obb_ctx *ctx, const uint8_t *tiles, int32_t B, int32_t h, int32_t w, float *head, float *cmax, obb_stream_t s);
/* Debug: text dump of the lowered forward for an (h, w) input -- one line per kernel launch (layer name, tiling,
 * grid, LDS bytes, MACs).  buf_host may be NULL to query *needed. */
int obb_debug_plan(obb_ctx *ctx, int32_t h, int32_t w, char *buf_host, int64_t buf_bytes, int64_t *needed);
/* Debug/parity tap: copy the activation called `name` (a conv's state-dict path such as "model.2.cv1", or a
 * layer output "x0".."x22") of the last forward to out as dense float[B*H*W*C] (bf16 widened).  out may be NULL to
 * query *n_elems / shape only. */
int obb_debug_activation(obb_ctx *ctx, int32_t h, int32_t w, int32_t B, const char *name, float *out, int64_t max_elems,
                         int64_t *n_elems, int32_t *shape_hwc_host, obb_stream_t s);
/* Decode (DFL, dist2rbox, angle, sigmoid; Appendix A4) + conf filter + class-offset ProbIoU Fast-NMS + max_det.
 * out float[B*max_det*7] rows (x, y, w, h, conf, cls, theta) in score order; count int32[B]. */
int obb_decode_nms(obb_ctx *ctx, const float *head, int32_t B, int32_t h, int32_t w, float conf_thres, float iou_thres,
                   int32_t max_det, float *out, int32_t *count, obb_stream_t s);
/* obb_decode_nms with the candidate gate read from cmax (obb_forward_gate's output for the SAME head tensor; NULL = obb_decode_nms): the
 * exact class scores are still taken from the head rows of the anchors that pass, so the result is identical. */
int obb_decode_nms_gate(obb_ctx *ctx, const float *head, const float *cmax, int32_t B, int32_t h, int32_t w, float conf_thres, float iou_thres,
                        int32_t max_det, float *det, int32_t *count, obb_stream_t s);
/* The same function in its full form (decode every anchor, then NMS sized for all of them): an independent implementation kept for the
 * parity tests, which require obb_decode_nms (candidate-first: conf filter on the class logits, decode + NMS of the survivors only) to
 * return the same rows bit for bit. */
int obb_decode_nms_full(obb_ctx *ctx, const float *head, int32_t B, int32_t h, int32_t w, float conf_thres, float iou_thres,
                        int32_t max_det, float *out, int32_t *count, obb_stream_t s);
/* Pieces of the above for parity tests: decoded predictions float[B*A*(4+nc+1)] (x,y,w,h, cls scores, theta). */
int obb_decode(obb_ctx *ctx, const float *head, int32_t B, int32_t h, int32_t w, float *pred, obb_stream_t s);
/* ProbIoU Fast-NMS on one candidate list: boxes float[n*5] (x,y,w,h,theta; class offset applied), scores float[n]. */
int obb_probiou_nms(obb_ctx *ctx, const float *boxes, const float *scores, int64_t n, float iou_thres, int32_t *order,
                    uint8_t *keep, obb_stream_t s);
/* Result construction (regularize_rboxes, scale_boxes(xywh=True), xywhr2xyxyxyxy; Appendix A6; consumed at
 * Detect_OBB.py:229-231).  det float[n*7] rows as written by obb_decode_nms; per-row letterbox params
 * lb float[n*3] = (gain, pad_x, pad_y) or NULL for identity.  xywhr float[n*5], pts float[n*8]. */
int obb_results(obb_ctx *ctx, const float *det, const float *lb, int64_t n, float *xywhr, float *pts, obb_stream_t s);

/* ------------------------------------------------------------------ f1 (first slice): the ProbIoU rotated-box loss of the training step */
/* `model.train(...)` (Train_OBB.py:796-841) -> ultralytics v8OBBLoss -> RotatedBboxLoss: loss_iou = sum((1 - probiou(pred, target)) * weight)
 * / target_scores_sum over the n matched (prediction, target) pairs, forward and backward in one pass.  pred, target float[n*5] rows
 * (x, y, w, h, theta); weight float[n] (NULL = 1); *loss (device float) receives the scalar, grad_pred float[n*5] = d loss / d pred. */
int obb_probiou_loss(obb_ctx *ctx, const float *pred, const float *target, const float *weight, int64_t n, float target_scores_sum,
                     float *loss, float *grad_pred, obb_stream_t s);

/* ------------------------------------------------------------------ f4: the training-set tiler's label assignment (Train_OBB.py:87-112) */
/* For every (tile, label) pair: mask uint8[ntiles*n] = 1 when the label goes into the tile's label file (midpoint of corners 1 and 4
 * inside the half-open tile AND >= min_fraction (object_boundary_threshold, :33) of its axis-aligned box inside), and then out
 * double[(t*n + l)*8] = its corners shifted to the tile, clipped to [0, tile] and divided by the tile size.  labels double[n*8] pixel
 * corners of one image; rects device int32[ntiles*4] (x, y, x2, y2) square tiles (the tiler only keeps full tiles, :83-84). */
int obb_tile_labels(obb_ctx *ctx, const double *labels, int64_t n, const int32_t *rects, int32_t ntiles, double min_fraction, uint8_t *mask,
                    double *out, obb_stream_t s);

/* f1 (second slice): the other two terms of v8OBBLoss, forward and backward in one pass each.
 * DFL (ultralytics DFLoss inside RotatedBboxLoss): pred_dist float[n*4*reg_max] logits of the n matched anchors' four sides, target_ltrb
 * float[n*4] = bbox2dist(anchor, xyxy(target), reg_max - 1) (clamped to [0, reg_max - 1.01] here as the reference does), weight float[n]
 * (NULL = 1); *loss = sum_i mean_side(CE(l) wl + CE(r) wr) weight_i / target_scores_sum; grad_pred float[n*4*reg_max].  reg_max must be 16. */
int obb_dfl_loss(obb_ctx *ctx, const float *pred_dist, const float *target_ltrb, const float *weight, int64_t n, int32_t reg_max,
                 float target_scores_sum, float *loss, float *grad_pred, obb_stream_t s);
/* classification term: BCEWithLogitsLoss(reduction="none")(logits, targets).sum() / target_scores_sum over n = batch * anchors * classes
 * elements; grad_logits float[n] = (sigmoid(logit) - target) / target_scores_sum. */
int obb_bce_loss(obb_ctx *ctx, const float *logits, const float *targets, int64_t n, float target_scores_sum, float *loss, float *grad_logits,
                 obb_stream_t s);

/* f1 (optimiser step): the update behind `model.train(...)` (Train_OBB.py:796-841; ultralytics `build_optimizer`: torch.optim.SGD with
 * nesterov momentum or torch.optim.AdamW, three parameter groups).  A group = flat fp32 device buffers of n elements, 16-byte aligned.
 * obb_sgd_step: d = grad + weight_decay * param; buf = first_step ? d : momentum * buf + d; d = nesterov ? d + momentum * buf : buf;
 * param -= lr * d (momentum 0: no buffer touched).  obb_adamw_step: torch.optim.AdamW's single-tensor order with step counted from 1. */
int obb_sgd_step(obb_ctx *ctx, float *param, const float *grad, float *momentum_buf, int64_t n, float lr, float momentum, float weight_decay,
                 int32_t nesterov, int32_t first_step, obb_stream_t s);
int obb_adamw_step(obb_ctx *ctx, float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, int64_t step, float lr, float beta1,
                   float beta2, float eps, float weight_decay, obb_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* OBBHIP_H */
