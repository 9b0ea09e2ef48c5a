/*
 * obb_oracle.c -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * Plain-C CPU restatement of the geometry half of the reference hot path
 * (Abolfazlmsl/Oriented-Object-Detection, Detect_OBB.py).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Each function cites the reference lines it follows:
 *   ora_poly_iou            Detect_OBB.py:144-154  (compute_polygon_iou; Shapely 2.0.7 / GEOS semantics restated)
 *   ora_merge_detections    Detect_OBB.py:176-200  (stable sort desc + greedy same-class suppression)
 *   ora_consensus           Detect_OBB.py:347-423  (cross_scale_consensus_filter)
 *   ora_center_inside       Detect_OBB.py:156-174  (margin_for / box_center_from_xyxyxyxy / center_inside_safe_region)
 *   ora_strike_angle        Detect_OBB.py:135-142  (compute_angle_from_bbox)
 *   ora_tile_grid           Detect_OBB.py:210-223  (detect_symbols tile enumeration)
 *   ora_match_ap            Detect_OBB.py:512-565, 489-499 (compute_pr_for_class + compute_ap_from_pr)
 *   ora_probiou / ora_fast_nms   ultralytics==8.3.196 (NOT under /root/reference; requirements.txt:3):
 *                           utils/metrics.py batch_probiou + utils/ops.py nms_rotated, restated from the
 *                           published algorithm (arXiv:2106.06072) -- parity unpinned, see DESIGN.md.
 *
 * Pinning: Shapely/GEOS is an un-vendored dependency (requirements.txt:4), so the polygon IoU is pinned by
 * analytic known answers + the golden xlsx invariants (tests/test_oracle_*.py); the control-flow functions are
 * pinned bit-exactly against the reference's own functions (AST-extracted, tests/golden/make_golden.py).
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile).  -ffp-contract=off matters: the HIP
 * kernels are built the same way so that fp64 results are bit-identical.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ polygon IoU (fp64) */

typedef struct { double x, y; } pt_t;

static double cross3(pt_t a, pt_t b, pt_t c) { /* (b-a) x (c-a) */
    return (b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x);
}

static double shoelace2(const pt_t *p, int n) { /* twice the signed area */
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        int j = (i + 1 == n) ? 0 : i + 1;
        s += p[i].x * p[j].y - p[j].x * p[i].y;
    }
    return s;
}

static int sgn(double v) { return (v > 0.0) - (v < 0.0); }

static int on_seg(pt_t a, pt_t b, pt_t c) { /* c collinear with ab: inside its bbox? */
    return fmin(a.x, b.x) <= c.x && c.x <= fmax(a.x, b.x) && fmin(a.y, b.y) <= c.y && c.y <= fmax(a.y, b.y);
}

static int seg_intersect(pt_t a, pt_t b, pt_t c, pt_t d) { /* closed segments share a point */
    int o1 = sgn(cross3(a, b, c)), o2 = sgn(cross3(a, b, d));
    int o3 = sgn(cross3(c, d, a)), o4 = sgn(cross3(c, d, b));
    if (o1 != o2 && o3 != o4) return 1;
    if (o1 == 0 && on_seg(a, b, c)) return 1;
    if (o2 == 0 && on_seg(a, b, d)) return 1;
    if (o3 == 0 && on_seg(c, d, a)) return 1;
    if (o4 == 0 && on_seg(c, d, b)) return 1;
    return 0;
}

/* Shapely Polygon(...).is_valid restated for a 4-vertex ring: finite coords, non-zero area, the two pairs of
 * non-adjacent edges do not touch, and no zero-width spike at a vertex. */
static int quad_valid(const pt_t *p) {
    for (int i = 0; i < 4; ++i)
        if (!isfinite(p[i].x) || !isfinite(p[i].y)) return 0;
    if (shoelace2(p, 4) == 0.0) return 0;
    if (seg_intersect(p[0], p[1], p[2], p[3])) return 0;
    if (seg_intersect(p[1], p[2], p[3], p[0])) return 0;
    for (int i = 0; i < 4; ++i) {
        pt_t a = p[(i + 3) & 3], b = p[i], c = p[(i + 1) & 3];
        if (cross3(a, b, c) == 0.0) {
            double dot = (a.x - b.x) * (c.x - b.x) + (a.y - b.y) * (c.y - b.y);
            if (dot > 0.0) return 0; /* spike */
        }
    }
    return 1;
}

static int is_convex(const pt_t *p, int n) {
    int pos = 0, neg = 0;
    for (int i = 0; i < n; ++i) {
        double c = cross3(p[i], p[(i + 1) % n], p[(i + 2) % n]);
        if (c > 0.0) pos = 1;
        if (c < 0.0) neg = 1;
    }
    return !(pos && neg);
}

/* Sutherland-Hodgman: clip subject (ns <= 8 verts) by CONVEX clip polygon given in counter-clockwise order
 * (positive shoelace).  Returns |area| of the result. */
static double clip_area(const pt_t *subj, int ns, const pt_t *clip, int nc) {
    pt_t bufa[16], bufb[16];
    pt_t *in = bufa, *out = bufb;
    int n = ns;
    for (int i = 0; i < ns; ++i) in[i] = subj[i];
    for (int e = 0; e < nc && n > 0; ++e) {
        pt_t a = clip[e], b = clip[(e + 1 == nc) ? 0 : e + 1];
        int m = 0;
        pt_t s = in[n - 1];
        double ds = cross3(a, b, s);
        for (int i = 0; i < n; ++i) {
            pt_t p = in[i];
            double dp = cross3(a, b, p);
            if (dp >= 0.0) {
                if (ds < 0.0) {
                    double t = ds / (ds - dp);
                    out[m].x = s.x + (p.x - s.x) * t;
                    out[m].y = s.y + (p.y - s.y) * t;
                    ++m;
                }
                out[m++] = p;
            } else if (ds >= 0.0) {
                double t = ds / (ds - dp);
                out[m].x = s.x + (p.x - s.x) * t;
                out[m].y = s.y + (p.y - s.y) * t;
                ++m;
            }
            s = p;
            ds = dp;
        }
        pt_t *tmp = in; in = out; out = tmp;
        n = m;
    }
    if (n < 3) return 0.0;
    return fabs(shoelace2(in, n)) * 0.5;
}

static void make_ccw(const pt_t *p, int n, pt_t *o) {
    if (shoelace2(p, n) < 0.0) for (int i = 0; i < n; ++i) o[i] = p[n - 1 - i];
    else for (int i = 0; i < n; ++i) o[i] = p[i];
}

/* split a simple quad into two triangles along an interior diagonal */
static void quad_tris(const pt_t *q /*ccw*/, pt_t t[2][3]) {
    /* a reflex vertex (cross < 0 in ccw order) must be a diagonal end point */
    int r = -1;
    for (int i = 0; i < 4; ++i)
        if (cross3(q[(i + 3) & 3], q[i], q[(i + 1) & 3]) < 0.0) r = i;
    int s = (r < 0) ? 0 : r;
    t[0][0] = q[s]; t[0][1] = q[(s + 1) & 3]; t[0][2] = q[(s + 2) & 3];
    t[1][0] = q[s]; t[1][1] = q[(s + 2) & 3]; t[1][2] = q[(s + 3) & 3];
}

double ora_poly_iou(const double *b1, const double *b2) {
    pt_t p[4], q[4], pc[4], qc[4];
    for (int i = 0; i < 4; ++i) { p[i].x = b1[2 * i]; p[i].y = b1[2 * i + 1]; q[i].x = b2[2 * i]; q[i].y = b2[2 * i + 1]; }
    if (!quad_valid(p) || !quad_valid(q)) return 0.0;
    { /* GEOS short-circuits on strictly disjoint envelopes -> empty intersection, area exactly 0 */
        double px0 = fmin(fmin(p[0].x, p[1].x), fmin(p[2].x, p[3].x)), px1 = fmax(fmax(p[0].x, p[1].x), fmax(p[2].x, p[3].x));
        double py0 = fmin(fmin(p[0].y, p[1].y), fmin(p[2].y, p[3].y)), py1 = fmax(fmax(p[0].y, p[1].y), fmax(p[2].y, p[3].y));
        double qx0 = fmin(fmin(q[0].x, q[1].x), fmin(q[2].x, q[3].x)), qx1 = fmax(fmax(q[0].x, q[1].x), fmax(q[2].x, q[3].x));
        double qy0 = fmin(fmin(q[0].y, q[1].y), fmin(q[2].y, q[3].y)), qy1 = fmax(fmax(q[0].y, q[1].y), fmax(q[2].y, q[3].y));
        if (px1 < qx0 || qx1 < px0 || py1 < qy0 || qy1 < py0) return 0.0;
    }
    double a1 = fabs(shoelace2(p, 4)) * 0.5, a2 = fabs(shoelace2(q, 4)) * 0.5;
    make_ccw(p, 4, pc);
    make_ccw(q, 4, qc);
    double inter;
    if (is_convex(qc, 4)) inter = clip_area(pc, 4, qc, 4);
    else if (is_convex(pc, 4)) inter = clip_area(qc, 4, pc, 4);
    else {
        pt_t ta[2][3], tb[2][3];
        quad_tris(pc, ta);
        quad_tris(qc, tb);
        inter = 0.0;
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) inter += clip_area(ta[i], 3, tb[j], 3);
    }
    double uni = a1 + a2 - inter;
    return uni > 0.0 ? inter / uni : 0.0;
}

void ora_poly_iou_pairs(const double *a, const double *b, int64_t m, double *out) {
    for (int64_t i = 0; i < m; ++i) out[i] = ora_poly_iou(a + 8 * i, b + 8 * i);
}

/* ------------------------------------------------------------------ sorting helper: stable, descending */

static void stable_sort_desc(const double *key, int64_t n, int32_t *order) {
    /* bottom-up merge sort on indices; ties keep input order (Python list.sort(reverse=True) is stable) */
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) order[i] = (int32_t)i;
    for (int64_t w = 1; w < n; w *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int64_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) {
                if (key[order[j]] > key[order[i]]) tmp[k++] = order[j++];
                else tmp[k++] = order[i++];
            }
            while (i < mid) tmp[k++] = order[i++];
            while (j < hi) tmp[k++] = order[j++];
        }
        memcpy(order, tmp, sizeof(int32_t) * (size_t)n);
    }
    free(tmp);
}

void ora_sort_desc_stable(const double *key, int64_t n, int32_t *order) { stable_sort_desc(key, n, order); }

/* ------------------------------------------------------------------ merge_detections (Detect_OBB.py:176-200) */

/* boxes [n,8], cls [n], conf [n] in caller order.  order[n] receives the stable conf-descending permutation,
 * keep[n] (indexed by sorted position) the greedy decisions.  Returns number kept. */
int64_t ora_merge_detections(const double *boxes, const int32_t *cls, const double *conf, int64_t n, double thr,
                             int32_t *order, uint8_t *keep) {
    if (n <= 0) return 0;
    stable_sort_desc(conf, n, order);
    int64_t nk = 0;
    int32_t *kept = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        int32_t di = order[i];
        int k = 1;
        for (int64_t j = 0; j < nk; ++j) {
            int32_t dj = kept[j];
            if (cls[di] == cls[dj] && ora_poly_iou(boxes + 8 * di, boxes + 8 * dj) >= thr) { k = 0; break; }
        }
        keep[i] = (uint8_t)k;
        if (k) kept[nk++] = di;
    }
    free(kept);
    return nk;
}

/* ------------------------------------------------------------------ consensus (Detect_OBB.py:347-423) */

/* Two or more scales, concatenated in ascending-scale order: scale s owns rows [off[s], off[s+1]).
 * out_idx receives row indices (into the concatenated arrays) of kept detections, in reference output order.
 * Constants CONS_IOU_PARTNER / CONS_LOW / CONS_HIGH are arguments so tests can sweep them. */
int64_t ora_consensus(const double *boxes, const int32_t *cls, const double *conf, const int64_t *off, int32_t nscales,
                      double iou_partner, double cons_low, double cons_high, int32_t *out_idx) {
    int64_t total = off[nscales];
    if (nscales == 1) { /* :357-358 passthrough (no conf filter) */
        for (int64_t i = 0; i < total; ++i) out_idx[i] = (int32_t)i;
        return total;
    }
    /* :361-364 filter conf >= CONS_LOW, keep per-scale order */
    uint8_t *alive = (uint8_t *)calloc((size_t)(total > 0 ? total : 1), 1);
    uint8_t *visited = (uint8_t *)calloc((size_t)(total > 0 ? total : 1), 1);
    for (int64_t i = 0; i < total; ++i) alive[i] = conf[i] >= cons_low;
    int64_t nout = 0;
    for (int32_t s = 0; s < nscales; ++s) {
        for (int64_t i = off[s]; i < off[s + 1]; ++i) {
            if (!alive[i] || visited[i]) continue;
            int64_t best = -1;
            double best_conf = -1.0, best_iou = 0.0;
            for (int32_t t = 0; t < nscales; ++t) {
                if (t == s) continue;
                for (int64_t j = off[t]; j < off[t + 1]; ++j) {
                    if (!alive[j] || visited[j]) continue;
                    if (cls[j] != cls[i]) continue;
                    double iou = ora_poly_iou(boxes + 8 * i, boxes + 8 * j);
                    if (iou >= iou_partner) {
                        double cp = conf[j];
                        if (cp > best_conf || (cp == best_conf && iou > best_iou)) { best = j; best_conf = cp; best_iou = iou; }
                    }
                }
            }
            if (best < 0 || best_conf < cons_low) { /* :406-410 */
                if (conf[i] >= cons_high) out_idx[nout++] = (int32_t)i;
                visited[i] = 1;
                continue;
            }
            out_idx[nout++] = (conf[i] >= best_conf) ? (int32_t)i : (int32_t)best; /* :414-417 */
            visited[i] = 1;
            visited[best] = 1;
        }
    }
    free(alive);
    free(visited);
    return nout;
}

/* ------------------------------------------------------------------ Center-Hit metric  (Detect_OBB.py:609-648) */

/* Shapely `Polygon(pts).is_valid and Polygon(pts).contains(Point(x, y))` restated for a 4-vertex ring: the point must lie in the
 * INTERIOR (a point on the boundary is not contained).  Boundary: exact zero of the edge cross product inside the edge's box;
 * interior: non-zero winding number (cross-product signs only, no division).  GEOS itself is absent offline -> parity unpinned at
 * this boundary; pinned by known answers in tests/test_oracle_geometry.py. */
int ora_point_in_quad(const double *p8, double x, double y) {
    pt_t p[4], q = {x, y};
    for (int i = 0; i < 4; ++i) { p[i].x = p8[2 * i]; p[i].y = p8[2 * i + 1]; }
    if (!quad_valid(p) || !isfinite(x) || !isfinite(y)) return 0;
    int wn = 0;
    for (int i = 0; i < 4; ++i) {
        pt_t a = p[i], b = p[(i + 1) & 3];
        double c = cross3(a, b, q);
        if (c == 0.0 && on_seg(a, b, q)) return 0; /* on the boundary */
        if (a.y <= y) { if (b.y > y && c > 0.0) ++wn; }
        else if (b.y <= y && c < 0.0) --wn;
    }
    return wn != 0;
}

/* ------------------------------------------------------------------ border filter + strike angle */

int ora_center_inside(const double *p8, double x0, double y0, double w, double h, double margin) { /* :159-174 */
    double cx = (p8[0] + p8[2] + p8[4] + p8[6]) / 4.0;
    double cy = (p8[1] + p8[3] + p8[5] + p8[7]) / 4.0;
    double cxr = cx - x0, cyr = cy - y0;
    return (margin <= cxr && cxr <= (w - margin)) && (margin <= cyr && cyr <= (h - margin));
}

double ora_strike_angle(const double *p8) { /* :135-142 */
    double a = atan2(p8[6] - p8[0], p8[7] - p8[1]) * (180.0 / 3.141592653589793);
    return a > 0 ? 180 - a : fabs(a);
}

/* ------------------------------------------------------------------ tiler (Detect_OBB.py:210-223) */

/* rects: [max,4] int32 rows (x, y, x2, y2) in the reference's visiting order.  Returns tile count. */
int64_t ora_tile_grid(int32_t H, int32_t W, int32_t tile, int32_t overlap, int32_t *rects, int64_t max_tiles) {
    int32_t step = tile - overlap;
    if (step < 1) step = 1;
    int64_t n = 0;
    for (int32_t y = 0; y < H; y += step)
        for (int32_t x = 0; x < W; x += step) {
            int32_t y2 = y + tile < H ? y + tile : H, x2 = x + tile < W ? x + tile : W;
            if (y2 - y == 0 || x2 - x == 0) continue;
            if (n < max_tiles) { rects[4 * n] = x; rects[4 * n + 1] = y; rects[4 * n + 2] = x2; rects[4 * n + 3] = y2; }
            ++n;
        }
    return n;
}

/* ------------------------------------------------------------------ AP instrument (Detect_OBB.py:489-565) */

/* One class.  dets sorted here by score desc (stable).  det_img[i]/gt_img[j] = image ids.  Outputs tp flags in
 * sorted order and returns AP.  tp_out may be NULL. */
double ora_ap_for_class(const double *det_boxes, const double *det_score, const int32_t *det_img, int64_t nd,
                        const double *gt_boxes, const int32_t *gt_img, int64_t ng, double iou_thr,
                        uint8_t *tp_out, int64_t *totals /* TP,FP,FN */) {
    if (totals) { totals[0] = totals[1] = 0; totals[2] = ng; }
    if (ng == 0) { if (totals) totals[2] = 0; return 0.0; }
    if (nd == 0) return 0.0;
    int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (size_t)nd);
    stable_sort_desc(det_score, nd, order);
    uint8_t *matched = (uint8_t *)calloc((size_t)ng, 1);
    double *rec = (double *)malloc(sizeof(double) * (size_t)(nd + 2));
    double *pre = (double *)malloc(sizeof(double) * (size_t)(nd + 2));
    double tpc = 0.0, fpc = 0.0;
    for (int64_t i = 0; i < nd; ++i) {
        int32_t d = order[i];
        double best = 0.0;
        int64_t bj = -1;
        for (int64_t j = 0; j < ng; ++j) {
            if (gt_img[j] != det_img[d] || matched[j]) continue;
            double iou = ora_poly_iou(det_boxes + 8 * d, gt_boxes + 8 * j);
            if (iou > best) { best = iou; bj = j; }
        }
        int tp = (best >= iou_thr && bj >= 0);
        if (tp) { matched[bj] = 1; tpc += 1.0; } else fpc += 1.0;
        if (tp_out) tp_out[i] = (uint8_t)tp;
        rec[i + 1] = tpc / ((double)ng + 1e-9);
        pre[i + 1] = tpc / (tpc + fpc + 1e-9);
    }
    if (totals) { totals[0] = (int64_t)tpc; totals[1] = (int64_t)fpc; totals[2] = ng - (int64_t)tpc; }
    /* compute_ap_from_pr :489-499 */
    rec[0] = 0.0; pre[0] = 0.0; rec[nd + 1] = 1.0; pre[nd + 1] = 0.0;
    for (int64_t i = nd; i >= 0; --i) pre[i] = pre[i] > pre[i + 1] ? pre[i] : pre[i + 1];
    double ap = 0.0;
    for (int64_t i = 0; i <= nd; ++i)
        if (rec[i + 1] != rec[i]) ap += (rec[i + 1] - rec[i]) * pre[i + 1];
    free(order); free(matched); free(rec); free(pre);
    return ap;
}

/* ------------------------------------------------------------------ ProbIoU + Fast-NMS (fp32; ultralytics 8.3.196) */

static void cov_abc(float w, float h, float r, float *A, float *B, float *C) {
    float a = w * w / 12.0f, b = h * h / 12.0f;
    float c = cosf(r), s = sinf(r);
    float c2 = c * c, s2 = s * s;
    *A = a * c2 + b * s2;
    *B = a * s2 + b * c2;
    *C = (a - b) * c * s;
}

float ora_probiou(const float *o1, const float *o2) { /* xywhr each */
    const float eps = 1e-7f;
    float a1, b1, c1, a2, b2, c2;
    cov_abc(o1[2], o1[3], o1[4], &a1, &b1, &c1);
    cov_abc(o2[2], o2[3], o2[4], &a2, &b2, &c2);
    float x1 = o1[0], y1 = o1[1], x2 = o2[0], y2 = o2[1];
    float sa = a1 + a2, sb = b1 + b2, sc = c1 + c2;
    float den = sa * sb - sc * sc;
    float t1 = ((sa * ((y1 - y2) * (y1 - y2)) + sb * ((x1 - x2) * (x1 - x2))) / (den + eps)) * 0.25f;
    float t2 = ((sc * (x2 - x1) * (y1 - y2)) / (den + eps)) * 0.5f;
    float d1 = a1 * b1 - c1 * c1, d2 = a2 * b2 - c2 * c2;
    d1 = d1 > 0.0f ? d1 : 0.0f;
    d2 = d2 > 0.0f ? d2 : 0.0f;
    float t3 = logf(den / (4.0f * sqrtf(d1 * d2) + eps) + eps) * 0.5f;
    float bd = t1 + t2 + t3;
    bd = bd < eps ? eps : (bd > 100.0f ? 100.0f : bd);
    float hd = sqrtf(1.0f - expf(-bd) + eps);
    return 1.0f - hd;
}

/* boxes [n,5] xywhr (class offset already applied), scores [n].  order = stable desc argsort; keep flags in
 * sorted order: keep j iff no i<j with probiou(i,j) >= thr (Fast-NMS, not greedy). */
int64_t ora_fast_nms(const float *boxes, const float *scores, int64_t n, float thr, int32_t *order, uint8_t *keep) {
    double *k = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    for (int64_t i = 0; i < n; ++i) k[i] = scores[i];
    stable_sort_desc(k, n, order);
    free(k);
    int64_t nk = 0;
    for (int64_t j = 0; j < n; ++j) {
        int kj = 1;
        for (int64_t i = 0; i < j; ++i)
            if (ora_probiou(boxes + 5 * order[i], boxes + 5 * order[j]) >= thr) { kj = 0; break; }
        keep[j] = (uint8_t)kj;
        nk += kj;
    }
    return nk;
}
