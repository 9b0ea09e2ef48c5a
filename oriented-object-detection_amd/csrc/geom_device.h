// fp64 polygon geometry for the post-processing kernels (device side).
//
// Replaces the Shapely/GEOS calls of Detect_OBB.py:144-154 (Polygon, is_valid, intersection().area, .area).
// Every expression is evaluated in the same order as the CPU oracle so that results are bit-identical when both
// are built with -ffp-contract=off: validity (finite, non-zero area, simple, no spike) -> envelope test (GEOS
// short-circuits on disjoint envelopes) -> Sutherland-Hodgman clip against the convex operand -> shoelace area.
#pragma once
#include <hip/hip_runtime.h>

namespace obb {

struct P2 { double x, y; };

__device__ __forceinline__ double cross3(P2 a, P2 b, P2 c) { return (b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x); }

template <int MAXN>
__device__ __forceinline__ double shoelace2(const P2 *p, int n) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        int j = (i + 1 == n) ? 0 : i + 1;
        s += p[i].x * p[j].y - p[j].x * p[i].y;
    }
    return s;
}

__device__ __forceinline__ int sgn(double v) { return (v > 0.0) - (v < 0.0); }

__device__ __forceinline__ bool on_seg(P2 a, P2 b, P2 c) {
    return fmin(a.x, b.x) <= c.x && c.x <= fmax(a.x, b.x) && fmin(a.y, b.y) <= c.y && c.y <= fmax(a.y, b.y);
}

__device__ __forceinline__ bool seg_intersect(P2 a, P2 b, P2 c, P2 d) {
    int o1 = sgn(cross3(a, b, c)), o2 = sgn(cross3(a, b, d));
    int o3 = sgn(cross3(c, d, a)), o4 = sgn(cross3(c, d, b));
    if (o1 != o2 && o3 != o4) return true;
    if (o1 == 0 && on_seg(a, b, c)) return true;
    if (o2 == 0 && on_seg(a, b, d)) return true;
    if (o3 == 0 && on_seg(c, d, a)) return true;
    if (o4 == 0 && on_seg(c, d, b)) return true;
    return false;
}

__device__ __forceinline__ bool quad_valid(const P2 *p) {
    for (int i = 0; i < 4; ++i)
        if (!isfinite(p[i].x) || !isfinite(p[i].y)) return false;
    if (shoelace2<4>(p, 4) == 0.0) return false;
    if (seg_intersect(p[0], p[1], p[2], p[3])) return false;
    if (seg_intersect(p[1], p[2], p[3], p[0])) return false;
    for (int i = 0; i < 4; ++i) {
        P2 a = p[(i + 3) & 3], b = p[i], c = p[(i + 1) & 3];
        if (cross3(a, b, c) == 0.0) {
            double dot = (a.x - b.x) * (c.x - b.x) + (a.y - b.y) * (c.y - b.y);
            if (dot > 0.0) return false;
        }
    }
    return true;
}

// Polygon(pts).is_valid and Polygon(pts).contains(Point(x, y)) of the Center-Hit metric (Detect_OBB.py:631-634): strictly interior
// (boundary excluded), winding number from cross-product signs only.  (The test suite checks it bit for bit against a plain-C restatement.)
__device__ __forceinline__ bool point_in_quad(const P2 *p, P2 q) {
    if (!quad_valid(p) || !isfinite(q.x) || !isfinite(q.y)) return false;
    int wn = 0;
    for (int i = 0; i < 4; ++i) {
        P2 a = p[i], b = p[(i + 1) & 3];
        double c = cross3(a, b, q);
        if (c == 0.0 && on_seg(a, b, q)) return false;
        if (a.y <= q.y) { if (b.y > q.y && c > 0.0) ++wn; }
        else if (b.y <= q.y && c < 0.0) --wn;
    }
    return wn != 0;
}

__device__ __forceinline__ bool is_convex(const P2 *p, int n) {
    bool pos = false, neg = false;
    for (int i = 0; i < n; ++i) {
        double c = cross3(p[i], p[(i + 1) % n], p[(i + 2) % n]);
        if (c > 0.0) pos = true;
        if (c < 0.0) neg = true;
    }
    return !(pos && neg);
}

__device__ inline double clip_area(const P2 *subj, int ns, const P2 *clip, int nc) {
    P2 bufa[16], bufb[16];
    P2 *in = bufa, *out = bufb;
    int n = ns;
    for (int i = 0; i < ns; ++i) in[i] = subj[i];
    for (int e = 0; e < nc && n > 0; ++e) {
        P2 a = clip[e], b = clip[(e + 1 == nc) ? 0 : e + 1];
        int m = 0;
        P2 s = in[n - 1];
        double ds = cross3(a, b, s);
        for (int i = 0; i < n; ++i) {
            P2 p = in[i];
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
        P2 *tmp = in; in = out; out = tmp;
        n = m;
    }
    if (n < 3) return 0.0;
    return fabs(shoelace2<16>(in, n)) * 0.5;
}

__device__ __forceinline__ void make_ccw(const P2 *p, int n, P2 *o) {
    if (shoelace2<4>(p, n) < 0.0) for (int i = 0; i < n; ++i) o[i] = p[n - 1 - i];
    else for (int i = 0; i < n; ++i) o[i] = p[i];
}

__device__ __forceinline__ void quad_tris(const P2 *q, P2 t[2][3]) {
    int r = -1;
    for (int i = 0; i < 4; ++i)
        if (cross3(q[(i + 3) & 3], q[i], q[(i + 1) & 3]) < 0.0) r = i;
    int s = (r < 0) ? 0 : r;
    t[0][0] = q[s]; t[0][1] = q[(s + 1) & 3]; t[0][2] = q[(s + 2) & 3];
    t[1][0] = q[s]; t[1][1] = q[(s + 2) & 3]; t[1][2] = q[(s + 3) & 3];
}

struct Aabb { double x0, y0, x1, y1; };

__device__ __forceinline__ Aabb quad_aabb(const P2 *p) {
    Aabb a;
    a.x0 = fmin(fmin(p[0].x, p[1].x), fmin(p[2].x, p[3].x));
    a.x1 = fmax(fmax(p[0].x, p[1].x), fmax(p[2].x, p[3].x));
    a.y0 = fmin(fmin(p[0].y, p[1].y), fmin(p[2].y, p[3].y));
    a.y1 = fmax(fmax(p[0].y, p[1].y), fmax(p[2].y, p[3].y));
    return a;
}

__device__ __forceinline__ bool aabb_disjoint(const Aabb &a, const Aabb &b) {
    return a.x1 < b.x0 || b.x1 < a.x0 || a.y1 < b.y0 || b.y1 < a.y0;
}

// IoU of two quads already known to be valid and with overlapping envelopes.
__device__ inline double poly_iou_core(const P2 *p, const P2 *q) {
    P2 pc[4], qc[4];
    double a1 = fabs(shoelace2<4>(p, 4)) * 0.5, a2 = fabs(shoelace2<4>(q, 4)) * 0.5;
    make_ccw(p, 4, pc);
    make_ccw(q, 4, qc);
    double inter;
    if (is_convex(qc, 4)) inter = clip_area(pc, 4, qc, 4);
    else if (is_convex(pc, 4)) inter = clip_area(qc, 4, pc, 4);
    else {
        P2 ta[2][3], tb[2][3];
        quad_tris(pc, ta);
        quad_tris(qc, tb);
        inter = 0.0;
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) inter += clip_area(ta[i], 3, tb[j], 3);
    }
    double uni = a1 + a2 - inter;
    return uni > 0.0 ? inter / uni : 0.0;
}

// ---- the same clip without private-memory arrays.  clip_area's two 16-vertex buffers are indexed by run-time counters, so the compiler
// keeps them in scratch (976 bytes per lane in every kernel that clips): each vertex is a round trip through the memory hierarchy, ~60 us
// for one quad pair per lane of a 1024-thread workgroup.  For two CONVEX quads -- every rotated rectangle -- a Sutherland-Hodgman stage
// adds at most one vertex (<= 5 + e after clip edge e), so: the stage's input lives in registers (static indices, fully unrolled), its
// output is appended to a lane-private LDS buffer of kClipCap vertices (the only run-time index) and read back into the registers with
// static offsets.  Every expression and the order of the shoelace sum are those of clip_area: the result is bit-identical.
// A sign pattern that only rounding can produce (more than one extra vertex at a stage) or a concave operand takes the general routine.
static constexpr int kClipCap = 8;

// lbuf: this lane's kClipCap vertices, vertex k at lbuf[k * lstride].  ok = false: not representable here (the caller falls back).
__device__ __forceinline__ double clip_area_convex(const P2 *subj, const P2 *clip, P2 *lbuf, int lstride, bool &ok) {
    P2 in[kClipCap];
#pragma unroll
    for (int i = 0; i < 4; ++i) in[i] = subj[i];
#pragma unroll
    for (int i = 4; i < kClipCap; ++i) in[i] = subj[0];
    int n = 4;
    P2 last = subj[3];
    ok = true;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const P2 a = clip[e], b = clip[(e + 1) & 3];
        int m = 0;
        P2 s = last;
        double ds = cross3(a, b, s);
#pragma unroll
        for (int i = 0; i < 4 + e; ++i) {
            if (i < n) {
                const P2 p = in[i];
                const double dp = cross3(a, b, p);
                if (dp >= 0.0) {
                    if (ds < 0.0) {
                        const double t = ds / (ds - dp);
                        P2 v;
                        v.x = s.x + (p.x - s.x) * t;
                        v.y = s.y + (p.y - s.y) * t;
                        lbuf[min(m, kClipCap - 1) * lstride] = v;
                        ++m;
                    }
                    lbuf[min(m, kClipCap - 1) * lstride] = p;
                    ++m;
                    last = p;
                } else if (ds >= 0.0) {
                    const double t = ds / (ds - dp);
                    P2 v;
                    v.x = s.x + (p.x - s.x) * t;
                    v.y = s.y + (p.y - s.y) * t;
                    lbuf[min(m, kClipCap - 1) * lstride] = v;
                    ++m;
                    last = v;
                }
                s = p;
                ds = dp;
            }
        }
        if (m > 5 + e) ok = false;
        n = min(m, 5 + e);
#pragma unroll
        for (int k = 0; k < 5 + e; ++k) in[k] = lbuf[k * lstride];
    }
    if (n < 3) return 0.0;
    double sum = 0.0;  // shoelace2, vertex by vertex in the same order
#pragma unroll
    for (int i = 0; i < kClipCap; ++i)
        if (i < n) {
            const P2 pj = (i + 1 == n) ? in[0] : in[(i + 1) & (kClipCap - 1)];
            sum += in[i].x * pj.y - pj.x * in[i].y;
        }
    return fabs(sum) * 0.5;
}

// poly_iou_core with the convex case clipped through the lane's LDS buffer
__device__ inline double poly_iou_core_lds(const P2 *p, const P2 *q, P2 *lbuf, int lstride) {
    P2 pc[4], qc[4];
    const double sp = shoelace2<4>(p, 4), sq = shoelace2<4>(q, 4);
    const bool rp = sp < 0.0, rq = sq < 0.0;  // make_ccw
#pragma unroll
    for (int i = 0; i < 4; ++i) { pc[i] = rp ? p[3 - i] : p[i]; qc[i] = rq ? q[3 - i] : q[i]; }
    if (is_convex(pc, 4) && is_convex(qc, 4)) {
        bool ok;
        const double inter = clip_area_convex(pc, qc, lbuf, lstride, ok);
        if (ok) {
            const double a1 = fabs(sp) * 0.5, a2 = fabs(sq) * 0.5;
            const double uni = a1 + a2 - inter;
            return uni > 0.0 ? inter / uni : 0.0;
        }
    }
    return poly_iou_core(p, q);
}

__device__ inline double poly_iou_lds(const double *b1, const double *b2, P2 *lbuf, int lstride) {
    P2 p[4], q[4];
    for (int i = 0; i < 4; ++i) { p[i].x = b1[2 * i]; p[i].y = b1[2 * i + 1]; q[i].x = b2[2 * i]; q[i].y = b2[2 * i + 1]; }
    if (!quad_valid(p) || !quad_valid(q)) return 0.0;
    if (aabb_disjoint(quad_aabb(p), quad_aabb(q))) return 0.0;
    return poly_iou_core_lds(p, q, lbuf, lstride);
}

__device__ inline double poly_iou(const double *b1, const double *b2) {
    P2 p[4], q[4];
    for (int i = 0; i < 4; ++i) { p[i].x = b1[2 * i]; p[i].y = b1[2 * i + 1]; q[i].x = b2[2 * i]; q[i].y = b2[2 * i + 1]; }
    if (!quad_valid(p) || !quad_valid(q)) return 0.0;
    if (aabb_disjoint(quad_aabb(p), quad_aabb(q))) return 0.0;
    return poly_iou_core(p, q);
}

}  // namespace obb
