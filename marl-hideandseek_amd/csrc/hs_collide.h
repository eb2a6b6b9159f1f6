// Convex narrowphase on the device: exact SAT over face normals and edge-direction cross products
// with reference-face clipping (boxes, scaled wall boxes and the ramp wedge), and hull-vs-plane.
// Replaces madrona::phys narrowphase (spliced in at src/sim.cpp:1162-1163; engine source absent
// — DESIGN.md "Engine decisions").  Hull vertices of movable bodies live in LDS (written once
// per substep by the owning lane); wall hulls are generated from (cx,cy,hx,hy) on the fly.
#pragma once
#include "hs_dev.h"

namespace hs {

// ---- hull topology (data/*_collision.obj): boxes 8v/6f/12e, wedge 6v/5f/9e ----
__constant__ int cBoxFaceIdx[6][4] = {{0, 4, 6, 2}, {1, 3, 7, 5}, {0, 1, 5, 4}, {2, 6, 7, 3}, {0, 2, 3, 1}, {4, 5, 7, 6}};
__constant__ int cBoxEdges[12][3] = {{0, 1, 0}, {2, 3, 0}, {4, 5, 0}, {6, 7, 0}, {0, 2, 1}, {1, 3, 1},
                                     {4, 6, 1}, {5, 7, 1}, {0, 4, 2}, {1, 5, 2}, {2, 6, 2}, {3, 7, 2}};
__constant__ float cWedgeV[6][3] = {{1, 1, 1}, {1, 1, -1}, {1, -2, -1}, {-1, 1, 1}, {-1, 1, -1}, {-1, -2, -1}};
__constant__ int cWedgeFaceCnt[5] = {4, 4, 4, 3, 3};
__constant__ int cWedgeFaceIdx[5][4] = {{4, 1, 2, 5}, {4, 3, 0, 1}, {2, 0, 3, 5}, {1, 0, 2, 0}, {5, 3, 4, 0}};
__constant__ float cWedgeFN[5][3] = {{0, 0, -1}, {0, 1, 0}, {0, -0.554700196f, 0.832050294f}, {1, 0, 0}, {-1, 0, 0}};
__constant__ int cWedgeEdges[9][3] = {{4, 1, 0}, {2, 5, 0}, {3, 0, 0}, {1, 2, 1}, {5, 4, 1}, {4, 3, 2}, {0, 1, 2}, {2, 0, 3}, {5, 3, 3}};

enum { HULL_BOX = 0, HULL_WEDGE = 1, HULL_WALL = 2 };

struct HullRef {
    int kind;
    const float (*v)[3];    // LDS vertices (HULL_BOX / HULL_WEDGE)
    V3 c, ax, ay, az, e;    // centre, rotation columns, half extents
};

HSD int hull_nv(const HullRef &h) { return h.kind == HULL_WEDGE ? 6 : 8; }
HSD int hull_nf(const HullRef &h) { return h.kind == HULL_WEDGE ? 5 : 6; }
HSD int hull_ned(const HullRef &h) { return h.kind == HULL_WEDGE ? 4 : 3; }
HSD int hull_ne(const HullRef &h) { return h.kind == HULL_WEDGE ? 9 : 12; }

HSD V3 hull_v(const HullRef &h, int i) {
    if (h.kind == HULL_WALL)
        return {h.c.x + ((i & 1) ? h.e.x : -h.e.x), h.c.y + ((i & 2) ? h.e.y : -h.e.y), h.c.z + ((i & 4) ? h.e.z : -h.e.z)};
    return {h.v[i][0], h.v[i][1], h.v[i][2]};
}
HSD V3 hull_fn(const HullRef &h, int f) {
    if (h.kind == HULL_WEDGE)
        return (h.ax * cWedgeFN[f][0] + h.ay * cWedgeFN[f][1]) + h.az * cWedgeFN[f][2];
    V3 a = (f >> 1) == 0 ? h.ax : ((f >> 1) == 1 ? h.ay : h.az);
    return (f & 1) ? a : -a;
}
HSD int hull_fcnt(const HullRef &h, int f) { return h.kind == HULL_WEDGE ? cWedgeFaceCnt[f] : 4; }
HSD int hull_fidx(const HullRef &h, int f, int k) { return h.kind == HULL_WEDGE ? cWedgeFaceIdx[f][k] : cBoxFaceIdx[f][k]; }
HSD float hull_fd(const HullRef &h, int f, V3 fn) { return dot(fn, hull_v(h, hull_fidx(h, f, 0))); }
HSD V3 hull_ed(const HullRef &h, int i) {
    if (i == 0) return h.ax;
    if (i == 1) return h.ay;
    if (i == 2) return h.az;
    return (h.ax * 0.f + h.ay * 0.832050294f) + h.az * 0.554700196f;
}
HSD void hull_edge(const HullRef &h, int e, int *v0, int *v1, int *dir) {
    if (h.kind == HULL_WEDGE) { *v0 = cWedgeEdges[e][0]; *v1 = cWedgeEdges[e][1]; *dir = cWedgeEdges[e][2]; }
    else { *v0 = cBoxEdges[e][0]; *v1 = cBoxEdges[e][1]; *dir = cBoxEdges[e][2]; }
}
// Local-frame vertex of a movable hull
HSD V3 hull_local_vertex(int obj, int i) {
    if (obj == OBJ_RAMP) return {cWedgeV[i][0], cWedgeV[i][1], cWedgeV[i][2]};
    V3 e = obj_half_extents(obj);
    return {(i & 1) ? e.x : -e.x, (i & 2) ? e.y : -e.y, (i & 4) ? e.z : -e.z};
}

// Write the world-space vertices of a movable body and return its AABB.
HSD void hull_build(int obj, V3 pos, Q rot, float (*v)[3], V3 *lo_out, V3 *hi_out) {
    M3 m = m3_from_quat(rot);
    V3 lo, hi;
    if (obj == OBJ_RAMP) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            V3 p = ((pos + m.c0 * cWedgeV[i][0]) + m.c1 * cWedgeV[i][1]) + m.c2 * cWedgeV[i][2];
            v[i][0] = p.x; v[i][1] = p.y; v[i][2] = p.z;
            if (i == 0) { lo = p; hi = p; }
            else { lo = {fminf(lo.x, p.x), fminf(lo.y, p.y), fminf(lo.z, p.z)}; hi = {fmaxf(hi.x, p.x), fmaxf(hi.y, p.y), fmaxf(hi.z, p.z)}; }
        }
    } else {
        V3 e = obj_half_extents(obj);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float sx = (i & 1) ? e.x : -e.x, sy = (i & 2) ? e.y : -e.y, sz = (i & 4) ? e.z : -e.z;
            V3 p = ((pos + m.c0 * sx) + m.c1 * sy) + m.c2 * sz;
            v[i][0] = p.x; v[i][1] = p.y; v[i][2] = p.z;
            if (i == 0) { lo = p; hi = p; }
            else { lo = {fminf(lo.x, p.x), fminf(lo.y, p.y), fminf(lo.z, p.z)}; hi = {fmaxf(hi.x, p.x), fmaxf(hi.y, p.y), fmaxf(hi.z, p.z)}; }
        }
    }
    *lo_out = lo; *hi_out = hi;
}

HSD HullRef hull_ref_body(int obj, V3 pos, Q rot, const float (*v)[3]) {
    M3 m = m3_from_quat(rot);
    HullRef h;
    h.kind = obj == OBJ_RAMP ? HULL_WEDGE : HULL_BOX;
    h.v = v; h.c = pos; h.ax = m.c0; h.ay = m.c1; h.az = m.c2; h.e = obj_half_extents(obj);
    return h;
}
HSD HullRef hull_ref_wall(float cx, float cy, float hx, float hy) {
    HullRef h;
    h.kind = HULL_WALL; h.v = nullptr;
    h.c = {cx, cy, 1.25f}; h.ax = {1.f, 0.f, 0.f}; h.ay = {0.f, 1.f, 0.f}; h.az = {0.f, 0.f, 1.f};
    h.e = {hx, hy, 1.25f};
    return h;
}

struct RawManifold { V3 n; int np; V3 pA[4]; V3 pB[4]; int vidx[4]; };

HSD float support_min(const HullRef &h, V3 n) {
    const int nv = hull_nv(h);
    float s = dot(n, hull_v(h, 0));
    for (int i = 1; i < nv; ++i) s = fminf(s, dot(n, hull_v(h, i)));
    return s;
}
HSD float support_max(const HullRef &h, V3 n) {
    const int nv = hull_nv(h);
    float s = dot(n, hull_v(h, 0));
    for (int i = 1; i < nv; ++i) s = fmaxf(s, dot(n, hull_v(h, i)));
    return s;
}

// Hull vs plane pn.p = pd: up to the 4 deepest vertices below the plane.
HSD bool collide_hull_plane(const HullRef &A, V3 pn, float pd, RawManifold &m) {
    int np = 0; float depth[4] = {0.f, 0.f, 0.f, 0.f};
    const int nv = hull_nv(A);
    for (int i = 0; i < nv; ++i) {
        V3 v = hull_v(A, i);
        float dist = dot(pn, v) - pd;
        if (!(dist < 0.f)) continue;
        float dep = -dist;
        V3 pb = v - pn * dist;
        int slot;
        if (np < 4) { slot = np; np++; }
        else {
            int mi = 0;
#pragma unroll
            for (int k = 1; k < 4; ++k) if (depth[k] < depth[mi]) mi = k;
            if (!(dep > depth[mi])) continue;
            slot = mi;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k == slot) { depth[k] = dep; m.pA[k] = v; m.pB[k] = pb; m.vidx[k] = i; }
    }
    m.np = np; m.n = -pn;
    return np > 0;
}

HSD void closest_seg_seg(V3 p1, V3 q1, V3 p2, V3 q2, V3 *c1, V3 *c2) {
    V3 d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
    float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r);
    float c = dot(d1, r), b = dot(d1, d2);
    float denom = a * e - b * b;
    float s = 0.f, t;
    if (denom > 1e-9f) { s = (b * f - c * e) / denom; s = fminf(fmaxf(s, 0.f), 1.f); }
    t = (b * s + f) / e;
    if (t < 0.f) { t = 0.f; s = fminf(fmaxf(-c / a, 0.f), 1.f); }
    else if (t > 1.f) { t = 1.f; s = fminf(fmaxf((b - c) / a, 0.f), 1.f); }
    *c1 = p1 + d1 * s; *c2 = p2 + d2 * t;
}

// Clip the incident face of I against the side planes of reference face fr of R; keep points on
// or below the reference plane; reduce to <= 4.
HSD int clip_face_contact(const HullRef &R, int fr, V3 nr, const HullRef &I, V3 *pInc, float *dist_out) {
    const float dr = hull_fd(R, fr, nr);
    const int inf = hull_nf(I);
    int fi = 0; float best = dot(nr, hull_fn(I, 0));
    for (int f = 1; f < inf; ++f) { float d = dot(nr, hull_fn(I, f)); if (d < best) { best = d; fi = f; } }
    V3 poly[8], tmp[8]; int n = hull_fcnt(I, fi);
    for (int k = 0; k < n; ++k) poly[k] = hull_v(I, hull_fidx(I, fi, k));
    const int rc = hull_fcnt(R, fr);
    for (int k = 0; k < rc && n > 0; ++k) {
        V3 v0 = hull_v(R, hull_fidx(R, fr, k)), v1 = hull_v(R, hull_fidx(R, fr, (k + 1) % rc));
        V3 s = cross(v1 - v0, nr);
        int m = 0;
        V3 prev = poly[n - 1]; float dprev = dot(s, prev - v0);
        for (int i = 0; i < n; ++i) {
            V3 cur = poly[i]; float dcur = dot(s, cur - v0);
            bool in_prev = dprev <= 0.f, in_cur = dcur <= 0.f;
            if (in_prev != in_cur) {
                float t = dprev / (dprev - dcur);
                if (m < 8) tmp[m++] = prev + (cur - prev) * t;
            }
            if (in_cur) { if (m < 8) tmp[m++] = cur; }
            prev = cur; dprev = dcur;
        }
        n = m;
        for (int i = 0; i < n; ++i) poly[i] = tmp[i];
    }
    V3 pts[8]; float dist[8]; int c = 0;
    for (int i = 0; i < n; ++i) {
        float d = dot(nr, poly[i]) - dr;
        if (d <= 0.f) { pts[c] = poly[i]; dist[c] = d; c++; }
    }
    if (c <= 4) {
        for (int i = 0; i < c; ++i) { pInc[i] = pts[i]; dist_out[i] = dist[i]; }
        return c;
    }
    int i0 = 0; for (int i = 1; i < c; ++i) if (dist[i] < dist[i0]) i0 = i;
    int i1 = -1; float bd = -1.f;
    for (int i = 0; i < c; ++i) { if (i == i0) continue; float d2 = len2(pts[i] - pts[i0]); if (d2 > bd) { bd = d2; i1 = i; } }
    int i2 = -1, i3 = -1; float amax = 0.f, amin = 0.f;
    for (int i = 0; i < c; ++i) {
        if (i == i0 || i == i1) continue;
        float ar = dot(cross(pts[i0] - pts[i], pts[i1] - pts[i]), nr);
        if (i2 < 0 || ar > amax) { amax = ar; i2 = i; }
    }
    for (int i = 0; i < c; ++i) {
        if (i == i0 || i == i1 || i == i2) continue;
        float ar = dot(cross(pts[i0] - pts[i], pts[i1] - pts[i]), nr);
        if (i3 < 0 || ar < amin) { amin = ar; i3 = i; }
    }
    const int sel[4] = {i0, i1, i2, i3};
    for (int k = 0; k < 4; ++k) { pInc[k] = pts[sel[k]]; dist_out[k] = dist[sel[k]]; }
    return 4;
}

HSD bool collide_hulls(const HullRef &A, const HullRef &B, RawManifold &m) {
    float bestA = 0.f; int fa = -1;
    const int anf = hull_nf(A), bnf = hull_nf(B);
    for (int f = 0; f < anf; ++f) {
        V3 fn = hull_fn(A, f);
        float s = support_min(B, fn) - hull_fd(A, f, fn);
        if (s > 0.f) return false;
        if (fa < 0 || s > bestA) { bestA = s; fa = f; }
    }
    float bestB = 0.f; int fb = -1;
    for (int f = 0; f < bnf; ++f) {
        V3 fn = hull_fn(B, f);
        float s = support_min(A, fn) - hull_fd(B, f, fn);
        if (s > 0.f) return false;
        if (fb < 0 || s > bestB) { bestB = s; fb = f; }
    }
    float bestE = 0.f; int ea = -1, eb = -1; V3 axE = {0.f, 0.f, 0.f};
    V3 ab = B.c - A.c;
    const int aned = hull_ned(A), bned = hull_ned(B);
    for (int i = 0; i < aned; ++i) {
        V3 ei = hull_ed(A, i);
        for (int j = 0; j < bned; ++j) {
            V3 ax = cross(ei, hull_ed(B, j));
            float l2 = len2(ax);
            if (l2 < 1e-6f) continue;
            ax = ax * (1.f / sqrtf(l2));
            if (dot(ax, ab) < 0.f) ax = -ax;
            float s = support_min(B, ax) - support_max(A, ax);
            if (s > 0.f) return false;
            if (ea < 0 || s > bestE) { bestE = s; ea = i; eb = j; axE = ax; }
        }
    }
    float bestF = fmaxf(bestA, bestB);
    if (ea >= 0 && bestE > 0.9f * bestF + 0.0025f) {
        int sa = -1, sb = -1; float va = 0.f, vb = 0.f;
        const int ane = hull_ne(A), bne = hull_ne(B);
        for (int e = 0; e < ane; ++e) {
            int v0, v1, dir; hull_edge(A, e, &v0, &v1, &dir);
            if (dir != ea) continue;
            float p = dot(axE, hull_v(A, v0)) + dot(axE, hull_v(A, v1));
            if (sa < 0 || p > va) { va = p; sa = e; }
        }
        for (int e = 0; e < bne; ++e) {
            int v0, v1, dir; hull_edge(B, e, &v0, &v1, &dir);
            if (dir != eb) continue;
            float p = dot(axE, hull_v(B, v0)) + dot(axE, hull_v(B, v1));
            if (sb < 0 || p < vb) { vb = p; sb = e; }
        }
        int a0, a1, b0, b1, dd;
        hull_edge(A, sa, &a0, &a1, &dd); hull_edge(B, sb, &b0, &b1, &dd);
        V3 c1, c2;
        closest_seg_seg(hull_v(A, a0), hull_v(A, a1), hull_v(B, b0), hull_v(B, b1), &c1, &c2);
        m.n = axE; m.np = 1; m.pA[0] = c1; m.pB[0] = c2;
        return true;
    }
    V3 pinc[4]; float dist[4];
    if (bestB > 0.98f * bestA + 0.00125f) {
        V3 nr = hull_fn(B, fb);
        int c = clip_face_contact(B, fb, nr, A, pinc, dist);
        if (c == 0) return false;
        m.n = -nr; m.np = c;
        for (int i = 0; i < c; ++i) { m.pA[i] = pinc[i]; m.pB[i] = pinc[i] - nr * dist[i]; }
    } else {
        V3 nr = hull_fn(A, fa);
        int c = clip_face_contact(A, fa, nr, B, pinc, dist);
        if (c == 0) return false;
        m.n = nr; m.np = c;
        for (int i = 0; i < c; ++i) { m.pB[i] = pinc[i]; m.pA[i] = pinc[i] - nr * dist[i]; }
    }
    return true;
}

}  // namespace hs
