// Convex narrowphase on the device: exact SAT over face normals and edge-direction cross products
// with reference-face clipping (boxes, scaled wall boxes and the ramp wedge), and hull-vs-plane.
// Replaces madrona::phys narrowphase (spliced in at src/sim.cpp:1162-1163; engine source absent
// — DESIGN.md "Engine decisions").
//
// Nothing here touches memory except the per-lane LDS clip buffer: hull vertices are computed from
// (centre, rotation columns, half extents) when needed, box supports are closed-form
// (centre projection -/+ projected radius), and the hull topology tables of data/*_collision.obj
// are bit-packed immediates instead of lookup tables.
#pragma once
#include "hs_dev.h"

namespace hs {

enum { HULL_BOX = 0, HULL_WEDGE = 1 };

struct HullRef {
    int kind;
    V3 c, ax, ay, az, e;    // centre, rotation columns, half extents (boxes)
};

// ---- packed topology ----
// box face loops {0,4,6,2},{1,3,7,5},{0,1,5,4},{2,6,7,3},{0,2,3,1},{4,5,7,6}: 3 bits per index
HSD int box_face_idx(int f, int k) {
    constexpr unsigned long long t0 = (0ull) | (4ull << 3) | (6ull << 6) | (2ull << 9) |
                                      (1ull << 12) | (3ull << 15) | (7ull << 18) | (5ull << 21) |
                                      (0ull << 24) | (1ull << 27) | (5ull << 30) | (4ull << 33);
    constexpr unsigned long long t1 = (2ull) | (6ull << 3) | (7ull << 6) | (3ull << 9) |
                                      (0ull << 12) | (2ull << 15) | (3ull << 18) | (1ull << 21) |
                                      (4ull << 24) | (5ull << 27) | (7ull << 30) | (6ull << 33);
    const unsigned long long t = f < 3 ? t0 : t1;
    const int ff = f < 3 ? f : f - 3;
    return (int)((t >> (ff * 12 + k * 3)) & 7ull);
}
// wedge face loops {4,1,2,5},{4,3,0,1},{2,0,3,5},{1,0,2,0},{5,3,4,0}
HSD int wedge_face_idx(int f, int k) {
    constexpr unsigned long long t = (4ull) | (1ull << 3) | (2ull << 6) | (5ull << 9) |
                                     (4ull << 12) | (3ull << 15) | (0ull << 18) | (1ull << 21) |
                                     (2ull << 24) | (0ull << 27) | (3ull << 30) | (5ull << 33) |
                                     (1ull << 36) | (0ull << 39) | (2ull << 42) | (0ull << 45) |
                                     (5ull << 48) | (3ull << 51) | (4ull << 54) | (0ull << 57);
    return (int)((t >> (f * 12 + k * 3)) & 7ull);
}
// wedge edges {4,1,0},{2,5,0},{3,0,0},{1,2,1},{5,4,1},{4,3,2},{0,1,2},{2,0,3},{5,3,3}
HSD void wedge_edge(int e, int *v0, int *v1, int *dir) {
    constexpr unsigned long long tv = (4ull | (1ull << 3)) | ((2ull | (5ull << 3)) << 6) | ((3ull | (0ull << 3)) << 12) |
                                      ((1ull | (2ull << 3)) << 18) | ((5ull | (4ull << 3)) << 24) | ((4ull | (3ull << 3)) << 30) |
                                      ((0ull | (1ull << 3)) << 36) | ((2ull | (0ull << 3)) << 42) | ((5ull | (3ull << 3)) << 48);
    constexpr unsigned td = 0u | (0u << 2) | (0u << 4) | (1u << 6) | (1u << 8) | (2u << 10) | (2u << 12) | (3u << 14) | (3u << 16);
    *v0 = (int)((tv >> (e * 6)) & 7ull); *v1 = (int)((tv >> (e * 6 + 3)) & 7ull); *dir = (int)((td >> (e * 2)) & 3u);
}
// box edges: x-direction {0,1},{2,3},{4,5},{6,7}; y {0,2},{1,3},{4,6},{5,7}; z {0,4},{1,5},{2,6},{3,7}
HSD void box_edge(int e, int *v0, int *v1, int *dir) {
    const int d = e >> 2, k = e & 3;
    int a, b;
    if (d == 0) { a = k << 1; b = a | 1; }
    else if (d == 1) { a = (k & 1) | ((k & 2) << 1); b = a | 2; }
    else { a = k; b = k | 4; }
    *v0 = a; *v1 = b; *dir = d;
}
HSD V3 wedge_local_v(int i) {     // {1,1,1},{1,1,-1},{1,-2,-1},{-1,1,1},{-1,1,-1},{-1,-2,-1}
    const int r = i >= 3 ? i - 3 : i;
    return {i < 3 ? 1.f : -1.f, r == 2 ? -2.f : 1.f, r == 0 ? 1.f : -1.f};
}
HSD V3 wedge_local_fn(int f) {    // {0,0,-1},{0,1,0},{0,-2,3}/sqrt13,{1,0,0},{-1,0,0}
    if (f == 0) return {0.f, 0.f, -1.f};
    if (f == 1) return {0.f, 1.f, 0.f};
    if (f == 2) return {0.f, -0.554700196f, 0.832050294f};
    if (f == 3) return {1.f, 0.f, 0.f};
    return {-1.f, 0.f, 0.f};
}

// One of four values by index, through bit masks: written as a chain of conditionals on a per-lane index the
// compiler parks the candidates in scratch memory and indexes them there — stores that show up as HBM write traffic.
HSD float sel4_bits(int i, float a, float b, float c, float d) {
    const unsigned m0 = i == 0 ? ~0u : 0u, m1 = i == 1 ? ~0u : 0u, m2 = i == 2 ? ~0u : 0u, m3 = i >= 3 ? ~0u : 0u;
    return __uint_as_float((__float_as_uint(a) & m0) | (__float_as_uint(b) & m1) | (__float_as_uint(c) & m2) | (__float_as_uint(d) & m3));
}
HSD int hull_nv(const HullRef &h) { return h.kind == HULL_WEDGE ? 6 : 8; }
HSD int hull_nf(const HullRef &h) { return h.kind == HULL_WEDGE ? 5 : 6; }
HSD int hull_ned(const HullRef &h) { return h.kind == HULL_WEDGE ? 4 : 3; }
HSD int hull_ne(const HullRef &h) { return h.kind == HULL_WEDGE ? 9 : 12; }

HSD V3 hull_v(const HullRef &h, int i) {
    V3 l;
    if (h.kind == HULL_WEDGE) l = wedge_local_v(i);
    else l = {(i & 1) ? h.e.x : -h.e.x, (i & 2) ? h.e.y : -h.e.y, (i & 4) ? h.e.z : -h.e.z};
    return madd(madd(madd(h.c, h.ax, l.x), h.ay, l.y), h.az, l.z);
}
HSD V3 hull_fn(const HullRef &h, int f) {
    if (h.kind == HULL_WEDGE) {
        V3 l = wedge_local_fn(f);
        return (h.ax * l.x + h.ay * l.y) + h.az * l.z;
    }
    const int k = f >> 1;
    const V3 a = {sel4_bits(k, h.ax.x, h.ay.x, h.az.x, h.az.x), sel4_bits(k, h.ax.y, h.ay.y, h.az.y, h.az.y), sel4_bits(k, h.ax.z, h.ay.z, h.az.z, h.az.z)};
    return (f & 1) ? a : -a;
}
HSD int hull_fcnt(const HullRef &h, int f) { return (h.kind == HULL_WEDGE && f >= 3) ? 3 : 4; }
HSD int hull_fidx(const HullRef &h, int f, int k) { return h.kind == HULL_WEDGE ? wedge_face_idx(f, k) : box_face_idx(f, k); }
// plane offset of face f (normal fn): closed form for boxes, first loop vertex for the wedge
HSD float hull_fd(const HullRef &h, int f, V3 fn) {
    if (h.kind == HULL_WEDGE) return dot(fn, hull_v(h, wedge_face_idx(f, 0)));
    const float ei = sel4_bits(f >> 1, h.e.x, h.e.y, h.e.z, h.e.z);
    return dot(fn, h.c) + ei;
}
HSD V3 hull_ed(const HullRef &h, int i) {
    const V3 s = (h.ax * 0.f + h.ay * 0.832050294f) + h.az * 0.554700196f;
    return {sel4_bits(i, h.ax.x, h.ay.x, h.az.x, s.x), sel4_bits(i, h.ax.y, h.ay.y, h.az.y, s.y), sel4_bits(i, h.ax.z, h.ay.z, h.az.z, s.z)};
}
HSD void hull_edge(const HullRef &h, int e, int *v0, int *v1, int *dir) {
    if (h.kind == HULL_WEDGE) wedge_edge(e, v0, v1, dir); else box_edge(e, v0, v1, dir);
}
HSD V3 hull_local_vertex(int obj, int i) {
    if (obj == OBJ_RAMP) return wedge_local_v(i);
    V3 e = obj_half_extents(obj);
    return {(i & 1) ? e.x : -e.x, (i & 2) ? e.y : -e.y, (i & 4) ? e.z : -e.z};
}

HSD HullRef hull_ref_body(int obj, V3 pos, Q rot) {
    M3 m = m3_from_quat(rot);
    HullRef h;
    h.kind = obj == OBJ_RAMP ? HULL_WEDGE : HULL_BOX;
    h.c = pos; h.ax = m.c0; h.ay = m.c1; h.az = m.c2;
    h.e = obj == OBJ_RAMP ? V3{1.f, 1.f, 1.f} : obj_half_extents(obj);
    return h;
}
// What a hull is built from: 11 words instead of HullRef's 16.  The convex test keeps the sources of its two hulls
// and builds the HullRefs it needs per stage (face normals: the lane's own X / Y assignment; edge directions and
// clipping: A / B), so that only one pair of HullRefs is live at a time.
struct HullSrc { int kind; V3 c; Q q; V3 e; };
HSD HullSrc hull_src_body(int obj, V3 pos, Q rot) {
    return {obj == OBJ_RAMP ? HULL_WEDGE : HULL_BOX, pos, rot, obj == OBJ_RAMP ? V3{1.f, 1.f, 1.f} : obj_half_extents(obj)};
}
// the identity rotation gives exactly hull_ref_wall's axes (m3_from_quat of {1,0,0,0} is the unit matrix)
HSD HullSrc hull_src_wall(float cx, float cy, float hx, float hy) {
    return {HULL_BOX, V3{cx, cy, 1.25f}, Q{1.f, 0.f, 0.f, 0.f}, V3{hx, hy, 1.25f}};
}
HSD HullSrc hull_src_sel(bool c, const HullSrc &a, const HullSrc &b) {
    HullSrc r;
    r.kind = c ? a.kind : b.kind; r.c = vsel(c, a.c, b.c); r.e = vsel(c, a.e, b.e);
    r.q = {c ? a.q.w : b.q.w, c ? a.q.x : b.q.x, c ? a.q.y : b.q.y, c ? a.q.z : b.q.z};
    return r;
}
HSD HullRef hull_from(const HullSrc &s) {
    M3 m = m3_from_quat(s.q);
    HullRef h;
    h.kind = s.kind; h.c = s.c; h.ax = m.c0; h.ay = m.c1; h.az = m.c2; h.e = s.e;
    return h;
}
HSD HullRef hull_ref_wall(float cx, float cy, float hx, float hy) {
    HullRef h;
    h.kind = HULL_BOX;
    h.c = {cx, cy, 1.25f}; h.ax = {1.f, 0.f, 0.f}; h.ay = {0.f, 1.f, 0.f}; h.az = {0.f, 0.f, 1.f};
    h.e = {hx, hy, 1.25f};
    return h;
}
// AABB of the hull's vertices
HSD void hull_aabb(const HullRef &h, V3 *lo_out, V3 *hi_out) {
    V3 lo = hull_v(h, 0), hi = lo;
    const int nv = hull_nv(h);
#pragma unroll
    for (int i = 1; i < 8; ++i) {
        if (i < nv) {
            V3 p = hull_v(h, i);
            lo = {fminf(lo.x, p.x), fminf(lo.y, p.y), fminf(lo.z, p.z)};
            hi = {fmaxf(hi.x, p.x), fmaxf(hi.y, p.y), fmaxf(hi.z, p.z)};
        }
    }
    *lo_out = lo; *hi_out = hi;
}

struct RawManifold { V3 n; int np; V3 pA[4]; V3 pB[4]; int vidx[4]; };

HSD float box_radius(const HullRef &h, V3 n) {
    return hs_fma(fabsf(dot(n, h.az)), h.e.z, hs_fma(fabsf(dot(n, h.ay)), h.e.y, fabsf(dot(n, h.ax)) * h.e.x));
}
// The wedge's six world vertices, computed once per convex test (its supports walk them ~46 times).
struct WedgeVerts { V3 v[6]; };
HSD void wedge_verts(const HullRef &h, WedgeVerts &wv) {
#pragma unroll
    for (int i = 0; i < 6; ++i) wv.v[i] = hull_v(h, i);
}
HSD float support_min(const HullRef &h, const WedgeVerts &wv, V3 n) {
    if (h.kind == HULL_BOX) return dot(n, h.c) - box_radius(h, n);
    float s = dot(n, wv.v[0]);
#pragma unroll
    for (int i = 1; i < 6; ++i) s = fminf(s, dot(n, wv.v[i]));
    return s;
}
HSD float support_max(const HullRef &h, const WedgeVerts &wv, V3 n) {
    if (h.kind == HULL_BOX) return dot(n, h.c) + box_radius(h, n);
    float s = dot(n, wv.v[0]);
#pragma unroll
    for (int i = 1; i < 6; ++i) s = fmaxf(s, dot(n, wv.v[i]));
    return s;
}
// plane offset of face f: closed form for boxes, first loop vertex for the wedge
HSD float hull_fd_w(const HullRef &h, const WedgeVerts &wv, int f, V3 fn) {
    if (h.kind == HULL_WEDGE) {
        // first loop vertex of face f: 4, 4, 2, 1, 5
        const int k = f < 2 ? 0 : f - 1;
        const V3 v = {sel4_bits(k, wv.v[4].x, wv.v[2].x, wv.v[1].x, wv.v[5].x), sel4_bits(k, wv.v[4].y, wv.v[2].y, wv.v[1].y, wv.v[5].y),
                      sel4_bits(k, wv.v[4].z, wv.v[2].z, wv.v[1].z, wv.v[5].z)};
        return dot(fn, v);
    }
    const float ei = sel4_bits(f >> 1, h.e.x, h.e.y, h.e.z, h.e.z);
    return dot(fn, h.c) + ei;
}

// Hull vs plane pn.p = pd: up to the 4 deepest vertices below the plane.
HSD bool collide_hull_plane(const HullRef &A, V3 pn, float pd, RawManifold &m) {
    int np = 0; float depth[4] = {0.f, 0.f, 0.f, 0.f};
    const int nv = hull_nv(A);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i >= nv) continue;
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

// The ground-manifold form of collide_hull_plane: same vertex order, same selection and the same expressions, but
// it keeps per contact only what the ground solve reads — the vertex index (the plane offset dot(pB, n) of the ground
// plane is a signed zero: hs_k_physics.h BodyReg) — instead of the points (this runs for every body in every substep).
// Returns np; idx packs 3 bits per contact.
HSD int ground_manifold(const HullRef &A, V3 pn, float pd, int *idx) {
    int np = 0, vi = 0; float depth[4] = {0.f, 0.f, 0.f, 0.f};
    const int nv = hull_nv(A);
    const V3 n = -pn;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i >= nv) continue;
        V3 v = hull_v(A, i);
        float dist = dot(pn, v) - pd;
        if (!(dist < 0.f)) continue;
        float dep = -dist;
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
            if (k == slot) { depth[k] = dep; vi = (vi & ~(7 << (3 * k))) | (i << (3 * k)); }
    }
    *idx = vi;
    return np;
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

// LDS scratch for polygon clipping: two ping-pong polygons of up to 8 vertices per lane, laid out
// [buffer][vertex][component][lane] so that the lanes of a wave hit different banks (a
// per-lane struct of 48 words would be a 16-way bank conflict on every access).
constexpr int kSatPairs = 32;           // pairs per round of the separating-axis search (two lanes per pair)
// lanes of a wave that generate contacts in one round (each owns a column of the scratch): 48 for a wave of 8 worlds, 24 for 4
constexpr int clip_lanes(int tile) { return tile == 8 ? 48 : 24; }
constexpr int clip_words(int lanes) { return 2 * 8 * 3 * lanes; }
struct ClipBuf { float *base; int lane; int stride; };          // stride = clip lanes of the kernel (a compile-time constant at every use)
HSD V3 cb_get(const ClipBuf &b, int w, int i) {
    const float *p = b.base + ((w * 8 + i) * 3) * b.stride + b.lane;
    return {p[0], p[b.stride], p[2 * b.stride]};
}
HSD void cb_set(const ClipBuf &b, int w, int i, V3 v) {
    float *p = b.base + ((w * 8 + i) * 3) * b.stride + b.lane;
    p[0] = v.x; p[b.stride] = v.y; p[2 * b.stride] = v.z;
}

// Clip the incident face of I against the side planes of reference face fr of R; keep points on
// or below the reference plane; reduce to <= 4.  Returns the count; pInc/dist_out have 4 slots.
HSD int clip_face_contact(const HullRef &R, int fr, V3 nr, const HullRef &I, const ClipBuf &cb, V3 *pInc, float *dist_out) {
    const float dr = hull_fd(R, fr, nr);
    const int inf = hull_nf(I);
    int fi = 0; float best = dot(nr, hull_fn(I, 0));
    for (int f = 1; f < inf; ++f) { float d = dot(nr, hull_fn(I, f)); if (d < best) { best = d; fi = f; } }
    int n = hull_fcnt(I, fi);
    int cur_buf = 0;
    for (int k = 0; k < n; ++k) cb_set(cb, 0, k, hull_v(I, hull_fidx(I, fi, k)));
    const int rc = hull_fcnt(R, fr);
    for (int k = 0; k < rc && n > 0; ++k) {
        const int k1 = (k + 1 == rc) ? 0 : k + 1;
        V3 v0 = hull_v(R, hull_fidx(R, fr, k)), v1 = hull_v(R, hull_fidx(R, fr, k1));
        V3 s = cross(v1 - v0, nr);
        int m = 0;
        const int src = cur_buf, dst = cur_buf ^ 1;
        V3 prev = cb_get(cb, src, n - 1); float dprev = dot(s, prev - v0);
        for (int i = 0; i < n; ++i) {
            V3 cur = cb_get(cb, src, i); float dcur = dot(s, cur - v0);
            bool in_prev = dprev <= 0.f, in_cur = dcur <= 0.f;
            if (in_prev != in_cur) {
                float t = dprev / (dprev - dcur);
                if (m < 8) { cb_set(cb, dst, m, madd(prev, cur - prev, t)); m++; }
            }
            if (in_cur) { if (m < 8) { cb_set(cb, dst, m, cur); m++; } }
            prev = cur; dprev = dcur;
        }
        n = m;
        cur_buf = dst;
    }
    // compact points on or below the reference plane into the other buffer
    const int src = cur_buf, pts = cur_buf ^ 1;
    int c = 0;
    for (int i = 0; i < n; ++i) {
        V3 p = cb_get(cb, src, i);
        float d = dot(nr, p) - dr;
        if (d <= 0.f) { cb_set(cb, pts, c, p); c++; }
    }
#define HS_PT(i) cb_get(cb, pts, (i))
#define HS_DIST(i) (dot(nr, HS_PT(i)) - dr)
    if (c <= 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i < c) { pInc[i] = HS_PT(i); dist_out[i] = HS_DIST(i); }
        return c;
    }
    int i0 = 0; float d0 = HS_DIST(0);
    for (int i = 1; i < c; ++i) { float d = HS_DIST(i); if (d < d0) { d0 = d; i0 = i; } }
    const V3 p0 = HS_PT(i0);
    int i1 = -1; float bd = -1.f;
    for (int i = 0; i < c; ++i) { if (i == i0) continue; float d2 = len2(HS_PT(i) - p0); if (d2 > bd) { bd = d2; i1 = i; } }
    const V3 p1 = HS_PT(i1);
    int i2 = -1, i3 = -1; float amax = 0.f, amin = 0.f;
    for (int i = 0; i < c; ++i) {
        if (i == i0 || i == i1) continue;
        V3 p = HS_PT(i);
        float ar = dot(cross(p0 - p, p1 - p), nr);
        if (i2 < 0 || ar > amax) { amax = ar; i2 = i; }
    }
    for (int i = 0; i < c; ++i) {
        if (i == i0 || i == i1 || i == i2) continue;
        V3 p = HS_PT(i);
        float ar = dot(cross(p0 - p, p1 - p), nr);
        if (i3 < 0 || ar < amin) { amin = ar; i3 = i; }
    }
    pInc[0] = p0; dist_out[0] = HS_DIST(i0);
    pInc[1] = p1; dist_out[1] = HS_DIST(i1);
    pInc[2] = HS_PT(i2); dist_out[2] = HS_DIST(i2);
    pInc[3] = HS_PT(i3); dist_out[3] = HS_DIST(i3);
#undef HS_PT
#undef HS_DIST
    return 4;
}

// The convex test in two stages, so that the (long, single-lane) contact generation runs only for the pairs that do
// collide, compacted over the wave's items, and neither stage has to keep the other's registers.
//
// Stage 1, sat_axes: the separating-axis search.  Two neighbouring lanes of a wave — 2k and 2k + 1, `hi` on the second —
// run it together on the SAME pair: the low lane tests A's face normals and the first half of the edge-direction
// pairs, its partner B's face normals and the second half; they exchange results with cross-lane shuffles and
// combine them exactly as the sequential loop would (strictly-greater updates, earlier axis wins ties).  The result
// is valid on the low lane:  code 0 = separated;  1 | ea << 4 | eb << 8 = edge contact along `ax`;
// 2 | face << 4 | refB << 8 = face contact with reference face `face` of B (refB) or A.
struct AxisResult { int code; V3 ax; };
// the partner lane of a pair is its neighbour (lanes 2k, 2k + 1): one DPP move instead of a trip through the LDS crossbar
HSD int pair_swap(int x) { return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false); }
HSD float pair_swap(float x) { return __int_as_float(pair_swap(__float_as_int(x))); }
HSD AxisResult sat_axes(const HullSrc &sa, const HullSrc &sb, const bool hi) {
    AxisResult res = {0, {0.f, 0.f, 0.f}};
    // ---- face normals: this lane takes the faces of X against the vertices of Y (X = A on the low lane, B on its partner)
    float bestA, bestB; int fa, fb;
    {
        const HullRef X = hull_from(hull_src_sel(hi, sb, sa)), Y = hull_from(hull_src_sel(hi, sa, sb));
        WedgeVerts wx = {}, wy = {};
        if (X.kind == HULL_WEDGE) wedge_verts(X, wx);
        if (Y.kind == HULL_WEDGE) wedge_verts(Y, wy);
        float best = 0.f; int fx = -1; int sep = 0;
        const int xnf = hull_nf(X);
        for (int f = 0; f < xnf; ++f) {
            V3 fn = hull_fn(X, f);
            float s = support_min(Y, wy, fn) - hull_fd_w(X, wx, f, fn);
            if (s > 0.f) { sep = 1; break; }
            if (fx < 0 || s > best) { best = s; fx = f; }
        }
        const float best_o = pair_swap(best); const int fx_o = pair_swap(fx);
        if (sep | pair_swap(sep)) return res;
        bestA = hi ? best_o : best; fa = hi ? fx_o : fx;
        bestB = hi ? best : best_o; fb = hi ? fx : fx_o;
    }
    const HullRef A = hull_from(sa), B = hull_from(sb);
    WedgeVerts wa = {}, wb = {};
    if (A.kind == HULL_WEDGE) wedge_verts(A, wa);
    if (B.kind == HULL_WEDGE) wedge_verts(B, wb);
    // ---- edge-direction crosses: pairs p = i * bned + j, first half on the low lane
    float bestE = 0.f; int ea = -1, eb = -1; V3 axE = {0.f, 0.f, 0.f};
    {
        V3 ab = B.c - A.c;
        const int aned = hull_ned(A), bned = hull_ned(B);
        const int tot = aned * bned, p0 = hi ? tot / 2 : 0, p1 = hi ? tot : tot / 2;
        int sep = 0;
        for (int p = p0; p < p1; ++p) {
            const int i = p / bned, j = p - i * bned;
            V3 ax = cross(hull_ed(A, i), hull_ed(B, j));
            float l2 = len2(ax);
            if (l2 < 1e-6f) continue;
            ax = ax * (1.f / sqrtf(l2));
            if (dot(ax, ab) < 0.f) ax = -ax;
            float s = support_min(B, wb, ax) - support_max(A, wa, ax);
            if (s > 0.f) { sep = 1; break; }
            if (ea < 0 || s > bestE) { bestE = s; ea = i; eb = j; axE = ax; }
        }
        if (sep | pair_swap(sep)) return res;
        const float bE_o = pair_swap(bestE); const int ea_o = pair_swap(ea), eb_o = pair_swap(eb);
        const V3 ax_o = {pair_swap(axE.x), pair_swap(axE.y), pair_swap(axE.z)};
        if (hi) return res;                   // the low lane holds the result
        // sequential semantics: the second half replaces the first half's best only if strictly greater
        if (ea_o >= 0 && (ea < 0 || bE_o > bestE)) { bestE = bE_o; ea = ea_o; eb = eb_o; axE = ax_o; }
    }
    const float bestF = fmaxf(bestA, bestB);
    if (ea >= 0 && bestE > 0.9f * bestF + 0.0025f) { res.code = 1 | (ea << 4) | (eb << 8); res.ax = axE; return res; }
    const bool refB = bestB > 0.98f * bestA + 0.00125f;
    res.code = 2 | ((refB ? fb : fa) << 4) | ((refB ? 1 : 0) << 8);
    return res;
}

// Stage 1 for the few pairs that involve the ramp's wedge (a couple per octet and substep, yet a round of their own:
// the wedge's supports walk its vertices): SIXTEEN lanes per pair, `part` = lane % 16.  A hull pair has at most 12 face
// normals and 16 edge-direction pairs, so every lane tests at most one of each; the sequential loop's "first maximum
// wins" is the group's maximum with ties to the lower index, found by a butterfly of shuffles.  Same expressions per
// axis as sat_axes, so the same result; it is valid on every lane of the group.
struct AxisCand { float s; int i; };
HSD AxisCand axis_better(AxisCand a, AxisCand b) { return (b.s > a.s || (b.s == a.s && b.i < a.i)) ? b : a; }
// all-reduce over the 16 lanes of a DPP row: pairs and quads by quad_perm, then the other quad of the half row
// (row_half_mirror) and the other half row (row_mirror) — after each step all lanes of the combined group agree
template <int CTRL> HSD AxisCand axis_dpp(AxisCand c) {
    return {__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c.s), CTRL, 0xF, 0xF, false)),
            __builtin_amdgcn_update_dpp(0, c.i, CTRL, 0xF, 0xF, false)};
}
HSD AxisCand axis_reduce16(AxisCand c) {
    c = axis_better(c, axis_dpp<0xB1>(c));      // quad_perm [1,0,3,2]
    c = axis_better(c, axis_dpp<0x4E>(c));      // quad_perm [2,3,0,1]
    c = axis_better(c, axis_dpp<0x141>(c));     // row_half_mirror
    c = axis_better(c, axis_dpp<0x140>(c));     // row_mirror
    return c;
}
HSD bool group16_any(bool x) {
    const unsigned long long b = __ballot(x);
    return ((b >> (hs_lane() & 48)) & 0xffffull) != 0ull;
}
HSD AxisResult sat_axes_wide(const HullSrc &sa, const HullSrc &sb, const int part) {
    AxisResult res = {0, {0.f, 0.f, 0.f}};
    const HullRef A = hull_from(sa), B = hull_from(sb);
    WedgeVerts wa = {}, wb = {};
    if (A.kind == HULL_WEDGE) wedge_verts(A, wa);
    if (B.kind == HULL_WEDGE) wedge_verts(B, wb);
    const float kNone = -3.0e38f;
    // ---- face normals: lanes [0, nfa) take A's faces against B's vertices, the next nfb lanes B's against A's
    float bestA, bestB; int fa, fb;
    {
        const int nfa = hull_nf(A), nfb = hull_nf(B);
        const bool mineA = part < nfa, mineB = !mineA && part < nfa + nfb;
        const int f = mineA ? part : part - nfa;
        float s = kNone;
        if (mineA) { const V3 fn = hull_fn(A, f); s = support_min(B, wb, fn) - hull_fd_w(A, wa, f, fn); }
        else if (mineB) { const V3 fn = hull_fn(B, f); s = support_min(A, wa, fn) - hull_fd_w(B, wb, f, fn); }
        if (group16_any((mineA || mineB) && s > 0.f)) return res;
        const AxisCand ca = axis_reduce16(AxisCand{mineA ? s : kNone, mineA ? f : 99});
        const AxisCand cb = axis_reduce16(AxisCand{mineB ? s : kNone, mineB ? f : 99});
        bestA = ca.s; fa = ca.i; bestB = cb.s; fb = cb.i;
    }
    // ---- edge-direction crosses: pair p = i * bned + j on lane p
    float bestE = 0.f; int ea = -1, eb = -1; V3 axE = {0.f, 0.f, 0.f};
    {
        const V3 ab = B.c - A.c;
        const int aned = hull_ned(A), bned = hull_ned(B);
        const int tot = aned * bned;
        const int i = part / bned, j = part - i * bned;
        float s = kNone; V3 ax = {0.f, 0.f, 0.f}; bool have = false;
        if (part < tot) {
            ax = cross(hull_ed(A, i), hull_ed(B, j));
            const float l2 = len2(ax);
            if (!(l2 < 1e-6f)) {
                ax = ax * (1.f / sqrtf(l2));
                if (dot(ax, ab) < 0.f) ax = -ax;
                s = support_min(B, wb, ax) - support_max(A, wa, ax);
                have = true;
            }
        }
        if (group16_any(have && s > 0.f)) return res;
        const AxisCand ce = axis_reduce16(AxisCand{have ? s : kNone, have ? part : 99});
        if (ce.i != 99) {
            const int src = (hs_lane() & 48) + ce.i;
            bestE = ce.s; ea = ce.i / bned; eb = ce.i - ea * bned;
            axE = {__shfl(ax.x, src), __shfl(ax.y, src), __shfl(ax.z, src)};
        }
    }
    const float bestF = fmaxf(bestA, bestB);
    if (ea >= 0 && bestE > 0.9f * bestF + 0.0025f) { res.code = 1 | (ea << 4) | (eb << 8); res.ax = axE; return res; }
    const bool refB = bestB > 0.98f * bestA + 0.00125f;
    res.code = 2 | ((refB ? fb : fa) << 4) | ((refB ? 1 : 0) << 8);
    return res;
}

// Stage 2, sat_contact: contact generation for a colliding pair, one lane (it owns a slice of the LDS clip scratch):
// the closest points of the two supporting edges, or the incident face clipped against the reference face.
HSD bool sat_contact(const HullSrc &sa, const HullSrc &sb, const AxisResult &res, const ClipBuf &cb, RawManifold &m) {
    if ((res.code & 3) == 1) {
        const HullRef A = hull_from(sa), B = hull_from(sb);
        const int ea = (res.code >> 4) & 15, eb = (res.code >> 8) & 15;
        const V3 axE = res.ax;
        int ia = -1, ib = -1; float va = 0.f, vb = 0.f;
        const int ane = hull_ne(A), bne = hull_ne(B);
        for (int e = 0; e < ane; ++e) {
            int v0, v1, dir; hull_edge(A, e, &v0, &v1, &dir);
            if (dir != ea) continue;
            float p = dot(axE, hull_v(A, v0)) + dot(axE, hull_v(A, v1));
            if (ia < 0 || p > va) { va = p; ia = e; }
        }
        for (int e = 0; e < bne; ++e) {
            int v0, v1, dir; hull_edge(B, e, &v0, &v1, &dir);
            if (dir != eb) continue;
            float p = dot(axE, hull_v(B, v0)) + dot(axE, hull_v(B, v1));
            if (ib < 0 || p < vb) { vb = p; ib = e; }
        }
        int a0, a1, b0, b1, dd;
        hull_edge(A, ia, &a0, &a1, &dd); hull_edge(B, ib, &b0, &b1, &dd);
        V3 c1, c2;
        closest_seg_seg(hull_v(A, a0), hull_v(A, a1), hull_v(B, b0), hull_v(B, b1), &c1, &c2);
        m.n = axE; m.np = 1; m.pA[0] = c1; m.pB[0] = c2;
        return true;
    }
    // face contact: the reference hull is chosen first so that the clipping code runs once whichever body owns the face
    V3 pinc[4]; float dist[4];
    const bool refB = ((res.code >> 8) & 1) != 0;
    const int fr = (res.code >> 4) & 15;
    const HullRef R = hull_from(hull_src_sel(refB, sb, sa)), I = hull_from(hull_src_sel(refB, sa, sb));
    const V3 nr = hull_fn(R, fr);
    const int c = clip_face_contact(R, fr, nr, I, cb, pinc, dist);
    if (c == 0) return false;
    m.n = refB ? -nr : nr; m.np = c;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (i < c) {
        const V3 on_ref = nmadd(pinc[i], nr, dist[i]);
        m.pA[i] = refB ? pinc[i] : on_ref;
        m.pB[i] = refB ? on_ref : pinc[i];
    }
    return true;
}

}  // namespace hs
