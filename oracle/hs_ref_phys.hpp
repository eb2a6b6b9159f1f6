// ORACLE — TEST INFRASTRUCTURE ONLY (see hs_ref_math.hpp header).
//
// PARITY UNPINNED.  madrona::phys (PhysicsSystem::setupBroadphaseTasks / setupPhysicsStepTasks /
// setupCleanupTasks, spliced in at src/sim.cpp:1156,1162-1163,1167-1168; BVH::traceRay used at
// src/sim.cpp:288,331,602,738,797; makeFixedJoint src/sim.cpp:354-356) is absent from the
// reference snapshot.  This is a clean-room rigid-body pipeline against the call-site
// semantics, following the published XPBD rigid-body formulation (Mueller et al. 2020,
// "Detailed Rigid Body Simulation with Extended Position Based Dynamics"):
//   per substep h = dt/4: integrate -> collide (exact convex SAT + face clipping) ->
//   position solve (normal + static friction) -> derive velocities -> velocity solve
//   (dynamic friction, restitution 0).
// Decisions that the engine would have made are listed in DESIGN.md §"Engine decisions".
// The Gauss-Seidel order is chosen so that the HIP kernels can run it in parallel without
// changing a single rounding: joints, then body-body manifolds in (i<j) order, then each
// body's manifolds against static geometry (those never couple two movable bodies).
#pragma once
#include "hs_ref_world.hpp"

namespace hsref {

constexpr float kSubstepH = (1.f / 30.f) / 4.f;
constexpr float kInvSubstepH = 120.f;
constexpr float kGravityZ = -9.8f;            // sim.cpp:1360
// Candidate pairs per world per substep (pairs whose AABBs overlap) are bounded only by the body counts, as in the
// reference (src/sim.cpp:1356-1361 sizes for every entity): 17 movable bodies give at most 136 body-body pairs, and a
// body meets at most 36 walls + 2 extra planes.  No pair is ever dropped.  (The HIP kernels keep the first 16 / 24
// candidates of a world in LDS and spill the rest to a global list — same pairs, same order, same results.)
constexpr int kMaxDDCand = kNumDSlots * (kNumDSlots - 1) / 2;          // 136 body-body candidate pairs
constexpr int kMaxSCand = kNumDSlots * (kMaxWalls + kMaxPlanes - 1);   // body-(wall | plane>=1) candidate pairs
constexpr float kMaxDepenVel = 3.f;         // m/s, rate limit for pre-existing overlap

// mgr.cpp:476-559 — inverse mass and friction per SimObject
static inline float obj_inv_mass(int32_t o) {
    switch (o) {
    case OBJ_CUBE: case OBJ_RAMP: case OBJ_BOX: return 0.5f;
    case OBJ_HIDER: case OBJ_SEEKER: case OBJ_SPHERE: return 1.f;   // (no level ever makes a sphere; the table has it)
    default: return 0.f;
    }
}
static inline float obj_mu_s(int32_t o) { return o == OBJ_PLANE ? 2.f : 0.5f; }
static inline float obj_mu_d(int32_t o) {
    switch (o) {
    case OBJ_PLANE: return 2.f;
    case OBJ_CUBE: case OBJ_WALL: return 2.f;
    case OBJ_HIDER: case OBJ_SEEKER: return 16.f;
    case OBJ_RAMP: return 1.f;
    case OBJ_BOX: return 4.f;
    default: return 0.5f;
    }
}
// Diagonal inverse inertia in the object frame (uniform density solids; the wedge's product of
// inertia and centre-of-mass offset are dropped — DESIGN.md).  Agents: x,y zeroed (mgr.cpp:577-584).
static inline V3 obj_inv_inertia(int32_t o) {
    switch (o) {
    case OBJ_CUBE: return {0.75f, 0.75f, 0.75f};                       // m=2, 2x2x2
    case OBJ_BOX: return {0.96f, 0.088235294f, 0.090566038f};          // m=2, 8x1.5x2
    case OBJ_RAMP: return {0.692307692f, 0.9f, 0.6f};                  // m=2 wedge
    case OBJ_HIDER: case OBJ_SEEKER: return {0.f, 0.f, 1.5f};          // m=1, 2x2x2, yaw only
    default: return {0.f, 0.f, 0.f};
    }
}

// ----------------------------------------------------------------------------------------
// World-space convex hulls: boxes (8 v / 6 f / 3 edge directions) and the ramp wedge
// (6 v / 5 f / 4 edge directions), data/*_collision.obj.
// ----------------------------------------------------------------------------------------
struct Hull {
    int nv, nf, ned, ne;
    V3 v[8];
    V3 fn[6]; float fd[6];
    int fcnt[6]; int fidx[6][4];
    V3 ed[4];
    int e0[12], e1[12], edir[12];
    V3 center;
    V3 lo, hi;
    bool is_box; V3 ax, ay, az, e;   // boxes: rotation columns and half extents (closed-form supports)
};

static const int kBoxFaceIdx[6][4] = {
    {0, 4, 6, 2}, {1, 3, 7, 5}, {0, 1, 5, 4}, {2, 6, 7, 3}, {0, 2, 3, 1}, {4, 5, 7, 6}};
static const int kBoxEdges[12][3] = {
    {0, 1, 0}, {2, 3, 0}, {4, 5, 0}, {6, 7, 0}, {0, 2, 1}, {1, 3, 1}, {4, 6, 1}, {5, 7, 1},
    {0, 4, 2}, {1, 5, 2}, {2, 6, 2}, {3, 7, 2}};
static const float kWedgeV[6][3] = {
    {1, 1, 1}, {1, 1, -1}, {1, -2, -1}, {-1, 1, 1}, {-1, 1, -1}, {-1, -2, -1}};
static const int kWedgeFaceCnt[5] = {4, 4, 4, 3, 3};
static const int kWedgeFaceIdx[5][4] = {{4, 1, 2, 5}, {4, 3, 0, 1}, {2, 0, 3, 5}, {1, 0, 2, 0}, {5, 3, 4, 0}};
static const float kWedgeFN[5][3] = {
    {0, 0, -1}, {0, 1, 0}, {0, -0.554700196f, 0.832050294f}, {1, 0, 0}, {-1, 0, 0}};
static const int kWedgeEdges[9][3] = {
    {4, 1, 0}, {2, 5, 0}, {3, 0, 0}, {1, 2, 1}, {5, 4, 1}, {4, 3, 2}, {0, 1, 2}, {2, 0, 3}, {5, 3, 3}};
static const float kWedgeSlant[3] = {0.f, 0.832050294f, 0.554700196f};   // (0,3,2)/sqrt(13)

static inline void hull_finish(Hull &h) {
    for (int f = 0; f < h.nf; ++f) {
        if (h.is_box) { float ei = (f >> 1) == 0 ? h.e.x : ((f >> 1) == 1 ? h.e.y : h.e.z); h.fd[f] = dot(h.fn[f], h.center) + ei; }
        else h.fd[f] = dot(h.fn[f], h.v[h.fidx[f][0]]);
    }
    V3 lo = h.v[0], hi = h.v[0];
    for (int i = 1; i < h.nv; ++i) {
        V3 p = h.v[i];
        lo = {fminf(lo.x, p.x), fminf(lo.y, p.y), fminf(lo.z, p.z)};
        hi = {fmaxf(hi.x, p.x), fmaxf(hi.y, p.y), fmaxf(hi.z, p.z)};
    }
    h.lo = lo; h.hi = hi;
}

static inline void hull_box(Hull &h, V3 c, V3 ax, V3 ay, V3 az, V3 e) {
    h.nv = 8; h.nf = 6; h.ned = 3; h.ne = 12;
    h.center = c; h.is_box = true; h.ax = ax; h.ay = ay; h.az = az; h.e = e;
    for (int i = 0; i < 8; ++i) {
        float sx = (i & 1) ? e.x : -e.x, sy = (i & 2) ? e.y : -e.y, sz = (i & 4) ? e.z : -e.z;
        h.v[i] = madd(madd(madd(c, ax, sx), ay, sy), az, sz);
    }
    h.fn[0] = -ax; h.fn[1] = ax; h.fn[2] = -ay; h.fn[3] = ay; h.fn[4] = -az; h.fn[5] = az;
    for (int f = 0; f < 6; ++f) { h.fcnt[f] = 4; for (int k = 0; k < 4; ++k) h.fidx[f][k] = kBoxFaceIdx[f][k]; }
    h.ed[0] = ax; h.ed[1] = ay; h.ed[2] = az;
    for (int i = 0; i < 12; ++i) { h.e0[i] = kBoxEdges[i][0]; h.e1[i] = kBoxEdges[i][1]; h.edir[i] = kBoxEdges[i][2]; }
    hull_finish(h);
}

static inline void hull_wedge(Hull &h, V3 c, const M3 &m) {
    h.nv = 6; h.nf = 5; h.ned = 4; h.ne = 9;
    h.center = c; h.is_box = false; h.ax = m.c0; h.ay = m.c1; h.az = m.c2; h.e = {1.f, 1.f, 1.f};
    for (int i = 0; i < 6; ++i)
        h.v[i] = madd(madd(madd(c, m.c0, kWedgeV[i][0]), m.c1, kWedgeV[i][1]), m.c2, kWedgeV[i][2]);
    for (int f = 0; f < 5; ++f) {
        h.fn[f] = (m.c0 * kWedgeFN[f][0] + m.c1 * kWedgeFN[f][1]) + m.c2 * kWedgeFN[f][2];
        h.fcnt[f] = kWedgeFaceCnt[f];
        for (int k = 0; k < 4; ++k) h.fidx[f][k] = kWedgeFaceIdx[f][k];
    }
    h.ed[0] = m.c0; h.ed[1] = m.c1; h.ed[2] = m.c2;
    h.ed[3] = (m.c0 * kWedgeSlant[0] + m.c1 * kWedgeSlant[1]) + m.c2 * kWedgeSlant[2];
    for (int i = 0; i < 9; ++i) { h.e0[i] = kWedgeEdges[i][0]; h.e1[i] = kWedgeEdges[i][1]; h.edir[i] = kWedgeEdges[i][2]; }
    hull_finish(h);
}

static inline V3 obj_half_extents(int32_t o) {
    return o == OBJ_BOX ? V3{4.f, 0.75f, 1.f} : V3{1.f, 1.f, 1.f};
}

static inline void hull_from_body(Hull &h, int32_t obj, V3 pos, Q rot) {
    M3 m = m3_from_quat(rot);
    if (obj == OBJ_RAMP) hull_wedge(h, pos, m);
    else hull_box(h, pos, m.c0, m.c1, m.c2, obj_half_extents(obj));
}
static inline void hull_from_wall(Hull &h, const WallS &w) {
    hull_box(h, {w.cx, w.cy, 1.25f}, {1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f},
             {w.hx, w.hy, 1.25f});
}
static inline bool hull_aabb_overlap(const Hull &a, const Hull &b) {
    return a.lo.x <= b.hi.x && b.lo.x <= a.hi.x && a.lo.y <= b.hi.y && b.lo.y <= a.hi.y &&
           a.lo.z <= b.hi.z && b.lo.z <= a.hi.z;
}

// ----------------------------------------------------------------------------------------
// Contact manifolds.  n points from body A toward body B; pA/pB are the world contact points
// on each surface, (pA-pB).n = penetration depth >= 0.
// ----------------------------------------------------------------------------------------
struct Manifold {
    int a, b;          // D-slots; b = -1: static geometry (B never moves)
    float muS, muD;
    V3 n;
    int np;
    V3 rA[4];          // contact point in A's local frame
    V3 rB[4];          // in B's local frame (b >= 0 only)
    float offB[4];     // b == -1: n . pB, the static surface point projected on the normal
    float lambdaN[4];
};

struct RawManifold { V3 n; int np; V3 pA[4]; V3 pB[4]; int vidx[4]; };

// Local-frame hull vertex i (data/*_collision.obj)
static inline V3 hull_local_vertex(int32_t obj, int i) {
    if (obj == OBJ_RAMP) return {kWedgeV[i][0], kWedgeV[i][1], kWedgeV[i][2]};
    V3 e = obj_half_extents(obj);
    return {(i & 1) ? e.x : -e.x, (i & 2) ? e.y : -e.y, (i & 4) ? e.z : -e.z};
}

// Hull vs infinite plane (pn.p = pd): up to the 4 deepest vertices below the plane.  The contact
// point on the hull is the vertex itself (vidx), the point on the plane its projection.
static inline bool collide_hull_plane(const Hull &A, V3 pn, float pd, RawManifold &m) {
    int np = 0; float depth[4];
    for (int i = 0; i < A.nv; ++i) {
        float dist = dot(pn, A.v[i]) - pd;
        if (!(dist < 0.f)) continue;
        float dep = -dist;
        V3 pa = A.v[i];
        V3 pb = A.v[i] - pn * dist;
        if (np < 4) { depth[np] = dep; m.pA[np] = pa; m.pB[np] = pb; m.vidx[np] = i; np++; }
        else {
            int mi = 0;
            for (int k = 1; k < 4; ++k) if (depth[k] < depth[mi]) mi = k;
            if (dep > depth[mi]) { depth[mi] = dep; m.pA[mi] = pa; m.pB[mi] = pb; m.vidx[mi] = i; }
        }
    }
    m.np = np; m.n = -pn;
    return np > 0;
}

// Boxes: centre projection -/+ projected radius; the wedge walks its 6 vertices.
static inline float box_radius(const Hull &h, V3 n) {
    return hs_fma(fabsf(dot(n, h.az)), h.e.z, hs_fma(fabsf(dot(n, h.ay)), h.e.y, fabsf(dot(n, h.ax)) * h.e.x));
}
static inline float support_min(const Hull &h, V3 n) {
    if (h.is_box) return dot(n, h.center) - box_radius(h, n);
    float s = dot(n, h.v[0]);
    for (int i = 1; i < h.nv; ++i) s = fminf(s, dot(n, h.v[i]));
    return s;
}
static inline float support_max(const Hull &h, V3 n) {
    if (h.is_box) return dot(n, h.center) + box_radius(h, n);
    float s = dot(n, h.v[0]);
    for (int i = 1; i < h.nv; ++i) s = fmaxf(s, dot(n, h.v[i]));
    return s;
}

// closest points between segments [p1,q1] and [p2,q2] (Ericson, RTCD 5.1.9)
static inline void closest_seg_seg(V3 p1, V3 q1, V3 p2, V3 q2, V3 *c1, V3 *c2) {
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

// Reference-face clipping.  R owns the reference face fr; I is the incident hull.
// Returns points on the incident hull below the reference plane (<= 4 after reduction).
static inline int clip_face_contact(const Hull &R, int fr, const Hull &I, V3 *pInc, float *dist_out) {
    V3 nr = R.fn[fr]; float dr = R.fd[fr];
    int fi = 0; float best = dot(nr, I.fn[0]);
    for (int f = 1; f < I.nf; ++f) { float d = dot(nr, I.fn[f]); if (d < best) { best = d; fi = f; } }
    V3 poly[8], tmp[8]; int n = I.fcnt[fi];
    for (int k = 0; k < n; ++k) poly[k] = I.v[I.fidx[fi][k]];
    int rc = R.fcnt[fr];
    for (int k = 0; k < rc && n > 0; ++k) {
        V3 v0 = R.v[R.fidx[fr][k]], v1 = R.v[R.fidx[fr][(k + 1) % rc]];
        V3 s = cross(v1 - v0, nr);
        int m = 0;
        V3 prev = poly[n - 1]; float dprev = dot(s, prev - v0);
        for (int i = 0; i < n; ++i) {
            V3 cur = poly[i]; float dcur = dot(s, cur - v0);
            bool in_prev = dprev <= 0.f, in_cur = dcur <= 0.f;
            if (in_prev != in_cur) {
                float t = dprev / (dprev - dcur);
                if (m < 8) tmp[m++] = madd(prev, cur - prev, t);
            }
            if (in_cur) { if (m < 8) tmp[m++] = cur; }
            prev = cur; dprev = dcur;
        }
        n = m;
        for (int i = 0; i < n; ++i) poly[i] = tmp[i];
    }
    // keep points on or below the reference plane
    V3 pts[8]; float dist[8]; int c = 0;
    for (int i = 0; i < n; ++i) {
        float d = dot(nr, poly[i]) - dr;
        if (d <= 0.f) { pts[c] = poly[i]; dist[c] = d; c++; }
    }
    if (c <= 4) {
        for (int i = 0; i < c; ++i) { pInc[i] = pts[i]; dist_out[i] = dist[i]; }
        return c;
    }
    // reduce to 4: deepest, farthest from it, then the extreme signed areas
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
    int sel[4] = {i0, i1, i2, i3};
    for (int k = 0; k < 4; ++k) { pInc[k] = pts[sel[k]]; dist_out[k] = dist[sel[k]]; }
    return 4;
}

// Exact convex-convex test: SAT over face normals and edge-direction cross products.
static inline bool collide_hulls(const Hull &A, const Hull &B, RawManifold &m) {
    float bestA = 0.f; int fa = -1;
    for (int f = 0; f < A.nf; ++f) {
        float s = support_min(B, A.fn[f]) - A.fd[f];
        if (s > 0.f) return false;
        if (fa < 0 || s > bestA) { bestA = s; fa = f; }
    }
    float bestB = 0.f; int fb = -1;
    for (int f = 0; f < B.nf; ++f) {
        float s = support_min(A, B.fn[f]) - B.fd[f];
        if (s > 0.f) return false;
        if (fb < 0 || s > bestB) { bestB = s; fb = f; }
    }
    float bestE = 0.f; int ea = -1, eb = -1; V3 axE = {0.f, 0.f, 0.f};
    V3 ab = B.center - A.center;
    for (int i = 0; i < A.ned; ++i) {
        for (int j = 0; j < B.ned; ++j) {
            V3 ax = cross(A.ed[i], B.ed[j]);
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
        // edge-edge: supporting edge of A along +axis, of B along -axis
        int sa = -1, sb = -1; float va = 0.f, vb = 0.f;
        for (int e = 0; e < A.ne; ++e) {
            if (A.edir[e] != ea) continue;
            float p = dot(axE, A.v[A.e0[e]]) + dot(axE, A.v[A.e1[e]]);
            if (sa < 0 || p > va) { va = p; sa = e; }
        }
        for (int e = 0; e < B.ne; ++e) {
            if (B.edir[e] != eb) continue;
            float p = dot(axE, B.v[B.e0[e]]) + dot(axE, B.v[B.e1[e]]);
            if (sb < 0 || p < vb) { vb = p; sb = e; }
        }
        V3 c1, c2;
        closest_seg_seg(A.v[A.e0[sa]], A.v[A.e1[sa]], B.v[B.e0[sb]], B.v[B.e1[sb]], &c1, &c2);
        m.n = axE; m.np = 1; m.pA[0] = c1; m.pB[0] = c2;
        return true;
    }
    V3 pinc[4]; float dist[4];
    if (bestB > 0.98f * bestA + 0.00125f) {
        int c = clip_face_contact(B, fb, A, pinc, dist);
        if (c == 0) return false;
        V3 nr = B.fn[fb];
        m.n = -nr; m.np = c;
        for (int i = 0; i < c; ++i) { m.pA[i] = pinc[i]; m.pB[i] = nmadd(pinc[i], nr, dist[i]); }
    } else {
        int c = clip_face_contact(A, fa, B, pinc, dist);
        if (c == 0) return false;
        V3 nr = A.fn[fa];
        m.n = nr; m.np = c;
        for (int i = 0; i < c; ++i) { m.pB[i] = pinc[i]; m.pA[i] = nmadd(pinc[i], nr, dist[i]); }
    }
    return true;
}

// ----------------------------------------------------------------------------------------
// XPBD solver
// ----------------------------------------------------------------------------------------
// World-space inverse inertia R diag(invI) R^T (symmetric, 6 values).  It is evaluated once per
// manifold (and per joint) from the body's rotation at that moment and kept while the manifold's
// contact points are solved.
struct Sym3 { float xx, xy, xz, yy, yz, zz; };
static inline Sym3 world_inv_inertia(Q q, V3 invI) {
    M3 m = m3_from_quat(q);
    V3 r0 = m.c0 * invI.x, r1 = m.c1 * invI.y, r2 = m.c2 * invI.z;
    Sym3 s;
    s.xx = hs_fma(r2.x, m.c2.x, hs_fma(r1.x, m.c1.x, r0.x * m.c0.x));
    s.xy = hs_fma(r2.x, m.c2.y, hs_fma(r1.x, m.c1.y, r0.x * m.c0.y));
    s.xz = hs_fma(r2.x, m.c2.z, hs_fma(r1.x, m.c1.z, r0.x * m.c0.z));
    s.yy = hs_fma(r2.y, m.c2.y, hs_fma(r1.y, m.c1.y, r0.y * m.c0.y));
    s.yz = hs_fma(r2.y, m.c2.z, hs_fma(r1.y, m.c1.z, r0.y * m.c0.z));
    s.zz = hs_fma(r2.z, m.c2.z, hs_fma(r1.z, m.c1.z, r0.z * m.c0.z));
    return s;
}
static inline V3 sym_mul(const Sym3 &s, V3 v) {
    return {hs_fma(s.xz, v.z, hs_fma(s.xy, v.y, s.xx * v.x)), hs_fma(s.yz, v.z, hs_fma(s.yy, v.y, s.xy * v.x)),
            hs_fma(s.zz, v.z, hs_fma(s.yz, v.y, s.xz * v.x))};
}
struct BodyMass { float invM; V3 invI; Sym3 Iw; };

static inline BodyMass body_mass(const DBody &b) {
    if (b.objType == OBJ_NONE || b.response != RESP_DYNAMIC) return {0.f, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
    V3 invI = obj_inv_inertia(b.objType);
    return {obj_inv_mass(b.objType), invI, world_inv_inertia(b.rot, invI)};
}
static inline V3 apply_inv_inertia(const BodyMass &bm, V3 v) { return sym_mul(bm.Iw, v); }
// For a direction d that is NOT normalised: |d|^2 times the generalised inverse mass along d/|d|
// (d2 = |d|^2).  The friction corrections below are written with it so that they need no sqrt / divide to
// normalise the tangent: correction along t = d/|d| of size |d| / w(t) equals d * (|d|^2 / w'(d)).
static inline float gen_inv_mass_sq(const BodyMass &bm, V3 r, V3 d, float d2) {
    V3 rd = cross(r, d);
    return hs_fma(bm.invM, d2, dot(rd, sym_mul(bm.Iw, rd)));
}
static inline float gen_inv_mass(const BodyMass &bm, V3 r, V3 n) {
    V3 rn = cross(r, n);
    return dot_add(rn, sym_mul(bm.Iw, rn), bm.invM);
}
// q += 0.5 * (0,dth) * q, then — for small updates — ONE Newton step of 1/sqrt(|q|^2) from 1 instead of an exact
// normalisation:
// |q|^2 = 1 + |dth|^2/4 after the update, so the step leaves a norm error of 3/8 (|dth|^2/4)^2 (< 2e-5 even for a
// body tumbling at 20 rad/s) that the next update corrects again; it costs 4 multiplies instead of sqrt + divide
// in the innermost loop of the solver.
static inline Q quat_add_rotation(Q q, V3 dth) {
    Q dq = qmul(Q{0.f, dth.x, dth.y, dth.z}, q);
    Q r = {hs_fma(0.5f, dq.w, q.w), hs_fma(0.5f, dq.x, q.x), hs_fma(0.5f, dq.y, q.y), hs_fma(0.5f, dq.z, q.z)};
    const float n2 = hs_fma(r.z, r.z, hs_fma(r.y, r.y, hs_fma(r.x, r.x, r.w * r.w)));
    // small updates (|dth| < 0.2 rad: every contact correction, ordinary integration); a joint that snaps a badly
    // misaligned body round can turn it by radians in one go and gets the exact normalisation
    const float k = n2 < 1.01f ? hs_fma(-0.5f, n2, 1.5f) : 1.f / sqrtf(n2);
    return {r.w * k, r.x * k, r.y * k, r.z * k};
}
// positional impulse p applied at rA (on A, gets -p) and rB (on B, gets +p); r's are world offsets
static inline void apply_pos_impulse(DBody *A, const BodyMass &ma, V3 rA, DBody *B, const BodyMass &mb,
                                     V3 rB, V3 p) {
    if (ma.invM != 0.f || ma.invI.z != 0.f || ma.invI.x != 0.f || ma.invI.y != 0.f) {
        A->pos = nmadd(A->pos, p, ma.invM);
        V3 dth = apply_inv_inertia(ma, cross(rA, p));
        A->rot = quat_add_rotation(A->rot, -dth);
    }
    if (B && (mb.invM != 0.f || mb.invI.z != 0.f || mb.invI.x != 0.f || mb.invI.y != 0.f)) {
        B->pos = madd(B->pos, p, mb.invM);
        V3 dth = apply_inv_inertia(mb, cross(rB, p));
        B->rot = quat_add_rotation(B->rot, dth);
    }
}

static inline void solve_manifold_positions(World &w, Manifold &m) {
    DBody *A = &w.d[m.a];
    DBody *B = m.b >= 0 ? &w.d[m.b] : nullptr;
    BodyMass ma = body_mass(*A);
    BodyMass mb = B ? body_mass(*B) : BodyMass{0.f, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
    const V3 n = m.n;
    for (int j = 0; j < m.np; ++j) {
        V3 rAw = qrot(A->rot, m.rA[j]);
        V3 pA = A->pos + rAw;
        V3 rBw = B ? qrot(B->rot, m.rB[j]) : V3{0.f, 0.f, 0.f};
        V3 pB = B ? B->pos + rBw : V3{0.f, 0.f, 0.f};
        float d = B ? dot(pA - pB, n) : dot_add(pA, n, -m.offB[j]);
        if (!(d > 0.f)) continue;
        // Overlap that already existed at the start of the substep (spawn overlaps after 20
        // rejected placements, level_gen.cpp:146) is resolved at kMaxDepenVel instead of in one
        // substep; penetration gained during this substep is always resolved in full.
        V3 pAprev = A->prevPos + qrot(A->prevRot, m.rA[j]);
        V3 pBprev = B ? B->prevPos + qrot(B->prevRot, m.rB[j]) : V3{0.f, 0.f, 0.f};
        float dprev = B ? dot(pAprev - pBprev, n) : dot_add(pAprev, n, -m.offB[j]);
        float excess = dprev - kMaxDepenVel * kSubstepH;
        if (excess > 0.f) d = d - excess;
        if (!(d > 0.f)) continue;
        float wA = gen_inv_mass(ma, rAw, n);
        float wB = B ? gen_inv_mass(mb, rBw, n) : 0.f;
        float wsum = wA + wB;
        if (!(wsum > 0.f)) continue;
        float lam = d / wsum;
        m.lambdaN[j] += lam;
        apply_pos_impulse(A, ma, rAw, B, mb, rBw, n * lam);
        // static friction: undo the tangential drift of the contact points over this substep
        rAw = qrot(A->rot, m.rA[j]);
        pA = A->pos + rAw;
        V3 dp;
        if (B) {
            rBw = qrot(B->rot, m.rB[j]);
            pB = B->pos + rBw;
            dp = (pA - pAprev) - (pB - pBprev);
        } else {
            dp = pA - pAprev;
        }
        V3 dpt = nmadd(dp, n, dot(dp, n));
        float lt2 = len2(dpt);
        if (lt2 > 1e-12f) {
            float wtA = gen_inv_mass_sq(ma, rAw, dpt, lt2);
            float wtB = B ? gen_inv_mass_sq(mb, rBw, dpt, lt2) : 0.f;
            float wts = wtA + wtB;
            if (wts > 0.f) {
                // static friction holds while |dpt| / w(t) < muS * lam  <=>  lt2^3 < (muS * lam * wts)^2
                float lim = (m.muS * lam) * wts;
                if ((lt2 * lt2) * lt2 < lim * lim) apply_pos_impulse(A, ma, rAw, B, mb, rBw, dpt * (lt2 / wts));
            }
        }
    }
}

// Ground-plane manifold of a YAW-ONLY body (an agent: inverse inertia x, y = 0, src/mgr.cpp:577-584).  Such a body cannot
// tilt, so its (up to four) floor contacts cannot be resolved one after the other: the first would take the whole normal
// correction — and with it the whole friction budget of the substep, at a lever arm, so that a straight push spun the agent
// up (tests/test_oracle_first_principles.py::test_a_straight_push_does_not_spin_the_agent).  Instead the penetrations of
// all points are evaluated BEFORE any correction; the body is lifted once, by the deepest of them, along the normal
// (translation only: for a body that cannot tilt the plane's reaction passes through its centre); and the normal
// multiplier lam = d / invM is shared equally by the touching points, each of which then gets its own static-friction
// correction — at its own lever arm, limited by muS times its share — exactly as in solve_manifold_positions.  The
// velocity pass is the ordinary one with the shared multipliers.
static inline void solve_ground_positions_yaw_only(World &w, Manifold &m) {
    DBody *A = &w.d[m.a];
    BodyMass ma = body_mass(*A);
    const BodyMass mb = {0.f, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
    const V3 n = m.n;
    float dj[4]; V3 pAprev[4];
    int k = 0; float dmax = 0.f;
    for (int j = 0; j < m.np; ++j) {
        V3 pA = A->pos + qrot(A->rot, m.rA[j]);
        float d = dot_add(pA, n, -m.offB[j]);
        pAprev[j] = A->prevPos + qrot(A->prevRot, m.rA[j]);
        if (d > 0.f) {
            float excess = dot_add(pAprev[j], n, -m.offB[j]) - kMaxDepenVel * kSubstepH;
            if (excess > 0.f) d = d - excess;
        }
        dj[j] = d;
        if (d > 0.f) { k++; dmax = fmaxf(dmax, d); }
    }
    if (k == 0 || !(ma.invM > 0.f)) return;
    const float lamT = dmax / ma.invM;
    A->pos = nmadd(A->pos, n * lamT, ma.invM);
    const float share = lamT / (float)k;
    for (int j = 0; j < m.np; ++j) {
        if (!(dj[j] > 0.f)) continue;
        m.lambdaN[j] += share;
        V3 rAw = qrot(A->rot, m.rA[j]);
        V3 dp = (A->pos + rAw) - pAprev[j];
        V3 dpt = nmadd(dp, n, dot(dp, n));
        float lt2 = len2(dpt);
        if (lt2 > 1e-12f) {
            float wts = gen_inv_mass_sq(ma, rAw, dpt, lt2);
            if (wts > 0.f) {
                float lim = (m.muS * share) * wts;
                if ((lt2 * lt2) * lt2 < lim * lim) apply_pos_impulse(A, ma, rAw, nullptr, mb, V3{0.f, 0.f, 0.f}, dpt * (lt2 / wts));
            }
        }
    }
}
static inline bool yaw_only(int32_t obj) { V3 i = obj_inv_inertia(obj); return i.x == 0.f && i.y == 0.f && i.z != 0.f; }

static inline void apply_vel_impulse(DBody *A, const BodyMass &ma, V3 rA, DBody *B, const BodyMass &mb,
                                     V3 rB, V3 p) {   // A gets +p, B gets -p
    A->lin = madd(A->lin, p, ma.invM);
    A->ang = A->ang + apply_inv_inertia(ma, cross(rA, p));
    if (B) {
        B->lin = nmadd(B->lin, p, mb.invM);
        B->ang = B->ang - apply_inv_inertia(mb, cross(rB, p));
    }
}

static inline void solve_manifold_velocities(World &w, const Manifold &m) {
    DBody *A = &w.d[m.a];
    DBody *B = m.b >= 0 ? &w.d[m.b] : nullptr;
    BodyMass ma = body_mass(*A);
    BodyMass mb = B ? body_mass(*B) : BodyMass{0.f, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
    const V3 n = m.n;
    for (int j = 0; j < m.np; ++j) {
        float lamN = m.lambdaN[j];
        if (!(lamN > 0.f)) continue;
        V3 rAw = qrot(A->rot, m.rA[j]);
        V3 rBw = B ? qrot(B->rot, m.rB[j]) : V3{0.f, 0.f, 0.f};
        // bodies with Static response keep a stale Velocity; it does not take part
        V3 v = {0.f, 0.f, 0.f};
        if (ma.invM + ma.invI.x + ma.invI.y + ma.invI.z != 0.f) v = cross_add(A->ang, rAw, A->lin);
        if (B && mb.invM + mb.invI.x + mb.invI.y + mb.invI.z != 0.f) v = v - cross_add(B->ang, rBw, B->lin);
        float vn = dot(n, v);
        V3 vt = nmadd(v, n, vn);
        float vt2 = len2(vt);
        V3 dv = -(n * vn);                      // restitution 0: kill the normal component
        if (vt2 > 1e-18f) {
            // dynamic friction: |dv_t| <= h * muD * f_n with f_n = lamN / h^2, i.e. muD * lamN / h
            float vtl = sqrtf(vt2);
            float mag = fminf((m.muD * lamN) * kInvSubstepH, vtl);
            dv = nmadd(dv, vt, mag / vtl);
        }
        float dv2 = len2(dv);
        if (!(dv2 > 1e-18f)) continue;
        float wA = gen_inv_mass_sq(ma, rAw, dv, dv2);
        float wB = B ? gen_inv_mass_sq(mb, rBw, dv, dv2) : 0.f;
        float ws = wA + wB;
        if (!(ws > 0.f)) continue;
        apply_vel_impulse(A, ma, rAw, B, mb, rBw, dv * (dv2 / ws));
    }
}

// Fixed "grab" joint between agent body A and grabbed body B (sim.cpp:343-356): angular part
// aligns qA*attach1 with qB*attach2, positional part pins A's anchor (r1 + separation*fwd) to B's r2.
static inline void solve_grab_joint(World &w, int agent) {
    GrabJoint &g = w.grab[agent];
    if (g.other < 0) return;
    DBody *A = &w.d[kAgentSlot0 + agent];
    DBody *B = &w.d[g.other];
    BodyMass ma = body_mass(*A), mb = body_mass(*B);
    {   // angular
        Q qa = qmul(A->rot, g.attach1), qb = qmul(B->rot, g.attach2);
        Q dq = qmul(qa, qinv(qb));
        V3 dphi = {2.f * dq.x, 2.f * dq.y, 2.f * dq.z};
        if (dq.w < 0.f) dphi = -dphi;
        float th2 = len2(dphi);
        if (th2 > 1e-12f) {
            float th = sqrtf(th2);
            V3 ax = dphi * (1.f / th);
            float wA = dot(ax, sym_mul(ma.Iw, ax));
            float wB = dot(ax, sym_mul(mb.Iw, ax));
            float ws = wA + wB;
            if (ws > 0.f) {
                V3 p = ax * (th / ws);
                A->rot = quat_add_rotation(A->rot, -apply_inv_inertia(ma, p));
                B->rot = quat_add_rotation(B->rot, apply_inv_inertia(mb, p));
            }
        }
    }
    {   // positional
        V3 anchorA = g.r1 + V3{0.f, g.separation, 0.f};     // + separation * fwd (fwd = +y)
        V3 rAw = qrot(A->rot, anchorA), rBw = qrot(B->rot, g.r2);
        V3 dx = (A->pos + rAw) - (B->pos + rBw);
        float c2 = len2(dx);
        if (c2 > 1e-12f) {
            float c = sqrtf(c2);
            V3 n = dx * (1.f / c);
            float ws = gen_inv_mass(ma, rAw, n) + gen_inv_mass(mb, rBw, n);
            if (ws > 0.f) apply_pos_impulse(A, ma, rAw, B, mb, rBw, n * (c / ws));
        }
    }
}

static inline void integrate_body(DBody &b) {
    const float h = kSubstepH;
    b.prevPos = b.pos; b.prevRot = b.rot;
    if (b.objType == OBJ_NONE || b.response != RESP_DYNAMIC) return;
    float invM = obj_inv_mass(b.objType);
    V3 invI = obj_inv_inertia(b.objType);
    b.lin = madd(b.lin, madd(V3{0.f, 0.f, kGravityZ}, b.extForce, invM), h);
    b.pos = madd(b.pos, b.lin, h);
    Q qi = qinv(b.rot);
    V3 wl = qrot(qi, b.ang), tl = qrot(qi, b.extTorque);
    V3 I = {invI.x > 0.f ? 1.f / invI.x : 0.f, invI.y > 0.f ? 1.f / invI.y : 0.f,
            invI.z > 0.f ? 1.f / invI.z : 0.f};
    V3 Iw = mulc(I, wl);
    wl = madd(wl, mulc(invI, tl - cross(wl, Iw)), h);
    b.ang = qrot(b.rot, wl);
    b.rot = quat_add_rotation(b.rot, b.ang * h);
}

static inline void derive_velocity(DBody &b) {
    const float h = kSubstepH;
    if (b.objType == OBJ_NONE || b.response != RESP_DYNAMIC) return;
    b.lin = (b.pos - b.prevPos) * (1.f / h);
    Q dq = qmul(b.rot, qinv(b.prevRot));
    V3 wv = V3{dq.x, dq.y, dq.z} * (2.f / h);
    b.ang = dq.w >= 0.f ? wv : -wv;
}

static inline void manifold_from_raw(const World &w, Manifold &m, int a, int b, int32_t objB,
                                     const RawManifold &raw, bool plane = false) {
    const DBody &A = w.d[a];
    m.a = a; m.b = b; m.n = raw.n; m.np = raw.np;
    m.muS = 0.5f * (obj_mu_s(A.objType) + obj_mu_s(objB));
    m.muD = 0.5f * (obj_mu_d(A.objType) + obj_mu_d(objB));
    Q qai = qinv(A.rot);
    for (int j = 0; j < raw.np; ++j) {
        m.rA[j] = plane ? hull_local_vertex(A.objType, raw.vidx[j]) : qrot(qai, raw.pA[j] - A.pos);
        if (b >= 0) { m.rB[j] = qrot(qinv(w.d[b].rot), raw.pB[j] - w.d[b].pos); m.offB[j] = 0.f; }
        else { m.rB[j] = {0.f, 0.f, 0.f}; m.offB[j] = dot(raw.pB[j], raw.n); }
        m.lambdaN[j] = 0.f;
    }
}

// One substep for one world.
static inline void physics_substep(World &w) {
    for (int i = 0; i < kNumDSlots; ++i) integrate_body(w.d[i]);

    Hull hulls[kNumDSlots];
    for (int i = 0; i < kNumDSlots; ++i)
        if (w.d[i].objType != OBJ_NONE) hull_from_body(hulls[i], w.d[i].objType, w.d[i].pos, w.d[i].rot);

    // --- candidate pairs (the "broadphase": all-pairs AABB tests)
    int ddA[kMaxDDCand], ddB[kMaxDDCand]; int ndd = 0;
    for (int i = 0; i < kNumDSlots; ++i) {
        if (w.d[i].objType == OBJ_NONE) continue;
        for (int j = i + 1; j < kNumDSlots; ++j) {
            if (w.d[j].objType == OBJ_NONE) continue;
            if (w.d[i].response != RESP_DYNAMIC && w.d[j].response != RESP_DYNAMIC) continue;
            if (!hull_aabb_overlap(hulls[i], hulls[j])) continue;
            ddA[ndd] = i; ddB[ndd] = j; ndd++;
        }
    }
    // static candidates per movable body: planes 1.. (always), then walls whose AABB overlaps;
    // plane 0 (the ground, present in every level) has a dedicated manifold per body.
    int scBody[kMaxSCand], scStatic[kMaxSCand]; int nsc = 0;
    for (int i = 0; i < kNumDSlots; ++i) {
        if (w.d[i].objType == OBJ_NONE || w.d[i].response != RESP_DYNAMIC) continue;
        for (int p = 1; p < w.numPlanes; ++p)
            { scBody[nsc] = i; scStatic[nsc] = kMaxWalls + p; nsc++; }
        const Hull &hb = hulls[i];
        for (int k = 0; k < w.numWalls; ++k) {
            const WallS &ws = w.walls[k];
            if (!(hb.lo.x <= ws.cx + ws.hx && ws.cx - ws.hx <= hb.hi.x &&
                  hb.lo.y <= ws.cy + ws.hy && ws.cy - ws.hy <= hb.hi.y && hb.lo.z <= 2.5f && 0.f <= hb.hi.z))
                continue;
            scBody[nsc] = i; scStatic[nsc] = k; nsc++;
        }
    }

    // --- narrowphase
    static thread_local Manifold dd[kMaxDDCand], sc[kMaxSCand], ground[kNumDSlots];
    RawManifold raw;
    for (int k = 0; k < ndd; ++k) {
        dd[k].np = 0;
        if (collide_hulls(hulls[ddA[k]], hulls[ddB[k]], raw))
            manifold_from_raw(w, dd[k], ddA[k], ddB[k], w.d[ddB[k]].objType, raw);
    }
    for (int i = 0; i < kNumDSlots; ++i) {
        ground[i].np = 0;
        if (w.d[i].objType == OBJ_NONE || w.d[i].response != RESP_DYNAMIC || w.numPlanes < 1) continue;
        if (collide_hull_plane(hulls[i], w.planes[0].n, w.planes[0].d, raw))
            manifold_from_raw(w, ground[i], i, -1, OBJ_PLANE, raw, true);
    }
    for (int k = 0; k < nsc; ++k) {
        sc[k].np = 0;
        int i = scBody[k], st = scStatic[k];
        if (st >= kMaxWalls) {
            const PlaneS &pl = w.planes[st - kMaxWalls];
            if (collide_hull_plane(hulls[i], pl.n, pl.d, raw)) manifold_from_raw(w, sc[k], i, -1, OBJ_PLANE, raw, true);
        } else {
            Hull hw; hull_from_wall(hw, w.walls[st]);
            if (collide_hulls(hulls[i], hw, raw)) manifold_from_raw(w, sc[k], i, -1, OBJ_WALL, raw);
        }
    }

    // --- position solve: joints, body-body in pair order, then each body's static manifolds
    for (int a = 0; a < kMaxAgents; ++a) if (w.agentActive[a]) solve_grab_joint(w, a);
    for (int k = 0; k < ndd; ++k) if (dd[k].np > 0) solve_manifold_positions(w, dd[k]);
    for (int i = 0; i < kNumDSlots; ++i) {
        if (ground[i].np > 0) {
            if (yaw_only(w.d[i].objType)) solve_ground_positions_yaw_only(w, ground[i]);
            else solve_manifold_positions(w, ground[i]);
        }
        for (int k = 0; k < nsc; ++k) if (scBody[k] == i && sc[k].np > 0) solve_manifold_positions(w, sc[k]);
    }

    for (int i = 0; i < kNumDSlots; ++i) derive_velocity(w.d[i]);

    for (int k = 0; k < ndd; ++k) if (dd[k].np > 0) solve_manifold_velocities(w, dd[k]);
    for (int i = 0; i < kNumDSlots; ++i) {
        if (ground[i].np > 0) solve_manifold_velocities(w, ground[i]);
        for (int k = 0; k < nsc; ++k) if (scBody[k] == i && sc[k].np > 0) solve_manifold_velocities(w, sc[k]);
    }
}

// ----------------------------------------------------------------------------------------
// Ray casting (BVH::traceRay call sites).  Closest front-face entry over every body; a ray that
// starts inside a hull does not hit that hull (the agents' own cubes: sim.cpp:579,720).
// Returns the body id: D-slot (0..16), 100+wall, 200+plane, or -1.
// ----------------------------------------------------------------------------------------
constexpr int kHitWallBase = 100;
constexpr int kHitPlaneBase = 200;

// slab test against a box given in its own frame (origin o, dir d local); entry t or -1
static inline float ray_box_local(V3 o, V3 d, V3 e) {
    float tn = -3.0e38f, tf = 3.0e38f;
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z}, ee[3] = {e.x, e.y, e.z};
    for (int k = 0; k < 3; ++k) {
        if (dd[k] == 0.f) { if (oo[k] < -ee[k] || oo[k] > ee[k]) return -1.f; continue; }
        // slab in centre / extent form: entry = -o/d - e/|d|, exit = -o/d + e/|d| (no near/far swap)
        float inv = 1.f / dd[k];
        const float r = ee[k] * fabsf(inv);
        tn = fmaxf(tn, hs_fma(-oo[k], inv, -r)); tf = fminf(tf, hs_fma(-oo[k], inv, r));
    }
    if (tn > tf || tn < 0.f) return -1.f;
    return tn;
}
static inline float ray_wedge_local(V3 o, V3 d) {
    float tn = -3.0e38f, tf = 3.0e38f;
    // plane offsets of the wedge faces in its own frame: n.p = off
    const float off[5] = {1.f, 1.f, 0.277350098f, 1.f, 1.f};
    for (int f = 0; f < 5; ++f) {
        V3 n = {kWedgeFN[f][0], kWedgeFN[f][1], kWedgeFN[f][2]};
        float dist = dot(n, o) - off[f];
        float dn = dot(n, d);
        if (dn == 0.f) { if (dist > 0.f) return -1.f; continue; }
        float t = -dist / dn;
        if (dn < 0.f) tn = fmaxf(tn, t); else tf = fminf(tf, t);
    }
    if (tn > tf || tn < 0.f) return -1.f;
    return tn;
}

static inline int trace_ray(const World &w, V3 o, V3 d, float tmax, float *t_out) {
    int hit = -1; float best = tmax;
    for (int i = 0; i < kNumDSlots; ++i) {
        const DBody &b = w.d[i];
        if (b.objType == OBJ_NONE) continue;
        Q qi = qinv(b.rot);
        V3 ol = qrot(qi, o - b.pos), dl = qrot(qi, d);
        float t = b.objType == OBJ_RAMP ? ray_wedge_local(ol, dl) : ray_box_local(ol, dl, obj_half_extents(b.objType));
        if (t >= 0.f && t <= best && (hit < 0 || t < best)) { best = t; hit = i; }
    }
    for (int k = 0; k < w.numWalls; ++k) {
        const WallS &ws = w.walls[k];
        V3 ol = {o.x - ws.cx, o.y - ws.cy, o.z - 1.25f};
        float t = ray_box_local(ol, d, {ws.hx, ws.hy, 1.25f});
        if (t >= 0.f && t <= best && (hit < 0 || t < best)) { best = t; hit = kHitWallBase + k; }
    }
    for (int p = 0; p < w.numPlanes; ++p) {
        const PlaneS &pl = w.planes[p];
        float dn = dot(pl.n, d);
        if (!(dn < 0.f)) continue;
        float dist = dot(pl.n, o) - pl.d;
        if (dist < 0.f) continue;
        float t = -dist / dn;
        if (t >= 0.f && t <= best && (hit < 0 || t < best)) { best = t; hit = kHitPlaneBase + p; }
    }
    *t_out = best;
    return hit;
}

}  // namespace hsref
