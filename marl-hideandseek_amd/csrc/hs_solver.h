// XPBD contact / joint solver pieces shared by the phases of the physics kernel
// (hs_k_pipeline.h).  Replaces madrona::phys' solver (spliced in at src/sim.cpp:1162-1163; engine
// source absent — DESIGN.md "Engine decisions"): per contact point a normal correction with static
// friction in the position pass, dynamic friction and restitution 0 in the velocity pass.
#pragma once
#include "hs_state.h"
#include "hs_rays.h"
#include "hs_collide.h"

namespace hs {

// Contact manifolds live in an HBM workspace (L2-resident): written once by the lane that ran the
// convex test, read by the lane that solves them.  16-byte multiples so they move as dwordx4.
struct alignas(16) ManDD { int a, b, np; float muS, muD; float n[3]; float rA[4][3]; float rB[4][3]; float lam[4]; };
struct alignas(16) ManS { int np; float muS, muD; float n[3]; float pad[2]; float rA[4][3]; float offB[4]; float lam[4]; };
static_assert(sizeof(ManDD) == 144 && sizeof(ManS) == 112, "manifold layout");

HSD bool has_mass_i(float invM, V3 invI) { return invM != 0.f || invI.z != 0.f || invI.x != 0.f || invI.y != 0.f; }

// World-space inverse inertia R diag(invI) R^T (symmetric): evaluated once per manifold / joint from the
// body's rotation at that moment and kept while the manifold's contact points are solved.
struct Sym3 { float xx, xy, xz, yy, yz, zz; };
struct BodyS { V3 pos; Q rot; V3 ppos; Q prot; V3 lin, ang; float invM; V3 invI; Sym3 Iw; };

HSD Sym3 world_inv_inertia(Q q, V3 invI) {
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
HSD V3 sym_mul(const Sym3 &s, V3 v) {
    return {hs_fma(s.xz, v.z, hs_fma(s.xy, v.y, s.xx * v.x)), hs_fma(s.yz, v.z, hs_fma(s.yy, v.y, s.xy * v.x)),
            hs_fma(s.zz, v.z, hs_fma(s.yz, v.y, s.xz * v.x))};
}
// call at the start of every manifold / joint (the oracle re-evaluates body_mass there)
HSD void body_refresh_inertia(BodyS &b) {
    b.Iw = has_mass_i(b.invM, b.invI) ? world_inv_inertia(b.rot, b.invI) : Sym3{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
}

HSD V3 ld3(const float *p) { return {p[0], p[1], p[2]}; }
HSD Q ld4(const float *p) { return {p[0], p[1], p[2], p[3]}; }
HSD void st3(float *p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
HSD void st4(float *p, Q q) { p[0] = q.w; p[1] = q.x; p[2] = q.y; p[3] = q.z; }

HSD V3 apply_inv_inertia(const BodyS &b, V3 v) { return sym_mul(b.Iw, v); }
// |d|^2 times the generalised inverse mass along d / |d| for an unnormalised d (d2 = |d|^2): lets the friction
// corrections skip the sqrt / divide of normalising the tangent (same expressions as the oracle's)
HSD float gen_inv_mass_sq(const BodyS &b, V3 r, V3 d, float d2) {
    V3 rd = cross(r, d);
    return hs_fma(b.invM, d2, dot(rd, sym_mul(b.Iw, rd)));
}
HSD float gen_inv_mass(const BodyS &b, V3 r, V3 n) {
    V3 rn = cross(r, n);
    return dot_add(rn, sym_mul(b.Iw, rn), b.invM);
}
// q += 0.5 * (0,dth) * q, then one Newton step of 1/sqrt(|q|^2) from 1 (DESIGN.md "Engine decisions")
HSD Q quat_add_rotation(Q q, V3 dth) {
    Q dq = qmul(Q{0.f, dth.x, dth.y, dth.z}, q);
    Q r = {hs_fma(0.5f, dq.w, q.w), hs_fma(0.5f, dq.x, q.x), hs_fma(0.5f, dq.y, q.y), hs_fma(0.5f, dq.z, q.z)};
    const float n2 = hs_fma(r.z, r.z, hs_fma(r.y, r.y, hs_fma(r.x, r.x, r.w * r.w)));
    // small updates (|dth| < 0.2 rad: every contact correction, ordinary integration); a joint that snaps a badly
    // misaligned body round can turn it by radians in one go and gets the exact normalisation
    const float k = n2 < 1.01f ? hs_fma(-0.5f, n2, 1.5f) : 1.f / sqrtf(n2);
    return {r.w * k, r.x * k, r.y * k, r.z * k};
}
HSD bool has_mass(const BodyS &b) { return b.invM != 0.f || b.invI.z != 0.f || b.invI.x != 0.f || b.invI.y != 0.f; }

template <bool HAS_B>
HSD void apply_pos_impulse(BodyS &A, V3 rA, BodyS &B, V3 rB, V3 p) {
    if (has_mass(A)) {
        A.pos = nmadd(A.pos, p, A.invM);
        V3 dth = apply_inv_inertia(A, cross(rA, p));
        A.rot = quat_add_rotation(A.rot, -dth);
    }
    if (HAS_B && has_mass(B)) {
        B.pos = madd(B.pos, p, B.invM);
        V3 dth = apply_inv_inertia(B, cross(rB, p));
        B.rot = quat_add_rotation(B.rot, dth);
    }
}

// One contact point of the XPBD position pass (normal + static friction).  Returns the normal
// multiplier added this pass.
template <bool HAS_B>
HSD float solve_point_position(BodyS &A, BodyS &B, V3 n, V3 rAl, V3 rBl, float offB, float muS) {
    V3 rAw = qrot(A.rot, rAl);
    V3 pA = A.pos + rAw;
    V3 rBw = HAS_B ? qrot(B.rot, rBl) : V3{0.f, 0.f, 0.f};
    V3 pB = HAS_B ? B.pos + rBw : V3{0.f, 0.f, 0.f};
    float d = HAS_B ? dot(pA - pB, n) : dot_add(pA, n, -offB);
    if (!(d > 0.f)) return 0.f;
    V3 pAprev = A.ppos + qrot(A.prot, rAl);
    V3 pBprev = HAS_B ? B.ppos + qrot(B.prot, rBl) : V3{0.f, 0.f, 0.f};
    float dprev = HAS_B ? dot(pAprev - pBprev, n) : dot_add(pAprev, n, -offB);
    float excess = dprev - kMaxDepenVel * kSubstepH;
    if (excess > 0.f) d = d - excess;
    if (!(d > 0.f)) return 0.f;
    float wA = gen_inv_mass(A, rAw, n);
    float wB = HAS_B ? gen_inv_mass(B, rBw, n) : 0.f;
    float wsum = wA + wB;
    if (!(wsum > 0.f)) return 0.f;
    float lam = d / wsum;
    apply_pos_impulse<HAS_B>(A, rAw, B, rBw, n * lam);
    rAw = qrot(A.rot, rAl);
    pA = A.pos + rAw;
    V3 dp;
    if (HAS_B) {
        rBw = qrot(B.rot, rBl);
        pB = B.pos + rBw;
        dp = (pA - pAprev) - (pB - pBprev);
    } else {
        dp = pA - pAprev;
    }
    V3 dpt = nmadd(dp, n, dot(dp, n));
    float lt2 = len2(dpt);
    if (lt2 > 1e-12f) {
        float wtA = gen_inv_mass_sq(A, rAw, dpt, lt2);
        float wtB = HAS_B ? gen_inv_mass_sq(B, rBw, dpt, lt2) : 0.f;
        float wts = wtA + wtB;
        if (wts > 0.f) {
            // static friction holds while |dpt| / w(t) < muS * lam  <=>  lt2^3 < (muS * lam * wts)^2
            float lim = (muS * lam) * wts;
            if ((lt2 * lt2) * lt2 < lim * lim) apply_pos_impulse<HAS_B>(A, rAw, B, rBw, dpt * (lt2 / wts));
        }
    }
    return lam;
}

// One contact point of a GROUND-PLANE manifold in the position pass (body against plane 0, no body B).  For an ordinary
// body it is exactly solve_point_position<false>.  For a YAW-ONLY body (an agent: inverse inertia x, y = 0,
// src/mgr.cpp:577-584) the normal part has been done for the whole manifold by yaw_ground_prepass (below): `dj` is
// the point's penetration before any correction and `share` its equal share of the manifold's normal multiplier; what
// remains per point is the static-friction correction, the same expressions for both kinds of body.
HSD float solve_point_position_ground(BodyS &A, V3 n, V3 rAl, float offB, float muS, bool yaw, float dj, float share) {
    BodyS none;
    V3 rAw = qrot(A.rot, rAl);
    V3 pA = A.pos + rAw;
    const V3 pAprev = A.ppos + qrot(A.prot, rAl);
    float lam;
    if (!yaw) {
        float d = dot_add(pA, n, -offB);
        if (!(d > 0.f)) return 0.f;
        const float dprev = dot_add(pAprev, n, -offB);
        const float excess = dprev - kMaxDepenVel * kSubstepH;
        if (excess > 0.f) d = d - excess;
        if (!(d > 0.f)) return 0.f;
        const float wsum = gen_inv_mass(A, rAw, n);
        if (!(wsum > 0.f)) return 0.f;
        lam = d / wsum;
        apply_pos_impulse<false>(A, rAw, none, V3{0.f, 0.f, 0.f}, n * lam);
        rAw = qrot(A.rot, rAl);
        pA = A.pos + rAw;
    } else {
        if (!(dj > 0.f)) return 0.f;
        lam = share;
    }
    const V3 dp = pA - pAprev;
    const V3 dpt = nmadd(dp, n, dot(dp, n));
    const float lt2 = len2(dpt);
    if (lt2 > 1e-12f) {
        const float wts = gen_inv_mass_sq(A, rAw, dpt, lt2);
        if (wts > 0.f) {
            const float lim = (muS * lam) * wts;
            if ((lt2 * lt2) * lt2 < lim * lim) apply_pos_impulse<false>(A, rAw, none, V3{0.f, 0.f, 0.f}, dpt * (lt2 / wts));
        }
    }
    return lam;
}
// The manifold-level part for a yaw-only body (oracle: solve_ground_positions_yaw_only).  Such a body cannot tilt, so
// its floor contacts cannot be resolved one after the other — the first would take the whole normal correction and
// with it the whole friction budget of the substep, at a lever arm (a straight push spun the agent up).  The
// penetrations of all points are evaluated before any correction, the body is lifted once by the deepest of them along
// the normal (translation only), and the multiplier dmax / invM is shared equally by the touching points.  Returns the
// share (0: nothing touches); r0..r3 are the contact points in the body frame.
HSD float yaw_ground_prepass(BodyS &A, V3 n, int np, V3 r0, V3 r1, V3 r2, V3 r3, float off, float (&dj)[4]) {
    int k = 0; float dmax = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        dj[j] = 0.f;
        if (j < np) {
            const V3 r = j == 0 ? r0 : j == 1 ? r1 : j == 2 ? r2 : r3;
            float d = dot_add(A.pos + qrot(A.rot, r), n, -off);
            if (d > 0.f) {
                const float excess = dot_add(A.ppos + qrot(A.prot, r), n, -off) - kMaxDepenVel * kSubstepH;
                if (excess > 0.f) d = d - excess;
            }
            dj[j] = d;
            if (d > 0.f) { k++; dmax = fmaxf(dmax, d); }
        }
    }
    if (k == 0 || !(A.invM > 0.f)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dj[j] = 0.f;
        return 0.f;
    }
    const float lamT = dmax / A.invM;
    A.pos = nmadd(A.pos, n * lamT, A.invM);
    return lamT / (float)k;
}

// One contact point of the velocity pass (dynamic friction, restitution 0).
template <bool HAS_B>
HSD void solve_point_velocity(BodyS &A, BodyS &B, V3 n, V3 rAl, V3 rBl, float lamN, float muD) {
    if (!(lamN > 0.f)) return;
    V3 rAw = qrot(A.rot, rAl);
    V3 rBw = HAS_B ? qrot(B.rot, rBl) : V3{0.f, 0.f, 0.f};
    V3 v = {0.f, 0.f, 0.f};
    if (A.invM + A.invI.x + A.invI.y + A.invI.z != 0.f) v = cross_add(A.ang, rAw, A.lin);
    if (HAS_B && B.invM + B.invI.x + B.invI.y + B.invI.z != 0.f) v = v - cross_add(B.ang, rBw, B.lin);
    float vn = dot(n, v);
    V3 vt = nmadd(v, n, vn);
    float vt2 = len2(vt);
    V3 dv = -(n * vn);
    if (vt2 > 1e-18f) {
        float vtl = sqrtf(vt2);
        float mag = fminf((muD * lamN) * kInvSubstepH, vtl);
        dv = nmadd(dv, vt, mag / vtl);
    }
    float dv2 = len2(dv);
    if (!(dv2 > 1e-18f)) return;
    float wA = gen_inv_mass_sq(A, rAw, dv, dv2);
    float wB = HAS_B ? gen_inv_mass_sq(B, rBw, dv, dv2) : 0.f;
    float ws = wA + wB;
    if (!(ws > 0.f)) return;
    V3 p = dv * (dv2 / ws);
    A.lin = madd(A.lin, p, A.invM);
    A.ang = A.ang + apply_inv_inertia(A, cross(rAw, p));
    if (HAS_B) {
        B.lin = nmadd(B.lin, p, B.invM);
        B.ang = B.ang - apply_inv_inertia(B, cross(rBw, p));
    }
}

}  // namespace hs
