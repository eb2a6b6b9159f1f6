// XPBD contact / joint solver pieces and the actionSystem, shared by the physics pipeline kernels
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

// One world's geometry + action state staged in LDS for the (rare) lock / grab ray casts.
struct ActWorld {
    WorldGeom g;
    int grabOther[kMaxAgents];
    float grabData[kMaxAgents][8];
    int actGL[kMaxAgents];
    int teams;
};

struct BodyS { V3 pos; Q rot; V3 ppos; Q prot; V3 lin, ang; float invM; V3 invI; };

HSD V3 ld3(const float *p) { return {p[0], p[1], p[2]}; }
HSD Q ld4(const float *p) { return {p[0], p[1], p[2], p[3]}; }
HSD void st3(float *p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
HSD void st4(float *p, Q q) { p[0] = q.w; p[1] = q.x; p[2] = q.y; p[3] = q.z; }

HSD V3 apply_inv_inertia(Q q, V3 invI, V3 v) {
    V3 l = qrot(qinv(q), v);
    l = mulc(l, invI);
    return qrot(q, l);
}
HSD float gen_inv_mass(Q q, float invM, V3 invI, V3 r, V3 n) {
    V3 rn = cross(r, n);
    V3 l = qrot(qinv(q), rn);
    return invM + ((l.x * l.x * invI.x + l.y * l.y * invI.y) + l.z * l.z * invI.z);
}
HSD Q quat_add_rotation(Q q, V3 dth) {
    Q dq = qmul(Q{0.f, dth.x, dth.y, dth.z}, q);
    Q r = {q.w + 0.5f * dq.w, q.x + 0.5f * dq.x, q.y + 0.5f * dq.y, q.z + 0.5f * dq.z};
    return qnormalize(r);
}
HSD bool has_mass(const BodyS &b) { return b.invM != 0.f || b.invI.z != 0.f || b.invI.x != 0.f || b.invI.y != 0.f; }

template <bool HAS_B>
HSD void apply_pos_impulse(BodyS &A, V3 rA, BodyS &B, V3 rB, V3 p) {
    if (has_mass(A)) {
        A.pos = A.pos - p * A.invM;
        V3 dth = apply_inv_inertia(A.rot, A.invI, cross(rA, p));
        A.rot = quat_add_rotation(A.rot, -dth);
    }
    if (HAS_B && has_mass(B)) {
        B.pos = B.pos + p * B.invM;
        V3 dth = apply_inv_inertia(B.rot, B.invI, cross(rB, p));
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
    float d = HAS_B ? dot(pA - pB, n) : dot(pA, n) - offB;
    if (!(d > 0.f)) return 0.f;
    V3 pAprev = A.ppos + qrot(A.prot, rAl);
    V3 pBprev = HAS_B ? B.ppos + qrot(B.prot, rBl) : V3{0.f, 0.f, 0.f};
    float dprev = HAS_B ? dot(pAprev - pBprev, n) : dot(pAprev, n) - offB;
    float excess = dprev - kMaxDepenVel * kSubstepH;
    if (excess > 0.f) d = d - excess;
    if (!(d > 0.f)) return 0.f;
    float wA = gen_inv_mass(A.rot, A.invM, A.invI, rAw, n);
    float wB = HAS_B ? gen_inv_mass(B.rot, B.invM, B.invI, rBw, n) : 0.f;
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
    V3 dpt = dp - n * dot(dp, n);
    float lt2 = len2(dpt);
    if (lt2 > 1e-12f) {
        float lt = sqrtf(lt2);
        V3 t = dpt * (1.f / lt);
        float wtA = gen_inv_mass(A.rot, A.invM, A.invI, rAw, t);
        float wtB = HAS_B ? gen_inv_mass(B.rot, B.invM, B.invI, rBw, t) : 0.f;
        float wts = wtA + wtB;
        if (wts > 0.f) {
            float lamT = lt / wts;
            if (lamT < muS * lam) apply_pos_impulse<HAS_B>(A, rAw, B, rBw, t * lamT);
        }
    }
    return lam;
}

// One contact point of the velocity pass (dynamic friction, restitution 0).
template <bool HAS_B>
HSD void solve_point_velocity(BodyS &A, BodyS &B, V3 n, V3 rAl, V3 rBl, float lamN, float muD) {
    const float h = kSubstepH;
    if (!(lamN > 0.f)) return;
    V3 rAw = qrot(A.rot, rAl);
    V3 rBw = HAS_B ? qrot(B.rot, rBl) : V3{0.f, 0.f, 0.f};
    V3 v = {0.f, 0.f, 0.f};
    if (A.invM + A.invI.x + A.invI.y + A.invI.z != 0.f) v = A.lin + cross(A.ang, rAw);
    if (HAS_B && B.invM + B.invI.x + B.invI.y + B.invI.z != 0.f) v = v - (B.lin + cross(B.ang, rBw));
    float vn = dot(n, v);
    V3 vt = v - n * vn;
    float vtl = len(vt);
    V3 dv = -(n * vn);
    if (vtl > 1e-9f) {
        float fn = lamN / (h * h);
        float mag = fminf(h * muD * fn, vtl);
        dv = dv - vt * (mag / vtl);
    }
    float dvl = len(dv);
    if (!(dvl > 1e-9f)) return;
    V3 dir = dv * (1.f / dvl);
    float wA = gen_inv_mass(A.rot, A.invM, A.invI, rAw, dir);
    float wB = HAS_B ? gen_inv_mass(B.rot, B.invM, B.invI, rBw, dir) : 0.f;
    float ws = wA + wB;
    if (!(ws > 0.f)) return;
    V3 p = dir * (dvl / ws);
    A.lin = A.lin + p * A.invM;
    A.ang = A.ang + apply_inv_inertia(A.rot, A.invI, cross(rAw, p));
    if (HAS_B) {
        B.lin = B.lin - p * B.invM;
        B.ang = B.ang - apply_inv_inertia(B.rot, B.invI, cross(rBw, p));
    }
}

// actionSystem for one world, agents in interface order (executed by lane 0 of the group).
HSD void action_system(ActWorld &pw, int A_) {
    for (int i = 0; i < A_; ++i) {
        const int fl = pw.actGL[i];
        if (fl == 0) continue;
        const int type = team_agent_type(pw.teams, i);
        const int slot = kAgentSlot0 + i;
        const V3 mpos = geom_pos(pw.g, slot);
        const Q mrot = geom_rot(pw.g, slot);
        if (fl & 2) {   // lock
            float t; V3 o = mpos + V3{0.f, 0.f, 0.5f};
            int hit = trace_ray(pw.g, o, qrot(mrot, {0.f, 1.f, 0.f}), 2.5f, &t);
            if (hit >= 0 && hit < kNumDSlots) {
                const int m = pw.g.meta[hit];
                const int obj = meta_obj(m), resp = meta_resp(m), owner = meta_owner(m);
                if (resp == RESP_STATIC) {
                    if ((type == AGENT_SEEKER && owner == OWNER_SEEKER) || (type == AGENT_HIDER && owner == OWNER_HIDER))
                        pw.g.meta[hit] = meta_pack(obj, RESP_DYNAMIC, OWNER_NONE);
                } else if (owner == OWNER_NONE) {
                    pw.g.meta[hit] = meta_pack(obj, RESP_STATIC, type == AGENT_HIDER ? OWNER_HIDER : OWNER_SEEKER);
                }
            }
        }
        if (fl & 1) {   // grab
            if (pw.grabOther[i] >= 0) {
                pw.grabOther[i] = -1;
            } else {
                float t; V3 o = mpos + V3{0.f, 0.f, 0.5f};
                V3 dir = qrot(mrot, {0.f, 1.f, 0.f});
                int hit = trace_ray(pw.g, o, dir, 2.5f, &t);
                if (hit >= 0 && hit < kNumDSlots) {
                    const int m = pw.g.meta[hit];
                    if (meta_owner(m) == OWNER_NONE && meta_resp(m) == RESP_DYNAMIC) {
                        V3 hit_pos = o + dir * t;
                        Q erot = geom_rot(pw.g, hit);
                        V3 r2 = qrot(qinv(erot), hit_pos - geom_pos(pw.g, hit));
                        Q at2 = qnormalize(qmul(qinv(erot), mrot));
                        pw.grabOther[i] = hit;
                        float *gd = pw.grabData[i];
                        gd[0] = r2.x; gd[1] = r2.y; gd[2] = r2.z;
                        gd[3] = at2.w; gd[4] = at2.x; gd[5] = at2.y; gd[6] = at2.z;
                        gd[7] = t - 1.25f;
                    }
                }
            }
        }
    }
}

}  // namespace hs
