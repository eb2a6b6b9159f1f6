// Physics + game-logic kernel: everything between the start of Manager::step and the reset node
// of the Step task graph (src/sim.cpp:1140-1201):
//   movementSystem | instantMovementSystem  (:202-254)
//   [broadphase]  actionSystem               (:270-370)
//   4 XPBD substeps (PhysicsSystem::setupPhysicsStepTasks, :1162-1163)
//   agentZeroVelSystem (:258-268), rewardsVisSystem (:763-804),
//   outputRewardsDonesSystem (:806-841), updateEpisodeResultsSystem (:843-893)
//
// Mapping: one GROUP of G lanes per world (G = 16 for <= 5 agents, 32 for 6), 64/G worlds per
// wave, one wave per workgroup.  Lane l of a group owns movable-body slot l.  The world's
// poses, velocities, static geometry, contact manifolds and joints stay in LDS for the whole
// step (4 substeps), so HBM sees each body column once in and once out per step.
// Candidate pairs are found by all-pairs AABB tests (<= 17 bodies, <= 36 walls: no BVH build),
// compacted in pair order with a per-group prefix sum, and the convex tests are then spread
// over the lanes of the group.  The Gauss-Seidel order is the oracle's: joints, body-body
// manifolds in pair order (lane 0), then every body's static manifolds (one lane per body).
#pragma once
#include "hs_state.h"
#include "hs_rays.h"
#include "hs_collide.h"

namespace hs {

// Developer-only phase timing (build with -DHS_PHASE_TIMING): per-phase shader-clock sums per wave.
#ifdef HS_PHASE_TIMING
#define HS_STAMP_INIT unsigned long long t_prev_ = clock64(); unsigned long long acc_[12] = {0,0,0,0,0,0,0,0,0,0,0,0};
#define HS_STAMP(i) { unsigned long long t_ = clock64(); acc_[i] += t_ - t_prev_; t_prev_ = t_; }
#define HS_STAMP_FLUSH if (threadIdx.x == 0 && S.dbg) { for (int i_ = 0; i_ < 12; ++i_) atomicAdd(&S.dbg[i_], acc_[i_]); }
#else
#define HS_STAMP_INIT
#define HS_STAMP(i)
#define HS_STAMP_FLUSH
#endif

// Contact manifolds live in an HBM workspace (L2-resident): written once by the lane that ran the
// convex test, read by the lane that solves them.  16-byte multiples so they move as dwordx4.
struct alignas(16) ManDD { int a, b, np; float muS, muD; float n[3]; float rA[4][3]; float rB[4][3]; float lam[4]; };
struct alignas(16) ManS { int np; float muS, muD; float n[3]; float pad[2]; float rA[4][3]; float offB[4]; float lam[4]; };
static_assert(sizeof(ManDD) == 144 && sizeof(ManS) == 112, "manifold layout");

struct PhysWorld {
    WorldGeom g;
    float lin[kNumDSlots][3], ang[kNumDSlots][3];
    float ppos[kNumDSlots][3], prot[kNumDSlots][4];
    float lo[kNumDSlots][3], hi[kNumDSlots][3];
    int cnt[kNumDSlots];          // per-slot candidate counts: body-body | body-static << 16
    int ndd, nsc;
    unsigned char ddA[kMaxDDCand], ddB[kMaxDDCand], scBody[kMaxSCand], scStatic[kMaxSCand];
    int grabOther[kMaxAgents];
    float grabData[kMaxAgents][8];
    float aforce[kMaxAgents][4];
    int actGL[kMaxAgents];
    int counts, teams, step;
    float hiderReward;
};

struct BodyS { V3 pos; Q rot; V3 ppos; Q prot; V3 lin, ang; float invM; V3 invI; };

HSD V3 ld3(const float *p) { return {p[0], p[1], p[2]}; }
HSD Q ld4(const float *p) { return {p[0], p[1], p[2], p[3]}; }
HSD void st3(float *p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
HSD void st4(float *p, Q q) { p[0] = q.w; p[1] = q.x; p[2] = q.y; p[3] = q.z; }

HSD void body_load(const PhysWorld &pw, int s, BodyS &b) {
    b.pos = ld3(pw.g.pos[s]); b.rot = ld4(pw.g.rot[s]);
    b.ppos = ld3(pw.ppos[s]); b.prot = ld4(pw.prot[s]);
    b.lin = ld3(pw.lin[s]); b.ang = ld3(pw.ang[s]);
    const int m = pw.g.meta[s];
    const bool dyn = m != 0 && meta_resp(m) == RESP_DYNAMIC;
    b.invM = dyn ? obj_inv_mass(meta_obj(m)) : 0.f;
    b.invI = dyn ? obj_inv_inertia(meta_obj(m)) : V3{0.f, 0.f, 0.f};
}
HSD void body_store_pose(PhysWorld &pw, int s, const BodyS &b) { st3(pw.g.pos[s], b.pos); st4(pw.g.rot[s], b.rot); }
HSD void body_store_vel(PhysWorld &pw, int s, const BodyS &b) { st3(pw.lin[s], b.lin); st3(pw.ang[s], b.ang); }

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

// Fixed grab joint (sim.cpp:343-356): angular alignment, then anchor coincidence.
HSD void solve_grab_joint(PhysWorld &pw, int agent) {
    const int other = pw.grabOther[agent];
    if (other < 0) return;
    const int sa = kAgentSlot0 + agent;
    BodyS A, B;
    body_load(pw, sa, A); body_load(pw, other, B);
    const float *gd = pw.grabData[agent];
    const V3 r2 = {gd[0], gd[1], gd[2]};
    const Q attach2 = {gd[3], gd[4], gd[5], gd[6]};
    const float sep = gd[7];
    {
        Q qa = qmul(A.rot, Q{1.f, 0.f, 0.f, 0.f}), qb = qmul(B.rot, attach2);
        Q dq = qmul(qa, qinv(qb));
        V3 dphi = {2.f * dq.x, 2.f * dq.y, 2.f * dq.z};
        if (dq.w < 0.f) dphi = -dphi;
        float th2 = len2(dphi);
        if (th2 > 1e-12f) {
            float th = sqrtf(th2);
            V3 ax = dphi * (1.f / th);
            V3 la = qrot(qinv(A.rot), ax), lb = qrot(qinv(B.rot), ax);
            float wA = (la.x * la.x * A.invI.x + la.y * la.y * A.invI.y) + la.z * la.z * A.invI.z;
            float wB = (lb.x * lb.x * B.invI.x + lb.y * lb.y * B.invI.y) + lb.z * lb.z * B.invI.z;
            float ws = wA + wB;
            if (ws > 0.f) {
                V3 p = ax * (th / ws);
                A.rot = quat_add_rotation(A.rot, -apply_inv_inertia(A.rot, A.invI, p));
                B.rot = quat_add_rotation(B.rot, apply_inv_inertia(B.rot, B.invI, p));
            }
        }
    }
    {
        V3 anchorA = V3{0.f, 1.25f, 0.5f} + V3{0.f, sep, 0.f};
        V3 rAw = qrot(A.rot, anchorA), rBw = qrot(B.rot, r2);
        V3 dx = (A.pos + rAw) - (B.pos + rBw);
        float c2 = len2(dx);
        if (c2 > 1e-12f) {
            float c = sqrtf(c2);
            V3 n = dx * (1.f / c);
            float ws = gen_inv_mass(A.rot, A.invM, A.invI, rAw, n) + gen_inv_mass(B.rot, B.invM, B.invI, rBw, n);
            if (ws > 0.f) apply_pos_impulse<true>(A, rAw, B, rBw, n * (c / ws));
        }
    }
    body_store_pose(pw, sa, A); body_store_pose(pw, other, B);
}

// actionSystem for one world, agents in interface order (executed by lane 0 of the group).
HSD void action_system(PhysWorld &pw, int A_) {
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

// dynamic read / accumulate of a 4-entry register array without scratch
HSD float sel4(const float (&a)[4], int j) { return j == 0 ? a[0] : (j == 1 ? a[1] : (j == 2 ? a[2] : a[3])); }
HSD void acc4(float (&a)[4], int j, float v) { a[0] += j == 0 ? v : 0.f; a[1] += j == 1 ? v : 0.f; a[2] += j == 2 ? v : 0.f; a[3] += j == 3 ? v : 0.f; }

// Work-list entry helpers: (world-in-block << 8) | index
HSD unsigned short item_pack(int world, int k) { return (unsigned short)((world << 8) | k); }

template <int NT, int NB>   // NB = body slots per lane: 1 when agents <= 5 (slots 0..15), else 2
__global__ void __launch_bounds__(NT, 2) k_physics(SimState S) {
    constexpr int G = 16;                           // lanes per world
    constexpr int WPB = NT / G;                     // worlds per workgroup
    constexpr int PACK = 64;                        // sparse phases are packed into the first wave
    __shared__ PhysWorld sh[WPB];
    __shared__ ClipBuf clipbuf[PACK];
    __shared__ unsigned short satItems[WPB * (kMaxDDCand + kMaxSCand)];
    __shared__ unsigned short wallItems[WPB * kMaxSCand];
    __shared__ unsigned char ddWorlds[WPB];
    __shared__ int nSat, nWall, nDDW;
    const int tid = threadIdx.x;
    const int grp = tid / G, l = tid % G;
    const int w = blockIdx.x * WPB + grp;
    const int wbase = blockIdx.x * WPB;
    const int N = S.N, A_ = S.A;
    const bool wok = w < N;
    PhysWorld &pw = sh[grp];
    ManDD *const wsDDb = (ManDD *)S.wsDD;
    ManS *const wsSCb = (ManS *)S.wsSC;
    const bool instant = (S.flags & FLAG_ZERO_AGENT_VELOCITY) == FLAG_ZERO_AGENT_VELOCITY;

    HS_STAMP_INIT
    // ---------------- stage the world into LDS ----------------
    if (wok) {
        for (int s = l; s < kNumDSlots; s += G) {
            pw.g.meta[s] = S.bmeta[s * N + w];
            pw.cnt[s] = 0;            // slots this block's lanes do not own (slot 16 when NB == 1) stay empty
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                pw.g.pos[s][c] = S.bpos[(c * kNumDSlots + s) * N + w];
                pw.lin[s][c] = S.blin[(c * kNumDSlots + s) * N + w];
                pw.ang[s][c] = S.bang[(c * kNumDSlots + s) * N + w];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) pw.g.rot[s][c] = S.brot[(c * kNumDSlots + s) * N + w];
        }
        const int nw = S.numWalls[w], npl = S.numPlanes[w];
        for (int k = l; k < nw; k += G) {
#pragma unroll
            for (int c = 0; c < 4; ++c) pw.g.wall[k][c] = S.walls[(c * kMaxWalls + k) * N + w];
        }
        for (int p = l; p < npl; p += G) {
#pragma unroll
            for (int c = 0; c < 4; ++c) pw.g.plane[p][c] = S.planes[(c * kMaxPlanes + p) * N + w];
        }
        for (int i = l; i < kMaxAgents; i += G) {
            pw.grabOther[i] = S.grabOther[i * N + w];
#pragma unroll
            for (int c = 0; c < 8; ++c) pw.grabData[i][c] = S.grabData[(c * kMaxAgents + i) * N + w];
#pragma unroll
            for (int c = 0; c < 4; ++c) pw.aforce[i][c] = S.aforce[(c * kMaxAgents + i) * N + w];
            pw.actGL[i] = 0;
        }
        if (l == 0) {
            pw.g.numWalls = nw; pw.g.numPlanes = npl;
            pw.counts = S.counts[w]; pw.teams = S.teams[w]; pw.step = S.curEpisodeStep[w];
            pw.hiderReward = S.hiderTeamReward[w];
            pw.ndd = 0; pw.nsc = 0;
        }
    } else if (l == 0) {
        pw.g.numWalls = 0; pw.g.numPlanes = 0; pw.ndd = 0; pw.nsc = 0; pw.teams = 0; pw.counts = 0;
        for (int s = 0; s < kNumDSlots; ++s) { pw.g.meta[s] = 0; pw.cnt[s] = 0; }
    }
    __syncthreads();

    HS_STAMP(0)
    // ---------------- movementSystem / instantMovementSystem (sim.cpp:202-254) ----------------
    if (wok) {
        for (int agent = l; agent < A_; agent += G) {
            const bool active = team_agent_active(pw.teams, agent) != 0;
            const int type = team_agent_type(pw.teams, agent);
            if (!active || (type == AGENT_SEEKER && pw.step < kNumPrepSteps - 1)) continue;
            int32_t *act_row = S.xAction + (w * A_ + agent) * 5;
            const int ax = act_row[0], ay = act_row[1], ar = act_row[2], ag = act_row[3], al = act_row[4];
            float fx, fy, tz;
            if (instant) { fx = 400.f * (float)(ax - 2); fy = 400.f * (float)(ay - 2); tz = 120.f * (float)(ar - 2); }
            else { fx = 12.f * (float)(ax - 5); fy = 12.f * (float)(ay - 5); tz = 3.f * (float)(ar - 5); }
            V3 f = qrot(geom_rot(pw.g, kAgentSlot0 + agent), {fx, fy, 0.f});
            pw.aforce[agent][0] = f.x; pw.aforce[agent][1] = f.y; pw.aforce[agent][2] = f.z; pw.aforce[agent][3] = tz;
            pw.actGL[agent] = (ag == 1 ? 1 : 0) | (al == 1 ? 2 : 0);
            // "consume" the action (sim.cpp:365-369)
            act_row[0] = 2; act_row[1] = 2; act_row[2] = 2; act_row[3] = 0; act_row[4] = 0;
        }
    }
    __syncthreads();
    // ---------------- actionSystem (sim.cpp:270-370): rare, serial per world ----------------
    if (wok && l == 0) action_system(pw, A_);
    __syncthreads();

    HS_STAMP(1)
    // ground-plane manifolds of the owned bodies live in registers
    int g_np[NB], g_vi[NB]; float g_off[NB][4], g_lam[NB][4];
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) { g_np[jb] = 0; g_vi[jb] = 0; }

    for (int sub = 0; sub < 4; ++sub) {
        if (tid == 0) { nSat = 0; nWall = 0; nDDW = 0; }
        // ---------- P1: integrate, AABB ----------
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
            const int slot = l + jb * G;
            if (!wok || slot >= kNumDSlots) continue;
            const int meta = pw.g.meta[slot];
            const int obj = meta_obj(meta);
            const bool present = meta != 0;
            const bool dynamic = present && meta_resp(meta) == RESP_DYNAMIC;
            V3 pos = ld3(pw.g.pos[slot]); Q rot = ld4(pw.g.rot[slot]);
            st3(pw.ppos[slot], pos); st4(pw.prot[slot], rot);
            if (dynamic) {
                const float h = kSubstepH;
                V3 lin = ld3(pw.lin[slot]), ang = ld3(pw.ang[slot]);
                const float invM = obj_inv_mass(obj);
                const V3 invI = obj_inv_inertia(obj);
                V3 force = {0.f, 0.f, 0.f}; float torque_z = 0.f;
                if (slot >= kAgentSlot0) { force = ld3(pw.aforce[slot - kAgentSlot0]); torque_z = pw.aforce[slot - kAgentSlot0][3]; }
                lin = lin + (force * invM + V3{0.f, 0.f, kGravityZ}) * h;
                pos = pos + lin * h;
                Q qi = qinv(rot);
                V3 wl = qrot(qi, ang), tl = qrot(qi, V3{0.f, 0.f, torque_z});
                V3 I = {invI.x > 0.f ? 1.f / invI.x : 0.f, invI.y > 0.f ? 1.f / invI.y : 0.f, invI.z > 0.f ? 1.f / invI.z : 0.f};
                V3 Iw = mulc(I, wl);
                wl = wl + mulc(invI, tl - cross(wl, Iw)) * h;
                ang = qrot(rot, wl);
                rot = quat_add_rotation(rot, ang * h);
                st3(pw.g.pos[slot], pos); st4(pw.g.rot[slot], rot);
                st3(pw.lin[slot], lin); st3(pw.ang[slot], ang);
            }
            if (present) {
                V3 lo, hi;
                hull_aabb(hull_ref_body(obj, pos, rot), &lo, &hi);
                st3(pw.lo[slot], lo); st3(pw.hi[slot], hi);
            }
        }
        __syncthreads();
        HS_STAMP(2)
        // ---------- P2: candidate pairs, compacted in (slot, partner) order ----------
        unsigned dd_mask[NB]; unsigned long long s_mask[NB]; int s_planes[NB];
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
            const int slot = l + jb * G;
            dd_mask[jb] = 0; s_mask[jb] = 0ull; s_planes[jb] = 0;
            if (!wok || slot >= kNumDSlots) continue;
            const int meta = pw.g.meta[slot];
            if (meta != 0) {
                const bool dynamic = meta_resp(meta) == RESP_DYNAMIC;
                const V3 lo = ld3(pw.lo[slot]), hi = ld3(pw.hi[slot]);
                for (int j = slot + 1; j < kNumDSlots; ++j) {
                    const int mj = pw.g.meta[j];
                    if (mj == 0) continue;
                    if (!dynamic && meta_resp(mj) != RESP_DYNAMIC) continue;
                    const V3 lj = ld3(pw.lo[j]), hj = ld3(pw.hi[j]);
                    if (lo.x <= hj.x && lj.x <= hi.x && lo.y <= hj.y && lj.y <= hi.y && lo.z <= hj.z && lj.z <= hi.z)
                        dd_mask[jb] |= 1u << j;
                }
                if (dynamic) {
                    s_planes[jb] = pw.g.numPlanes > 1 ? pw.g.numPlanes - 1 : 0;
                    const int nw = pw.g.numWalls;
                    for (int k = 0; k < nw; ++k) {
                        const float cx = pw.g.wall[k][0], cy = pw.g.wall[k][1], hx = pw.g.wall[k][2], hy = pw.g.wall[k][3];
                        if (lo.x <= cx + hx && cx - hx <= hi.x && lo.y <= cy + hy && cy - hy <= hi.y && lo.z <= 2.5f && 0.f <= hi.z)
                            s_mask[jb] |= 1ull << k;
                    }
                }
            }
            pw.cnt[slot] = __popc(dd_mask[jb]) | ((s_planes[jb] + __popcll(s_mask[jb])) << 16);
        }
        __syncthreads();
        if (wok) {
            if (l == 0) {
                // totals of this world, and its entries in the block-wide sparse work lists
                int td = 0, ts = 0; bool grab = false;
                for (int k = 0; k < kNumDSlots; ++k) { const int c = pw.cnt[k]; td += c & 0xffff; ts += c >> 16; }
                const int ndd = td < kMaxDDCand ? td : kMaxDDCand, nsc = ts < kMaxSCand ? ts : kMaxSCand;
                pw.ndd = ndd; pw.nsc = nsc;
                for (int a = 0; a < kMaxAgents; ++a) grab |= pw.grabOther[a] >= 0;
                if (ndd + nsc > 0) {
                    const int base = atomicAdd(&nSat, ndd + nsc);
                    for (int k = 0; k < ndd + nsc; ++k) satItems[base + k] = item_pack(grp, k);
                }
                if (ndd > 0 || grab) ddWorlds[atomicAdd(&nDDW, 1)] = (unsigned char)grp;
            }
#pragma unroll
            for (int jb = 0; jb < NB; ++jb) {
                const int slot = l + jb * G;
                if (slot >= kNumDSlots) continue;
                if (dd_mask[jb] == 0 && s_mask[jb] == 0ull && s_planes[jb] == 0) continue;
                int off_dd = 0, off_sc = 0;
                for (int k = 0; k < slot; ++k) { const int c = pw.cnt[k]; off_dd += c & 0xffff; off_sc += c >> 16; }
                if ((s_mask[jb] != 0ull || s_planes[jb] != 0) && off_sc < kMaxSCand)
                    wallItems[atomicAdd(&nWall, 1)] = item_pack(grp, slot);
                unsigned mm = dd_mask[jb];
                while (mm) {
                    const int j = __ffs(mm) - 1; mm &= mm - 1;
                    if (off_dd < kMaxDDCand) { pw.ddA[off_dd] = (unsigned char)slot; pw.ddB[off_dd] = (unsigned char)j; }
                    off_dd++;
                }
                for (int p = 1; p <= s_planes[jb]; ++p) {
                    if (off_sc < kMaxSCand) { pw.scBody[off_sc] = (unsigned char)slot; pw.scStatic[off_sc] = (unsigned char)(kMaxWalls + p); }
                    off_sc++;
                }
                unsigned long long sm = s_mask[jb];
                while (sm) {
                    const int k = __ffsll((long long)sm) - 1; sm &= sm - 1;
                    if (off_sc < kMaxSCand) { pw.scBody[off_sc] = (unsigned char)slot; pw.scStatic[off_sc] = (unsigned char)k; }
                    off_sc++;
                }
            }
        }
        HS_STAMP(3)
        // ---------- P3a: ground plane (plane 0) vs every owned body; result stays in registers ----------
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
            const int slot = l + jb * G;
            g_np[jb] = 0;
            if (!wok || slot >= kNumDSlots || pw.g.numPlanes < 1) continue;
            const int meta = pw.g.meta[slot];
            if (meta == 0 || meta_resp(meta) != RESP_DYNAMIC) continue;
            const int obj = meta_obj(meta);
            HullRef hb = hull_ref_body(obj, ld3(pw.g.pos[slot]), ld4(pw.g.rot[slot]));
            RawManifold raw;
            if (collide_hull_plane(hb, ld3(pw.g.plane[0]), pw.g.plane[0][3], raw)) {
                g_np[jb] = raw.np;
                int vi = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j < raw.np) { vi |= raw.vidx[j] << (3 * j); g_off[jb][j] = dot(raw.pB[j], raw.n); }
                    g_lam[jb][j] = 0.f;
                }
                g_vi[jb] = vi;
            }
        }
        __syncthreads();
        HS_STAMP(4)
        // ---------- P3b: convex tests of the whole block, packed into the first wave ----------
        if (tid < PACK) {
            const int total = nSat;
            ClipBuf &cb = clipbuf[tid];
            for (int it = tid; it < total; it += PACK) {
                const int item = satItems[it];
                const int g2 = item >> 8, k = item & 0xff;
                PhysWorld &q = sh[g2];
                ManDD *const wsDD = wsDDb + (size_t)(wbase + g2) * kMaxDDCand;
                ManS *const wsSC = wsSCb + (size_t)(wbase + g2) * kMaxSCand;
                const int ndd = q.ndd;
                const bool isdd = k < ndd;
                const int kk = isdd ? k : k - ndd;
                const int a = isdd ? q.ddA[kk] : q.scBody[kk];
                const int bsel = isdd ? q.ddB[kk] : q.scStatic[kk];
                const int oa = meta_obj(q.g.meta[a]);
                const V3 pa = ld3(q.g.pos[a]);
                const Q qa = ld4(q.g.rot[a]);
                const HullRef ha = hull_ref_body(oa, pa, qa);
                RawManifold raw;
                if (isdd) wsDD[kk].np = 0; else wsSC[kk].np = 0;
                if (!isdd && bsel >= kMaxWalls) {
                    const float *pl = q.g.plane[bsel - kMaxWalls];
                    if (collide_hull_plane(ha, ld3(pl), pl[3], raw)) {
                        ManS m;
                        m.np = raw.np; st3(m.n, raw.n); m.pad[0] = 0.f; m.pad[1] = 0.f;
                        m.muS = 0.5f * (obj_mu_s(oa) + obj_mu_s(OBJ_PLANE));
                        m.muD = 0.5f * (obj_mu_d(oa) + obj_mu_d(OBJ_PLANE));
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const bool on = j < raw.np;
                            st3(m.rA[j], on ? hull_local_vertex(oa, raw.vidx[j]) : V3{0.f, 0.f, 0.f});
                            m.offB[j] = on ? dot(raw.pB[j], raw.n) : 0.f; m.lam[j] = 0.f;
                        }
                        wsSC[kk] = m;
                    }
                    continue;
                }
                int ob; V3 pb = {0.f, 0.f, 0.f}; Q qb = {1.f, 0.f, 0.f, 0.f};
                HullRef hb;
                if (isdd) {
                    ob = meta_obj(q.g.meta[bsel]); pb = ld3(q.g.pos[bsel]); qb = ld4(q.g.rot[bsel]);
                    hb = hull_ref_body(ob, pb, qb);
                } else {
                    ob = OBJ_WALL;
                    hb = hull_ref_wall(q.g.wall[bsel][0], q.g.wall[bsel][1], q.g.wall[bsel][2], q.g.wall[bsel][3]);
                }
                if (!collide_hulls(ha, hb, cb, raw)) continue;
                const float muS = 0.5f * (obj_mu_s(oa) + obj_mu_s(ob)), muD = 0.5f * (obj_mu_d(oa) + obj_mu_d(ob));
                const Q qai = qinv(qa);
                if (isdd) {
                    ManDD m;
                    m.a = a; m.b = bsel; m.np = raw.np; m.muS = muS; m.muD = muD;
                    st3(m.n, raw.n);
                    const Q qbi = qinv(qb);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool on = j < raw.np;
                        st3(m.rA[j], on ? qrot(qai, raw.pA[j] - pa) : V3{0.f, 0.f, 0.f});
                        st3(m.rB[j], on ? qrot(qbi, raw.pB[j] - pb) : V3{0.f, 0.f, 0.f});
                        m.lam[j] = 0.f;
                    }
                    wsDD[kk] = m;
                } else {
                    ManS m;
                    m.np = raw.np; m.muS = muS; m.muD = muD; m.pad[0] = 0.f; m.pad[1] = 0.f;
                    st3(m.n, raw.n);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool on = j < raw.np;
                        st3(m.rA[j], on ? qrot(qai, raw.pA[j] - pa) : V3{0.f, 0.f, 0.f});
                        m.offB[j] = on ? dot(raw.pB[j], raw.n) : 0.f; m.lam[j] = 0.f;
                    }
                    wsSC[kk] = m;
                }
            }
        }
        __syncthreads();
        HS_STAMP(5)
        // ---------- P4a: joints + body-body manifolds, one lane per world that has any ----------
        if (tid < PACK) {
            const int total = nDDW;
            for (int it = tid; it < total; it += PACK) {
                const int g2 = ddWorlds[it];
                PhysWorld &q = sh[g2];
                ManDD *const wsDD = wsDDb + (size_t)(wbase + g2) * kMaxDDCand;
                for (int a = 0; a < kMaxAgents; ++a)
                    if (team_agent_active(q.teams, a)) solve_grab_joint(q, a);
                const int ndd = q.ndd;
                for (int k = 0; k < ndd; ++k) {
                    if (wsDD[k].np <= 0) continue;
                    ManDD &m = wsDD[k];
                    const int ma = m.a, mb = m.b, mnp = m.np;
                    const float muS = m.muS;
                    BodyS Ab, Bb;
                    body_load(q, ma, Ab); body_load(q, mb, Bb);
                    const V3 n = ld3(m.n);
#pragma unroll 1
                    for (int j = 0; j < mnp; ++j)
                        m.lam[j] += solve_point_position<true>(Ab, Bb, n, ld3(m.rA[j]), ld3(m.rB[j]), 0.f, muS);
                    body_store_pose(q, ma, Ab); body_store_pose(q, mb, Bb);
                }
            }
        }
        __syncthreads();
        HS_STAMP(6)
        // ---------- P4b: ground manifold of every owned body ----------
        BodyS none;
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
            const int slot = l + jb * G;
            if (!wok || slot >= kNumDSlots || g_np[jb] == 0) continue;
            const int obj = meta_obj(pw.g.meta[slot]);
            BodyS me;
            body_load(pw, slot, me);
            const V3 gn = -ld3(pw.g.plane[0]);
            const float gmuS = 0.5f * (obj_mu_s(obj) + obj_mu_s(OBJ_PLANE));
#pragma unroll 1
            for (int j = 0; j < g_np[jb]; ++j)
                acc4(g_lam[jb], j, solve_point_position<false>(me, none, gn, hull_local_vertex(obj, (g_vi[jb] >> (3 * j)) & 7),
                                                                V3{0.f, 0.f, 0.f}, sel4(g_off[jb], j), gmuS));
            body_store_pose(pw, slot, me);
        }
        __syncthreads();
        HS_STAMP(7)
        // ---------- P4c: wall / extra-plane manifolds, one lane per body that has candidates ----------
        if (tid < PACK) {
            const int total = nWall;
            for (int it = tid; it < total; it += PACK) {
                const int item = wallItems[it];
                const int g2 = item >> 8, slot = item & 0xff;
                PhysWorld &q = sh[g2];
                ManS *const wsSC = wsSCb + (size_t)(wbase + g2) * kMaxSCand;
                BodyS me;
                body_load(q, slot, me);
                const int nsc = q.nsc;
                for (int k = 0; k < nsc; ++k) {
                    if (q.scBody[k] != slot) continue;
                    if (wsSC[k].np <= 0) continue;
                    ManS &m = wsSC[k];
                    const int mnp = m.np; const float muS = m.muS;
                    const V3 n = ld3(m.n);
#pragma unroll 1
                    for (int j = 0; j < mnp; ++j)
                        m.lam[j] += solve_point_position<false>(me, none, n, ld3(m.rA[j]), V3{0.f, 0.f, 0.f}, m.offB[j], muS);
                }
                body_store_pose(q, slot, me);
            }
        }
        __syncthreads();
        HS_STAMP(8)
        // ---------- P5: derive velocities ----------
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
            const int slot = l + jb * G;
            if (!wok || slot >= kNumDSlots) continue;
            const int meta = pw.g.meta[slot];
            if (meta == 0 || meta_resp(meta) != RESP_DYNAMIC) continue;
            const float h = kSubstepH;
            const V3 pos = ld3(pw.g.pos[slot]), ppos = ld3(pw.ppos[slot]);
            const Q rot = ld4(pw.g.rot[slot]), prot = ld4(pw.prot[slot]);
            const V3 lin = (pos - ppos) * (1.f / h);
            Q dq = qmul(rot, qinv(prot));
            V3 wv = V3{dq.x, dq.y, dq.z} * (2.f / h);
            st3(pw.lin[slot], lin); st3(pw.ang[slot], dq.w >= 0.f ? wv : -wv);
        }
        __syncthreads();
        // ---------- P6a: body-body velocity pass ----------
        if (tid < PACK) {
            const int total = nDDW;
            for (int it = tid; it < total; it += PACK) {
                const int g2 = ddWorlds[it];
                PhysWorld &q = sh[g2];
                ManDD *const wsDD = wsDDb + (size_t)(wbase + g2) * kMaxDDCand;
                const int ndd = q.ndd;
                for (int k = 0; k < ndd; ++k) {
                    if (wsDD[k].np <= 0) continue;
                    const ManDD &m = wsDD[k];
                    const int ma = m.a, mb = m.b, mnp = m.np;
                    const float muD = m.muD;
                    BodyS Ab, Bb;
                    body_load(q, ma, Ab); body_load(q, mb, Bb);
                    const V3 n = ld3(m.n);
#pragma unroll 1
                    for (int j = 0; j < mnp; ++j)
                        solve_point_velocity<true>(Ab, Bb, n, ld3(m.rA[j]), ld3(m.rB[j]), m.lam[j], muD);
                    body_store_vel(q, ma, Ab); body_store_vel(q, mb, Bb);
                }
            }
        }
        __syncthreads();
        HS_STAMP(9)
        // ---------- P6b: ground velocity pass ----------
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
            const int slot = l + jb * G;
            if (!wok || slot >= kNumDSlots || g_np[jb] == 0) continue;
            const int obj = meta_obj(pw.g.meta[slot]);
            BodyS me;
            body_load(pw, slot, me);
            const V3 gn = -ld3(pw.g.plane[0]);
            const float gmuD = 0.5f * (obj_mu_d(obj) + obj_mu_d(OBJ_PLANE));
#pragma unroll 1
            for (int j = 0; j < g_np[jb]; ++j)
                solve_point_velocity<false>(me, none, gn, hull_local_vertex(obj, (g_vi[jb] >> (3 * j)) & 7), V3{0.f, 0.f, 0.f},
                                            sel4(g_lam[jb], j), gmuD);
            body_store_vel(pw, slot, me);
        }
        __syncthreads();
        HS_STAMP(10)
        // ---------- P6c: wall / extra-plane velocity pass ----------
        if (tid < PACK) {
            const int total = nWall;
            for (int it = tid; it < total; it += PACK) {
                const int item = wallItems[it];
                const int g2 = item >> 8, slot = item & 0xff;
                PhysWorld &q = sh[g2];
                ManS *const wsSC = wsSCb + (size_t)(wbase + g2) * kMaxSCand;
                BodyS me;
                body_load(q, slot, me);
                const int nsc = q.nsc;
                for (int k = 0; k < nsc; ++k) {
                    if (q.scBody[k] != slot) continue;
                    if (wsSC[k].np <= 0) continue;
                    const ManS &m = wsSC[k];
                    const int mnp = m.np; const float muD = m.muD;
                    const V3 n = ld3(m.n);
#pragma unroll 1
                    for (int j = 0; j < mnp; ++j)
                        solve_point_velocity<false>(me, none, n, ld3(m.rA[j]), V3{0.f, 0.f, 0.f}, m.lam[j], muD);
                }
                body_store_vel(q, slot, me);
            }
        }
        __syncthreads();
        HS_STAMP(11)
    }

    // ---------------- agentZeroVelSystem (sim.cpp:258-268) ----------------
    if (instant && wok) {
        for (int agent = l; agent < kMaxAgents; agent += G) {
            const int slot = kAgentSlot0 + agent;
            if (pw.g.meta[slot] == 0) continue;
            pw.lin[slot][0] = 0.f; pw.lin[slot][1] = 0.f; pw.lin[slot][2] = fminf(pw.lin[slot][2], 0.f);
            pw.ang[slot][0] = 0.f; pw.ang[slot][1] = 0.f; pw.ang[slot][2] = 0.f;
        }
    }
    __syncthreads();

    // ---------------- rewardsVisSystem (sim.cpp:763-804): work items = (seeker, hider) pairs ----------------
    bool seen = false;
    if (wok) {
        for (int it = l; it < 9; it += G) {
            const int si = it / 3, hi_ = it % 3;
            if (si < cnt_seekers(pw.counts) && hi_ < cnt_hiders(pw.counts)) {
                const int ss = kAgentSlot0 + team_seeker(pw.teams, si), hs_ = kAgentSlot0 + team_hider(pw.teams, hi_);
                const V3 spos = geom_pos(pw.g, ss);
                const V3 fwd = qrot(geom_rot(pw.g, ss), {0.f, 1.f, 0.f});
                V3 to = geom_pos(pw.g, hs_) - spos;
                float c = dot(normalize(to), fwd);
                if (!(c < kCosFovHalf)) {
                    float t;
                    if (trace_ray(pw.g, spos, to, 1.f, &t) == hs_) seen = true;
                }
            }
        }
    }
    if (seen) pw.hiderReward = -1.f;      // every writer stores the same value
    __syncthreads();

    // ---------------- outputRewardsDonesSystem (sim.cpp:806-841) ----------------
    if (wok) {
        for (int agent = l; agent < A_; agent += G) {
            if (!team_agent_active(pw.teams, agent)) continue;
            const int slot = kAgentSlot0 + agent;
            const int row = w * A_ + agent;
            const int step = pw.step;
            if (step == 0) S.xDone[row] = 0;
            if (step < kNumPrepSteps - 1) {
                S.xReward[row] = 0.f;
            } else {
                if (step == kEpisodeLen - 1) S.xDone[row] = 1;
                float r = pw.hiderReward;
                if (team_agent_type(pw.teams, agent) == AGENT_SEEKER) r *= -1.f;
                if (fabsf(pw.g.pos[slot][0]) >= 18.f || fabsf(pw.g.pos[slot][1]) >= 18.f) r -= 10.f;
                S.xReward[row] = r;
            }
        }
    }
    // ---------------- updateEpisodeResultsSystem (sim.cpp:843-893) ----------------
    if (wok && l == 0) {
        float *res = S.xEpisodeResult + w * 2;
        const int step = pw.step;
        int s0 = S.runningScores[0 * N + w], s1 = S.runningScores[1 * N + w];
        if (step == 0) { res[0] = 0.f; res[1] = 0.f; s0 = 0; s1 = 0; }
        if (step >= kNumPrepSteps) {
            const bool hidden = pw.hiderReward == 1.f;
            const bool sf = cnt_seekers_first(pw.counts) != 0;
            const int win = hidden ? (sf ? 1 : 0) : (sf ? 0 : 1);
            if (win == 0) s0 += 1; else s1 += 1;
        }
        if (step == kEpisodeLen - 1) {
            if (s0 > s1) { res[0] = 1.f; res[1] = 0.f; }
            else if (s0 < s1) { res[0] = 0.f; res[1] = 1.f; }
            else { res[0] = 0.5f; res[1] = 0.5f; }
        }
        S.runningScores[0 * N + w] = s0; S.runningScores[1 * N + w] = s1;
        S.hiderTeamReward[w] = pw.hiderReward;
    }

    // ---------------- write the world back to HBM ----------------
    if (wok) {
        for (int s = l; s < kNumDSlots; s += G) {
            S.bmeta[s * N + w] = pw.g.meta[s];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                S.bpos[(c * kNumDSlots + s) * N + w] = pw.g.pos[s][c];
                S.blin[(c * kNumDSlots + s) * N + w] = pw.lin[s][c];
                S.bang[(c * kNumDSlots + s) * N + w] = pw.ang[s][c];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) S.brot[(c * kNumDSlots + s) * N + w] = pw.g.rot[s][c];
        }
        for (int i = l; i < kMaxAgents; i += G) {
            S.grabOther[i * N + w] = pw.grabOther[i];
#pragma unroll
            for (int c = 0; c < 8; ++c) S.grabData[(c * kMaxAgents + i) * N + w] = pw.grabData[i][c];
#pragma unroll
            for (int c = 0; c < 4; ++c) S.aforce[(c * kMaxAgents + i) * N + w] = pw.aforce[i][c];
        }
    }
    HS_STAMP_FLUSH
}

}  // namespace hs
