// Physics step: movement / actions, PhysicsSystem::setupPhysicsStepTasks (src/sim.cpp:1162-1163; engine
// source absent — DESIGN.md "Engine decisions") and the reward systems as ONE kernel, ONE WAVE PER OCTET.
// (Everything below is templated on the tile T = worlds per wave: T = 8, the octet, is the default and what the text
// describes; T = 4 — HS_TILE=4, a wave per half octet, 16 lanes per world, four waves per SIMD — is the same code, bit-exact,
// and measured 8 - 10 % slower: DESIGN.md §5.)
//
// A workgroup is a single 64-lane wave that owns 8 consecutive worlds (an octet, hs_state.h) for the whole step.
// The octet's working set — pose, previous pose, velocities and meta word of every body — is copied from its
// contiguous blocks of the tiled columns into LDS once (struct OctResT<8>, 20 KiB: 8 such waves share a CU's 160 KiB),
// stays there through movement, the four XPBD substeps and the rewards, and is copied back once.  Nothing in the
// step is shared between waves, so there is NO barrier between waves anywhere: a wave walks its own worlds
// through the phases at its own pace while the other wave of its SIMD fills its stalls.  (The round-1 kernel ran
// 64 worlds per 8-wave workgroup with a workgroup barrier after every phase: each of the ~26 phases of a step
// then lasted as long as its slowest item among 64 worlds, and the average wave issued one instruction every
// ~27 clocks.)
//
// Lane mappings inside the wave (L = lane):
//   bodies     compact list of the octet's existing bodies, 64 per round (2 rounds for the benchmark's ~12 bodies per
//              world; bodies with a wall contact in the previous step first, agents before boxes): integrate, ground
//              contacts, velocity pass; the last round's idle lanes solve the wall manifolds of earlier rounds'
//              bodies.  A body keeps its (round, lane) for the whole step: its ground manifold lives in registers.
//   worlds     8 lanes per world (world = L / 8): broadphase, body-body solve, rewards.
//   pairs      2 neighbouring lanes per candidate pair (16 per pair with a wedge): exact convex test.
// Substep s:  [integrate s=0] -> detect -> sat -> dd<pos> -> body_pos -> dd<vel> -> body_vel (+ integrate s+1)
// The Gauss-Seidel order and every rounding are the oracle's (joints, body-body in pair order, then per body:
// ground, walls by static id).  The contact manifolds of a substep live in LDS (manifold slots in the clip buffers),
// the level generator at the tail of the step works in the octet's LDS as well: the kernel uses no scratch memory.
#pragma once
#include "hs_state.h"
#include "hs_rays.h"
#include "hs_collide.h"
#include "hs_solver.h"
#include "hs_k_reset.h"

// weight of a body-body candidate pair against a body-static one in the load estimate k_balance sorts the worlds by
#ifndef HS_LOAD_DD_WEIGHT
#define HS_LOAD_DD_WEIGHT 1
#endif

namespace hs {

constexpr int kPhysThreads = 64;                   // one wave
#ifdef HS_EXP_REGCAP
constexpr int kPhysDynLds = 20 * 1024;
#else
constexpr int kPhysDynLds = 0;
#endif
// The kernel exists for two tile sizes T (template parameter of the resident set and of every phase): T = 8, a wave per octet
// of the tiled columns, two waves per SIMD (20 KiB of LDS, <= 256 registers) — and T = 4, a wave per half octet, four waves
// per SIMD (10 KiB, 128 registers).  G = 64 / T lanes per world in the world-mapped phases.
constexpr int kLdsWalls = 32;                      // walls per world staged in LDS for the broadphase (a world has
                                                   // 4..34; the rare ones beyond 32 are read from global memory)
static_assert(kMaxSCand <= 32 && kMaxDDCand <= 16, "accepted-manifold masks are one word per world");
// Candidate pairs of a world beyond the LDS capacities (kMaxDDCand / kMaxSCand) are NOT dropped: they spill.  The
// reference has no cap below its entity count (src/sim.cpp:1356-1361), so the global workspace holds a manifold for
// EVERY possible pair of a world — 17 bodies: 136 body-body pairs; 17 x (36 walls + 2 extra planes) body-static pairs —
// indexed by the pair's place in the world's candidate order (the oracle's order).  Places below the capacity are the
// fast path (LDS lists, lane-parallel solve); a place at or beyond it has its pair code in S.spPair and is tested and
// solved after the fast ones of its kind, one after the other — which IS the oracle's order, because candidates are
// generated in pair order (body-body) and per body in candidate order (statics).  Rare (two pairs in 15 M world-steps
// of the training configuration), counted in S.status, bit-identical to the unbounded oracle
// (libhideseek_smallcap.so runs every pair but the first through this path: tests/test_gpu_status.py).
constexpr int kAllDD = kNumDSlots * (kNumDSlots - 1) / 2;                 // 136
constexpr int kAllSC = kNumDSlots * (kMaxWalls + kMaxPlanes - 1);         // 646
constexpr int kSpInfoWords = kNumDSlots + 1;      // S.spInfo per world: totals (dd | sc << 16), then per body: first spilled static | count << 16

// ---- the octet's resident working set (LDS) ----
// Every column is [row][world of the octet] exactly like its block in HBM (hs_state.h Col), so loading and storing
// are linear copies, and a wave whose lanes are (slot, world) or (world, slot) pairs touches 64 distinct banks.
template <int T> struct alignas(16) OctResT {
    static constexpr int kT = T;
    static constexpr int kG = 64 / T;                                 // lanes per world
    static constexpr int kClip = clip_lanes(T);                       // contact lanes per round
    static constexpr int kClipWords = clip_words(kClip);
    static constexpr int kSatPairs = kClip < hs::kSatPairs ? kClip : hs::kSatPairs;   // pairs per axis-search round (never more than a contact round can take)
    static constexpr int kMaxItems = T * (kMaxDDCand + kMaxSCand);    // convex-test items of a tile
    static constexpr int kSlots = kClip < 31 ? kClip : 31;            // manifold slots in the clip buffers (location 31 = global)
    static constexpr int kVelOffsetWords = T == 8 ? 2 * 8 * 3 * 32 : 36 * 24;   // beyond the manifold slots and the hull AABBs
    float pos[3][kNumDSlots][T];
    float rot[4][kNumDSlots][T];        // w, x, y, z
    float ppos[3][kNumDSlots][T];       // pose at the start of the substep
    float prot[4][kNumDSlots][T];
    int meta[kNumDSlots][T];            // meta_pack(); 0 = empty slot
    // Scratch of the broadphase / the convex tests.  The work list of the convex tests is written when the
    // broadphase loops are over (the walls are dead by then) and lies beyond the clip buffers that the convex
    // tests use, so both views can be live where they need to be.
    union {
        struct {
            float lo[3][kNumDSlots][T], hi[3][kNumDSlots][T];   // hull AABBs: integrate -> detect
            float wall[4][kLdsWalls][T];                            // cx, cy, hx, hy: staged by detect
        } det;
        struct {
            float clip[kClipWords];                                     // polygon clipping of the convex tests
            unsigned short items[kMaxItems + 32];                       // world << 6 | candidate (32+ = static); ramp items start at a multiple of 32
            int pend[4][kClip];                                    // colliding pairs waiting for contact generation: item | axis code << 16, axis xyz
        } sat;
        // The velocities share the memory of the convex tests' scratch: between the integration of a substep (which consumes
        // them) and the derivation of the new ones from the pose change (after the position solve) the velocity of a DYNAMIC
        // body is dead, and that is exactly when the broadphase stages the walls and the convex tests clip their polygons
        // here — which buys the scratch of 48 contact lanes per round instead of 32 (most substeps of the benchmark have
        // about 32 colliding pairs per octet: one contact round instead of two).  The velocity of a body that is NOT
        // dynamic (a locked box, an empty slot) is frozen (DESIGN.md "Engine decisions"); it is destroyed here and
        // therefore never written back: copy_out_vel stores the rows of dynamic bodies only, HBM keeps the others.
        // The manifold slots (clip words 0 .. 31 * 36) lie below the velocities and stay valid through the solver phases.
        struct {
            float pad[kVelOffsetWords];
            float lin[3][kNumDSlots][T];
            float ang[3][kNumDSlots][T];
        } vel;
    } u;
    unsigned short ddPair[kMaxDDCand][T];   // a | b << 5 | manifold location << 11 (pair_*() below)
    unsigned short scPair[kMaxSCand][T];    // body | static << 5 | location << 11  (static = wall index, 36 + plane index)
    unsigned short scInfo[kNumDSlots][T];   // per body: first static candidate | count << 8
    unsigned int scAcc[T];                  // bit k: static candidate k of the world has a manifold (set by the convex test)
    unsigned int ddAcc[T];                  // bit k: body-body candidate k has a manifold
    unsigned int ddOrd[2][T];               // the accepted body-body candidates in solve order, 4 bits each (phase_dd)
    float plane0[4][T];                     // the ground plane nx, ny, nz, d of every world
    unsigned char bodies[kNumDSlots * T];   // compact list of existing bodies: slot << 3 | world
    union {
        unsigned char wallBodies[kNumDSlots * T];   // bodies with a wall / extra-plane manifold in this substep
        unsigned char actGL[kMaxAgents][T];         // grab / lock requests (phase_pre only)
    };
    unsigned int wallSeen[T];               // bit s: body slot s had a wall / extra-plane manifold (previous step's while the
                                                // body list is built, then this step's: SimState::wallHist)
    unsigned char numWalls[T], numPlanes[T], ndd[T], nsc[T], seen[T], hasGrab[T];
    unsigned char spill[T];                 // this substep: bit 0 body-body, bit 1 body-static candidates beyond the LDS capacity
    int wid[T];                             // world id of each slot of the octet (SimState::worldOfSlot), -1 = empty slot
};
static_assert(sizeof(OctResT<8>) <= 20 * 1024, "8 waves of 8 worlds share the CU's 160 KiB of LDS");
static_assert(sizeof(OctResT<4>) <= 10 * 1024, "16 waves of 4 worlds share the CU's 160 KiB of LDS");
static_assert(offsetof(OctResT<8>, u.sat.items) >= offsetof(OctResT<8>, u.det.wall) && offsetof(OctResT<4>, u.sat.items) >= offsetof(OctResT<4>, u.det.wall), "items must not overlap the AABBs");
static_assert(sizeof(ManDD) / 4 * OctResT<8>::kSlots <= OctResT<8>::kVelOffsetWords && sizeof(ManDD) / 4 * OctResT<4>::kSlots <= OctResT<4>::kVelOffsetWords, "manifold slots lie below the velocities");
static_assert(2 * 3 * kNumDSlots * 8 <= OctResT<8>::kVelOffsetWords && 2 * 3 * kNumDSlots * 4 <= OctResT<4>::kVelOffsetWords, "the hull AABBs lie below the velocities");

// ---- candidate pairs and where their contact manifolds live ----
// A manifold produced by the LAST contact-generation round of a substep (nearly always the only one) stays in LDS:
// when every lane of that round is done with its clipping scratch, lane i writes its manifold into slot i of the clip
// buffers (36 consecutive words, moved 16 bytes at a time), where the solver phases of the substep read it.
// Slots 0..30; location 31 = the manifold is in the global workspace (S.wsDD / S.wsSC): earlier rounds of a substep
// with more than 32 colliding pairs, the 32nd lane of the last round, extra planes of the debug levels.
constexpr int kLocGlobal = 31;
HSD int pair_a(int p) { return p & 31; }
HSD int pair_b(int p) { return (p >> 5) & 63; }
HSD int pair_loc(int p) { return (p >> 11) & 31; }
HSD int pair_pack(int a, int b) { return a | (b << 5) | (kLocGlobal << 11); }
static_assert(kNumDSlots <= 32 && kMaxWalls + kMaxPlanes <= 64, "pair encoding");
static_assert(sizeof(ManDD) % 16 == 0 && sizeof(ManS) % 16 == 0, "manifold records move 16 bytes at a time");
constexpr int kManWords = sizeof(ManDD) / 4;          // slot stride (a ManS record uses the first 28 words)
template <typename M> HSD void man_lds_load(const float *clip, int slot, M &m) {
    m = *reinterpret_cast<const M *>(clip + slot * kManWords);          // 16-byte reads
}
// accumulated normal multipliers of a manifold in LDS: the last 4 words of either record
template <typename M> HSD void man_lds_set_lam(float *clip, int slot, int j, float v) {
    static_assert(offsetof(M, lam) == sizeof(M) - 16, "lam is the record's tail");
    clip[slot * kManWords + (sizeof(M) / 4 - 4 + j)] = v;
}

// ---- accessors of the resident columns ----
template <int C, int T> HSD V3 rld3(const float (&a)[C][kNumDSlots][T], int slot, int g) { return {a[0][slot][g], a[1][slot][g], a[2][slot][g]}; }
template <int T> HSD Q rld4(const float (&a)[4][kNumDSlots][T], int slot, int g) { return {a[0][slot][g], a[1][slot][g], a[2][slot][g], a[3][slot][g]}; }
template <int T> HSD void rst3(float (&a)[3][kNumDSlots][T], int slot, int g, V3 v) { a[0][slot][g] = v.x; a[1][slot][g] = v.y; a[2][slot][g] = v.z; }
template <int T> HSD void rst4(float (&a)[4][kNumDSlots][T], int slot, int g, Q q) { a[0][slot][g] = q.w; a[1][slot][g] = q.x; a[2][slot][g] = q.y; a[3][slot][g] = q.z; }

template <class OR> HSD void rbody_load(const OR &R, int g, int slot, BodyS &b) {
    b.pos = rld3(R.pos, slot, g); b.rot = rld4(R.rot, slot, g);
    b.ppos = rld3(R.ppos, slot, g); b.prot = rld4(R.prot, slot, g);
    b.lin = rld3(R.u.vel.lin, slot, g); b.ang = rld3(R.u.vel.ang, slot, g);
    const int m = R.meta[slot][g];
    const bool dyn = m != 0 && meta_resp(m) == RESP_DYNAMIC;
    b.invM = dyn ? obj_inv_mass(meta_obj(m)) : 0.f;
    b.invI = dyn ? obj_inv_inertia(meta_obj(m)) : V3{0.f, 0.f, 0.f};
    body_refresh_inertia(b);
}
template <class OR> HSD void rbody_store_pose(OR &R, int g, int slot, const BodyS &b) { rst3(R.pos, slot, g, b.pos); rst4(R.rot, slot, g, b.rot); }
template <class OR> HSD void rbody_store_vel(OR &R, int g, int slot, const BodyS &b) { rst3(R.u.vel.lin, slot, g, b.lin); rst3(R.u.vel.ang, slot, g, b.ang); }
HSD void derive_velocity(BodyS &b) {
    const float h = kSubstepH;
    b.lin = (b.pos - b.ppos) * (1.f / h);
    Q dq = qmul(b.rot, qinv(b.prot));
    V3 wv = V3{dq.x, dq.y, dq.z} * (2.f / h);
    b.ang = dq.w >= 0.f ? wv : -wv;
}

// Hand-offs between the lanes of the wave.  Through LDS: the LDS operations of a wave execute in order, so all it
// takes is that the compiler keeps them in program order and the data has landed (lgkmcnt).  Through global memory
// (the manifold workspace, the agents' forces): the stores must have left the wave first (vmcnt).
HSD void wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
HSD void mem_sync() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

// Prefix sums and totals across lanes by DPP (one vector instruction per step) instead of shuffles through the LDS
// crossbar (a round trip each, and the broadphase chains a dozen of them).
template <int D> HSD int dpp_row_shr(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x110 + D, 0xF, 0xF, true); }   // lane i <- lane i - D of its row of 16, 0 beyond
// inclusive prefix sum over the G = 8 or 16 lanes of a world (l = lane % G): the steps' reach never leaves the group where it counts
template <int G> HSD int scanG_incl(int x, int l) {
    int y;
    y = dpp_row_shr<1>(x); if (l >= 1) x += y;
    y = dpp_row_shr<2>(x); if (l >= 2) x += y;
    y = dpp_row_shr<4>(x); if (l >= 4) x += y;
    if (G == 16) { y = dpp_row_shr<8>(x); if (l >= 8) x += y; }
    return x;
}
// sum over the G = 8 or 16 lanes of a world, in every lane: pairs, quads (quad_perm), then the two quads of the half row (mirror)
template <int G> HSD int sumG_all(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false);      // quad_perm [1,0,3,2]
    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false);      // quad_perm [2,3,0,1]
    x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false);     // row_half_mirror
    if (G == 16) x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false);    // row_mirror: the other half row
    return x;
}
// inclusive prefix sum over the wave's 64 lanes
HSD int scan64_incl(int x) {
    x += dpp_row_shr<1>(x); x += dpp_row_shr<2>(x); x += dpp_row_shr<4>(x); x += dpp_row_shr<8>(x);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);     // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);     // row_bcast:31 -> rows 2, 3
    return x;
}

// Geometry view of one world of the octet for trace_ray (hs_rays.h): bodies from the resident columns, walls and
// the (at most 3) planes from global memory (the rays of the physics kernel — lock / grab, seeker -> hider line
// of sight — are few; the lidar / visibility rays are k_observe's).
template <class OR> struct ResGeom {
    const OR &R; const SimState &S; int g; int w;
    HSD int g_meta(int i) const { return R.meta[i][g]; }
    HSD V3 g_pos(int i) const { return rld3(R.pos, i, g); }
    HSD Q g_rot(int i) const { return rld4(R.rot, i, g); }
    HSD int g_num_walls() const { return R.numWalls[g]; }
    HSD float g_wall(int k, int c) const { return S.walls(c * kMaxWalls + k, w); }
    HSD int g_num_planes() const { return R.numPlanes[g]; }
    HSD float g_plane(int p, int c) const { return S.planes(c * kMaxPlanes + p, w); }
};

// What a body's lane keeps in registers for the whole step: which body it is and the
// ground-plane manifold of the current substep (up to 4 deepest vertices, hs_collide.h ground_manifold).
constexpr float kGroundOff = 0.f;
struct BodyReg {
    int np, vidx;            // ground manifold: contact count, 3 bits of vertex index per contact
    float lam[4];            // accumulated normal multipliers
    // (No plane offsets dot(pB, n) per contact: plane 0 is the ground n = (0, 0, 1), d = 0 in every level — makePlane,
    // level_gen.cpp:68-71, 294-295 and each debug level — and for that plane the offset is a signed zero in exact IEEE
    // arithmetic: pB = v - pn * dist with dist = pn.v - 0 = v.z gives pB.z = +0, and dot(pB, -pn) sums products with
    // zero.  `x - (+-0)` is x for every x but a zero, whose sign no comparison or sum below can see.  kGroundOff stands for
    // it; the oracle computes the offset from the plane it is given and gets the same bits.)
};

// ------------------------------------------------------------------------------------------
// Copies between the tile's part of the octet's blocks of the tiled columns and LDS, 16 bytes per lane and trip.  A block is
// ROWS x 8 words; a tile of 8 takes whole rows (a linear copy), a tile of 4 the lower or upper half of every row.
// p0 = the tile's first slot (a multiple of T).
template <int T, typename E, int ROWS>
HSD void copy_in(E *dst, const Col<E, ROWS> &col, int p0) {
    static_assert(sizeof(E) == 4 && (T == 4 || T == 8), "16-byte pieces");
    constexpr int PR = T / 4;                        // pieces per row
    const float4 *src = reinterpret_cast<const float4 *>(col.octet(p0 >> 3)) + ((p0 & 7) >> 2);
    float4 *d = reinterpret_cast<float4 *>(dst);
    for (int i = hs_lane(); i < ROWS * PR; i += 64) d[i] = src[(i / PR) * 2 + i % PR];
}
template <int T, typename E, int ROWS>
HSD void copy_out(const Col<E, ROWS> &col, int p0, const E *src) {
    static_assert(sizeof(E) == 4 && (T == 4 || T == 8), "16-byte pieces");
    constexpr int PR = T / 4;
    float4 *dst = reinterpret_cast<float4 *>(col.octet(p0 >> 3)) + ((p0 & 7) >> 2);
    const float4 *s = reinterpret_cast<const float4 *>(src);
    for (int i = hs_lane(); i < ROWS * PR; i += 64) dst[(i / PR) * 2 + i % PR] = s[i];
}
// The velocity columns: only the rows of DYNAMIC bodies go back (OctResT::u.vel: the others were scratch during the step).
template <class OR>
HSD void copy_out_vel(const Col<float, 3 * kNumDSlots> &col, int p0, const float *src, const OR &R) {
    constexpr int T = OR::kT;
    float *dst = col.octet(p0 >> 3) + (p0 & 7);
    for (int i = hs_lane(); i < 3 * kNumDSlots * T; i += 64) {
        const int row = i / T, g = i % T, slot = row % kNumDSlots;
        const int m = R.meta[slot][g];
        if (m != 0 && meta_resp(m) == RESP_DYNAMIC) dst[row * kTile + g] = src[i];
    }
}

// ------------------------------------------------------------------------------------------
// Start of a substep for one body: remember the pose, semi-implicit Euler step (gravity, agent force and
// torque, gyroscopic term), refresh the hull AABB and the ground-plane manifold (registers of the body's lane).
// `aforce`: the octet's block of S.aforce (ExternalForce xyz + ExternalTorque z of the agents, written by phase_pre);
// read here, four times per step by the agents' lanes, rather than kept in four registers for the whole step — the step
// sits at its register budget.
template <class OR> HSD void integrate_body(OR &R, BodyReg &b, int slot, int g, int meta, const float *aforce) {
    const int obj = meta_obj(meta);
    const bool dyn = meta_resp(meta) == RESP_DYNAMIC;
    V3 force = {0.f, 0.f, 0.f}; float torque = 0.f;
    if (dyn && slot >= kAgentSlot0) {
        const float *f = aforce + (slot - kAgentSlot0) * kTile + g;          // (rows of the octet's block are 8 worlds wide whatever the tile)
        force = {f[0 * kMaxAgents * kTile], f[1 * kMaxAgents * kTile], f[2 * kMaxAgents * kTile]};
        torque = f[3 * kMaxAgents * kTile];
    }
    V3 pos = rld3(R.pos, slot, g); Q rot = rld4(R.rot, slot, g);
    rst3(R.ppos, slot, g, pos); rst4(R.prot, slot, g, rot);
    if (dyn) {
        V3 lin = rld3(R.u.vel.lin, slot, g), ang = rld3(R.u.vel.ang, slot, g);
        const float h = kSubstepH;
        const float invM = obj_inv_mass(obj);
        const V3 invI = obj_inv_inertia(obj);
        lin = madd(lin, madd(V3{0.f, 0.f, kGravityZ}, force, invM), h);
        pos = madd(pos, lin, h);
        Q qi = qinv(rot);
        V3 wloc = qrot(qi, ang), tl = qrot(qi, V3{0.f, 0.f, torque});
        const V3 I = obj_inertia(obj);            // 1 / invI per axis, 0 where invI is 0
        V3 Iw = mulc(I, wloc);
        wloc = madd(wloc, mulc(invI, tl - cross(wloc, Iw)), h);
        ang = qrot(rot, wloc);
        rot = quat_add_rotation(rot, ang * h);
        rst3(R.pos, slot, g, pos); rst4(R.rot, slot, g, rot);
        rst3(R.u.vel.lin, slot, g, lin); rst3(R.u.vel.ang, slot, g, ang);
    }
    V3 lo, hi;
    const HullRef hb = hull_ref_body(obj, pos, rot);
    hull_aabb(hb, &lo, &hi);
    rst3(R.u.det.lo, slot, g, lo); rst3(R.u.det.hi, slot, g, hi);
    // ground plane (plane 0) manifold at the integrated pose
    b.np = 0; b.vidx = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) b.lam[j] = 0.f;
    if (dyn && R.numPlanes[g] >= 1) {
        const V3 pn = {R.plane0[0][g], R.plane0[1][g], R.plane0[2][g]};
        b.np = ground_manifold(hb, pn, R.plane0[3][g], &b.vidx);
    }
}

// ------------------------------------------------------------------------------------------
// All-pairs AABB candidates (<= 17 bodies, <= 36 walls per world: no BVH), 8 lanes per world: lane l of a world
// owns body slots l, l + 8, l + 16.  Also builds the octet's work list of convex tests.  Returns the number of
// box-only items in .x and of items that involve a ramp (wedge hull) in .y; the ramp items start at the next
// multiple of 32 so that most rounds of the convex test run the box code only.
struct ItemCounts { int nbox, nwedge; bool anySpill; };
template <int JB, class OR>   // JB body slots per lane: with 8 lanes per world 2 cover 16 slots (<= 5 agents), 3 all 17; with 16 lanes 1 / 2
HSD ItemCounts phase_detect(const SimState &S, OR &R, int NS) {
    constexpr int T = OR::kT, G = OR::kG;
    const int L = hs_lane(), g = L / G, l = L % G;
    // the tile's walls -> LDS, 16 bytes per lane and component (rows beyond a world's count are never read)
    {
        constexpr int PR = T / 4;                                   // 16-byte pieces per row of the tile
        static_assert(kLdsWalls * PR <= 64, "one piece per lane per component");
        const float *src = S.walls.octet(S.wbeg >> 3) + (S.wbeg & 7);
        float4 v[4];
        const bool on = L < kLdsWalls * PR;
        const int row = L / PR, part = L % PR;
#pragma unroll
        for (int c = 0; c < 4; ++c) if (on) v[c] = *(const float4 *)(src + (c * kMaxWalls + row) * kTile + 4 * part);
#pragma unroll
        for (int c = 0; c < 4; ++c) if (on) *(float4 *)(&R.u.det.wall[c][row][4 * part]) = v[c];
    }
    if (l == 0) { R.scAcc[g] = 0u; R.ddAcc[g] = 0u; }
    wave_sync();
    const int nwl = R.numWalls[g], npl = R.numPlanes[g];
    int tot_items = 0;
    unsigned dd_mask[JB] = {}; unsigned long long s_mask[JB] = {};
    int bdd[JB] = {}, bsc[JB] = {}, add[JB] = {}, asc[JB] = {};
    bool have[JB], dynamic[JB]; V3 lo[JB], hi[JB];
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) {
        const int slot = l + jb * G;
        const int meta = slot < NS ? R.meta[slot][g] : 0;
        have[jb] = meta != 0;
        dynamic[jb] = have[jb] && meta_resp(meta) == RESP_DYNAMIC;
        lo[jb] = {0.f, 0.f, 0.f}; hi[jb] = {0.f, 0.f, 0.f};
        if (have[jb]) { lo[jb] = rld3(R.u.det.lo, slot, g); hi[jb] = rld3(R.u.det.hi, slot, g); }
    }
    // The loops run over the OTHER body / the wall, each read from LDS once and tested against all of the lane's slots.
    // (no branch on the other body's presence: the loads of several trips are in flight together)
    // (one body slot per lane — the 4-world wave — rolled: unrolled, its comparisons are kept as lane masks in scalar
    // registers, a thousand of which then spill)
    constexpr int kDetUnroll = JB == 1 ? 1 : 4;
#pragma unroll kDetUnroll
    for (int j = 1; j < NS; ++j) {
        const int mj = R.meta[j][g];
        const bool dynj = meta_resp(mj) == RESP_DYNAMIC;
        const V3 loj = rld3(R.u.det.lo, j, g), hij = rld3(R.u.det.hi, j, g);
#pragma unroll
        for (int jb = 0; jb < JB; ++jb) {
            const bool hit = (mj != 0) & have[jb] & (l + jb * G < j) & (dynamic[jb] | dynj) &
                (lo[jb].x <= hij.x) & (loj.x <= hi[jb].x) & (lo[jb].y <= hij.y) & (loj.y <= hi[jb].y) &
                (lo[jb].z <= hij.z) & (loj.z <= hi[jb].z);
            dd_mask[jb] |= hit ? 1u << j : 0u;
        }
    }
    bool zin[JB];                                       // the walls all span z in [0, 2.5]
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) zin[jb] = dynamic[jb] & (lo[jb].z <= 2.5f) & (0.f <= hi[jb].z);
    auto wall_test = [&](int k, float cx, float cy, float hx, float hy) {
        const float wx0 = cx - hx, wx1 = cx + hx, wy0 = cy - hy, wy1 = cy + hy;
#pragma unroll
        for (int jb = 0; jb < JB; ++jb) {
            const bool hit = zin[jb] & (lo[jb].x <= wx1) & (wx0 <= hi[jb].x) & (lo[jb].y <= wy1) & (wy0 <= hi[jb].y);
            s_mask[jb] |= hit ? 1ull << k : 0ull;
        }
    };
    const int nwLds = nwl < kLdsWalls ? nwl : kLdsWalls;
#pragma unroll 4
    for (int k = 0; k < nwLds; ++k) wall_test(k, R.u.det.wall[0][k][g], R.u.det.wall[1][k][g], R.u.det.wall[2][k][g], R.u.det.wall[3][k][g]);
    for (int k = kLdsWalls; k < nwl; ++k) {             // (the few walls beyond the staged ones, from global memory)
        const int w = S.wbeg + g;
        wall_test(k, S.walls(0 * kMaxWalls + k, w), S.walls(1 * kMaxWalls + k, w), S.walls(2 * kMaxWalls + k, w), S.walls(3 * kMaxWalls + k, w));
    }
    // candidate slots in the world's lists: prefix sums in body-slot order (slots l of all lanes, then l + 8, ...),
    // i.e. the oracle's candidate order — a pair's place in it is where its record lives in the workspace
    int tot_dd = 0, tot_sc = 0;
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) {
        if (dynamic[jb]) for (int p = 1; p < npl; ++p) s_mask[jb] |= 1ull << (kMaxWalls + p);
        const int cdd = __popc(dd_mask[jb]), csc = __popcll(s_mask[jb]);
        const int in_dd = scanG_incl<G>(cdd, l), in_sc = scanG_incl<G>(csc, l);
        bdd[jb] = tot_dd + in_dd - cdd; bsc[jb] = tot_sc + in_sc - csc;
        tot_dd += sumG_all<G>(cdd); tot_sc += sumG_all<G>(csc);
        add[jb] = cdd ? max(0, min(cdd, kMaxDDCand - bdd[jb])) : 0;
        asc[jb] = csc ? max(0, min(csc, kMaxSCand - bsc[jb])) : 0;
        tot_items += add[jb] + asc[jb];
        if (add[jb] != cdd) atomicAdd(&S.status[0], cdd - add[jb]);      // beyond the capacity: the spill path, counted
        if (asc[jb] != csc) atomicAdd(&S.status[1], csc - asc[jb]);
    }
    const bool spillDD = tot_dd > kMaxDDCand, spillSC = tot_sc > kMaxSCand;
    const bool anySpill = __ballot(spillDD || spillSC) != 0ull;
    unsigned short *const spPair = S.spPair + (size_t)(S.wbeg + g) * (kAllDD + kAllSC);
    int *const spInfo = S.spInfo + (size_t)(S.wbeg + g) * kSpInfoWords;
    // ---- the octet's work list: box-only items from the front, items with a ramp behind them (from a multiple of 32)
    int n_wedge = 0;
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) {
        const int slot = l + jb * G;
        const bool ramp = slot >= kRampSlot0 && slot < kRampSlot0 + kMaxRamps;
        if (ramp) n_wedge += add[jb] + asc[jb];
        else { unsigned mm = dd_mask[jb]; for (int i = 0; mm && i < add[jb]; ++i) { const int j = __ffs(mm) - 1; mm &= mm - 1; n_wedge += (j >= kRampSlot0 && j < kRampSlot0 + kMaxRamps) ? 1 : 0; } }
    }
    const int n_box = tot_items - n_wedge;
    const int inc_box = scan64_incl(n_box), inc_wedge = scan64_incl(n_wedge);
    ItemCounts ic;
    ic.anySpill = anySpill;
    ic.nbox = __builtin_amdgcn_readlane(inc_box, 63); ic.nwedge = __builtin_amdgcn_readlane(inc_wedge, 63);
    const int wedge0 = (ic.nbox + 31) / 32 * 32;
    int ibox = inc_box - n_box, iwedge = wedge0 + inc_wedge - n_wedge;
    wave_sync();                          // every lane is done with the walls: the work list may overwrite them
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) {
        const int slot = l + jb * G;
        const bool ramp = slot >= kRampSlot0 && slot < kRampSlot0 + kMaxRamps;
        unsigned mm = dd_mask[jb]; int i = 0;
        while (mm && i < add[jb]) {
            const int j = __ffs(mm) - 1; mm &= mm - 1;
            R.ddPair[bdd[jb] + i][g] = (unsigned short)pair_pack(slot, j);
            const unsigned short item = (unsigned short)((g << 6) | (bdd[jb] + i));
            if (ramp || (j >= kRampSlot0 && j < kRampSlot0 + kMaxRamps)) R.u.sat.items[iwedge++] = item; else R.u.sat.items[ibox++] = item;
            ++i;
        }
        while (mm) {                      // (beyond the capacity: the pair's code goes to the spill list, at its place)
            const int j = __ffs(mm) - 1; mm &= mm - 1;
            spPair[bdd[jb] + i] = (unsigned short)pair_pack(slot, j);
            ++i;
        }
        // oracle order inside a body: extra planes first, then walls by index; the body's candidates
        // occupy the contiguous range [bsc, bsc + asc) of the world's list
        unsigned long long sm = (s_mask[jb] >> kMaxWalls) | (s_mask[jb] << (64 - kMaxWalls) >> (64 - kMaxWalls) << kMaxPlanes); i = 0;
        while (sm && i < asc[jb]) {
            const int bit = __ffsll((long long)sm) - 1; sm &= sm - 1;
            const int k = bit < kMaxPlanes ? kMaxWalls + bit : bit - kMaxPlanes;
            R.scPair[bsc[jb] + i][g] = (unsigned short)pair_pack(slot, k);
            const unsigned short item = (unsigned short)((g << 6) | 32 | (bsc[jb] + i));
            if (ramp) R.u.sat.items[iwedge++] = item; else R.u.sat.items[ibox++] = item;
            ++i;
        }
        const int nspill = sm ? __popcll(sm) : 0;
        while (sm) {
            const int bit = __ffsll((long long)sm) - 1; sm &= sm - 1;
            const int k = bit < kMaxPlanes ? kMaxWalls + bit : bit - kMaxPlanes;
            spPair[kAllDD + bsc[jb] + i] = (unsigned short)pair_pack(slot, k);
            ++i;
        }
        if (spillSC && slot < kNumDSlots) spInfo[1 + slot] = nspill ? ((bsc[jb] + asc[jb]) | (nspill << 16)) : 0;
        if (slot < NS) R.scInfo[slot][g] = (unsigned short)(asc[jb] > 0 ? (bsc[jb] | (asc[jb] << 8)) : 0);
    }
    if (l == 0) {
        R.spill[g] = (unsigned char)((spillDD ? 1 : 0) | (spillSC ? 2 : 0));
        if (spillDD || spillSC) spInfo[0] = tot_dd | (tot_sc << 16);
        R.ndd[g] = (unsigned char)min(tot_dd, kMaxDDCand); R.nsc[g] = (unsigned char)min(tot_sc, kMaxSCand);
        // what k_balance sorts the worlds by: candidate pairs are where an octet's time differs from another's
        if (R.wid[g] >= 0 && tot_dd + tot_sc > 0) S.loadAcc[R.wid[g]] += HS_LOAD_DD_WEIGHT * tot_dd + tot_sc;
#ifdef HS_LOAD_STUDY
        if (R.wid[g] >= 0) {
            long long *const st = S.phaseTicks + phase_ticks_study_base(S.N) + (size_t)R.wid[g] * kStudyWords;
            st[0] += tot_dd; st[1] += tot_sc; st[4] = blockIdx.x;
            // (where the wave runs: HW_ID = wave | simd << 4 | pipe << 6 | cu << 8 | sh << 12 | se << 13 ..., and the XCC)
            if (g == 0) { st[6] = __builtin_amdgcn_s_getreg(4 | (31 << 11)); st[7] = __builtin_amdgcn_s_getreg(20 | (31 << 11)); }
        }
#endif
    }
    wave_sync();
    return ic;
}

// ------------------------------------------------------------------------------------------
// The SPILL PATH: candidate pairs beyond the LDS capacities.  Cold code behind wave-uniform branches.
// (Template parameter SPILL of the phases: false compiles the spill path out — the register report of the fast path alone.)
#define HS_COLD __device__ __forceinline__
struct SpillCtx {
    const float *walls, *planes;         // the tiled columns (hs_state.h Col)
    ManDD *wsDD; ManS *wsSC; unsigned short *spPair; int *spInfo;
    int wbeg;
};
HSD SpillCtx spill_ctx(const SimState &S) { return {S.walls.p, S.planes.p, (ManDD *)S.wsDD, (ManS *)S.wsSC, S.spPair, S.spInfo, S.wbeg}; }

// Convex tests of the spilled pairs: world by world, body-body then body-static, 32 pairs per trip, two lanes per pair
// for the axis search (sat_axes) and the pair's first lane for the contact generation right away (no compaction: this
// is the rare path).  Every spilled pair gets a record at its place of the workspace; np = 0 says "no manifold".
template <class OR> HS_COLD void spill_sat(SpillCtx c, OR *Rp) {
    OR &R = *Rp;
    const int lane = hs_lane() & 63;
    const bool hi = (lane & 1) != 0;
    const Col<float, 4 * kMaxWalls> walls = {const_cast<float *>(c.walls)};
    const Col<float, 4 * kMaxPlanes> planes = {const_cast<float *>(c.planes)};
#pragma unroll 1
    for (int g = 0; g < OR::kT; ++g) {
        const int fl = __builtin_amdgcn_readfirstlane((int)R.spill[g]);
        if (fl == 0) continue;
        const int w = c.wbeg + g;
        const int tot = __builtin_amdgcn_readfirstlane(c.spInfo[(size_t)w * kSpInfoWords]);
#pragma unroll 1
        for (int kind = 0; kind < 2; ++kind) {
            if ((fl & (1 << kind)) == 0) continue;
            const bool isdd = kind == 0;
            const int n = isdd ? tot & 0xffff : tot >> 16;
#pragma unroll 1
            for (int base = isdd ? kMaxDDCand : kMaxSCand; base < n; base += OR::kSatPairs) {
                const int kk = base + (lane >> 1);
                if (kk < n && (lane >> 1) < OR::kSatPairs) {
                    const int pair = c.spPair[(size_t)w * (kAllDD + kAllSC) + (isdd ? 0 : kAllDD) + kk];
                    const int a = pair_a(pair), bsel = pair_b(pair);
                    const int oa = meta_obj(R.meta[a][g]);
                    const V3 pa = rld3(R.pos, a, g);
                    const Q qa = rld4(R.rot, a, g);
                    ManDD *const md = c.wsDD + (size_t)w * kAllDD + kk;
                    ManS *const ms = c.wsSC + (size_t)w * kAllSC + kk;
                    RawManifold raw;
                    raw.np = 0;
                    bool plane = false;
                    if (!isdd && bsel >= kMaxWalls) {           // extra planes (debug levels only): hull against plane
                        const int p = bsel - kMaxWalls;
                        const V3 pn = {planes(0 * kMaxPlanes + p, w), planes(1 * kMaxPlanes + p, w), planes(2 * kMaxPlanes + p, w)};
                        plane = true;
                        if (!hi && !collide_hull_plane(hull_ref_body(oa, pa, qa), pn, planes(3 * kMaxPlanes + p, w), raw)) raw.np = 0;
                    } else {
                        const HullSrc ha = hull_src_body(oa, pa, qa);
                        const HullSrc hb = isdd ? hull_src_body(meta_obj(R.meta[bsel][g]), rld3(R.pos, bsel, g), rld4(R.rot, bsel, g))
                                                : hull_src_wall(walls(0 * kMaxWalls + bsel, w), walls(1 * kMaxWalls + bsel, w),
                                                                walls(2 * kMaxWalls + bsel, w), walls(3 * kMaxWalls + bsel, w));
                        const AxisResult res = sat_axes(ha, hb, hi);
                        const ClipBuf cb = {R.u.sat.clip, lane >> 1, OR::kClip};
                        if (!hi && res.code != 0 && !sat_contact(ha, hb, res, cb, raw)) raw.np = 0;
                    }
                    if (!hi) {
                        const int ob = isdd ? meta_obj(R.meta[bsel][g]) : plane ? OBJ_PLANE : OBJ_WALL;
                        const float muS = 0.5f * (obj_mu_s(oa) + obj_mu_s(ob)), muD = 0.5f * (obj_mu_d(oa) + obj_mu_d(ob));
                        const Q qai = qinv(qa);
                        if (isdd) {
                            md->a = a; md->b = bsel; md->np = raw.np; md->muS = muS; md->muD = muD;
                            st3(md->n, raw.n);
                            const V3 pb = rld3(R.pos, bsel, g);
                            const Q qbi = qinv(rld4(R.rot, bsel, g));
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (j < raw.np) {
                                    st3(md->rA[j], qrot(qai, raw.pA[j] - pa)); st3(md->rB[j], qrot(qbi, raw.pB[j] - pb));
                                    md->lam[j] = 0.f;
                                }
                        } else {
                            ms->np = raw.np; ms->muS = muS; ms->muD = muD;
                            st3(ms->n, raw.n);
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (j < raw.np) {
                                    st3(ms->rA[j], plane ? hull_local_vertex(oa, raw.vidx[j]) : qrot(qai, raw.pA[j] - pa));
                                    ms->offB[j] = dot(raw.pB[j], raw.n); ms->lam[j] = 0.f;
                                }
                        }
                    }
                }
                wave_sync();            // (the clip scratch of this trip is free again)
            }
        }
    }
    mem_sync();
}

// The spilled body-body manifolds of a world, one after the other on ONE lane: the oracle's loop
// (solve_manifold_positions / solve_manifold_velocities), which the lane-pair form of phase_dd reproduces bit for bit.
// (World by world in a wave-uniform loop, here and below: the addresses into the large per-world workspaces are then
// scalar arithmetic.  Per-lane 64-bit pointers would be hoisted to the top of the kernel as loop invariants and
// spilled to scratch memory there — the step's hot path sits exactly at the 256-register budget of two waves per SIMD.)
template <bool POS, class OR>
HS_COLD void spill_dd(SpillCtx c, OR *Rp) {
    OR &R = *Rp;
#pragma unroll 1
    for (int g = 0; g < OR::kT; ++g) {
        if ((__builtin_amdgcn_readfirstlane((int)R.spill[g]) & 1) == 0) continue;
        const int w = c.wbeg + g;
        const int nall = __builtin_amdgcn_readfirstlane(c.spInfo[(size_t)w * kSpInfoWords]) & 0xffff;
        if (hs_lane() == 0) {
#pragma unroll 1
            for (int k = kMaxDDCand; k < nall; ++k) {
                ManDD *const m = c.wsDD + (size_t)w * kAllDD + k;       // (read field by field: a private copy indexed by j would live in scratch memory)
                const int np = m->np;
                if (np <= 0) continue;
                const int ba = m->a, bb = m->b;
                BodyS Ab, Bb;
                rbody_load(R, g, ba, Ab); rbody_load(R, g, bb, Bb);
                const V3 n = ld3(m->n);
                const float mu = POS ? m->muS : m->muD;
#pragma unroll 1
                for (int j = 0; j < np; ++j) {
                    const float lamj = m->lam[j];
                    if (POS) m->lam[j] = lamj + solve_point_position<true>(Ab, Bb, n, ld3(m->rA[j]), ld3(m->rB[j]), 0.f, mu);
                    else solve_point_velocity<true>(Ab, Bb, n, ld3(m->rA[j]), ld3(m->rB[j]), lamj, mu);
                }
                if (POS) { rbody_store_pose(R, g, ba, Ab); rbody_store_pose(R, g, bb, Bb); }
                else { rbody_store_vel(R, g, ba, Ab); rbody_store_vel(R, g, bb, Bb); }
                wave_sync();
            }
        }
    }
}

// The spilled static manifolds: a lane per body of the world walks the body's in order, after the body's ground manifold
// and its static manifolds of the fast path — the oracle's order per body.
template <bool POS, class OR>
HS_COLD void spill_static(SpillCtx c, OR *Rp, int NS) {
    OR &R = *Rp;
#pragma unroll 1
    for (int g = 0; g < OR::kT; ++g) {
        if ((__builtin_amdgcn_readfirstlane((int)R.spill[g]) & 2) == 0) continue;
        const int w = c.wbeg + g;
        const int slot = hs_lane();
        if (slot >= NS) continue;
        const int info = c.spInfo[(size_t)w * kSpInfoWords + 1 + slot];
        const int first = info & 0xffff, cnt = info >> 16;
        if (cnt == 0) continue;
        BodyS me, none;
        rbody_load(R, g, slot, me);
#pragma unroll 1
        for (int k = first; k < first + cnt; ++k) {
            ManS *const m = c.wsSC + (size_t)w * kAllSC + k;
            const int np = m->np;
            if (np <= 0) continue;
            body_refresh_inertia(me);
            const V3 n = ld3(m->n);
            const float mu = POS ? m->muS : m->muD;
#pragma unroll 1
            for (int j = 0; j < np; ++j) {
                const float lamj = m->lam[j];
                if (POS) m->lam[j] = lamj + solve_point_position<false>(me, none, n, ld3(m->rA[j]), V3{0.f, 0.f, 0.f}, m->offB[j], mu);
                else solve_point_velocity<false>(me, none, n, ld3(m->rA[j]), V3{0.f, 0.f, 0.f}, lamj, mu);
            }
        }
        if (POS) rbody_store_pose(R, g, slot, me); else rbody_store_vel(R, g, slot, me);
    }
}

#ifdef HS_PHASE_TIMING
#define HS_TICK_PARAMS , long long &tk, long long (&acc)[10]
#define HS_TICK_ARGS , tk, acc
#define HS_TICK(i) { const long long now_ = wall_clock64(); acc[i] += now_ - tk; tk = now_; }
#else
#define HS_TICK_PARAMS
#define HS_TICK_ARGS
#define HS_TICK(i)
#endif
// HS_FINE_TIMING (with HS_PHASE_TIMING; development aid): the ten slots are re-assigned to look inside the convex tests —
// 0 everything before them, 1 axis search of the box rounds, 2 of the wedge rounds, 3 pending-list upkeep, 4 contact
// generation, 5 manifold records, 6 dd_pos, 7 static position passes, 8 the rest of the substep, 9 post / load / store.
#ifdef HS_FINE_TIMING
#define HS_FTICK(i) HS_TICK(i)
#define HS_CTICK(i, j) HS_TICK(j)
#else
#define HS_FTICK(i)
#define HS_CTICK(i, j) HS_TICK(i)
#endif
// ------------------------------------------------------------------------------------------
// Exact convex tests, in two stages (hs_collide.h).  Stage 1 — the separating-axis search — runs over all the
// octet's candidate pairs, two lanes per pair (2k and 2k + 1), 32 pairs per round; the pairs that collide are appended
// to a pending list.  Stage 2 — contact generation, the long part: polygon clipping on one lane — runs on the pending
// pairs only, compacted, 32 per round: about a third of the candidates collide, so one round serves the whole octet.
template <class OR> HSD HullSrc sat_hull_a(const OR &R, int g, int a) { return hull_src_body(meta_obj(R.meta[a][g]), rld3(R.pos, a, g), rld4(R.rot, a, g)); }
template <class OR> HSD HullSrc sat_hull_b(const SimState &S, const OR &R, int g, int w, bool isdd, int bsel) {
    if (isdd) return hull_src_body(meta_obj(R.meta[bsel][g]), rld3(R.pos, bsel, g), rld4(R.rot, bsel, g));
    // (the staged walls share their LDS with the clip buffers: from global memory here)
    return hull_src_wall(S.walls(0 * kMaxWalls + bsel, w), S.walls(1 * kMaxWalls + bsel, w),
                         S.walls(2 * kMaxWalls + bsel, w), S.walls(3 * kMaxWalls + bsel, w));
}
// stage 2 for the first `npend` pending pairs: lane i < 32 takes pair i.  `last`: this is the substep's last round,
// whose manifolds stay in LDS (in the lane's own clip column); returns whether a manifold went to global memory.
template <class OR> HSD bool sat_flush(const SimState &S, OR &R, int npend, bool last HS_TICK_PARAMS) {
    const int lane = hs_lane() & 63;
    const bool toLds = last && lane < OR::kSlots;
    bool wroteGlobal = false;
    int mw[kManWords]; int mkind = 0, mkk = 0, mg = 0, mpair = 0;      // the manifold for LDS: 1 body-body, 2 body-static
    if (lane < npend) {
        const int item = R.u.sat.pend[0][lane] & 0xffff;
        AxisResult res;
        res.code = R.u.sat.pend[0][lane] >> 16;
        res.ax = {__int_as_float(R.u.sat.pend[1][lane]), __int_as_float(R.u.sat.pend[2][lane]), __int_as_float(R.u.sat.pend[3][lane])};
        const int g = item >> 6, idx = item & 63;
        const int w = S.wbeg + g;
        const bool isdd = idx < 32;
        const int kk = idx & 31;
        const int pair = isdd ? R.ddPair[kk][g] : R.scPair[kk][g];
        const int a = pair_a(pair), bsel = pair_b(pair);
        const ClipBuf cb = {R.u.sat.clip, lane, OR::kClip};
        RawManifold raw;
        const bool plane = res.code == 3;                 // an extra plane of a debug level (static candidate kMaxWalls + p)
        bool have;
        if (plane) {
            const int p = bsel - kMaxWalls;
            const V3 pn = {S.planes(0 * kMaxPlanes + p, w), S.planes(1 * kMaxPlanes + p, w), S.planes(2 * kMaxPlanes + p, w)};
            have = collide_hull_plane(hull_ref_body(meta_obj(R.meta[a][g]), rld3(R.pos, a, g), rld4(R.rot, a, g)), pn, S.planes(3 * kMaxPlanes + p, w), raw);
        } else {
            have = sat_contact(sat_hull_a(R, g, a), sat_hull_b(S, R, g, w, isdd, bsel), res, cb, raw);
        }
        if (have) {
            const int oa = meta_obj(R.meta[a][g]);
            const int ob = isdd ? meta_obj(R.meta[bsel][g]) : plane ? OBJ_PLANE : OBJ_WALL;
            const V3 pa = rld3(R.pos, a, g);
            const Q qai = qinv(rld4(R.rot, a, g));
            const float muS = 0.5f * (obj_mu_s(oa) + obj_mu_s(ob)), muD = 0.5f * (obj_mu_d(oa) + obj_mu_d(ob));
            if (isdd) {
                ManDD m;
                m.a = a; m.b = bsel; m.np = raw.np; m.muS = muS; m.muD = muD;
                st3(m.n, raw.n);
                const V3 pb = rld3(R.pos, bsel, g);
                const Q qbi = qinv(rld4(R.rot, bsel, g));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool on = j < raw.np;
                    st3(m.rA[j], on ? qrot(qai, raw.pA[j] - pa) : V3{0.f, 0.f, 0.f});
                    st3(m.rB[j], on ? qrot(qbi, raw.pB[j] - pb) : V3{0.f, 0.f, 0.f});
                    m.lam[j] = 0.f;
                }
                if (toLds) { __builtin_memcpy(mw, &m, sizeof(m)); mkind = 1; mkk = kk; mg = g; mpair = pair; }
                else { ((ManDD *)S.wsDD + (size_t)w * kAllDD)[kk] = m; wroteGlobal = true; }
                atomicOr(&R.ddAcc[g], 1u << kk);
            } else {
                ManS m;
                m.np = raw.np; m.muS = muS; m.muD = muD; m.pad[0] = 0.f; m.pad[1] = 0.f;
                st3(m.n, raw.n);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool on = j < raw.np;
                    st3(m.rA[j], !on ? V3{0.f, 0.f, 0.f} : plane ? hull_local_vertex(oa, raw.vidx[j]) : qrot(qai, raw.pA[j] - pa));
                    m.offB[j] = on ? dot(raw.pB[j], raw.n) : 0.f; m.lam[j] = 0.f;
                }
                if (toLds) { __builtin_memcpy(mw, &m, sizeof(m)); mkind = 2; mkk = kk; mg = g; mpair = pair; }
                else { ((ManS *)S.wsSC + (size_t)w * kAllSC)[kk] = m; wroteGlobal = true; }
                atomicOr(&R.scAcc[g], 1u << kk);
            }
        }
    }
    // every lane is done with its clipping scratch: the slots may overwrite it
    wave_sync(); __builtin_amdgcn_wave_barrier();
    HS_FTICK(4)
    if (mkind != 0) {
        int4 *dst = reinterpret_cast<int4 *>(R.u.sat.clip + lane * kManWords);
#pragma unroll
        for (int k = 0; k < (int)(sizeof(ManS) / 16); ++k) dst[k] = int4{mw[4 * k], mw[4 * k + 1], mw[4 * k + 2], mw[4 * k + 3]};
        if (mkind == 1) {
#pragma unroll
            for (int k = (int)(sizeof(ManS) / 16); k < (int)(sizeof(ManDD) / 16); ++k) dst[k] = int4{mw[4 * k], mw[4 * k + 1], mw[4 * k + 2], mw[4 * k + 3]};
            R.ddPair[mkk][mg] = (unsigned short)((mpair & 0x7ff) | (lane << 11));
        } else {
            R.scPair[mkk][mg] = (unsigned short)((mpair & 0x7ff) | (lane << 11));
        }
    }
    wave_sync();
    HS_FTICK(5)
    return __ballot(wroteGlobal) != 0ull;
}
// Returns whether any manifold of the substep lies in the global workspace (wave-uniform): only then do the solver
// phases have to wait for global memory at all.
template <bool SPILL, class OR>
HSD bool phase_sat(const SimState &S, OR &R, ItemCounts ic HS_TICK_PARAMS) {
    const int wedge0 = (ic.nbox + 31) / 32 * 32;
    const int lane = hs_lane() & 63;
    const bool hi = (lane & 1) != 0;                  // the second lane of a pair (box rounds: lanes 2k, 2k + 1)
    int npend = 0;
    bool usedGlobal = false;
    int nhit_total = 0, nflush = 0;      // (counters of the HS_PHASE_TIMING build)
#ifdef HS_SAT_COUNTERS
    long long tflush = 0, twedge = 0; const long long tsat0 = wall_clock64();
#define HS_SAT_T(x) x
#else
#define HS_SAT_T(x)
#endif
    // rounds over the box-only items, 2 lanes per pair and 32 pairs per round, then over the items with a wedge, 16 lanes
    // per pair and 4 pairs per round (sat_axes_wide); `lead`: the lane of a pair that files its result
    const int boxRounds = (ic.nbox + OR::kSatPairs - 1) / OR::kSatPairs, wedgeRounds = (ic.nwedge + 3) / 4;
    // (pairs beyond the LDS capacities, if any world of the octet has them: tested and turned into manifolds first, while
    // the clip buffers are free)
    if (SPILL && __builtin_expect(ic.anySpill, 0)) { spill_sat(spill_ctx(S), &R); usedGlobal = true; }
    // The axis-search rounds and the contact rounds alternate as two plain loops — as many axis-search rounds as the pending
    // list (OR::kClip entries) is sure to hold, then one contact round — instead of a contact round nested inside the
    // axis-search loop: the two bodies never share registers.  (A round files at most as many pairs as it has items.)
    const int totalRounds = boxRounds + wedgeRounds;
    int round = 0;
    while (round < totalRounds) {
    for (; round < totalRounds; ++round) {
        HS_SAT_T(const long long tr0_ = wall_clock64();)
        const bool wide = round >= boxRounds;
        {
            const int itemsHere = wide ? min(4, wedge0 + ic.nwedge - (wedge0 + (round - boxRounds) * 4)) : min(OR::kSatPairs, ic.nbox - round * OR::kSatPairs);
            if (npend + itemsHere > OR::kClip) break;            // (the pending list goes through a contact round first)
        }
        const int base = wide ? wedge0 + (round - boxRounds) * 4 : round * OR::kSatPairs;
        const int it = wide ? base + (lane >> 4) : base + (lane >> 1);
        const bool lead = wide ? (lane & 15) == 0 : !hi;
        AxisResult res = {0, {0.f, 0.f, 0.f}};
        int item = 0;
        if (wide ? it < wedge0 + ic.nwedge : (it < ic.nbox && (lane >> 1) < OR::kSatPairs)) {
            item = R.u.sat.items[it];
            const int g = item >> 6, idx = item & 63;
            const int w = S.wbeg + g;
            const bool isdd = idx < 32;
            const int kk = idx & 31;
            const int pair = isdd ? R.ddPair[kk][g] : R.scPair[kk][g];
            const int a = pair_a(pair), bsel = pair_b(pair);
            if (!isdd && bsel >= kMaxWalls) {
                // extra planes (debug levels only): no axis search — the contact round tests the hull against the plane
                if (lead) res.code = 3;
            } else if (wide) {
                res = sat_axes_wide(sat_hull_a(R, g, a), sat_hull_b(S, R, g, w, isdd, bsel), lane & 15);
            } else {
                res = sat_axes(sat_hull_a(R, g, a), sat_hull_b(S, R, g, w, isdd, bsel), hi);
            }
        }
        HS_SAT_T(if (wide) twedge += wall_clock64() - tr0_;)
#ifdef HS_FINE_TIMING
        if (wide) HS_TICK(2) else HS_TICK(1)
#endif
        // the colliding pairs of this round join the pending list (their lead lanes file the results)
        const bool hit = lead && res.code != 0;
        const unsigned long long m = __ballot(hit);
        const int nhit = __popcll(m);
        nhit_total += nhit;
        if (hit) {
            const int pos = npend + __popcll(m & ((1ull << lane) - 1ull));
            R.u.sat.pend[0][pos] = item | (res.code << 16);         // (item < 2^9, axis code < 2^12)
            R.u.sat.pend[1][pos] = __float_as_int(res.ax.x); R.u.sat.pend[2][pos] = __float_as_int(res.ax.y); R.u.sat.pend[3][pos] = __float_as_int(res.ax.z);
        }
        npend += nhit;
        HS_FTICK(3)
    }
    wave_sync();
    if (npend > 0) { HS_SAT_T(const long long t0_ = wall_clock64();) usedGlobal |= sat_flush(S, R, npend, round >= totalRounds HS_TICK_ARGS); HS_SAT_T(tflush += wall_clock64() - t0_;) ++nflush; npend = 0; }
    }
    if (usedGlobal) mem_sync();          // the manifolds in global memory are complete for the lanes that solve them
#ifdef HS_SAT_COUNTERS
    if (lane == 0) {      // work counters of the convex tests (tools/phase_timing.py; their atomics disturb the phase times)
        unsigned long long *c = (unsigned long long *)S.phaseTicks + phase_ticks_obs_base(S.N) + 16 * 1024;
        atomicAdd(&c[0], 1ull); atomicAdd(&c[1], (unsigned long long)ic.nbox); atomicAdd(&c[2], (unsigned long long)ic.nwedge);
        atomicAdd(&c[3], (unsigned long long)(boxRounds + wedgeRounds)); atomicAdd(&c[4], (unsigned long long)nhit_total);
        atomicAdd(&c[5], (unsigned long long)nflush);
        atomicAdd(&c[6], (unsigned long long)tflush); atomicAdd(&c[7], (unsigned long long)(wall_clock64() - tsat0)); atomicAdd(&c[8], (unsigned long long)twedge);
    }
#endif
    return usedGlobal;
}

// Fixed grab joint on two loaded bodies (sim.cpp:343-356): angular alignment, then anchor coincidence.
HSD void solve_grab_joint_bodies(BodyS &A, BodyS &B, V3 r2, Q attach2, float sep, V3 r1, Q attach1) {
    {
        Q qa = qmul(A.rot, attach1), qb = qmul(B.rot, attach2);
        Q dq = qmul(qa, qinv(qb));
        V3 dphi = {2.f * dq.x, 2.f * dq.y, 2.f * dq.z};
        if (dq.w < 0.f) dphi = -dphi;
        float th2 = len2(dphi);
        if (th2 > 1e-12f) {
            float th = sqrtf(th2);
            V3 ax = dphi * (1.f / th);
            float wA = dot(ax, sym_mul(A.Iw, ax));
            float wB = dot(ax, sym_mul(B.Iw, ax));
            float ws = wA + wB;
            if (ws > 0.f) {
                V3 p = ax * (th / ws);
                A.rot = quat_add_rotation(A.rot, -apply_inv_inertia(A, p));
                B.rot = quat_add_rotation(B.rot, apply_inv_inertia(B, p));
            }
        }
    }
    {
        V3 anchorA = r1 + V3{0.f, sep, 0.f};
        V3 rAw = qrot(A.rot, anchorA), rBw = qrot(B.rot, r2);
        V3 dx = (A.pos + rAw) - (B.pos + rBw);
        float c2 = len2(dx);
        if (c2 > 1e-12f) {
            float c = sqrtf(c2);
            V3 n = dx * (1.f / c);
            float ws = gen_inv_mass(A, rAw, n) + gen_inv_mass(B, rBw, n);
            if (ws > 0.f) apply_pos_impulse<true>(A, rAw, B, rBw, n * (c / ws));
        }
    }
}

// ------------------------------------------------------------------------------------------
// Body-body manifolds (and grab joints), 8 lanes per world, all 8 worlds of the octet at once.  The oracle solves a
// world's manifolds one after the other in (i<j) pair order; manifolds that share no body commute exactly, so the
// h-th pair of lanes takes the h-th accepted manifold of the sorted order and runs as soon as no EARLIER manifold that
// is still pending touches one of its bodies.  Disjoint pairs are solved in one round instead of one after the other;
// the result is bit-identical to the sequential order.
//
// A manifold is solved by TWO lanes, one per body: everything a contact point does to body A is independent of what
// it does to body B between the few scalars and points the two sides share (penetration, generalised inverse masses,
// tangential drift), which the lanes exchange with a DPP swap.  Each lane evaluates exactly the expressions the
// one-lane form (solve_point_position / solve_point_velocity <true>) evaluates for its body; the shared quantities
// are sums and differences whose value does not depend on which side computes them (a + b = b + a, -(a - b) = b - a,
// the sign of a dot product follows the sign of its vector argument exactly), so the results are the same bits.
HSD float swap1(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, false)); }   // lanes 2k <-> 2k+1
HSD V3 swap1(V3 v) { return {swap1(v.x), swap1(v.y), swap1(v.z)}; }

// me: this lane's body; rl: its contact point (body frame); isA: this lane holds body A of the manifold.
HSD float pair_point_position(BodyS &me, bool isA, V3 n, V3 rl, float muS) {
    V3 rw = qrot(me.rot, rl);
    V3 pm = me.pos + rw;
    const float dm = dot(pm - swap1(pm), n);
    float d = isA ? dm : -dm;                                     // dot(pA - pB, n)
    if (!(d > 0.f)) return 0.f;
    const V3 pprev = me.ppos + qrot(me.prot, rl);
    const float dpm = dot(pprev - swap1(pprev), n);
    const float dprev = isA ? dpm : -dpm;
    const float excess = dprev - kMaxDepenVel * kSubstepH;
    if (excess > 0.f) d = d - excess;
    if (!(d > 0.f)) return 0.f;
    const float wm = gen_inv_mass(me, rw, n);
    const float wo = swap1(wm);                                   // (every lane swaps: never inside a branch on isA)
    const float wsum = isA ? wm + wo : wo + wm;                   // wA + wB
    if (!(wsum > 0.f)) return 0.f;
    const float lam = d / wsum;
    auto apply = [&](V3 p) {
        if (!has_mass(me)) return;
        const V3 dth = apply_inv_inertia(me, cross(rw, p));
        if (isA) { me.pos = nmadd(me.pos, p, me.invM); me.rot = quat_add_rotation(me.rot, -dth); }
        else { me.pos = madd(me.pos, p, me.invM); me.rot = quat_add_rotation(me.rot, dth); }
    };
    apply(n * lam);
    rw = qrot(me.rot, rl);
    pm = me.pos + rw;
    const V3 mv = pm - pprev;                                     // how far this side's point has moved in the substep
    const V3 ov = swap1(mv);
    const V3 dp = isA ? mv - ov : ov - mv;                        // (pA - pAprev) - (pB - pBprev)
    const V3 dpt = nmadd(dp, n, dot(dp, n));
    const float lt2 = len2(dpt);
    if (lt2 > 1e-12f) {
        const float wt = gen_inv_mass_sq(me, rw, dpt, lt2);
        const float wto = swap1(wt);
        const float wts = isA ? wt + wto : wto + wt;
        if (wts > 0.f) {
            const float lim = (muS * lam) * wts;
            if ((lt2 * lt2) * lt2 < lim * lim) apply(dpt * (lt2 / wts));
        }
    }
    return lam;
}
HSD void pair_point_velocity(BodyS &me, bool isA, V3 n, V3 rl, float lamN, float muD) {
    if (!(lamN > 0.f)) return;
    const V3 rw = qrot(me.rot, rl);
    const bool hm = me.invM + me.invI.x + me.invI.y + me.invI.z != 0.f;
    const V3 vm = cross_add(me.ang, rw, me.lin);
    const V3 vo = swap1(vm);
    const bool ho = swap1(hm ? 1.f : 0.f) != 0.f;
    const V3 vA = isA ? vm : vo, vB = isA ? vo : vm;
    const bool hA = isA ? hm : ho, hB = isA ? ho : hm;
    V3 v = {0.f, 0.f, 0.f};
    if (hA) v = vA;
    if (hB) v = v - vB;
    const float vn = dot(n, v);
    const V3 vt = nmadd(v, n, vn);
    const float vt2 = len2(vt);
    V3 dv = -(n * vn);
    if (vt2 > 1e-18f) {
        const float vtl = sqrtf(vt2);
        const float mag = fminf((muD * lamN) * kInvSubstepH, vtl);
        dv = nmadd(dv, vt, mag / vtl);
    }
    const float dv2 = len2(dv);
    if (!(dv2 > 1e-18f)) return;
    const float wm = gen_inv_mass_sq(me, rw, dv, dv2);
    const float wo = swap1(wm);
    const float ws = isA ? wm + wo : wo + wm;
    if (!(ws > 0.f)) return;
    const V3 p = dv * (dv2 / ws);
    const V3 da = apply_inv_inertia(me, cross(rw, p));
    if (isA) { me.lin = madd(me.lin, p, me.invM); me.ang = me.ang + da; }
    else { me.lin = nmadd(me.lin, p, me.invM); me.ang = me.ang - da; }
}

template <bool POS, bool SPILL, class OR>
HSD void phase_dd(const SimState &S, OR &R, bool anySpill) {
    constexpr int GL = OR::kG, PAIRS = GL / 2;                    // lanes per world (8 or 16); manifolds of a world in flight at once
    const int L = hs_lane(), g = L / GL, q = L % GL;
    const int h = q >> 1;                                         // this lane's pair within the world's lanes
    const bool isA = (q & 1) == 0;
    const int gbit0 = g * GL;                                     // first lane of this group in the wave
    const int w = S.wbeg + g;                                     // the world's slot in the tiled columns
    const int ndd = R.ndd[g];
    const bool grab = R.hasGrab[g] != 0;
    if (!(SPILL && anySpill) && __ballot(R.ddAcc[g] != 0u || (POS && grab)) == 0ull) return;        // nothing to do in the whole octet
    if (POS && grab && q == 0) {
        const int teams = S.teams[R.wid[g]];
        for (int a = 0; a < kMaxAgents; ++a) {
            if (!team_agent_active(teams, a)) continue;
            const int other = S.grabOther(a, w);
            if (other < 0) continue;
            BodyS A, B;
            rbody_load(R, g, kAgentSlot0 + a, A); rbody_load(R, g, other, B);
            float gd[kGrabWords];
#pragma unroll
            for (int c = 0; c < kGrabWords; ++c) gd[c] = S.grabData(c * kMaxAgents + a, w);
            solve_grab_joint_bodies(A, B, {gd[0], gd[1], gd[2]}, {gd[3], gd[4], gd[5], gd[6]}, gd[7],
                                    {gd[8], gd[9], gd[10]}, {gd[11], gd[12], gd[13], gd[14]});
            rbody_store_pose(R, g, kAgentSlot0 + a, A); rbody_store_pose(R, g, other, B);
        }
    }
    ManDD *const wsDD = (ManDD *)S.wsDD + (size_t)w * kAllDD;
    const unsigned acc = R.ddAcc[g];
    const int nacc = __popc(acc);
#ifdef HS_LOAD_STUDY
    if (POS && q == 0 && R.wid[g] >= 0) S.phaseTicks[phase_ticks_study_base(S.N) + (size_t)R.wid[g] * kStudyWords + 2] += nacc;
#endif
    if (POS) {
        // The solve order, once per substep (the velocity pass reuses it): kMaxDDCand = 16 candidates, lane q ranks the
        // keys of the accepted ones among q and q+8 (keys are unique: distinct pairs) and enters them at their rank.
        if (q < 2) R.ddOrd[q][g] = 0u;
        int key0 = 0x7fffffff, key1 = 0x7fffffff;
        if (q < ndd && ((acc >> q) & 1u)) { const int p = R.ddPair[q][g]; key0 = (pair_a(p) << 8) | pair_b(p); }
        if (q + GL < ndd && ((acc >> (q + GL)) & 1u)) { const int p = R.ddPair[q + GL][g]; key1 = (pair_a(p) << 8) | pair_b(p); }
        int rank0 = 0, rank1 = 0;
#pragma unroll
        for (int p = 0; p < GL; ++p) {
            const int k0 = __shfl(key0, gbit0 + p), k1 = __shfl(key1, gbit0 + p);
            rank0 += (k0 < key0) + (k1 < key0); rank1 += (k0 < key1) + (k1 < key1);
        }
        wave_sync();
        if (key0 != 0x7fffffff) atomicOr(&R.ddOrd[rank0 >> 3][g], (unsigned)q << ((rank0 & 7) * 4));
        if (key1 != 0x7fffffff) atomicOr(&R.ddOrd[rank1 >> 3][g], (unsigned)(q + GL) << ((rank1 & 7) * 4));
    }
    wave_sync();
    for (int base = 0; base < kMaxDDCand; base += PAIRS) {
        if (__ballot(base < nacc) == 0ull) break;
        // the h-th pair of lanes takes the manifold of rank base + h
        const int rk = base + h;
        const int mine = rk < nacc ? (int)((R.ddOrd[rk >> 3][g] >> ((rk & 7) * 4)) & 15u) : -1;
        int ma = -1, mb = -1, loc = kLocGlobal;
        ManDD m;
        if (mine >= 0) {
            const int pr = R.ddPair[mine][g];
            ma = pair_a(pr); mb = pair_b(pr); loc = pair_loc(pr);
            if (loc != kLocGlobal) man_lds_load(R.u.sat.clip, loc, m); else m = wsDD[mine];
        }
        // the earlier pairs of this batch that touch one of its bodies
        unsigned dep = 0u;
#pragma unroll
        for (int p = 0; p < PAIRS - 1; ++p) {
            const int pa = __shfl(ma, gbit0 + 2 * p), pb = __shfl(mb, gbit0 + 2 * p);
            if (p < h && pa >= 0 && (pa == ma || pa == mb || pb == ma || pb == mb)) dep |= 1u << p;
        }
        bool pending = mine >= 0;
        BodyS me;
        const int myBody = isA ? ma : mb;
        const V3 n = ld3(m.n);
        while (true) {
            const unsigned long long pend_mask = __ballot(pending && isA);
            if (pend_mask == 0ull) break;
#ifdef HS_LOAD_STUDY
            if (POS) {
                long long *const st = S.phaseTicks + phase_ticks_study_base(S.N);
                if (q == 0 && R.wid[g] >= 0 && ((pend_mask >> gbit0) & ((1ull << GL) - 1ull)) != 0ull) st[(size_t)R.wid[g] * kStudyWords + 3] += 1;
                if (L == 0 && R.wid[0] >= 0) st[(size_t)R.wid[0] * kStudyWords + 5] += 1;
            }
#endif
            // (bit 2p of the world's 8 bits: pair p is pending)
            const unsigned wp = (unsigned)(pend_mask >> gbit0);
            unsigned pendPairs = 0u;
#pragma unroll
            for (int p = 0; p < PAIRS; ++p) pendPairs |= ((wp >> (2 * p)) & 1u) << p;
#ifdef HS_DD_ONE_LANE
            if (pending && (dep & pendPairs) == 0u) {
                if (isA) {
                    BodyS Ab, Bb;
                    rbody_load(R, g, m.a, Ab); rbody_load(R, g, m.b, Bb);
                    if (POS) {
                        for (int j = 0; j < 4; ++j)
                            if (j < m.np) {
                                const float lam = m.lam[j] + solve_point_position<true>(Ab, Bb, n, ld3(m.rA[j]), ld3(m.rB[j]), 0.f, m.muS);
                                if (loc != kLocGlobal) man_lds_set_lam<ManDD>(R.u.sat.clip, loc, j, lam); else wsDD[mine].lam[j] = lam;
                            }
                        rbody_store_pose(R, g, m.a, Ab); rbody_store_pose(R, g, m.b, Bb);
                    } else {
                        for (int j = 0; j < 4; ++j)
                            if (j < m.np) solve_point_velocity<true>(Ab, Bb, n, ld3(m.rA[j]), ld3(m.rB[j]), m.lam[j], m.muD);
                        rbody_store_vel(R, g, m.a, Ab); rbody_store_vel(R, g, m.b, Bb);
                    }
                }
                pending = false;
            }
#else
            if (pending && (dep & pendPairs) == 0u) {
                rbody_load(R, g, myBody, me);
                if (POS) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j < m.np) {
                            const float lam = pair_point_position(me, isA, n, isA ? ld3(m.rA[j]) : ld3(m.rB[j]), m.muS);
                            if (isA) { if (loc != kLocGlobal) man_lds_set_lam<ManDD>(R.u.sat.clip, loc, j, m.lam[j] + lam); else wsDD[mine].lam[j] = m.lam[j] + lam; }
                        }
                    rbody_store_pose(R, g, myBody, me);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (j < m.np) pair_point_velocity(me, isA, n, isA ? ld3(m.rA[j]) : ld3(m.rB[j]), m.lam[j], m.muD);
                    rbody_store_vel(R, g, myBody, me);
                }
                pending = false;
            }
#endif
            wave_sync();                  // later rounds must see the poses / velocities just written
        }
    }
    wave_sync();
    // (the world's SPILLED body-body pairs come after every pair above in pair order)
    if (SPILL && __builtin_expect(anySpill, 0)) { spill_dd<POS>(spill_ctx(S), &R); wave_sync(); }
}

// ------------------------------------------------------------------------------------------
// Static contacts, in the oracle's order per body: ground manifold, then the body's other static candidates (extra
// planes, walls by index), then the velocities of the substep from the pose change.  The ground manifold of every
// body is solved by the body's own lane (registers).  The few bodies that also touch a wall would make their whole
// round wait, so their wall manifolds are solved in a round of their own — one lane per such body, compacted over
// the octet (wallBodies) — between the ground pass and the velocity derivation.
struct WallLists { int nwb, nEarly; };     // listed bodies; how many of them belong to a round before the last one
template <int ROUNDS, class OR>
HSD WallLists list_wall_bodies(OR &R, int nbodies, int lastRound) {
    const int L = hs_lane();
    int n = 0, nEarly = 0;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (r == lastRound) nEarly = n;
        const bool valid = r * 64 + L < nbodies;
        const int t = valid ? R.bodies[r * 64 + L] : 0;
        const int sci = valid ? R.scInfo[t >> 3][t & 7] : 0;
        const bool on = sci != 0 && ((R.scAcc[t & 7] >> (sci & 0xff)) & ((1u << (sci >> 8)) - 1u)) != 0u;
        const unsigned long long m = __ballot(on);
        if (on) { R.wallBodies[n + __popcll(m & ((1ull << L) - 1ull))] = (unsigned char)t; atomicOr(&R.wallSeen[t & 7], 1u << (t >> 3)); }
        n += __popcll(m);
    }
    wave_sync();
    return {n, nEarly};
}

template <class OR> HSD void ground_pos(OR &R, BodyReg &b, int slot, int g, int meta) {
    if (meta_resp(meta) != RESP_DYNAMIC || b.np == 0) return;
    const int obj = meta_obj(meta);
    BodyS me;
    rbody_load(R, g, slot, me);
    const V3 gn = -V3{R.plane0[0][g], R.plane0[1][g], R.plane0[2][g]};
    const float gmuS = 0.5f * (obj_mu_s(obj) + obj_mu_s(OBJ_PLANE));
    // agents are yaw-only bodies (obj_inv_inertia): their floor contacts share the manifold's normal multiplier
    const bool yaw = slot >= kAgentSlot0;
    float dj[4] = {0.f, 0.f, 0.f, 0.f}, share = 0.f;
    if (yaw) share = yaw_ground_prepass(me, gn, b.np, hull_local_vertex(obj, b.vidx & 7), hull_local_vertex(obj, (b.vidx >> 3) & 7),
                                        hull_local_vertex(obj, (b.vidx >> 6) & 7), hull_local_vertex(obj, (b.vidx >> 9) & 7), kGroundOff, dj);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < b.np) b.lam[j] += solve_point_position_ground(me, gn, hull_local_vertex(obj, (b.vidx >> (3 * j)) & 7), kGroundOff, gmuS,
                                                              yaw, dj[j], share);
    rbody_store_pose(R, g, slot, me);
}
template <class OR> HSD void ground_vel(OR &R, const BodyReg &b, int slot, int g, int meta) {
    if (meta_resp(meta) != RESP_DYNAMIC || b.np == 0) return;
    const int obj = meta_obj(meta);
    BodyS me, none;
    rbody_load(R, g, slot, me);
    const V3 gn = -V3{R.plane0[0][g], R.plane0[1][g], R.plane0[2][g]};
    const float gmuD = 0.5f * (obj_mu_d(obj) + obj_mu_d(OBJ_PLANE));
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < b.np) solve_point_velocity<false>(me, none, gn, hull_local_vertex(obj, (b.vidx >> (3 * j)) & 7), V3{0.f, 0.f, 0.f},
                                                  b.lam[j], gmuD);
    rbody_store_vel(R, g, slot, me);
}
// The wall / extra-plane manifolds of the listed bodies, one lane per body, candidates in solve order.
template <bool POS, class OR>
HSD void wall_round(const SimState &S, OR &R, int first, int nwb) {
    for (int i = first + hs_lane(); i < nwb; i += 64) {
        const int t = R.wallBodies[i];
        const int slot = t >> 3, g = t & 7;
        const int sci = R.scInfo[slot][g];
        const int bsc = sci & 0xff, asc = sci >> 8;
        const unsigned acc = R.scAcc[g];
        ManS *const wsSC = (ManS *)S.wsSC + (size_t)(S.wbeg + g) * kAllSC;
        BodyS me, none;
        rbody_load(R, g, slot, me);
        // the body's ACCEPTED candidates in candidate order, a lane at its own pace: a trip of the loop is a whole manifold
        // solve for the wave, so the trips are the largest number of manifolds any listed body has (mostly one), not
        // the span of candidate positions over the lanes
        unsigned todo = (acc >> bsc) & ((1u << asc) - 1u);
#pragma unroll 1
        while (todo != 0u) {
            const int k = bsc + __ffs((int)todo) - 1;
            todo &= todo - 1u;
            const int loc = pair_loc(R.scPair[k][g]);
            ManS m;
            if (loc != kLocGlobal) man_lds_load(R.u.sat.clip, loc, m); else m = wsSC[k];
            body_refresh_inertia(me);
            const V3 n = ld3(m.n);
            if (POS) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < m.np) {
                        const float lam = m.lam[j] + solve_point_position<false>(me, none, n, ld3(m.rA[j]), V3{0.f, 0.f, 0.f}, m.offB[j], m.muS);
                        if (loc != kLocGlobal) man_lds_set_lam<ManS>(R.u.sat.clip, loc, j, lam); else wsSC[k].lam[j] = lam;
                    }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < m.np) solve_point_velocity<false>(me, none, n, ld3(m.rA[j]), V3{0.f, 0.f, 0.f}, m.lam[j], m.muD);
            }
        }
        if (POS) rbody_store_pose(R, g, slot, me); else rbody_store_vel(R, g, slot, me);
    }
}
// The LAST round of bodies is rarely full (96 bodies on average: 64 + 32), and a round costs the wave the same whatever
// the number of busy lanes.  Its idle lanes therefore take the wall / extra-plane manifolds of listed bodies that belong
// to an EARLIER round (their ground manifold is done by then), `nMerged` of them: a lane of this round walks a list of
// static manifolds — the ground manifold from its registers, or a listed body's accepted candidates from LDS — through
// one and the same solve code.  What does not fit (bodies of the last round itself, more listed bodies than idle lanes)
// is left to wall_round.  Per body the order stays ground, then its static candidates by index.
template <bool POS, class OR>
HSD void last_round(const SimState &S, OR &R, BodyReg &b, bool valid, int slot, int g, int meta, int nLast, int nMerged) {
    const int mi = (int)hs_lane() - nLast;
    const bool walls = mi >= 0 && mi < nMerged;
    unsigned todo = 0u; int bsc = 0;
    if (walls) {
        const int t = R.wallBodies[mi];
        slot = t >> 3; g = t & 7; meta = R.meta[slot][g];
        const int sci = R.scInfo[slot][g];
        bsc = sci & 0xff;
        todo = (R.scAcc[g] >> bsc) & ((1u << (sci >> 8)) - 1u);
    } else if (valid && meta_resp(meta) == RESP_DYNAMIC && b.np != 0) {
        todo = 1u;
    }
    if (todo == 0u) return;
    const int obj = meta_obj(meta);
    BodyS me, none;
    rbody_load(R, g, slot, me);
    ManS *const wsSC = (ManS *)S.wsSC + (size_t)(S.wbeg + g) * kAllSC;
#pragma unroll 1
    while (todo != 0u) {
        const int k = bsc + __ffs((int)todo) - 1;
        todo &= todo - 1u;
        // (no arrays indexed by the contact number here: in scratch memory they cost a store and a load per contact)
        V3 n; float mu; int np, loc = kLocGlobal;
        ManS m;
        if (walls) {
            loc = pair_loc(R.scPair[k][g]);
            if (loc != kLocGlobal) man_lds_load(R.u.sat.clip, loc, m); else m = wsSC[k];
            n = ld3(m.n); mu = POS ? m.muS : m.muD; np = m.np;
        } else {
            n = -V3{R.plane0[0][g], R.plane0[1][g], R.plane0[2][g]};
            mu = POS ? 0.5f * (obj_mu_s(obj) + obj_mu_s(OBJ_PLANE)) : 0.5f * (obj_mu_d(obj) + obj_mu_d(OBJ_PLANE));
            np = b.np;
        }
        body_refresh_inertia(me);
        // the ground manifold of an agent (a yaw-only body): manifold-level normal part first (hs_solver.h)
        const bool yaw = POS && !walls && slot >= kAgentSlot0;
        float dj[4] = {0.f, 0.f, 0.f, 0.f}, share = 0.f;
        if (yaw) share = yaw_ground_prepass(me, n, np, hull_local_vertex(obj, b.vidx & 7), hull_local_vertex(obj, (b.vidx >> 3) & 7),
                                            hull_local_vertex(obj, (b.vidx >> 6) & 7), hull_local_vertex(obj, (b.vidx >> 9) & 7), kGroundOff, dj);
#define HS_LAST_POINT(j)                                                                                                   \
        if ((j) < np) {                                                                                                    \
            const V3 rAj = walls ? ld3(m.rA[j]) : hull_local_vertex(obj, (b.vidx >> (3 * (j))) & 7);                       \
            const float lamj = walls ? m.lam[j] : b.lam[j];                                                                \
            if (POS) {                                                                                                     \
                const float nl = lamj + solve_point_position_ground(me, n, rAj, walls ? m.offB[j] : kGroundOff, mu, yaw, dj[j], share); \
                if (!walls) b.lam[j] = nl;                                                                                 \
                else if (loc != kLocGlobal) man_lds_set_lam<ManS>(R.u.sat.clip, loc, (j), nl);                             \
                else wsSC[k].lam[j] = nl;                                                                                  \
            } else {                                                                                                       \
                solve_point_velocity<false>(me, none, n, rAj, V3{0.f, 0.f, 0.f}, lamj, mu);                                \
            }                                                                                                              \
        }
        HS_LAST_POINT(0) HS_LAST_POINT(1) HS_LAST_POINT(2) HS_LAST_POINT(3)
#undef HS_LAST_POINT
    }
    if (POS) rbody_store_pose(R, g, slot, me); else rbody_store_vel(R, g, slot, me);
}
template <class OR> HSD void derive_body_velocity(OR &R, int slot, int g, int meta) {
    if (meta_resp(meta) != RESP_DYNAMIC) return;
    BodyS me;
    me.pos = rld3(R.pos, slot, g); me.rot = rld4(R.rot, slot, g);
    me.ppos = rld3(R.ppos, slot, g); me.prot = rld4(R.prot, slot, g);
    derive_velocity(me);
    rbody_store_vel(R, g, slot, me);
}

// ------------------------------------------------------------------------------------------
// actionSystem (sim.cpp:270-370) for one world, agents in interface order, run by ONE lane: lock / grab ray casts
// against the resident geometry, joint create / destroy.  Meta words change in LDS (copied back at the end of
// the launch); the joint table lives in global memory.
template <class OR> HSD void action_system(const SimState &S, OR &R, int g, int A_, int teams) {
    const int w = S.wbeg + g;
    const ResGeom<OR> geom = {R, S, g, w};
    for (int i = 0; i < A_; ++i) {
        const int fl = R.actGL[i][g];
        if (fl == 0) continue;
        const int type = team_agent_type(teams, i);
        const int slot = kAgentSlot0 + i;
        const V3 mpos = geom.g_pos(slot);
        const Q mrot = geom.g_rot(slot);
        if (fl & 2) {   // lock
            float t; V3 o = mpos + V3{0.f, 0.f, 0.5f};
            int hit = trace_ray(geom, o, qrot(mrot, {0.f, 1.f, 0.f}), 2.5f, &t);
            if (hit >= 0 && hit < kNumDSlots) {
                const int m = R.meta[hit][g];
                const int obj = meta_obj(m), resp = meta_resp(m), owner = meta_owner(m);
                if (resp == RESP_STATIC) {
                    if ((type == AGENT_SEEKER && owner == OWNER_SEEKER) || (type == AGENT_HIDER && owner == OWNER_HIDER))
                        R.meta[hit][g] = meta_pack(obj, RESP_DYNAMIC, OWNER_NONE);
                } else if (owner == OWNER_NONE) {
                    R.meta[hit][g] = meta_pack(obj, RESP_STATIC, type == AGENT_HIDER ? OWNER_HIDER : OWNER_SEEKER);
                }
            }
        }
        if (fl & 1) {   // grab
            if (S.grabOther(i, w) >= 0) {
                S.grabOther(i, w) = -1;
            } else {
                float t; V3 o = mpos + V3{0.f, 0.f, 0.5f};
                V3 dir = qrot(mrot, {0.f, 1.f, 0.f});
                int hit = trace_ray(geom, o, dir, 2.5f, &t);
                if (hit >= 0 && hit < kNumDSlots) {
                    const int m = R.meta[hit][g];
                    if (meta_owner(m) == OWNER_NONE && meta_resp(m) == RESP_DYNAMIC) {
                        V3 hit_pos = o + dir * t;
                        Q erot = geom.g_rot(hit);
                        V3 r2 = qrot(qinv(erot), hit_pos - geom.g_pos(hit));
                        Q at2 = qnormalize(qmul(qinv(erot), mrot));
                        S.grabOther(i, w) = hit;
                        const float gd[kGrabWords] = {r2.x, r2.y, r2.z, at2.w, at2.x, at2.y, at2.z, t - 1.25f,
                                                      0.f, 1.25f, 0.5f,            // r1 = 1.25 fwd + 0.5 up (sim.cpp:343-344)
                                                      1.f, 0.f, 0.f, 0.f};         // attachRot1 = identity
#pragma unroll
                        for (int c = 0; c < kGrabWords; ++c) S.grabData(c * kMaxAgents + i, w) = gd[c];
                    }
                }
            }
        }
    }
}

// Before the substeps: movementSystem | instantMovementSystem (sim.cpp:202-254) and actionSystem (:270-370).
// A lane per (agent, world) maps the action row to a force; a lane per world then runs the action system for the
// worlds in which an agent locks or grabs (scripts/benchmark.py never does, scripts/jax_train.py does all the time).
template <class OR> HSD void phase_pre(const SimState &S, OR &R) {
    constexpr int T = OR::kT;
    const int A_ = S.A, L = hs_lane();
    const bool instant = (S.flags & FLAG_ZERO_AGENT_VELOCITY) == FLAG_ZERO_AGENT_VELOCITY;
    bool need_action = false;
    if (L < kMaxAgents * T) {
        const int agent = L / T, g = L % T;
        const int w = R.wid[g], p = S.wbeg + g;          // world id (exports, per-world scalars) / slot (columns)
        int fl = 0;
        if (agent < A_ && w >= 0) {
            const int teams = S.teams[w], step = S.curEpisodeStep[w];
            const bool active = team_agent_active(teams, agent) != 0;
            const int type = team_agent_type(teams, agent);
            if (active && !(type == AGENT_SEEKER && step < kNumPrepSteps - 1)) {
                int32_t *act_row = S.xAction + (w * A_ + agent) * 5;
                const int ax = act_row[0], ay = act_row[1], ar = act_row[2], ag = act_row[3], al = act_row[4];
                float fx, fy, tz;
                if (instant) { fx = 400.f * (float)(ax - 2); fy = 400.f * (float)(ay - 2); tz = 120.f * (float)(ar - 2); }
                else { fx = 12.f * (float)(ax - 5); fy = 12.f * (float)(ay - 5); tz = 3.f * (float)(ar - 5); }
                V3 f = qrot(rld4(R.rot, kAgentSlot0 + agent, g), {fx, fy, 0.f});
                S.aforce(0 * kMaxAgents + agent, p) = f.x; S.aforce(1 * kMaxAgents + agent, p) = f.y;
                S.aforce(2 * kMaxAgents + agent, p) = f.z; S.aforce(3 * kMaxAgents + agent, p) = tz;
                fl = (ag == 1 ? 1 : 0) | (al == 1 ? 2 : 0);
                act_row[0] = 2; act_row[1] = 2; act_row[2] = 2; act_row[3] = 0; act_row[4] = 0;   // sim.cpp:365-369
            }
        }
        R.actGL[agent][g] = (unsigned char)fl;
        need_action = fl != 0;
    }
    const bool any = __ballot(need_action) != 0ull;
    wave_sync();                          // actGL and the forces (global memory) are in place
    if (any && L < T && R.wid[L] >= 0) {   // one lane per world: the ray casts of a world are sequential
        bool want = false;
        for (int a = 0; a < A_; ++a) want |= R.actGL[a][L] != 0;
        if (want) action_system(S, R, L, A_, S.teams[R.wid[L]]);
    }
    // worlds with a grab joint take part in the body-body phase of every substep
    if (L < T) {
        bool grab = false;
        if (R.wid[L] >= 0) for (int a = 0; a < kMaxAgents; ++a) grab |= S.grabOther(a, S.wbeg + L) >= 0;
        R.hasGrab[L] = grab ? 1 : 0;
    }
    wave_sync();
}

// After the substeps: agentZeroVelSystem (sim.cpp:258-268), rewardsVisSystem (:763-804),
// outputRewardsDonesSystem (:806-841), updateEpisodeResultsSystem (:843-893).  8 lanes per world.
template <class OR> HSD void phase_post(const SimState &S, OR &R) {
    constexpr int G = OR::kG;
    const int L = hs_lane(), g = L / G, l = L % G;
    const int w = R.wid[g], p = S.wbeg + g;          // world id (exports, per-world scalars) / slot (columns)
    const int A_ = S.A;
    const bool wok = w >= 0;
    const bool instant = (S.flags & FLAG_ZERO_AGENT_VELOCITY) == FLAG_ZERO_AGENT_VELOCITY;
    int teams = 0, step = 0, counts = 0;
    if (wok) {
        teams = S.teams[w]; step = S.curEpisodeStep[w]; counts = S.counts[w];
        if (l == 0) R.seen[g] = 0;
        if (instant && l < kMaxAgents && R.meta[kAgentSlot0 + l][g] != 0) {
            const int slot = kAgentSlot0 + l;
            R.u.vel.lin[0][slot][g] = 0.f; R.u.vel.lin[1][slot][g] = 0.f;
            R.u.vel.lin[2][slot][g] = fminf(R.u.vel.lin[2][slot][g], 0.f);
            rst3(R.u.vel.ang, slot, g, V3{0.f, 0.f, 0.f});
        }
    }
    wave_sync();
    // The seen flag feeds the reward (from episode step 95 on) and the episode result (from 96 on); the reset overwrites
    // hiderTeamReward every step, so during the preparation phase the rays would change nothing anyone can read.
    for (int pr = l; wok && step >= kNumPrepSteps - 1 && pr < 9; pr += G) {      // (seeker, hider) pairs
        const int si = pr / 3, hi_ = pr % 3;
        if (si < cnt_seekers(counts) && hi_ < cnt_hiders(counts)) {
            const ResGeom<OR> geom = {R, S, g, p};
            const int ss = kAgentSlot0 + team_seeker(teams, si), hs_ = kAgentSlot0 + team_hider(teams, hi_);
            const V3 spos = geom.g_pos(ss);
            const V3 fwd = qrot(geom.g_rot(ss), {0.f, 1.f, 0.f});
            V3 to = geom.g_pos(hs_) - spos;
            float c = dot(normalize(to), fwd);
            if (!(c < kCosFovHalf)) {
                float t;
                if (trace_ray(geom, spos, to, 1.f, &t) == hs_) R.seen[g] = 1;   // every writer stores the same value
            }
        }
    }
    wave_sync();
    if (!wok) return;
    float hider_reward = S.hiderTeamReward[w];
    if (R.seen[g]) hider_reward = -1.f;
    if (l < A_ && team_agent_active(teams, l)) {
        const int agent = l, slot = kAgentSlot0 + agent, row = w * A_ + agent;
        if (step == 0) S.xDone[row] = 0;
        if (step < kNumPrepSteps - 1) {
            S.xReward[row] = 0.f;
        } else {
            if (step == kEpisodeLen - 1) S.xDone[row] = 1;
            float r = hider_reward;
            if (team_agent_type(teams, agent) == AGENT_SEEKER) r *= -1.f;
            if (fabsf(R.pos[0][slot][g]) >= 18.f || fabsf(R.pos[1][slot][g]) >= 18.f) r -= 10.f;
            S.xReward[row] = r;
        }
    }
    if (l == 0) {
        float *res = S.xEpisodeResult + w * 2;
        int s0 = S.runningScores(0, p), s1 = S.runningScores(1, p);
        if (step == 0) { res[0] = 0.f; res[1] = 0.f; s0 = 0; s1 = 0; }
        if (step >= kNumPrepSteps) {
            const bool hidden = hider_reward == 1.f;
            const bool sf = cnt_seekers_first(counts) != 0;
            const int win = hidden ? (sf ? 1 : 0) : (sf ? 0 : 1);
            if (win == 0) s0 += 1; else s1 += 1;
        }
        if (step == kEpisodeLen - 1) {
            if (s0 > s1) { res[0] = 1.f; res[1] = 0.f; }
            else if (s0 < s1) { res[0] = 0.f; res[1] = 1.f; }
            else { res[0] = 0.5f; res[1] = 0.5f; }
        }
        S.runningScores(0, p) = s0; S.runningScores(1, p) = s1;
        S.hiderTeamReward[w] = hider_reward;
    }
}

// ------------------------------------------------------------------------------------------
// The static-contact passes of a substep: ground_pos / ground_vel for the rounds before LASTR, then last_round for round
// LASTR (the last round that holds bodies) with the wall manifolds of earlier rounds' bodies in its idle lanes.
template <int ROUNDS, int LASTR, bool POS, class OR>
HSD void static_passes(const SimState &S, OR &R, BodyReg (&br)[ROUNDS], int nbodies, int nLast, int nMerged) {
    const int L = hs_lane();
#pragma unroll
    for (int r = 0; r < LASTR; ++r) {
        const bool valid = r * 64 + L < nbodies; const int t_ = valid ? R.bodies[r * 64 + L] : 0;
        const int slot = t_ >> 3, g = t_ & 7; const int meta = valid ? R.meta[slot][g] : 0;
        if (valid) { if (POS) ground_pos(R, br[r], slot, g, meta); else ground_vel(R, br[r], slot, g, meta); }
    }
    wave_sync();                          // (the poses / velocities of the earlier rounds, for the merged lanes)
    {
        const bool valid = LASTR * 64 + L < nbodies; const int t_ = valid ? R.bodies[LASTR * 64 + L] : 0;
        const int slot = t_ >> 3, g = t_ & 7; const int meta = valid ? R.meta[slot][g] : 0;
        last_round<POS>(S, R, br[LASTR], valid, slot, g, meta, nLast, nMerged);
    }
    wave_sync();
}

// What follows the broadphase in a substep: convex tests -> body-body position solve -> static position passes ->
// velocities from the pose change -> body-body and static velocity passes (-> integration for the next substep).
#define HS_BODY(r) const bool valid = (r) * 64 + L < nbodies; const int t_ = valid ? R.bodies[(r) * 64 + L] : 0; \
                   const int slot = t_ >> 3, g = t_ & 7; const int meta = valid ? R.meta[slot][g] : 0;
template <int ROUNDS, bool SPILL, class OR>
HSD void substep_rest(const SimState &S, OR &R, BodyReg (&br)[ROUNDS], int nbodies, int NS, ItemCounts ic, bool integrateNext,
                      const float *aforce HS_TICK_PARAMS) {
    const int L = hs_lane();
    const bool manGlobal = phase_sat<SPILL>(S, R, ic HS_TICK_ARGS);
    HS_CTICK(3, 3)
    phase_dd<true, SPILL>(S, R, ic.anySpill);
    HS_CTICK(4, 6)
    // (with 6 agents a third round exists for up to 136 bodies, but an octet rarely holds more than 128: then round 1 is
    // the last one that holds bodies, and the passes are two, not three)
    // (likewise the second round of a 4-world wave with 6 agents: 68 bodies at most, rarely more than 64)
    constexpr int kShortLastR = ROUNDS >= 2 ? ROUNDS - 2 : 0;
    const bool shortLast = (ROUNDS == 3 || (ROUNDS == 2 && OR::kT == 4)) && nbodies <= 64 * (ROUNDS - 1);
    const int lastRound = shortLast ? kShortLastR : ROUNDS - 1;
    const WallLists wl = list_wall_bodies<ROUNDS>(R, nbodies, lastRound);
    const int nLast = max(nbodies - 64 * lastRound, 0);            // bodies of the last round
    const int nMerged = min(wl.nEarly, 64 - nLast);                 // listed bodies its idle lanes take
    if (shortLast) static_passes<ROUNDS, kShortLastR, true>(S, R, br, nbodies, nLast, nMerged);
    else static_passes<ROUNDS, ROUNDS - 1, true>(S, R, br, nbodies, nLast, nMerged);
    if (wl.nwb > nMerged) { wall_round<true>(S, R, nMerged, wl.nwb); wave_sync(); }
    if (SPILL && __builtin_expect(ic.anySpill, 0)) { spill_static<true>(spill_ctx(S), &R, NS); wave_sync(); }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) { HS_BODY(r) if (valid) derive_body_velocity(R, slot, g, meta); }
    if (manGlobal) mem_sync(); else wave_sync();   // (the multipliers of manifolds in the global workspace, for the velocity pass)
    HS_CTICK(5, 7)
    phase_dd<false, SPILL>(S, R, ic.anySpill);
    HS_CTICK(6, 8)
    if (shortLast) static_passes<ROUNDS, kShortLastR, false>(S, R, br, nbodies, nLast, nMerged);
    else static_passes<ROUNDS, ROUNDS - 1, false>(S, R, br, nbodies, nLast, nMerged);
    if (wl.nwb > nMerged) { wall_round<false>(S, R, nMerged, wl.nwb); wave_sync(); }
    if (SPILL && __builtin_expect(ic.anySpill, 0)) { spill_static<false>(spill_ctx(S), &R, NS); wave_sync(); }
    if (integrateNext) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) { HS_BODY(r) if (valid) integrate_body(R, br[r], slot, g, meta, aforce); }
        wave_sync();
    }
    HS_CTICK(7, 8)
}

// ROUNDS = rounds of 64 lanes that cover the octet's bodies: 2 up to 16 body slots per world (<= 5 agents), 3 with
// 6 agents (17 slots x 8 worlds = 136 bodies at most).
#ifndef HS_EXP_PRIO
#define HS_EXP_PRIO 1
#endif
template <int ROUNDS, class OR>
HSD void physics_step(SimState &S, OR &R, GenScratch *gen) {
    constexpr int T = OR::kT;
    const int L = hs_lane(), o = blockIdx.x;            // o: this wave's tile
    S.wbeg = o * T;                                   // first slot of the tile in the tiled columns
    const int p0 = S.wbeg;
    const int NS = kAgentSlot0 + S.A;                 // body slots in use
    const int noct = gridDim.x;
    // The launch ends with its slowest wave: the physics waves that were slow in the previous step (the same worlds:
    // contact piles persist) go first on their SIMD, the partner of a slow wave has slack.
    // Measured: k_physics 0.387 -> 0.380 ms; boosting a wave while it is inside a chain of manifolds or an extra round
    // of convex tests instead gave 0.382.
    const long long tStart = wall_clock64();
    // (S.stepIdx < 0: a launch replayed from a HIP graph, whose arguments are frozen at capture — no rotating sums, no hint)
    if (S.stepIdx < 0 || HS_EXP_PRIO == 0) __builtin_amdgcn_s_setprio(2);
    else {
        const int sidx = S.stepIdx, prevIdx = sidx == 0 ? 2 : sidx - 1, nextIdx = sidx == 2 ? 0 : sidx + 1;
        if (o == 0 && L == 0) S.tickSum[nextIdx] = 0ull;
        const float mean = (float)S.tickSum[prevIdx] / (float)noct, mine = (float)S.octTicks[o];
        const bool known = mean > 0.f;
        if (known && mine > 1.10f * mean) __builtin_amdgcn_s_setprio(3);
        else if (known && mine > 0.98f * mean) __builtin_amdgcn_s_setprio(2);
        else if (known && mine > 0.88f * mean) __builtin_amdgcn_s_setprio(1);
        else if (known) __builtin_amdgcn_s_setprio(0);
        else __builtin_amdgcn_s_setprio(2);
    }
#ifdef HS_PHASE_TIMING
    // development aid: wall-clock ticks (100 MHz) per phase of every octet -> S.phaseTicks[octet][10]
    long long tk = wall_clock64(); long long acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // ---- the octet's columns -> LDS (the blocks are contiguous; worlds beyond N are zero padding = empty slots)
    copy_in<T>(&R.pos[0][0][0], S.bpos, p0); copy_in<T>(&R.rot[0][0][0], S.brot, p0);
    copy_in<T>(&R.u.vel.lin[0][0][0], S.blin, p0); copy_in<T>(&R.u.vel.ang[0][0][0], S.bang, p0);
    copy_in<T>(&R.meta[0][0], S.bmeta, p0);
    if (L < T) {
        const int w = S.worldOfSlot[S.wbeg + L];      // which world lives in this slot (k_balance moves them)
        R.wid[L] = w;
        R.numWalls[L] = w >= 0 ? (unsigned char)S.numWalls[w] : 0;
        R.numPlanes[L] = w >= 0 ? (unsigned char)S.numPlanes[w] : 0;
        R.seen[L] = 0; R.ndd[L] = 0; R.nsc[L] = 0;
        R.wallSeen[L] = w >= 0 ? (unsigned)S.wallHist[w] : 0u;
    }
    if (L < 4 * T) (&R.plane0[0][0])[L] = S.planes((L / T) * kMaxPlanes, S.wbeg + (L % T));
    wave_sync();
    HS_TICK(9)
    phase_pre(S, R);
    HS_TICK(0)
    // ---- compact list of the octet's existing bodies (a third of the box slots are empty).  Its order decides which
    // round a body belongs to and nothing else.  First the bodies that had a wall manifold in the previous step (contacts
    // persist: an agent pushing against a wall, a box resting at one), agents before boxes within each group: the wall
    // manifolds of a body of an earlier round ride in the idle lanes of the last round (last_round) instead of costing the
    // wave a round of their own.
    int nbodies;
    {
        int base = 0;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int c = 0; c < ROUNDS; ++c) {
                const int t = c * 64 + L;
                const int ord = t / T, g = t % T;
                const int slot = ord < S.A ? kAgentSlot0 + ord : ord - S.A;
                const bool hist = ord < NS && ((R.wallSeen[g] >> slot) & 1u) != 0u;
                const bool on = ord < NS && R.meta[slot][g] != 0 && hist == (pass == 0);
                const unsigned long long m = __ballot(on);
                if (on) R.bodies[base + __popcll(m & ((1ull << L) - 1ull))] = (unsigned char)(slot << 3 | g);
                base += __popcll(m);
            }
        }
        nbodies = base;
    }
    wave_sync();
    if (L < T) R.wallSeen[L] = 0u;                // from here on: this step's
    wave_sync();
    // A body keeps its (round, lane) for the whole step: which body it is comes from the list, its ground manifold
    // stays in registers.
    BodyReg br[ROUNDS];
    const float *const aforce = S.aforce.octet(p0 >> 3) + (p0 & 7);
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        BodyReg &b = br[r];
        b.np = 0; b.vidx = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) b.lam[j] = 0.f;
        HS_BODY(r)
        // integrate for the first substep (the later ones happen at the end of the velocity pass)
        if (valid) integrate_body(R, b, slot, g, meta, aforce);
    }
    wave_sync();
    HS_CTICK(1, 0)
    // The substeps.  (The spill path — pairs beyond the LDS capacities, a handful in millions of world-steps — sits behind
    // wave-uniform branches on ic.anySpill inside the phases.  While the kernel kept its lane constants alive across all
    // phases it sat at its 256-register budget and this code cost the hot path 12 spilled registers whichever way it was
    // packaged: as a second instantiation of the substep entered only when a world spills (12), in a loop of its own that
    // finishes the step (18), as a real call (93).  With the opaque lane index — hs_dev.h hs_lane() — the kernel needs 221
    // registers and the plain form spills nothing.)
#pragma unroll 1
    for (int sub = 0; sub < kNumSubsteps; ++sub) {
        const ItemCounts ic = phase_detect<ROUNDS>(S, R, NS);
        HS_TICK(2)
        substep_rest<ROUNDS, true>(S, R, br, nbodies, NS, ic, sub + 1 < kNumSubsteps, aforce HS_TICK_ARGS);
    }
#undef HS_BODY
    phase_post(S, R);
    wave_sync();
    // ---- LDS -> the octet's columns
    copy_out<T>(S.bpos, p0, &R.pos[0][0][0]); copy_out<T>(S.brot, p0, &R.rot[0][0][0]);
    copy_out_vel(S.blin, p0, &R.u.vel.lin[0][0][0], R); copy_out_vel(S.bang, p0, &R.u.vel.ang[0][0][0], R);
    copy_out<T>(S.bmeta, p0, &R.meta[0][0]);
    mem_sync();                           // the write-back is complete before a regenerated level overwrites it
    HS_CTICK(8, 9)
    // resetSystem, one lane per world: step counter, or a whole new level on the 240th step / on request
    // (the generator works in the LDS the octet no longer needs: hs_k_reset.h GenScratch)
    {
        const int myWorld = L < T ? R.wid[L] : -1;
        const int mySeen = L < T ? (int)R.wallSeen[L] : 0;
        wave_sync();                      // every lane has read what it needs from the resident set
        if (myWorld >= 0) {
            S.wallHist[myWorld] = mySeen;
            reset_world(S, myWorld, gen[L]);
        }
    }
    if (L == 0 && S.stepIdx >= 0) {
        const int now = (int)(wall_clock64() - tStart);
        const int dt = HS_EXP_PRIO == 2 ? (S.octTicks[o] + now) >> 1 : HS_EXP_PRIO == 3 ? (3 * S.octTicks[o] + now) >> 2 : now;
        S.octTicks[o] = dt;
        atomicAdd(&S.tickSum[S.stepIdx], (unsigned long long)dt);
    }
#ifdef HS_PHASE_TIMING
    if (L == 0) for (int i = 0; i < 10; ++i) S.phaseTicks[(size_t)o * 10 + i] += acc[i];
#endif
#undef HS_TICK
}

// -DHS_EXP_REGCAP=W (development aid): the kernel under the register budget of W waves per SIMD — the LDS becomes a dynamic
// allocation (with the static 20 KiB the compiler knows that W > 2 is unattainable and ignores the request); occupancy on the
// part stays at two waves per SIMD (LDS), so what the run shows is the cost of the spills alone.
// -DHS_EXP_REGCAP=W (development aid): the 8-world kernel under the register budget of W waves per SIMD — its LDS becomes a
// dynamic allocation (with the static 20 KiB the compiler knows that W > 2 is unattainable and ignores the request);
// occupancy on the part stays at two waves per SIMD (LDS), so what the run shows is the cost of the spills alone.
template <int ROUNDS, int T>
#ifdef HS_EXP_REGCAP
__global__ void __launch_bounds__(kPhysThreads) __attribute__((amdgpu_waves_per_eu(HS_EXP_REGCAP, HS_EXP_REGCAP))) k_physics(SimState S) {
    union PhysLds { OctResT<T> R; GenScratch gen[T]; };
    extern __shared__ __attribute__((aligned(16))) char dynlds[];
    PhysLds &lds = *reinterpret_cast<PhysLds *>(dynlds);
#else
__global__ void __launch_bounds__(kPhysThreads, T == 8 ? 2 : 4) k_physics(SimState S) {
    // (the generator's working memory — the reset at the tail of the step — shares the tile's LDS: hs_k_reset.h GenScratch)
    __shared__ union PhysLds { OctResT<T> R; GenScratch gen[T]; } lds;
    static_assert(sizeof(PhysLds) <= (T == 8 ? 20 : 10) * 1024, "8 waves of 8 worlds / 16 waves of 4 worlds share the CU's 160 KiB of LDS");
#endif
    physics_step<ROUNDS>(S, lds.R, lds.gen);
}

}  // namespace hs
