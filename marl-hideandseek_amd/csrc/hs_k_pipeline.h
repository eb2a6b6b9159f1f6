// Physics step: movement / actions, PhysicsSystem::setupPhysicsStepTasks (src/sim.cpp:1162-1163; engine
// source absent — DESIGN.md "Engine decisions") and the reward systems as ONE persistent kernel.
//
// A workgroup of kPhysThreads threads owns kPhysWorlds consecutive worlds for the whole step.  Their working set
// — pose, previous pose, velocities and meta word of every body — is loaded from the SoA-across-worlds columns
// ONCE at the start of the launch, lives in the CU's LDS through the four substeps (struct PhysRes, ~153 KiB of
// the 160 KiB, together with the hull AABBs and the walls that the broadphase reads), and is written back once
// at the end: the HBM-side traffic of the step is the state in and out, not a re-read per phase, and a phase pays
// LDS latency (~64 clocks) instead of an L2 / Infinity Cache round trip (~200-550) for every body it touches.
//
// The phases below are separated by workgroup barriers only.  Worlds never interact, so no phase has to wait for
// the slowest item of the whole batch (which is what a kernel boundary per phase costs: measured on MI355X,
// every sparse phase then takes as long as its worst wave, 3-4x the average one).
//
// Per-body work runs with SLOT-MAJOR lanes over the workgroup's worlds (the compact list of existing bodies keeps
// that order), so the lanes of a wave hold the same body slot of consecutive worlds: conflict-free LDS rows,
// lanes share the hull type.  The sparse work — convex tests of candidate pairs, body-body manifolds, bodies with
// wall candidates — is compacted into the workgroup's slice of the work lists (wavefront scan + one LDS atomic
// per list) and processed by one, two or eight lanes per item.
//
// Substep s:
//   integrate        slot-major      (substep 0; later ones happen at the end of body_vel); ground-plane manifold
//   detect           8 lanes/world   all-pairs AABB candidates -> per-world lists (slot order) + work lists
//   sat              2 lanes/pair    exact convex test -> manifold workspace
//   dd<pos>          8 lanes/world*  joints, then body-body manifolds in (i<j) order   (*worlds that have any)
//   body_pos         lane/body       ground manifold, the body's wall / extra-plane manifolds; velocity derivation
//   dd<vel>          8 lanes/world*  body-body velocity pass
//   body_vel         lane/body       ground + wall velocity pass; integrate for substep s+1
// The Gauss-Seidel order and every rounding are the oracle's (joints, body-body in pair order,
// then per body: ground, walls by static id).
#pragma once
#include "hs_state.h"
#include "hs_rays.h"
#include "hs_collide.h"
#include "hs_solver.h"
#include "hs_k_reset.h"

namespace hs {

#ifndef HS_PHYS_THREADS
#define HS_PHYS_THREADS 512
#define HS_PHYS_WORLDS 64
#endif
#ifndef HS_PHYS_MIN_WAVES
#define HS_PHYS_MIN_WAVES 2
#endif
constexpr int kPhysThreads = HS_PHYS_THREADS;      // 8 waves per workgroup: 2 per SIMD, so the convex test keeps its ~220 VGPRs
constexpr int kPhysWorlds = HS_PHYS_WORLDS;        // worlds per workgroup (16 000 worlds -> 250 workgroups on 256 CUs)
constexpr int kPhysWaves = kPhysThreads / 64;
constexpr int kW = kPhysWorlds;

// ---- the workgroup's resident working set (LDS) ----
// Every array is [component][slot][world of the workgroup]: a wave that walks consecutive worlds of one slot
// reads / writes consecutive words.
struct PhysRes {
    float pos[3][kNumDSlots][kW];
    float rot[4][kNumDSlots][kW];        // w, x, y, z
    float ppos[3][kNumDSlots][kW];       // pose at the start of the substep
    float prot[4][kNumDSlots][kW];
    float lin[3][kNumDSlots][kW];
    float ang[3][kNumDSlots][kW];
    int meta[kNumDSlots][kW];            // meta_pack(); 0 = empty slot
    int numWalls[kW], numPlanes[kW];
    int actGL[kMaxAgents][kW];           // grab / lock requests of phase_pre
    // phase-local scratch sharing one allocation (each use sits between two workgroup barriers):
    union {
        struct {
            float lo[3][kNumDSlots][kW], hi[3][kNumDSlots][kW];    // hull AABBs: integrate -> detect
            float wall[4][kMaxWalls][kW];                          // cx, cy, hx, hy: staged for detect (stage_walls)
        } det;
        float clip[kPhysWaves][kClipWords];                        // polygon clipping of phase_sat
    } u;
    int wtot[4][kPhysWaves];             // phase_detect: per-wave totals of the four work lists
    int bbase[4];
    int seen[kW];                        // phase_post: a seeker sees a hider
    int list_len[2 * 4];                 // work-list lengths (sat box, wall bodies, ddw, sat ramp) x substep parity
    int chunk_ctr[2];                    // next slot-major chunk of phase_body_pos / phase_body_vel
    int integ_scratch[kNumDSlots + 2];   // chunk offsets of the compact body list
};
static_assert(sizeof(PhysRes) <= 160 * 1024, "the resident working set must fit the CU's 160 KiB of LDS");

// List lengths are workgroup-local (LDS), bumped with atomics and read by other waves in a later phase.
HSD int load_counter(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// ---- accessors of the resident columns ----
template <int C> HSD V3 rld3(const float (&a)[C][kNumDSlots][kW], int slot, int wl) { return {a[0][slot][wl], a[1][slot][wl], a[2][slot][wl]}; }
HSD Q rld4(const float (&a)[4][kNumDSlots][kW], int slot, int wl) { return {a[0][slot][wl], a[1][slot][wl], a[2][slot][wl], a[3][slot][wl]}; }
HSD void rst3(float (&a)[3][kNumDSlots][kW], int slot, int wl, V3 v) { a[0][slot][wl] = v.x; a[1][slot][wl] = v.y; a[2][slot][wl] = v.z; }
HSD void rst4(float (&a)[4][kNumDSlots][kW], int slot, int wl, Q q) { a[0][slot][wl] = q.w; a[1][slot][wl] = q.x; a[2][slot][wl] = q.y; a[3][slot][wl] = q.z; }

// index of (component c, slot, world w) in a world-fastest global column
HSD int bidx(const SimState &S, int c, int slot, int w) { return (c * kNumDSlots + slot) * S.N + w; }

HSD void rbody_load(const PhysRes &R, int wl, int slot, BodyS &b) {
    b.pos = rld3(R.pos, slot, wl); b.rot = rld4(R.rot, slot, wl);
    b.ppos = rld3(R.ppos, slot, wl); b.prot = rld4(R.prot, slot, wl);
    b.lin = rld3(R.lin, slot, wl); b.ang = rld3(R.ang, slot, wl);
    const int m = R.meta[slot][wl];
    const bool dyn = m != 0 && meta_resp(m) == RESP_DYNAMIC;
    b.invM = dyn ? obj_inv_mass(meta_obj(m)) : 0.f;
    b.invI = dyn ? obj_inv_inertia(meta_obj(m)) : V3{0.f, 0.f, 0.f};
    body_refresh_inertia(b);
}
HSD void rbody_store_pose(PhysRes &R, int wl, int slot, const BodyS &b) { rst3(R.pos, slot, wl, b.pos); rst4(R.rot, slot, wl, b.rot); }
HSD void rbody_store_vel(PhysRes &R, int wl, int slot, const BodyS &b) { rst3(R.lin, slot, wl, b.lin); rst3(R.ang, slot, wl, b.ang); }
HSD void derive_velocity(BodyS &b) {
    const float h = kSubstepH;
    b.lin = (b.pos - b.ppos) * (1.f / h);
    Q dq = qmul(b.rot, qinv(b.prot));
    V3 wv = V3{dq.x, dq.y, dq.z} * (2.f / h);
    b.ang = dq.w >= 0.f ? wv : -wv;
}

// Geometry view of one world of the workgroup for trace_ray (hs_rays.h): bodies from the resident columns, walls
// and the (at most 3) planes from global memory (the rays of the physics kernel — lock / grab, seeker -> hider
// line of sight — are few; the lidar / visibility rays are k_observe's).
struct ResGeom {
    const PhysRes &R; const SimState &S; int wl;
    HSD int g_meta(int i) const { return R.meta[i][wl]; }
    HSD V3 g_pos(int i) const { return rld3(R.pos, i, wl); }
    HSD Q g_rot(int i) const { return rld4(R.rot, i, wl); }
    HSD int g_num_walls() const { return R.numWalls[wl]; }
    HSD float g_wall(int k, int c) const { return S.walls[(c * kMaxWalls + k) * S.N + S.wbeg + wl]; }
    HSD int g_num_planes() const { return R.numPlanes[wl]; }
    HSD float g_plane(int p, int c) const { return S.planes[(c * kMaxPlanes + p) * S.N + S.wbeg + wl]; }
};

// per-body manifold word: ground np | ground vertex indices << 4 | first static candidate << 16 |
// static candidate count << 21 | hasStaticCandidates << 30
constexpr int kGndHasWall = 1 << 30;
constexpr int kGndScBegShift = 16, kGndScCntShift = 21;
// The word is double-buffered by substep parity: the end of phase_body_vel writes the next substep's word while
// other lanes of the phase still read this substep's.
HSD int gman_idx(const SimState &S, int par, int slot, int w) { return (par * kNumDSlots + slot) * S.N + w; }

// ------------------------------------------------------------------------------------------
// Launch prologue / epilogue: the workgroup's columns HBM -> LDS and back.  Rows of kW consecutive worlds are
// contiguous in the world-fastest columns, so every wave moves whole 256-byte rows.
HSD void load_resident(const SimState &S, PhysRes &R, int NS) {
    const int N = S.N, tid = threadIdx.x;
    for (int i = tid; i < kW; i += kPhysThreads) {
        const bool ok = i < S.wcnt;
        R.numWalls[i] = ok ? S.numWalls[S.wbeg + i] : 0;
        R.numPlanes[i] = ok ? S.numPlanes[S.wbeg + i] : 0;
        R.seen[i] = 0;
    }
    for (int i = tid; i < NS * kW; i += kPhysThreads) {
        const int slot = i / kW, wl = i - slot * kW;
        const bool ok = wl < S.wcnt;
        const int w = ok ? S.wbeg + wl : S.wbeg;
        R.meta[slot][wl] = ok ? S.bmeta[slot * N + w] : 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            R.pos[c][slot][wl] = ok ? S.bpos[bidx(S, c, slot, w)] : 0.f;
            R.lin[c][slot][wl] = ok ? S.blin[bidx(S, c, slot, w)] : 0.f;
            R.ang[c][slot][wl] = ok ? S.bang[bidx(S, c, slot, w)] : 0.f;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) R.rot[c][slot][wl] = ok ? S.brot[bidx(S, c, slot, w)] : 0.f;
    }
}
// The walls of the workgroup's worlds -> the detect scratch (phase_sat's clip buffers share that allocation, so
// they are staged again for every substep: by the launch prologue for the first, by phase_dd<vel> — whose waves
// are mostly idle — for the following ones).
HSD void stage_walls(const SimState &S, PhysRes &R) {
    const int N = S.N;
    for (int i = threadIdx.x; i < kMaxWalls * kW; i += kPhysThreads) {
        const int k = i / kW, wl = i - k * kW;
        if (wl >= S.wcnt || k >= R.numWalls[wl]) continue;
        const int w = S.wbeg + wl;
#pragma unroll
        for (int c = 0; c < 4; ++c) R.u.det.wall[c][k][wl] = S.walls[(c * kMaxWalls + k) * N + w];
    }
}
HSD void store_resident(const SimState &S, const PhysRes &R, int NS) {
    const int N = S.N, tid = threadIdx.x;
    for (int i = tid; i < NS * kW; i += kPhysThreads) {
        const int slot = i / kW, wl = i - slot * kW;
        if (wl >= S.wcnt) continue;
        const int w = S.wbeg + wl;
        const int m = R.meta[slot][wl];
        S.bmeta[slot * N + w] = m;
        if (m == 0) continue;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            S.bpos[bidx(S, c, slot, w)] = R.pos[c][slot][wl];
            S.blin[bidx(S, c, slot, w)] = R.lin[c][slot][wl];
            S.bang[bidx(S, c, slot, w)] = R.ang[c][slot][wl];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) S.brot[bidx(S, c, slot, w)] = R.rot[c][slot][wl];
    }
}

// ------------------------------------------------------------------------------------------
// Start of a substep for one body: remember the pose, semi-implicit Euler step (gravity, agent force
// and torque, gyroscopic term), refresh the hull AABB.
HSD void integrate_body(const SimState &S, PhysRes &R, int wl, int slot, int meta, V3 pos, Q rot, V3 lin, V3 ang, int par) {
    const int N = S.N, w = S.wbeg + wl;
    const int obj = meta_obj(meta);
    rst3(R.ppos, slot, wl, pos); rst4(R.prot, slot, wl, rot);
    if (meta_resp(meta) == RESP_DYNAMIC) {
        const float h = kSubstepH;
        const float invM = obj_inv_mass(obj);
        const V3 invI = obj_inv_inertia(obj);
        V3 force = {0.f, 0.f, 0.f}; float torque_z = 0.f;
        if (slot >= kAgentSlot0) {
            const int a = slot - kAgentSlot0;
            force = {S.aforce[(0 * kMaxAgents + a) * N + w], S.aforce[(1 * kMaxAgents + a) * N + w], S.aforce[(2 * kMaxAgents + a) * N + w]};
            torque_z = S.aforce[(3 * kMaxAgents + a) * N + w];
        }
        lin = lin + (force * invM + V3{0.f, 0.f, kGravityZ}) * h;
        pos = pos + lin * h;
        Q qi = qinv(rot);
        V3 wloc = qrot(qi, ang), tl = qrot(qi, V3{0.f, 0.f, torque_z});
        const V3 I = obj_inertia(obj);            // 1 / invI per axis, 0 where invI is 0
        V3 Iw = mulc(I, wloc);
        wloc = wloc + mulc(invI, tl - cross(wloc, Iw)) * h;
        ang = qrot(rot, wloc);
        rot = quat_add_rotation(rot, ang * h);
        rst3(R.pos, slot, wl, pos); rst4(R.rot, slot, wl, rot);
        rst3(R.lin, slot, wl, lin); rst3(R.ang, slot, wl, ang);
    }
    V3 lo, hi;
    const HullRef hb = hull_ref_body(obj, pos, rot);
    hull_aabb(hb, &lo, &hi);
    rst3(R.u.det.lo, slot, wl, lo); rst3(R.u.det.hi, slot, wl, hi);
    // ground plane (plane 0) manifold at the integrated pose; phase_detect adds the static-candidate range
    int gword = 0;
    if (meta_resp(meta) == RESP_DYNAMIC && R.numPlanes[wl] >= 1) {
        const V3 pn = {S.planes[(0 * kMaxPlanes) * N + w], S.planes[(1 * kMaxPlanes) * N + w], S.planes[(2 * kMaxPlanes) * N + w]};
        int vidx; float off[4] = {0.f, 0.f, 0.f, 0.f};
        const int np = ground_manifold(hb, pn, S.planes[(3 * kMaxPlanes) * N + w], &vidx, off);
        if (np > 0) {
            gword |= np | (vidx << 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < np) S.goff[bidx(S, j, slot, w)] = off[j];
                S.glam[bidx(S, j, slot, w)] = 0.f;
            }
        }
    }
    S.gman[gman_idx(S, par, slot, w)] = gword;
}

// Per-world bookkeeping at the start of substep `par`: clear the candidate counts, queue worlds with a
// grab joint for phase_dd.  (The list lengths of a parity are cleared by phase_detect of the other one.)
HSD void substep_begin_worlds(const SimState &S, PhysRes &R, int par) {
    const int N = S.N;
    for (int i = threadIdx.x; i < S.wcnt; i += kPhysThreads) {
        const int w = S.wbeg + i;
        S.ndd[w] = 0; S.nsc[w] = 0;
        bool grab = false;
        for (int a = 0; a < kMaxAgents; ++a) grab |= S.grabOther[a * N + w] >= 0;
        S.wflags[w] = grab ? 1 : 0;
        if (grab) S.ddwList[atomicAdd(&R.list_len[par * 4 + 2], 1)] = i;
    }
}

// First substep only; the later substeps are integrated at the end of phase_body_vel.  Also compacts the
// workgroup's existing bodies into bodyList (slot-major order kept, so the lanes of a wave still mostly share
// a hull type): about a third of the box slots are empty, and the slot-major passes of body_pos / body_vel
// would carry them as idle lanes in every substep.
HSD int phase_integrate(const SimState &S, PhysRes &R, int NS, int par) {
    substep_begin_worlds(S, R, par);
    const int total = NS * S.wcnt, nchunks = (total + 63) / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int *const chunk_base = R.integ_scratch;         // [nchunks + 1], nchunks <= 17
    for (int c = wave; c < nchunks; c += kPhysWaves) {
        const int t = c * 64 + lane;
        int meta = 0, slot = 0, wl = 0;
        if (t < total) { slot = t / S.wcnt; wl = t - slot * S.wcnt; meta = R.meta[slot][wl]; }
        const unsigned long long m = __ballot(meta != 0);
        if (lane == 0) chunk_base[c + 1] = __popcll(m);
        if (meta != 0) {
            V3 lin = {0.f, 0.f, 0.f}, ang = {0.f, 0.f, 0.f};
            if (meta_resp(meta) == RESP_DYNAMIC) { lin = rld3(R.lin, slot, wl); ang = rld3(R.ang, slot, wl); }
            integrate_body(S, R, wl, slot, meta, rld3(R.pos, slot, wl), rld4(R.rot, slot, wl), lin, ang, par);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) { chunk_base[0] = 0; for (int c = 0; c < nchunks; ++c) chunk_base[c + 1] += chunk_base[c]; }
    __syncthreads();
    for (int c = wave; c < nchunks; c += kPhysWaves) {
        const int t = c * 64 + lane;
        int meta = 0, slot = 0, wl = 0;
        if (t < total) { slot = t / S.wcnt; wl = t - slot * S.wcnt; meta = R.meta[slot][wl]; }
        const unsigned long long m = __ballot(meta != 0);
        if (meta != 0) S.bodyList[chunk_base[c] + __popcll(m & ((1ull << lane) - 1ull))] = (wl << 5) | slot;
    }
    const int nbodies = chunk_base[nchunks];
    __syncthreads();
    return nbodies;
}

// ------------------------------------------------------------------------------------------
// All-pairs AABB candidates of one world (<= 17 bodies, <= 36 walls: no BVH), 8 lanes per world.  The AABBs and
// walls are resident, so there is nothing to stage.  List space is reserved with ONE atomic per list per pass
// (wave scans + workgroup scan).
constexpr int kDetectLanes = 8;        // lanes per world in phase_detect: lane l owns body slots l, l+8, l+16
HSD void detect_pass(const SimState &S, PhysRes &R, int wfirst, int NS, int par) {
    constexpr int G = kDetectLanes, JB = (kNumDSlots + G - 1) / G, NW = kPhysWaves;
    const int tid = threadIdx.x, grp = tid / G, l = tid % G;
    const int wl = wfirst + grp;
    const int w = S.wbeg + wl;
    const int N = S.N;
    const bool wok = wl < S.wcnt;
    const int wq = wok ? wl : 0;                       // a valid column for the (masked) reads of idle lanes
    int *cnt = R.list_len + par * 4;
    if (tid == 0 && wfirst == 0) { int *c = R.list_len + ((par ^ 1) * 4); c[0] = 0; c[1] = 0; c[2] = 0; c[3] = 0; }
    const int nwl = wok ? R.numWalls[wq] : 0, npl = wok ? R.numPlanes[wq] : 0;
    // lane l owns body slots l, l + 8 (and lane 0 slot 16 when 6 agents are configured).  The loops run over the
    // OTHER body / the wall, each read from LDS once and tested against all of the lane's slots.
    int tot_items = 0;
    unsigned dd_mask[JB] = {}; unsigned long long s_mask[JB] = {};
    int bdd[JB] = {}, bsc[JB] = {}, add[JB] = {}, asc[JB] = {};
    bool have[JB], dynamic[JB]; V3 lo[JB], hi[JB];
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) {
        const int slot = l + jb * G;
        const int meta = (wok && slot < NS) ? R.meta[slot][wq] : 0;
        have[jb] = meta != 0;
        dynamic[jb] = have[jb] && meta_resp(meta) == RESP_DYNAMIC;
        lo[jb] = {0.f, 0.f, 0.f}; hi[jb] = {0.f, 0.f, 0.f};
        if (have[jb]) { lo[jb] = rld3(R.u.det.lo, slot, wq); hi[jb] = rld3(R.u.det.hi, slot, wq); }
    }
    if (wok) {
        for (int j = 1; j < NS; ++j) {
            const int mj = R.meta[j][wq];
            if (mj == 0) continue;
            const bool dynj = meta_resp(mj) == RESP_DYNAMIC;
            const V3 loj = rld3(R.u.det.lo, j, wq), hij = rld3(R.u.det.hi, j, wq);
#pragma unroll
            for (int jb = 0; jb < JB; ++jb) {
                if (have[jb] && l + jb * G < j && (dynamic[jb] || dynj) &&
                    lo[jb].x <= hij.x && loj.x <= hi[jb].x && lo[jb].y <= hij.y && loj.y <= hi[jb].y &&
                    lo[jb].z <= hij.z && loj.z <= hi[jb].z) dd_mask[jb] |= 1u << j;
            }
        }
        for (int k = 0; k < nwl; ++k) {
            const float cx = R.u.det.wall[0][k][wq], cy = R.u.det.wall[1][k][wq], hx = R.u.det.wall[2][k][wq], hy = R.u.det.wall[3][k][wq];
            const float wx0 = cx - hx, wx1 = cx + hx, wy0 = cy - hy, wy1 = cy + hy;
#pragma unroll
            for (int jb = 0; jb < JB; ++jb) {
                if (dynamic[jb] && lo[jb].x <= wx1 && wx0 <= hi[jb].x && lo[jb].y <= wy1 && wy0 <= hi[jb].y &&
                    lo[jb].z <= 2.5f && 0.f <= hi[jb].z) s_mask[jb] |= 1ull << k;
            }
        }
    }
    // candidate slots in the world's lists: prefix sums in body-slot order (slots l of all lanes, then l + 8, ...),
    // i.e. the oracle's candidate order — so even the pairs dropped beyond the capacity are the oracle's
    int tot_dd = 0, tot_sc = 0;
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) {
        if (dynamic[jb]) for (int p = 1; p < npl; ++p) s_mask[jb] |= 1ull << (kMaxWalls + p);
        const int cdd = __popc(dd_mask[jb]), csc = __popcll(s_mask[jb]);
        int in_dd = cdd, in_sc = csc;
#pragma unroll
        for (int d = 1; d < G; d <<= 1) {
            const int y0 = __shfl_up(in_dd, d, G), y1 = __shfl_up(in_sc, d, G);
            if (l >= d) { in_dd += y0; in_sc += y1; }
        }
        bdd[jb] = tot_dd + in_dd - cdd; bsc[jb] = tot_sc + in_sc - csc;
        tot_dd += __shfl(in_dd, G - 1, G); tot_sc += __shfl(in_sc, G - 1, G);
        add[jb] = cdd ? max(0, min(cdd, kMaxDDCand - bdd[jb])) : 0;
        asc[jb] = csc ? max(0, min(csc, kMaxSCand - bsc[jb])) : 0;
        tot_items += add[jb] + asc[jb];
        if (add[jb] != cdd || asc[jb] != csc) {      // beyond the capacity: dropped (as the CPU restatement does), and counted
            if (add[jb] != cdd) atomicAdd(&S.status[0], cdd - add[jb]);
            if (asc[jb] != csc) atomicAdd(&S.status[1], csc - asc[jb]);
            *S.hostFlag = 1;
        }
    }
    // ---- reserve space in the work lists: wave scans, workgroup scan, one atomic per list.
    // Convex-test items that involve a ramp (wedge hull) are kept apart from the box-only ones — they go
    // to the far end of the workgroup's list slice — so that most waves of phase_sat run the box code only.
    const int lane = tid & 63, wv = tid >> 6;
    bool push_ddw = false;
    int n_wall = 0, n_wedge = 0;
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) {
        push_ddw |= add[jb] > 0 && bdd[jb] == 0; n_wall += asc[jb] > 0 ? 1 : 0;
        const int slot = l + jb * G;
        const bool ramp = slot >= kRampSlot0 && slot < kRampSlot0 + kMaxRamps;
        if (ramp) n_wedge += add[jb] + asc[jb];
        else { unsigned mm = dd_mask[jb]; for (int i = 0; mm && i < add[jb]; ++i) { const int j = __ffs(mm) - 1; mm &= mm - 1; n_wedge += (j >= kRampSlot0 && j < kRampSlot0 + kMaxRamps) ? 1 : 0; } }
    }
    push_ddw = push_ddw && wok && S.wflags[w] == 0;
    int mine[4] = {tot_items - n_wedge, n_wall, push_ddw ? 1 : 0, n_wedge};
    int incl[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int x = mine[q];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d); if (lane >= d) x += y; }
        incl[q] = x;
        if (lane == 63) R.wtot[q][wv] = x;
    }
    __syncthreads();
    if (tid < 4) {
        int tot = 0;
        for (int k = 0; k < NW; ++k) { const int c = R.wtot[tid][k]; R.wtot[tid][k] = tot; tot += c; }
        R.bbase[tid] = tot > 0 ? atomicAdd(&cnt[tid], tot) : 0;
    }
    __syncthreads();
    int gbase = R.bbase[0] + R.wtot[0][wv] + incl[0] - mine[0];
    int wbase2 = R.bbase[1] + R.wtot[1][wv] + incl[1] - mine[1];
    int gback = S.wcnt * (kMaxDDCand + kMaxSCand) - 1 - (R.bbase[3] + R.wtot[3][wv] + incl[3] - mine[3]);
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) {
        const int slot = l + jb * G;
        const bool ramp = slot >= kRampSlot0 && slot < kRampSlot0 + kMaxRamps;
        unsigned mm = dd_mask[jb]; int i = 0;
        while (mm && i < add[jb]) {
            const int j = __ffs(mm) - 1; mm &= mm - 1;
            S.ddPair[(bdd[jb] + i) * N + w] = slot | (j << 8);
            const int item = (wl << 6) | (bdd[jb] + i);
            if (ramp || (j >= kRampSlot0 && j < kRampSlot0 + kMaxRamps)) S.satList[gback--] = item; else S.satList[gbase++] = item;
            ++i;
        }
        // oracle order inside a body: extra planes first, then walls by index; the body's candidates
        // occupy the contiguous range [bsc, bsc + asc) of the world's list
        unsigned long long sm = (s_mask[jb] >> kMaxWalls) | (s_mask[jb] << (64 - kMaxWalls) >> (64 - kMaxWalls) << kMaxPlanes); i = 0;
        while (sm && i < asc[jb]) {
            const int bit = __ffsll((long long)sm) - 1; sm &= sm - 1;
            const int k = bit < kMaxPlanes ? kMaxWalls + bit : bit - kMaxPlanes;
            S.scPair[(bsc[jb] + i) * N + w] = slot | (k << 8);
            const int item = (wl << 6) | 32 | (bsc[jb] + i);
            if (ramp) S.satList[gback--] = item; else S.satList[gbase++] = item;
            ++i;
        }
        if (asc[jb] > 0) S.wallList[wbase2++] = (wl << 5) | slot;      // bodies with static candidates: own work items
    }
    if (push_ddw) S.ddwList[R.bbase[2] + R.wtot[2][wv] + incl[2] - 1] = wl;
    if (wok && l == 0) { S.ndd[w] = tot_dd; S.nsc[w] = tot_sc; }
    // ---- the static-candidate range of the owned bodies joins their ground-manifold word (phase_integrate)
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) {
        const int slot = l + jb * G;
        if (!wok || slot >= NS || asc[jb] <= 0) continue;
        S.gman[gman_idx(S, par, slot, w)] |= kGndHasWall | (bsc[jb] << kGndScBegShift) | (asc[jb] << kGndScCntShift);
    }
}
HSD void phase_detect(const SimState &S, PhysRes &R, int NS, int par) {
    for (int wfirst = 0; wfirst < S.wcnt; wfirst += kPhysThreads / kDetectLanes) {
        detect_pass(S, R, wfirst, NS, par);
        __syncthreads();                  // the next pass reuses wtot / bbase; phase_sat reuses the AABB rows
    }
}

// ------------------------------------------------------------------------------------------
// Two lanes per item: lane L < kClipLanes and lane L + 32 run the convex test of the same pair together
// (collide_hulls); the low lane owns the clip scratch in the wave's LDS slice and writes the manifold.
HSD void phase_sat(const SimState &S, PhysRes &R, int par) {
    static_assert(kClipLanes == 32, "lane L pairs with lane L + 32");
    const int N = S.N;
    // box-only items from the front of the list, then (starting at a fresh wave) the ramp items from its far end
    const int nbox = load_counter(&R.list_len[par * 4 + 0]), nwedge = load_counter(&R.list_len[par * 4 + 3]);
    const int wedge0 = (nbox + kClipLanes - 1) / kClipLanes * kClipLanes;
    const int total = wedge0 + nwedge, cap = S.wcnt * (kMaxDDCand + kMaxSCand);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool hi = lane >= kClipLanes;
    const ClipBuf cb = {R.u.clip[wave], lane & (kClipLanes - 1)};
    for (int it = wave * kClipLanes + (lane & (kClipLanes - 1)); it < total; it += kPhysWaves * kClipLanes) {
        if (it >= nbox && it < wedge0) continue;
        const int item = it < nbox ? S.satList[it] : S.satList[cap - 1 - (it - wedge0)];
        const int wl = item >> 6, idx = item & 63;
        const int w = S.wbeg + wl;
        const bool isdd = idx < 32;
        const int kk = idx & 31;
        const int pair = isdd ? S.ddPair[kk * N + w] : S.scPair[kk * N + w];
        const int a = pair & 0xff, bsel = pair >> 8;
        ManDD *const wsDD = (ManDD *)S.wsDD + (size_t)w * kMaxDDCand;
        ManS *const wsSC = (ManS *)S.wsSC + (size_t)w * kMaxSCand;
        const int oa = meta_obj(R.meta[a][wl]);
        const V3 pa = rld3(R.pos, a, wl);
        const Q qa = rld4(R.rot, a, wl);
        const HullRef ha = hull_ref_body(oa, pa, qa);
        RawManifold raw;
        if (!hi) { if (isdd) wsDD[kk].np = 0; else wsSC[kk].np = 0; }
        if (!isdd && bsel >= kMaxWalls) {
            const int p = bsel - kMaxWalls;
            const V3 pn = {S.planes[(0 * kMaxPlanes + p) * N + w], S.planes[(1 * kMaxPlanes + p) * N + w], S.planes[(2 * kMaxPlanes + p) * N + w]};
            if (!hi && collide_hull_plane(ha, pn, S.planes[(3 * kMaxPlanes + p) * N + w], raw)) {
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
            ob = meta_obj(R.meta[bsel][wl]); pb = rld3(R.pos, bsel, wl); qb = rld4(R.rot, bsel, wl);
            hb = hull_ref_body(ob, pb, qb);
        } else {
            ob = OBJ_WALL;
            // (the staged walls share their LDS with this phase's clip buffers: from global memory here)
            hb = hull_ref_wall(S.walls[(0 * kMaxWalls + bsel) * N + w], S.walls[(1 * kMaxWalls + bsel) * N + w],
                               S.walls[(2 * kMaxWalls + bsel) * N + w], S.walls[(3 * kMaxWalls + bsel) * N + w]);
        }
        if (!collide_hulls(ha, hb, cb, raw, hi)) continue;
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
// Body-body manifolds (and grab joints) of one world, 8 lanes per world.  The oracle solves them
// one after the other in (i<j) pair order; manifolds that share no body commute exactly, so lane q
// takes the q-th accepted manifold of the sorted order and runs as soon as no EARLIER manifold that
// is still pending touches one of its bodies.  Disjoint pairs are solved in one round instead of
// one after the other; the result is bit-identical to the sequential order.
template <bool POS>
HSD void phase_dd(const SimState &S, PhysRes &R, int par) {
    constexpr int GL = 8;
    const int N = S.N;
    const int total = load_counter(&R.list_len[par * 4 + 2]);
    const int q = threadIdx.x % GL;
    const int gbit0 = (threadIdx.x & 63) / GL * GL;               // first lane of this group in the wave
    // consecutive worlds of the list go to different waves: a wave runs until the slowest of its worlds is done
    for (int it = ((threadIdx.x & 63) / GL) * kPhysWaves + (threadIdx.x >> 6); ; it += kPhysThreads / GL) {
        if (__ballot(it < total) == 0ull) break;                  // wave-uniform exit
        const bool live = it < total;
        const int wl = live ? S.ddwList[it] : 0;
        const int w = S.wbeg + wl;
        if (POS && live && q == 0 && S.wflags[w]) {
            const int teams = S.teams[w];
            for (int a = 0; a < kMaxAgents; ++a) {
                if (!team_agent_active(teams, a)) continue;
                const int other = S.grabOther[a * N + w];
                if (other < 0) continue;
                BodyS A, B;
                rbody_load(R, wl, kAgentSlot0 + a, A); rbody_load(R, wl, other, B);
                float gd[kGrabWords];
#pragma unroll
                for (int c = 0; c < kGrabWords; ++c) gd[c] = S.grabData[(c * kMaxAgents + a) * N + w];
                solve_grab_joint_bodies(A, B, {gd[0], gd[1], gd[2]}, {gd[3], gd[4], gd[5], gd[6]}, gd[7],
                                        {gd[8], gd[9], gd[10]}, {gd[11], gd[12], gd[13], gd[14]});
                rbody_store_pose(R, wl, kAgentSlot0 + a, A); rbody_store_pose(R, wl, other, B);
            }
            __threadfence_block();
        }
        ManDD *const wsDD = (ManDD *)S.wsDD + (size_t)w * kMaxDDCand;
        int ndd = 0;
        if (live) ndd = S.ndd[w] < kMaxDDCand ? S.ndd[w] : kMaxDDCand;
        // keys of the accepted candidates, kMaxDDCand = 16: lane q inspects candidates q and q+8
        int key0 = 0x7fffffff, key1 = 0x7fffffff;
        if (q < ndd && wsDD[q].np > 0) { const int p = S.ddPair[q * N + w]; key0 = ((p & 0xff) << 8) | (p >> 8); }
        if (q + GL < ndd && wsDD[q + GL].np > 0) { const int p = S.ddPair[(q + GL) * N + w]; key1 = ((p & 0xff) << 8) | (p >> 8); }
        // rank of every accepted candidate in sorted key order (keys are unique: distinct pairs)
        int rank0 = 0, rank1 = 0, nacc = 0;
#pragma unroll
        for (int p = 0; p < GL; ++p) {
            const int k0 = __shfl(key0, gbit0 + p), k1 = __shfl(key1, gbit0 + p);
            rank0 += (k0 < key0) + (k1 < key0); rank1 += (k0 < key1) + (k1 < key1);
            nacc += (k0 != 0x7fffffff) + (k1 != 0x7fffffff);
        }
        for (int base = 0; base < kMaxDDCand; base += GL) {
            if (__ballot(base < nacc) == 0ull) break;
            // lane q takes the manifold of rank base+q: find which lane/slot holds it
            int mine = -1;
#pragma unroll
            for (int p = 0; p < GL; ++p) {
                const int r0 = __shfl(rank0, gbit0 + p), r1 = __shfl(rank1, gbit0 + p);
                const int k0 = __shfl(key0, gbit0 + p), k1 = __shfl(key1, gbit0 + p);
                if (k0 != 0x7fffffff && r0 == base + q) mine = p;
                if (k1 != 0x7fffffff && r1 == base + q) mine = p + GL;
            }
            int ma = -1, mb = -1;
            if (mine >= 0) { ma = wsDD[mine].a; mb = wsDD[mine].b; }
            bool pending = mine >= 0;
            while (true) {
                const unsigned long long pend_mask = __ballot(pending);
                if (pend_mask == 0ull) break;
                bool ready = pending;
#pragma unroll
                for (int p = 0; p < GL; ++p) {
                    const int pa = __shfl(ma, gbit0 + p), pb = __shfl(mb, gbit0 + p);
                    const bool ppend = (pend_mask >> (gbit0 + p)) & 1ull;
                    if (p < q && ppend && (pa == ma || pa == mb || pb == ma || pb == mb)) ready = false;
                }
                if (ready) {
                    ManDD m = wsDD[mine];
                    BodyS Ab, Bb;
                    rbody_load(R, wl, m.a, Ab); rbody_load(R, wl, m.b, Bb);
                    const V3 n = ld3(m.n);
                    if (POS) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (j < m.np) wsDD[mine].lam[j] = m.lam[j] + solve_point_position<true>(Ab, Bb, n, ld3(m.rA[j]), ld3(m.rB[j]), 0.f, m.muS);
                        rbody_store_pose(R, wl, m.a, Ab); rbody_store_pose(R, wl, m.b, Bb);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (j < m.np) solve_point_velocity<true>(Ab, Bb, n, ld3(m.rA[j]), ld3(m.rB[j]), m.lam[j], m.muD);
                        rbody_store_vel(R, wl, m.a, Ab); rbody_store_vel(R, wl, m.b, Bb);
                    }
                    pending = false;
                }
                __threadfence_block();        // later rounds of this wave must see the poses just written
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Static contacts of one body, in the oracle's order: ground manifold, then the body's other static
// candidates (extra planes, walls by index) — one thread per body, slot-major.
template <bool WALLED>
HSD void body_pos_item(const SimState &S, PhysRes &R, int slot, int wl, int par) {
    const int N = S.N, w = S.wbeg + wl;
    const int meta = R.meta[slot][wl];
    if (meta == 0 || meta_resp(meta) != RESP_DYNAMIC) return;
    const int gword = S.gman[gman_idx(S, par, slot, w)];
    const int np = gword & 7;
    const bool has_wall = (gword & kGndHasWall) != 0;
    if (has_wall != WALLED) return;       // bodies with static candidates are separate work items
    const int obj = meta_obj(meta);
    BodyS me, none;
    rbody_load(R, wl, slot, me);
    if (np > 0) {
        const V3 gn = -V3{S.planes[(0 * kMaxPlanes) * N + w], S.planes[(1 * kMaxPlanes) * N + w], S.planes[(2 * kMaxPlanes) * N + w]};
        const float gmuS = 0.5f * (obj_mu_s(obj) + obj_mu_s(OBJ_PLANE));
#pragma unroll 1
        for (int j = 0; j < np; ++j) {
            const float lam = solve_point_position<false>(me, none, gn, hull_local_vertex(obj, (gword >> (4 + 3 * j)) & 7),
                                                          V3{0.f, 0.f, 0.f}, S.goff[bidx(S, j, slot, w)], gmuS);
            S.glam[bidx(S, j, slot, w)] += lam;
        }
    }
    if (has_wall) {
        ManS *const wsSC = (ManS *)S.wsSC + (size_t)w * kMaxSCand;
        const int bsc = (gword >> kGndScBegShift) & 31, asc = (gword >> kGndScCntShift) & 31;
#pragma unroll 1
        for (int k = bsc; k < bsc + asc; ++k) {       // the body's candidates, already in solve order
            ManS m = wsSC[k];
            if (m.np <= 0) continue;
            body_refresh_inertia(me);
            const V3 n = ld3(m.n);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < m.np) wsSC[k].lam[j] = m.lam[j] + solve_point_position<false>(me, none, n, ld3(m.rA[j]), V3{0.f, 0.f, 0.f}, m.offB[j], m.muS);
        }
    }
    if (np > 0 || has_wall) rbody_store_pose(R, wl, slot, me);
    derive_velocity(me);
    rbody_store_vel(R, wl, slot, me);
}
// The few bodies with wall / extra-plane candidates take 2-3x longer than the others, so they are not
// left inside the slot-major waves (where 63 lanes would wait for one): the first waves take them packed
// from the wall-body list while the other waves start on the slot-major chunks, handed out 64 items at a
// time from an LDS counter; whoever finishes first takes more chunks.
HSD int next_chunk(int *ctr) {
    int c = 0;
    if ((threadIdx.x & 63) == 0) c = atomicAdd(ctr, 1);
    return __shfl(c, 0);
}
HSD void phase_body_pos(const SimState &S, PhysRes &R, int nbodies, int par) {
    const int nwall = load_counter(&R.list_len[par * 4 + 1]);
    for (int it = threadIdx.x; it < nwall; it += kPhysThreads) {
        const int item = S.wallList[it];
        body_pos_item<true>(S, R, item & 31, item >> 5, par);
    }
    for (int c = next_chunk(&R.chunk_ctr[0]); c * 64 < nbodies; c = next_chunk(&R.chunk_ctr[0])) {
        const int t = c * 64 + (threadIdx.x & 63);
        if (t >= nbodies) continue;
        const int item = S.bodyList[t];
        body_pos_item<false>(S, R, item & 31, item >> 5, par);
    }
}

// Velocity pass over a body's static contacts; with NEXT, also the start of the following substep
// (parity par_next) for every body, so the body is integrated from registers.
template <bool NEXT, bool WALLED>
HSD void body_vel_item(const SimState &S, PhysRes &R, int slot, int wl, int par, int par_next) {
    const int N = S.N, w = S.wbeg + wl;
    const int meta = R.meta[slot][wl];
    if (meta == 0) return;
    if (meta_resp(meta) != RESP_DYNAMIC) {
        if (NEXT) integrate_body(S, R, wl, slot, meta, rld3(R.pos, slot, wl), rld4(R.rot, slot, wl), V3{0.f, 0.f, 0.f}, V3{0.f, 0.f, 0.f}, par_next);
        return;
    }
    const int gword = S.gman[gman_idx(S, par, slot, w)];
    const int np = gword & 7;
    const bool has_wall = (gword & kGndHasWall) != 0;
    if (has_wall != WALLED) return;       // bodies with static candidates are separate work items
    if (np == 0 && !has_wall) {
        if (NEXT) integrate_body(S, R, wl, slot, meta, rld3(R.pos, slot, wl), rld4(R.rot, slot, wl), rld3(R.lin, slot, wl), rld3(R.ang, slot, wl), par_next);
        return;
    }
    const int obj = meta_obj(meta);
    BodyS me, none;
    rbody_load(R, wl, slot, me);
    if (np > 0) {
        const V3 gn = -V3{S.planes[(0 * kMaxPlanes) * N + w], S.planes[(1 * kMaxPlanes) * N + w], S.planes[(2 * kMaxPlanes) * N + w]};
        const float gmuD = 0.5f * (obj_mu_d(obj) + obj_mu_d(OBJ_PLANE));
#pragma unroll 1
        for (int j = 0; j < np; ++j)
            solve_point_velocity<false>(me, none, gn, hull_local_vertex(obj, (gword >> (4 + 3 * j)) & 7), V3{0.f, 0.f, 0.f},
                                        S.glam[bidx(S, j, slot, w)], gmuD);
    }
    if (has_wall) {
        const ManS *const wsSC = (const ManS *)S.wsSC + (size_t)w * kMaxSCand;
        const int bsc = (gword >> kGndScBegShift) & 31, asc = (gword >> kGndScCntShift) & 31;
#pragma unroll 1
        for (int k = bsc; k < bsc + asc; ++k) {
            const ManS m = wsSC[k];
            if (m.np <= 0) continue;
            body_refresh_inertia(me);
            const V3 n = ld3(m.n);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < m.np) solve_point_velocity<false>(me, none, n, ld3(m.rA[j]), V3{0.f, 0.f, 0.f}, m.lam[j], m.muD);
        }
    }
    if (NEXT) integrate_body(S, R, wl, slot, meta, me.pos, me.rot, me.lin, me.ang, par_next);
    else rbody_store_vel(R, wl, slot, me);
}
template <bool NEXT>
HSD void phase_body_vel(const SimState &S, PhysRes &R, int nbodies, int par, int par_next) {
    const int nwall = load_counter(&R.list_len[par * 4 + 1]);
    for (int it = threadIdx.x; it < nwall; it += kPhysThreads) {
        const int item = S.wallList[it];
        body_vel_item<NEXT, true>(S, R, item & 31, item >> 5, par, par_next);
    }
    for (int c = next_chunk(&R.chunk_ctr[1]); c * 64 < nbodies; c = next_chunk(&R.chunk_ctr[1])) {
        const int t = c * 64 + (threadIdx.x & 63);
        if (t >= nbodies) continue;
        const int item = S.bodyList[t];
        body_vel_item<NEXT, false>(S, R, item & 31, item >> 5, par, par_next);
    }
    if (NEXT) substep_begin_worlds(S, R, par_next);
}

// ------------------------------------------------------------------------------------------
// actionSystem (sim.cpp:270-370) for one world, agents in interface order, run by ONE lane: lock / grab ray casts
// against the resident geometry, joint create / destroy.  Meta words change in LDS (written back at the end of
// the launch); the joint table lives in global memory.
HSD void action_system(const SimState &S, PhysRes &R, int wl, int A_, int teams) {
    const int N = S.N, w = S.wbeg + wl;
    const ResGeom g = {R, S, wl};
    for (int i = 0; i < A_; ++i) {
        const int fl = R.actGL[i][wl];
        if (fl == 0) continue;
        const int type = team_agent_type(teams, i);
        const int slot = kAgentSlot0 + i;
        const V3 mpos = g.g_pos(slot);
        const Q mrot = g.g_rot(slot);
        if (fl & 2) {   // lock
            float t; V3 o = mpos + V3{0.f, 0.f, 0.5f};
            int hit = trace_ray(g, o, qrot(mrot, {0.f, 1.f, 0.f}), 2.5f, &t);
            if (hit >= 0 && hit < kNumDSlots) {
                const int m = R.meta[hit][wl];
                const int obj = meta_obj(m), resp = meta_resp(m), owner = meta_owner(m);
                if (resp == RESP_STATIC) {
                    if ((type == AGENT_SEEKER && owner == OWNER_SEEKER) || (type == AGENT_HIDER && owner == OWNER_HIDER))
                        R.meta[hit][wl] = meta_pack(obj, RESP_DYNAMIC, OWNER_NONE);
                } else if (owner == OWNER_NONE) {
                    R.meta[hit][wl] = meta_pack(obj, RESP_STATIC, type == AGENT_HIDER ? OWNER_HIDER : OWNER_SEEKER);
                }
            }
        }
        if (fl & 1) {   // grab
            if (S.grabOther[i * N + w] >= 0) {
                S.grabOther[i * N + w] = -1;
            } else {
                float t; V3 o = mpos + V3{0.f, 0.f, 0.5f};
                V3 dir = qrot(mrot, {0.f, 1.f, 0.f});
                int hit = trace_ray(g, o, dir, 2.5f, &t);
                if (hit >= 0 && hit < kNumDSlots) {
                    const int m = R.meta[hit][wl];
                    if (meta_owner(m) == OWNER_NONE && meta_resp(m) == RESP_DYNAMIC) {
                        V3 hit_pos = o + dir * t;
                        Q erot = g.g_rot(hit);
                        V3 r2 = qrot(qinv(erot), hit_pos - g.g_pos(hit));
                        Q at2 = qnormalize(qmul(qinv(erot), mrot));
                        S.grabOther[i * N + w] = hit;
                        const float gd[kGrabWords] = {r2.x, r2.y, r2.z, at2.w, at2.x, at2.y, at2.z, t - 1.25f,
                                                      0.f, 1.25f, 0.5f,            // r1 = 1.25 fwd + 0.5 up (sim.cpp:343-344)
                                                      1.f, 0.f, 0.f, 0.f};         // attachRot1 = identity
#pragma unroll
                        for (int c = 0; c < kGrabWords; ++c) S.grabData[(c * kMaxAgents + i) * N + w] = gd[c];
                    }
                }
            }
        }
    }
}

// Before the substeps: movementSystem | instantMovementSystem (sim.cpp:202-254) and actionSystem (:270-370).
// A lane per (world, agent) maps the action row to a force; a lane per world then runs the action system for the
// worlds in which an agent locks or grabs (scripts/benchmark.py never does, scripts/jax_train.py does all the time).
HSD void phase_pre(const SimState &S, PhysRes &R) {
    const int N = S.N, A_ = S.A;
    const bool instant = (S.flags & FLAG_ZERO_AGENT_VELOCITY) == FLAG_ZERO_AGENT_VELOCITY;
    bool need_action = false;
    for (int it = threadIdx.x; it < kMaxAgents * kW; it += kPhysThreads) {
        const int agent = it / kW, wl = it - agent * kW;
        R.actGL[agent][wl] = 0;
        if (agent >= A_ || wl >= S.wcnt) continue;
        const int w = S.wbeg + wl;
        const int teams = S.teams[w], step = S.curEpisodeStep[w];
        const bool active = team_agent_active(teams, agent) != 0;
        const int type = team_agent_type(teams, agent);
        if (active && !(type == AGENT_SEEKER && step < kNumPrepSteps - 1)) {
            int32_t *act_row = S.xAction + (w * A_ + agent) * 5;
            const int ax = act_row[0], ay = act_row[1], ar = act_row[2], ag = act_row[3], al = act_row[4];
            float fx, fy, tz;
            if (instant) { fx = 400.f * (float)(ax - 2); fy = 400.f * (float)(ay - 2); tz = 120.f * (float)(ar - 2); }
            else { fx = 12.f * (float)(ax - 5); fy = 12.f * (float)(ay - 5); tz = 3.f * (float)(ar - 5); }
            V3 f = qrot(rld4(R.rot, kAgentSlot0 + agent, wl), {fx, fy, 0.f});
            S.aforce[(0 * kMaxAgents + agent) * N + w] = f.x; S.aforce[(1 * kMaxAgents + agent) * N + w] = f.y;
            S.aforce[(2 * kMaxAgents + agent) * N + w] = f.z; S.aforce[(3 * kMaxAgents + agent) * N + w] = tz;
            const int fl = (ag == 1 ? 1 : 0) | (al == 1 ? 2 : 0);
            R.actGL[agent][wl] = fl;
            need_action |= fl != 0;
            act_row[0] = 2; act_row[1] = 2; act_row[2] = 2; act_row[3] = 0; act_row[4] = 0;   // sim.cpp:365-369
        }
    }
    // any lock / grab request in this workgroup?  (workgroup-uniform decision; the barrier also publishes actGL)
    const int any = __syncthreads_or(need_action ? 1 : 0);
    if (any) {
        // one lane per world, the worlds spread over all waves: the ray casts of a world are sequential
        constexpr int per = (kW + kPhysWaves - 1) / kPhysWaves;
        const int lane = threadIdx.x & 63, wl = (threadIdx.x >> 6) * per + lane;
        if (lane < per && wl < S.wcnt) {
            bool want = false;
            for (int a = 0; a < A_; ++a) want |= R.actGL[a][wl] != 0;
            if (want) action_system(S, R, wl, A_, S.teams[S.wbeg + wl]);
        }
    }
    __syncthreads();
}

// After the substeps: agentZeroVelSystem (sim.cpp:258-268), rewardsVisSystem (:763-804),
// outputRewardsDonesSystem (:806-841), updateEpisodeResultsSystem (:843-893).
constexpr int kPostLanes = 8;           // lanes per world in phase_post: the workgroup's 64 worlds in one pass
HSD void post_pass(const SimState &S, PhysRes &R, int wfirst) {
    constexpr int G = kPostLanes;
    const int tid = threadIdx.x, grp = tid / G, l = tid % G;
    const int wl = wfirst + grp;
    const int w = S.wbeg + wl;
    const int N = S.N, A_ = S.A;
    const bool wok = wl < S.wcnt;
    const bool instant = (S.flags & FLAG_ZERO_AGENT_VELOCITY) == FLAG_ZERO_AGENT_VELOCITY;
    int teams = 0, step = 0, counts = 0;
    if (wok) {
        teams = S.teams[w]; step = S.curEpisodeStep[w]; counts = S.counts[w];
        if (l == 0) R.seen[wl] = 0;
        if (instant && l < kMaxAgents && R.meta[kAgentSlot0 + l][wl] != 0) {
            const int slot = kAgentSlot0 + l;
            R.lin[0][slot][wl] = 0.f; R.lin[1][slot][wl] = 0.f;
            R.lin[2][slot][wl] = fminf(R.lin[2][slot][wl], 0.f);
            rst3(R.ang, slot, wl, V3{0.f, 0.f, 0.f});
        }
    }
    __syncthreads();
    // The seen flag feeds the reward (from episode step 95 on) and the episode result (from 96 on); the reset overwrites
    // hiderTeamReward every step, so during the preparation phase the rays would change nothing anyone can read.
    for (int pr = l; wok && step >= kNumPrepSteps - 1 && pr < 9; pr += G) {      // (seeker, hider) pairs
        const int si = pr / 3, hi_ = pr % 3;
        if (si < cnt_seekers(counts) && hi_ < cnt_hiders(counts)) {
            const ResGeom g = {R, S, wl};
            const int ss = kAgentSlot0 + team_seeker(teams, si), hs_ = kAgentSlot0 + team_hider(teams, hi_);
            const V3 spos = g.g_pos(ss);
            const V3 fwd = qrot(g.g_rot(ss), {0.f, 1.f, 0.f});
            V3 to = g.g_pos(hs_) - spos;
            float c = dot(normalize(to), fwd);
            if (!(c < kCosFovHalf)) {
                float t;
                if (trace_ray(g, spos, to, 1.f, &t) == hs_) R.seen[wl] = 1;   // every writer stores the same value
            }
        }
    }
    __syncthreads();
    if (!wok) return;
    float hider_reward = S.hiderTeamReward[w];
    if (R.seen[wl]) hider_reward = -1.f;
    if (l < A_ && team_agent_active(teams, l)) {
        const int agent = l, slot = kAgentSlot0 + agent, row = w * A_ + agent;
        if (step == 0) S.xDone[row] = 0;
        if (step < kNumPrepSteps - 1) {
            S.xReward[row] = 0.f;
        } else {
            if (step == kEpisodeLen - 1) S.xDone[row] = 1;
            float r = hider_reward;
            if (team_agent_type(teams, agent) == AGENT_SEEKER) r *= -1.f;
            if (fabsf(R.pos[0][slot][wl]) >= 18.f || fabsf(R.pos[1][slot][wl]) >= 18.f) r -= 10.f;
            S.xReward[row] = r;
        }
    }
    if (l == 0) {
        float *res = S.xEpisodeResult + w * 2;
        int s0 = S.runningScores[0 * N + w], s1 = S.runningScores[1 * N + w];
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
        S.runningScores[0 * N + w] = s0; S.runningScores[1 * N + w] = s1;
        S.hiderTeamReward[w] = hider_reward;
    }
}

HSD void phase_post(const SimState &S, PhysRes &R) {
    for (int wfirst = 0; wfirst < S.wcnt; wfirst += kPhysThreads / kPostLanes) {
        post_pass(S, R, wfirst);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kPhysThreads, HS_PHYS_MIN_WAVES) k_physics(SimState S) {
    __shared__ PhysRes R;
    // this workgroup's worlds, and its slices of the work lists
    S.wbeg = blockIdx.x * kPhysWorlds;
    S.wcnt = min(kPhysWorlds, S.N - S.wbeg);
    S.satList += (size_t)S.wbeg * (kMaxDDCand + kMaxSCand);
    S.ddwList += (size_t)S.wbeg * 2;
    S.wallList += (size_t)S.wbeg * kNumDSlots;
    S.bodyList += (size_t)S.wbeg * kNumDSlots;
    // the list lengths live in LDS: every phase starts by reading one, and an L2 round trip there is pure latency
    if (threadIdx.x < 8) R.list_len[threadIdx.x] = 0;
    const int NS = kAgentSlot0 + S.A;                 // body slots in use
#ifdef HS_PHASE_TIMING
    // development aid: wall-clock ticks (100 MHz) per phase of every workgroup -> S.phaseTicks[workgroup][10]
    long long tk = wall_clock64(); long long acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define HS_TICK(i) { __syncthreads(); const long long now_ = wall_clock64(); acc[i] += now_ - tk; tk = now_; }
#else
#define HS_TICK(i)
#endif
    load_resident(S, R, NS);
    __syncthreads();                      // numWalls is in place
    stage_walls(S, R);
    __syncthreads();
    HS_TICK(9)
    phase_pre(S, R);
    HS_TICK(0)
    int nbodies = 0;
    for (int sub = 0; sub < kNumSubsteps; ++sub) {
        const int par = sub & 1;
        if (sub == 0) nbodies = phase_integrate(S, R, NS, par);
        HS_TICK(1)
        if (threadIdx.x == 0) { R.chunk_ctr[0] = 0; R.chunk_ctr[1] = 0; }
        phase_detect(S, R, NS, par);
        HS_TICK(2)
        phase_sat(S, R, par);
        __syncthreads();
        HS_TICK(3)
        phase_dd<true>(S, R, par);
        __syncthreads();
        HS_TICK(4)
        phase_body_pos(S, R, nbodies, par);
        __syncthreads();
        HS_TICK(5)
        phase_dd<false>(S, R, par);
        if (sub + 1 < kNumSubsteps) stage_walls(S, R);        // for the next substep's detect
        __syncthreads();
        HS_TICK(6)
        if (sub + 1 < kNumSubsteps) phase_body_vel<true>(S, R, nbodies, par, par ^ 1);
        else phase_body_vel<false>(S, R, nbodies, par, 0);
        __syncthreads();
        HS_TICK(7)
    }
    phase_post(S, R);
    store_resident(S, R, NS);
    __syncthreads();                      // the write-back is complete before a regenerated level overwrites it
    HS_TICK(8)
    // resetSystem for the workgroup's worlds: step counter, or a whole new level on the 240th step / on request.
    // The level generator is serial per world and diverges between worlds, so the worlds are spread over all the
    // workgroup's waves (kW / kPhysWaves lanes each) instead of filling one.
    {
        constexpr int per = (kW + kPhysWaves - 1) / kPhysWaves;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int wl = wave * per + lane;
        if (lane < per && wl < S.wcnt) reset_world(S, S.wbeg + wl);
    }
#ifdef HS_PHASE_TIMING
    if (threadIdx.x == 0) for (int i = 0; i < 10; ++i) S.phaseTicks[blockIdx.x * 10 + i] += acc[i];
#endif
#undef HS_TICK
}

}  // namespace hs
