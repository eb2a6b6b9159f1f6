// Physics step: movement / actions, PhysicsSystem::setupPhysicsStepTasks (src/sim.cpp:1162-1163; engine
// source absent — DESIGN.md "Engine decisions") and the reward systems as ONE persistent kernel.
//
// A workgroup of kPhysThreads threads owns kPhysWorlds consecutive worlds for the whole step and walks
// them through the phases below, separated by workgroup barriers only.  Worlds never interact, so no
// phase has to wait for the slowest item of the whole batch (which is what a kernel boundary per
// phase costs: measured on MI355X, every sparse phase then takes as long as its worst wave, 3-4x
// the average one), and a world's columns stay in the CU's L1 / the XCD's L2 from phase to phase.
//
// Per-body work runs with SLOT-MAJOR lanes over the workgroup's worlds (the compact list of existing bodies keeps
// that order), so the lanes of a wave hold the same body slot of consecutive worlds: coalesced dword accesses
// of world-fastest columns, lanes share the hull type.  The sparse work — convex tests of candidate pairs,
// body-body manifolds, bodies with wall candidates — is compacted into the workgroup's slice of the work lists
// (wavefront scan + one LDS atomic per list) and processed by one, two or eight lanes per item.
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
constexpr int kPhysThreads = HS_PHYS_THREADS;      // 8 waves per workgroup: 2 per SIMD, so the convex test keeps its ~220 VGPRs
constexpr int kPhysWorlds = HS_PHYS_WORLDS;        // worlds per workgroup (16 000 worlds -> 250 workgroups on 256 CUs)
constexpr int kPhysWaves = kPhysThreads / 64;

// List lengths are workgroup-local (LDS, reached through SimState::counters), bumped with atomics and read by
// other waves in a later phase.
HSD int load_counter(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// ---- SoA accessors ----
HSD int bidx(const SimState &S, int c, int slot, int w) { return (c * kNumDSlots + slot) * S.N + w; }
HSD V3 gld3(const float *col, const SimState &S, int slot, int w) {
    return {col[bidx(S, 0, slot, w)], col[bidx(S, 1, slot, w)], col[bidx(S, 2, slot, w)]};
}
HSD Q gld4(const float *col, const SimState &S, int slot, int w) {
    return {col[bidx(S, 0, slot, w)], col[bidx(S, 1, slot, w)], col[bidx(S, 2, slot, w)], col[bidx(S, 3, slot, w)]};
}
HSD void gst3(float *col, const SimState &S, int slot, int w, V3 v) {
    col[bidx(S, 0, slot, w)] = v.x; col[bidx(S, 1, slot, w)] = v.y; col[bidx(S, 2, slot, w)] = v.z;
}
HSD void gst4(float *col, const SimState &S, int slot, int w, Q q) {
    col[bidx(S, 0, slot, w)] = q.w; col[bidx(S, 1, slot, w)] = q.x; col[bidx(S, 2, slot, w)] = q.y; col[bidx(S, 3, slot, w)] = q.z;
}
HSD void gbody_load(const SimState &S, int w, int slot, BodyS &b) {
    b.pos = gld3(S.bpos, S, slot, w); b.rot = gld4(S.brot, S, slot, w);
    b.ppos = gld3(S.bppos, S, slot, w); b.prot = gld4(S.bprot, S, slot, w);
    b.lin = gld3(S.blin, S, slot, w); b.ang = gld3(S.bang, S, slot, w);
    const int m = S.bmeta[slot * S.N + w];
    const bool dyn = m != 0 && meta_resp(m) == RESP_DYNAMIC;
    b.invM = dyn ? obj_inv_mass(meta_obj(m)) : 0.f;
    b.invI = dyn ? obj_inv_inertia(meta_obj(m)) : V3{0.f, 0.f, 0.f};
    body_refresh_inertia(b);
}
HSD void gbody_store_pose(const SimState &S, int w, int slot, const BodyS &b) { gst3(S.bpos, S, slot, w, b.pos); gst4(S.brot, S, slot, w, b.rot); }
HSD void gbody_store_vel(const SimState &S, int w, int slot, const BodyS &b) { gst3(S.blin, S, slot, w, b.lin); gst3(S.bang, S, slot, w, b.ang); }
HSD void derive_velocity(BodyS &b) {
    const float h = kSubstepH;
    b.lin = (b.pos - b.ppos) * (1.f / h);
    Q dq = qmul(b.rot, qinv(b.prot));
    V3 wv = V3{dq.x, dq.y, dq.z} * (2.f / h);
    b.ang = dq.w >= 0.f ? wv : -wv;
}

// Append `value` to a global list for every lane with pred set: one atomic per wave
// (wavefront ballot + prefix count).  Must be called by all lanes of the wave.
HSD void wave_push(int *list, int *counter, int value, bool pred) {
    const unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(mask));
    base = __shfl(base, leader);
    if (pred) list[base + __popcll(mask & ((1ull << lane) - 1ull))] = value;
}

// per-body manifold word: ground np | ground vertex indices << 4 | first static candidate << 16 |
// static candidate count << 21 | hasStaticCandidates << 30
constexpr int kGndHasWall = 1 << 30;
constexpr int kGndScBegShift = 16, kGndScCntShift = 21;
// The word is double-buffered by substep parity: the end of phase_body_vel writes the next substep's word while
// other lanes of the phase still read this substep's.
HSD int gman_idx(const SimState &S, int par, int slot, int w) { return (par * kNumDSlots + slot) * S.N + w; }

// ------------------------------------------------------------------------------------------
// Start of a substep for one body: remember the pose, semi-implicit Euler step (gravity, agent force
// and torque, gyroscopic term), refresh the hull AABB.
HSD void integrate_body(const SimState &S, int w, int slot, int meta, V3 pos, Q rot, V3 lin, V3 ang, int par) {
    const int N = S.N;
    const int obj = meta_obj(meta);
    gst3(S.bppos, S, slot, w, pos); gst4(S.bprot, S, slot, w, rot);
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
        V3 wl = qrot(qi, ang), tl = qrot(qi, V3{0.f, 0.f, torque_z});
        const V3 I = obj_inertia(obj);            // 1 / invI per axis, 0 where invI is 0
        V3 Iw = mulc(I, wl);
        wl = wl + mulc(invI, tl - cross(wl, Iw)) * h;
        ang = qrot(rot, wl);
        rot = quat_add_rotation(rot, ang * h);
        gst3(S.bpos, S, slot, w, pos); gst4(S.brot, S, slot, w, rot);
        gst3(S.blin, S, slot, w, lin); gst3(S.bang, S, slot, w, ang);
    }
    V3 lo, hi;
    const HullRef hb = hull_ref_body(obj, pos, rot);
    hull_aabb(hb, &lo, &hi);
    gst3(S.blo, S, slot, w, lo); gst3(S.bhi, S, slot, w, hi);
    // ground plane (plane 0) manifold at the integrated pose; phase_detect adds the static-candidate range
    int gword = 0;
    if (meta_resp(meta) == RESP_DYNAMIC && S.numPlanes[w] >= 1) {
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
HSD void substep_begin_worlds(const SimState &S, int par) {
    const int N = S.N;
    for (int i = threadIdx.x; i < S.wcnt; i += kPhysThreads) {
        const int w = S.wbeg + i;
        S.ndd[w] = 0; S.nsc[w] = 0;
        bool grab = false;
        for (int a = 0; a < kMaxAgents; ++a) grab |= S.grabOther[a * N + w] >= 0;
        S.wflags[w] = grab ? 1 : 0;
        if (grab) S.ddwList[atomicAdd(&S.counters[par * 4 + 2], 1)] = w;
    }
}

// First substep only; the later substeps are integrated at the end of phase_body_vel.  Also compacts the
// workgroup's existing bodies into bodyList (slot-major order kept, so the lanes of a wave still mostly share
// a hull type): about a third of the box slots are empty, and the slot-major passes of body_pos / body_vel
// would carry them as idle lanes in every substep.  `scratch` is kPhysWaves * 4 + 1 ints of LDS.
HSD int phase_integrate(const SimState &S, int NS, int par, int *scratch) {
    const int N = S.N;
    substep_begin_worlds(S, par);
    const int total = NS * S.wcnt, nchunks = (total + 63) / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int *const chunk_base = scratch;                 // [nchunks + 1], nchunks <= 17
    for (int c = wave; c < nchunks; c += kPhysWaves) {
        const int t = c * 64 + lane;
        int meta = 0, slot = 0, w = 0;
        if (t < total) { slot = t / S.wcnt; w = S.wbeg + (t - slot * S.wcnt); meta = S.bmeta[slot * N + w]; }
        const unsigned long long m = __ballot(meta != 0);
        if (lane == 0) chunk_base[c + 1] = __popcll(m);
        if (meta != 0) {
            V3 lin = {0.f, 0.f, 0.f}, ang = {0.f, 0.f, 0.f};
            if (meta_resp(meta) == RESP_DYNAMIC) { lin = gld3(S.blin, S, slot, w); ang = gld3(S.bang, S, slot, w); }
            integrate_body(S, w, slot, meta, gld3(S.bpos, S, slot, w), gld4(S.brot, S, slot, w), lin, ang, par);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) { chunk_base[0] = 0; for (int c = 0; c < nchunks; ++c) chunk_base[c + 1] += chunk_base[c]; }
    __syncthreads();
    for (int c = wave; c < nchunks; c += kPhysWaves) {
        const int t = c * 64 + lane;
        int meta = 0, slot = 0, w = 0;
        if (t < total) { slot = t / S.wcnt; w = S.wbeg + (t - slot * S.wcnt); meta = S.bmeta[slot * N + w]; }
        const unsigned long long m = __ballot(meta != 0);
        if (meta != 0) S.bodyList[chunk_base[c] + __popcll(m & ((1ull << lane) - 1ull))] = (w << 5) | slot;
    }
    const int nbodies = chunk_base[nchunks];
    __syncthreads();
    return nbodies;
}

// ------------------------------------------------------------------------------------------
// World-major here (16 lanes per world, lane = body slot): the all-pairs AABB tests re-read every
// wall and every other body's AABB, so the world's boxes are staged once in LDS.
struct DetectWorld {
    int meta[kNumDSlots];
    float lo[kNumDSlots][3], hi[kNumDSlots][3];
    float wall[kMaxWalls][4];
};

// One pass: kPhysThreads / kDetectLanes worlds starting at local index wfirst.  List space is
// reserved with ONE atomic per list per pass (wave scans + workgroup scan).
constexpr int kDetectLanes = 8;        // lanes per world in phase_detect: lane l owns body slots l, l+8, l+16
struct DetectLds {
    DetectWorld sh[kPhysThreads / kDetectLanes];
    int wtot[4][kPhysWaves];
    int bbase[4];
};
HSD void detect_pass(const SimState &S, DetectLds &L, int wfirst, int NS, int par) {
    constexpr int G = kDetectLanes, JB = (kNumDSlots + G - 1) / G, NW = kPhysWaves;
    DetectWorld *const sh = L.sh;
    int (*const wtot)[NW] = L.wtot;
    int *const bbase = L.bbase;
    const int tid = threadIdx.x, grp = tid / G, l = tid % G;
    const int w = S.wbeg + wfirst + grp;
    const int N = S.N;
    const bool wok = wfirst + grp < S.wcnt;
    DetectWorld &dw = sh[grp];
    int *cnt = S.counters + par * 4;
    if (tid == 0 && wfirst == 0) { int *c = S.counters + ((par ^ 1) * 4); c[0] = 0; c[1] = 0; c[2] = 0; c[3] = 0; }
    int nwl = 0, npl = 0;
    if (wok) {
        nwl = S.numWalls[w]; npl = S.numPlanes[w];
        for (int s = l; s < kNumDSlots; s += G) {
            const int m = s < NS ? S.bmeta[s * N + w] : 0;
            dw.meta[s] = m;
            if (m != 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) { dw.lo[s][c] = S.blo[bidx(S, c, s, w)]; dw.hi[s][c] = S.bhi[bidx(S, c, s, w)]; }
            }
        }
        for (int k = l; k < nwl; k += G) {
#pragma unroll
            for (int c = 0; c < 4; ++c) dw.wall[k][c] = S.walls[(c * kMaxWalls + k) * N + w];
        }
    }
    __syncthreads();
    // lane l owns body slots l, l + 8 (and lane 0 slot 16 when 6 agents are configured).  The loops run over the
    // OTHER body / the wall, each read from LDS once and tested against all of the lane's slots.
    int tot_items = 0;
    unsigned dd_mask[JB] = {}; unsigned long long s_mask[JB] = {};
    int bdd[JB] = {}, bsc[JB] = {}, add[JB] = {}, asc[JB] = {};
    bool have[JB], dynamic[JB]; V3 lo[JB], hi[JB];
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) {
        const int slot = l + jb * G;
        const int meta = (wok && slot < NS) ? dw.meta[slot] : 0;
        have[jb] = meta != 0;
        dynamic[jb] = have[jb] && meta_resp(meta) == RESP_DYNAMIC;
        lo[jb] = {0.f, 0.f, 0.f}; hi[jb] = {0.f, 0.f, 0.f};
        if (have[jb]) { lo[jb] = {dw.lo[slot][0], dw.lo[slot][1], dw.lo[slot][2]}; hi[jb] = {dw.hi[slot][0], dw.hi[slot][1], dw.hi[slot][2]}; }
    }
    if (wok) {
        for (int j = 1; j < NS; ++j) {
            const int mj = dw.meta[j];
            if (mj == 0) continue;
            const bool dynj = meta_resp(mj) == RESP_DYNAMIC;
            const V3 loj = {dw.lo[j][0], dw.lo[j][1], dw.lo[j][2]}, hij = {dw.hi[j][0], dw.hi[j][1], dw.hi[j][2]};
#pragma unroll
            for (int jb = 0; jb < JB; ++jb) {
                if (have[jb] && l + jb * G < j && (dynamic[jb] || dynj) &&
                    lo[jb].x <= hij.x && loj.x <= hi[jb].x && lo[jb].y <= hij.y && loj.y <= hi[jb].y &&
                    lo[jb].z <= hij.z && loj.z <= hi[jb].z) dd_mask[jb] |= 1u << j;
            }
        }
        for (int k = 0; k < nwl; ++k) {
            const float cx = dw.wall[k][0], cy = dw.wall[k][1], hx = dw.wall[k][2], hy = dw.wall[k][3];
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
        if (add[jb] != cdd || asc[jb] != csc) {      // beyond the capacity: dropped (as the oracle does), and counted
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
        if (lane == 63) wtot[q][wv] = x;
    }
    __syncthreads();
    if (tid < 4) {
        int tot = 0;
        for (int k = 0; k < NW; ++k) { const int c = wtot[tid][k]; wtot[tid][k] = tot; tot += c; }
        bbase[tid] = tot > 0 ? atomicAdd(&cnt[tid], tot) : 0;
    }
    __syncthreads();
    int gbase = bbase[0] + wtot[0][wv] + incl[0] - mine[0];
    int wbase2 = bbase[1] + wtot[1][wv] + incl[1] - mine[1];
    int gback = S.wcnt * (kMaxDDCand + kMaxSCand) - 1 - (bbase[3] + wtot[3][wv] + incl[3] - mine[3]);
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) {
        const int slot = l + jb * G;
        const bool ramp = slot >= kRampSlot0 && slot < kRampSlot0 + kMaxRamps;
        unsigned mm = dd_mask[jb]; int i = 0;
        while (mm && i < add[jb]) {
            const int j = __ffs(mm) - 1; mm &= mm - 1;
            S.ddPair[(bdd[jb] + i) * N + w] = slot | (j << 8);
            const int item = (w << 6) | (bdd[jb] + i);
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
            const int item = (w << 6) | 32 | (bsc[jb] + i);
            if (ramp) S.satList[gback--] = item; else S.satList[gbase++] = item;
            ++i;
        }
        if (asc[jb] > 0) S.wallList[wbase2++] = (w << 5) | slot;      // bodies with static candidates: own work items
    }
    if (push_ddw) S.ddwList[bbase[2] + wtot[2][wv] + incl[2] - 1] = w;
    if (wok && l == 0) { S.ndd[w] = tot_dd; S.nsc[w] = tot_sc; }
    // ---- the static-candidate range of the owned bodies joins their ground-manifold word (phase_integrate)
#pragma unroll
    for (int jb = 0; jb < JB; ++jb) {
        const int slot = l + jb * G;
        if (!wok || slot >= NS || asc[jb] <= 0) continue;
        S.gman[gman_idx(S, par, slot, w)] |= kGndHasWall | (bsc[jb] << kGndScBegShift) | (asc[jb] << kGndScCntShift);
    }
}
HSD void phase_detect(const SimState &S, DetectLds &L, int NS, int par) {
    for (int wfirst = 0; wfirst < S.wcnt; wfirst += kPhysThreads / kDetectLanes) {
        detect_pass(S, L, wfirst, NS, par);
        __syncthreads();                  // the next pass reuses the LDS slots
    }
}

// ------------------------------------------------------------------------------------------
// Two lanes per item: lane L < kClipLanes and lane L + 32 run the convex test of the same pair together
// (collide_hulls); the low lane owns the clip scratch in the wave's LDS slice and writes the manifold.
struct SatLds { float clipmem[kPhysWaves][kClipWords]; };
HSD void phase_sat(const SimState &S, SatLds &L, int par) {
    static_assert(kClipLanes == 32, "lane L pairs with lane L + 32");
    const int N = S.N;
    // box-only items from the front of the list, then (starting at a fresh wave) the ramp items from its far end
    const int nbox = load_counter(&S.counters[par * 4 + 0]), nwedge = load_counter(&S.counters[par * 4 + 3]);
    const int wedge0 = (nbox + kClipLanes - 1) / kClipLanes * kClipLanes;
    const int total = wedge0 + nwedge, cap = S.wcnt * (kMaxDDCand + kMaxSCand);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool hi = lane >= kClipLanes;
    const ClipBuf cb = {L.clipmem[wave], lane & (kClipLanes - 1)};
    for (int it = wave * kClipLanes + (lane & (kClipLanes - 1)); it < total; it += kPhysWaves * kClipLanes) {
        if (it >= nbox && it < wedge0) continue;
        const int item = it < nbox ? S.satList[it] : S.satList[cap - 1 - (it - wedge0)];
        const int w = item >> 6, idx = item & 63;
        const bool isdd = idx < 32;
        const int kk = idx & 31;
        const int pair = isdd ? S.ddPair[kk * N + w] : S.scPair[kk * N + w];
        const int a = pair & 0xff, bsel = pair >> 8;
        ManDD *const wsDD = (ManDD *)S.wsDD + (size_t)w * kMaxDDCand;
        ManS *const wsSC = (ManS *)S.wsSC + (size_t)w * kMaxSCand;
        const int oa = meta_obj(S.bmeta[a * N + w]);
        const V3 pa = gld3(S.bpos, S, a, w);
        const Q qa = gld4(S.brot, S, a, w);
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
            ob = meta_obj(S.bmeta[bsel * N + w]); pb = gld3(S.bpos, S, bsel, w); qb = gld4(S.brot, S, bsel, w);
            hb = hull_ref_body(ob, pb, qb);
        } else {
            ob = OBJ_WALL;
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
HSD void phase_dd(const SimState &S, int par) {
    constexpr int GL = 8;
    const int N = S.N;
    const int total = load_counter(&S.counters[par * 4 + 2]);
    const int q = threadIdx.x % GL;
    const int gbit0 = (threadIdx.x & 63) / GL * GL;               // first lane of this group in the wave
    // consecutive worlds of the list go to different waves: a wave runs until the slowest of its worlds is done
    for (int it = ((threadIdx.x & 63) / GL) * kPhysWaves + (threadIdx.x >> 6); ; it += kPhysThreads / GL) {
        if (__ballot(it < total) == 0ull) break;                  // wave-uniform exit
        const bool live = it < total;
        const int w = live ? S.ddwList[it] : 0;
        if (POS && live && q == 0 && S.wflags[w]) {
            const int teams = S.teams[w];
            for (int a = 0; a < kMaxAgents; ++a) {
                if (!team_agent_active(teams, a)) continue;
                const int other = S.grabOther[a * N + w];
                if (other < 0) continue;
                BodyS A, B;
                gbody_load(S, w, kAgentSlot0 + a, A); gbody_load(S, w, other, B);
                float gd[kGrabWords];
#pragma unroll
                for (int c = 0; c < kGrabWords; ++c) gd[c] = S.grabData[(c * kMaxAgents + a) * N + w];
                solve_grab_joint_bodies(A, B, {gd[0], gd[1], gd[2]}, {gd[3], gd[4], gd[5], gd[6]}, gd[7],
                                        {gd[8], gd[9], gd[10]}, {gd[11], gd[12], gd[13], gd[14]});
                gbody_store_pose(S, w, kAgentSlot0 + a, A); gbody_store_pose(S, w, other, B);
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
                    gbody_load(S, w, m.a, Ab); gbody_load(S, w, m.b, Bb);
                    const V3 n = ld3(m.n);
                    if (POS) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (j < m.np) wsDD[mine].lam[j] = m.lam[j] + solve_point_position<true>(Ab, Bb, n, ld3(m.rA[j]), ld3(m.rB[j]), 0.f, m.muS);
                        gbody_store_pose(S, w, m.a, Ab); gbody_store_pose(S, w, m.b, Bb);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (j < m.np) solve_point_velocity<true>(Ab, Bb, n, ld3(m.rA[j]), ld3(m.rB[j]), m.lam[j], m.muD);
                        gbody_store_vel(S, w, m.a, Ab); gbody_store_vel(S, w, m.b, Bb);
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
// candidates (extra planes, walls by index) — one thread per body, slot-major.  The wall part runs on
// the few lanes whose body has candidates; a packed per-body kernel was tried first and lost: those
// kernels are bound by the latency of one lane's sequential solve, not by lane utilisation, so the
// extra launch (drain + dispatch + reloading the body) cost more than the idle lanes do here.
template <bool WALLED>
HSD void body_pos_item(const SimState &S, int slot, int w, int par) {
    const int N = S.N;
    const int meta = S.bmeta[slot * N + w];
    if (meta == 0 || meta_resp(meta) != RESP_DYNAMIC) return;
    const int gword = S.gman[gman_idx(S, par, slot, w)];
    const int np = gword & 7;
    const bool has_wall = (gword & kGndHasWall) != 0;
    if (has_wall != WALLED) return;       // bodies with static candidates are separate work items
    const int obj = meta_obj(meta);
    BodyS me, none;
    gbody_load(S, w, slot, me);
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
    if (np > 0 || has_wall) gbody_store_pose(S, w, slot, me);
    derive_velocity(me);
    gbody_store_vel(S, w, slot, me);
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
HSD void phase_body_pos(const SimState &S, int nbodies, int par, int *chunk_ctr) {
    const int nwall = load_counter(&S.counters[par * 4 + 1]);
    for (int it = threadIdx.x; it < nwall; it += kPhysThreads) {
        const int item = S.wallList[it];
        body_pos_item<true>(S, item & 31, item >> 5, par);
    }
    for (int c = next_chunk(chunk_ctr); c * 64 < nbodies; c = next_chunk(chunk_ctr)) {
        const int t = c * 64 + (threadIdx.x & 63);
        if (t >= nbodies) continue;
        const int item = S.bodyList[t];
        body_pos_item<false>(S, item & 31, item >> 5, par);
    }
}

// Velocity pass over a body's static contacts; with NEXT, also the start of the following substep
// (parity par_next) for every body, so the body is integrated from registers instead of by a
// separate launch.
template <bool NEXT, bool WALLED>
HSD void body_vel_item(const SimState &S, int slot, int w, int par, int par_next) {
    const int N = S.N;
    const int meta = S.bmeta[slot * N + w];
    if (meta == 0) return;
    if (meta_resp(meta) != RESP_DYNAMIC) {
        if (NEXT) integrate_body(S, w, slot, meta, gld3(S.bpos, S, slot, w), gld4(S.brot, S, slot, w), V3{0.f, 0.f, 0.f}, V3{0.f, 0.f, 0.f}, par_next);
        return;
    }
    const int gword = S.gman[gman_idx(S, par, slot, w)];
    const int np = gword & 7;
    const bool has_wall = (gword & kGndHasWall) != 0;
    if (has_wall != WALLED) return;       // bodies with static candidates are separate work items
    if (np == 0 && !has_wall) {
        if (NEXT) integrate_body(S, w, slot, meta, gld3(S.bpos, S, slot, w), gld4(S.brot, S, slot, w), gld3(S.blin, S, slot, w), gld3(S.bang, S, slot, w), par_next);
        return;
    }
    const int obj = meta_obj(meta);
    BodyS me, none;
    gbody_load(S, w, slot, me);
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
    if (NEXT) integrate_body(S, w, slot, meta, me.pos, me.rot, me.lin, me.ang, par_next);
    else gbody_store_vel(S, w, slot, me);
}
template <bool NEXT>
HSD void phase_body_vel(const SimState &S, int nbodies, int par, int par_next, int *chunk_ctr) {
    const int nwall = load_counter(&S.counters[par * 4 + 1]);
    for (int it = threadIdx.x; it < nwall; it += kPhysThreads) {
        const int item = S.wallList[it];
        body_vel_item<NEXT, true>(S, item & 31, item >> 5, par, par_next);
    }
    for (int c = next_chunk(chunk_ctr); c * 64 < nbodies; c = next_chunk(chunk_ctr)) {
        const int t = c * 64 + (threadIdx.x & 63);
        if (t >= nbodies) continue;
        const int item = S.bodyList[t];
        body_vel_item<NEXT, false>(S, item & 31, item >> 5, par, par_next);
    }
    if (NEXT) substep_begin_worlds(S, par_next);
}

// ------------------------------------------------------------------------------------------
// Before the substeps: movementSystem | instantMovementSystem (sim.cpp:202-254) and actionSystem
// (:270-370).  One 16-lane group per world; the world is staged in LDS only when an agent
// actually locks or grabs (needs ray casts), which scripts/benchmark.py never does.
struct PreLds { ActWorld sh[kPhysThreads / 16]; };
HSD void pre_pass(const SimState &S, PreLds &L, int wfirst) {
    constexpr int G = 16;
    ActWorld *const sh = L.sh;
    const int tid = threadIdx.x, grp = tid / G, l = tid % G;
    const int w = S.wbeg + wfirst + grp;
    const int N = S.N, A_ = S.A;
    const bool wok = wfirst + grp < S.wcnt;
    ActWorld &pw = sh[grp];
    const bool instant = (S.flags & FLAG_ZERO_AGENT_VELOCITY) == FLAG_ZERO_AGENT_VELOCITY;
    int teams = 0, step = 0;
    if (wok) { teams = S.teams[w]; step = S.curEpisodeStep[w]; }
    bool need_action = false;
    if (wok && l < A_) {
        const int agent = l;
        const bool active = team_agent_active(teams, agent) != 0;
        const int type = team_agent_type(teams, agent);
        pw.actGL[agent] = 0;
        if (active && !(type == AGENT_SEEKER && step < kNumPrepSteps - 1)) {
            int32_t *act_row = S.xAction + (w * A_ + agent) * 5;
            const int ax = act_row[0], ay = act_row[1], ar = act_row[2], ag = act_row[3], al = act_row[4];
            float fx, fy, tz;
            if (instant) { fx = 400.f * (float)(ax - 2); fy = 400.f * (float)(ay - 2); tz = 120.f * (float)(ar - 2); }
            else { fx = 12.f * (float)(ax - 5); fy = 12.f * (float)(ay - 5); tz = 3.f * (float)(ar - 5); }
            V3 f = qrot(gld4(S.brot, S, kAgentSlot0 + agent, w), {fx, fy, 0.f});
            S.aforce[(0 * kMaxAgents + agent) * N + w] = f.x; S.aforce[(1 * kMaxAgents + agent) * N + w] = f.y;
            S.aforce[(2 * kMaxAgents + agent) * N + w] = f.z; S.aforce[(3 * kMaxAgents + agent) * N + w] = tz;
            const int fl = (ag == 1 ? 1 : 0) | (al == 1 ? 2 : 0);
            pw.actGL[agent] = fl;
            need_action = fl != 0;
            act_row[0] = 2; act_row[1] = 2; act_row[2] = 2; act_row[3] = 0; act_row[4] = 0;   // sim.cpp:365-369
        }
    }
    // any lock/grab request in this pass?  (workgroup-uniform decision so the barriers below are safe)
    const int any = __syncthreads_or(need_action ? 1 : 0);
    if (!any) return;
    if (wok) {
        for (int s = l; s < kNumDSlots; s += G) {
            pw.g.meta[s] = S.bmeta[s * N + w];
#pragma unroll
            for (int c = 0; c < 3; ++c) pw.g.pos[s][c] = S.bpos[(c * kNumDSlots + s) * N + w];
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
            for (int c = 0; c < kGrabWords; ++c) pw.grabData[i][c] = S.grabData[(c * kMaxAgents + i) * N + w];
            if (i >= A_) pw.actGL[i] = 0;
        }
        if (l == 0) { pw.g.numWalls = nw; pw.g.numPlanes = npl; pw.teams = teams; }
    }
    __syncthreads();
    if (wok && l == 0) action_system(pw, A_);
    __syncthreads();
    if (wok) {
        for (int s = l; s < kNumDSlots; s += G) S.bmeta[s * N + w] = pw.g.meta[s];
        for (int i = l; i < kMaxAgents; i += G) {
            S.grabOther[i * N + w] = pw.grabOther[i];
#pragma unroll
            for (int c = 0; c < kGrabWords; ++c) S.grabData[(c * kMaxAgents + i) * N + w] = pw.grabData[i][c];
        }
    }
}

HSD void phase_pre(const SimState &S, PreLds &L) {
    for (int wfirst = 0; wfirst < S.wcnt; wfirst += kPhysThreads / 16) {
        pre_pass(S, L, wfirst);
        __syncthreads();
    }
}

// After the substeps: agentZeroVelSystem (sim.cpp:258-268), rewardsVisSystem (:763-804),
// outputRewardsDonesSystem (:806-841), updateEpisodeResultsSystem (:843-893).
constexpr int kPostLanes = 8;           // lanes per world in phase_post: the workgroup's 64 worlds in one pass
struct PostLds { WorldGeom sh[kPhysThreads / kPostLanes]; int seen_flag[kPhysThreads / kPostLanes]; };
HSD void post_pass(const SimState &S, PostLds &L, int wfirst) {
    constexpr int G = kPostLanes;
    WorldGeom *const sh = L.sh;
    int *const seen_flag = L.seen_flag;
    const int tid = threadIdx.x, grp = tid / G, l = tid % G;
    const int w = S.wbeg + wfirst + grp;
    const int N = S.N, A_ = S.A;
    const bool wok = wfirst + grp < S.wcnt;
    WorldGeom &g = sh[grp];
    const bool instant = (S.flags & FLAG_ZERO_AGENT_VELOCITY) == FLAG_ZERO_AGENT_VELOCITY;
    int teams = 0, step = 0, counts = 0;
    if (wok) {
        teams = S.teams[w]; step = S.curEpisodeStep[w]; counts = S.counts[w];
        for (int s = l; s < kNumDSlots; s += G) {
            g.meta[s] = S.bmeta[s * N + w];
#pragma unroll
            for (int c = 0; c < 3; ++c) g.pos[s][c] = S.bpos[(c * kNumDSlots + s) * N + w];
#pragma unroll
            for (int c = 0; c < 4; ++c) g.rot[s][c] = S.brot[(c * kNumDSlots + s) * N + w];
        }
        const int nw = S.numWalls[w], npl = S.numPlanes[w];
        for (int k = l; k < nw; k += G) {
#pragma unroll
            for (int c = 0; c < 4; ++c) g.wall[k][c] = S.walls[(c * kMaxWalls + k) * N + w];
        }
        for (int p = l; p < npl; p += G) {
#pragma unroll
            for (int c = 0; c < 4; ++c) g.plane[p][c] = S.planes[(c * kMaxPlanes + p) * N + w];
        }
        if (l == 0) { g.numWalls = nw; g.numPlanes = npl; seen_flag[grp] = 0; }
        if (instant && l < kMaxAgents && S.bmeta[(kAgentSlot0 + l) * N + w] != 0) {
            const int slot = kAgentSlot0 + l;
            S.blin[bidx(S, 0, slot, w)] = 0.f; S.blin[bidx(S, 1, slot, w)] = 0.f;
            S.blin[bidx(S, 2, slot, w)] = fminf(S.blin[bidx(S, 2, slot, w)], 0.f);
            gst3(S.bang, S, slot, w, V3{0.f, 0.f, 0.f});
        }
    }
    __syncthreads();
    // The seen flag feeds the reward (from episode step 95 on) and the episode result (from 96 on); k_reset overwrites
    // hiderTeamReward every step, so during the preparation phase the rays would change nothing anyone can read.
    for (int pr = l; wok && step >= kNumPrepSteps - 1 && pr < 9; pr += G) {      // (seeker, hider) pairs
        const int si = pr / 3, hi_ = pr % 3;
        if (si < cnt_seekers(counts) && hi_ < cnt_hiders(counts)) {
            const int ss = kAgentSlot0 + team_seeker(teams, si), hs_ = kAgentSlot0 + team_hider(teams, hi_);
            const V3 spos = geom_pos(g, ss);
            const V3 fwd = qrot(geom_rot(g, ss), {0.f, 1.f, 0.f});
            V3 to = geom_pos(g, hs_) - spos;
            float c = dot(normalize(to), fwd);
            if (!(c < kCosFovHalf)) {
                float t;
                if (trace_ray(g, spos, to, 1.f, &t) == hs_) seen_flag[grp] = 1;   // every writer stores the same value
            }
        }
    }
    __syncthreads();
    if (!wok) return;
    float hider_reward = S.hiderTeamReward[w];
    if (seen_flag[grp]) hider_reward = -1.f;
    if (l < A_ && team_agent_active(teams, l)) {
        const int agent = l, slot = kAgentSlot0 + agent, row = w * A_ + agent;
        if (step == 0) S.xDone[row] = 0;
        if (step < kNumPrepSteps - 1) {
            S.xReward[row] = 0.f;
        } else {
            if (step == kEpisodeLen - 1) S.xDone[row] = 1;
            float r = hider_reward;
            if (team_agent_type(teams, agent) == AGENT_SEEKER) r *= -1.f;
            if (fabsf(g.pos[slot][0]) >= 18.f || fabsf(g.pos[slot][1]) >= 18.f) r -= 10.f;
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

HSD void phase_post(const SimState &S, PostLds &L) {
    for (int wfirst = 0; wfirst < S.wcnt; wfirst += kPhysThreads / kPostLanes) {
        post_pass(S, L, wfirst);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// The phases share one LDS allocation (each uses it between two workgroup barriers).
union PhysLds { DetectLds det; SatLds sat; PreLds pre; PostLds post; };

__global__ void __launch_bounds__(kPhysThreads, 2) k_physics(SimState S) {
    __shared__ PhysLds lds;
    __shared__ int chunk_ctr[2];           // next slot-major chunk of phase_body_pos / phase_body_vel
    __shared__ int integ_scratch[kNumDSlots + 2];   // chunk offsets of the compact body list
    __shared__ int list_len[2 * 4];        // work-list lengths (sat box, wall bodies, ddw, sat ramp) x substep parity
    // this workgroup's worlds, and its slices of the work lists / list-length counters
    S.wbeg = blockIdx.x * kPhysWorlds;
    S.wcnt = min(kPhysWorlds, S.N - S.wbeg);
    S.satList += (size_t)S.wbeg * (kMaxDDCand + kMaxSCand);
    S.ddwList += (size_t)S.wbeg * 2;
    S.wallList += (size_t)S.wbeg * kNumDSlots;
    S.bodyList += (size_t)S.wbeg * kNumDSlots;
    // the list lengths live in LDS: every phase starts by reading one, and an L2 round trip there is pure latency
    if (threadIdx.x < 8) list_len[threadIdx.x] = 0;
    S.counters = list_len;
    const int NS = kAgentSlot0 + S.A;                 // body slots in use
    const int ngroups = gridDim.x;
    if (S.stepPar >= 0 && threadIdx.x == 0) {          // clear the next step's half of the finish list
        const int pn = S.stepPar ^ 1;
        S.doneList[pn * ngroups + blockIdx.x] = -1;
        if (blockIdx.x == 0) { S.doneTickets[pn] = 0; S.startedCount[pn] = 0; }
        __hip_atomic_fetch_add(&S.startedCount[S.stepPar], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef HS_PHASE_TIMING
    // development aid: wall-clock ticks (100 MHz) per phase of every workgroup -> S.phaseTicks[workgroup][10]
    long long tk = wall_clock64(); long long acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define HS_TICK(i) { __syncthreads(); const long long now_ = wall_clock64(); acc[i] += now_ - tk; tk = now_; }
#else
#define HS_TICK(i)
#endif
    phase_pre(S, lds.pre);
    HS_TICK(0)
    int nbodies = 0;
    for (int sub = 0; sub < kNumSubsteps; ++sub) {
        const int par = sub & 1;
        if (sub == 0) nbodies = phase_integrate(S, NS, par, integ_scratch);
        HS_TICK(1)
        if (threadIdx.x == 0) { chunk_ctr[0] = 0; chunk_ctr[1] = 0; }
        phase_detect(S, lds.det, NS, par);
        HS_TICK(2)
        phase_sat(S, lds.sat, par);
        __syncthreads();
        HS_TICK(3)
        phase_dd<true>(S, par);
        __syncthreads();
        HS_TICK(4)
        phase_body_pos(S, nbodies, par, &chunk_ctr[0]);
        __syncthreads();
        HS_TICK(5)
        phase_dd<false>(S, par);
        __syncthreads();
        HS_TICK(6)
        if (sub + 1 < kNumSubsteps) phase_body_vel<true>(S, nbodies, par, par ^ 1, &chunk_ctr[1]);
        else phase_body_vel<false>(S, nbodies, par, 0, &chunk_ctr[1]);
        __syncthreads();
        HS_TICK(7)
    }
    phase_post(S, lds.post);
    HS_TICK(8)
    // resetSystem for the workgroup's worlds (one wave; the level generator diverges per world anyway): step
    // counter, or a whole new level on the 240th step / on request
    if (threadIdx.x < S.wcnt) reset_world(S, S.wbeg + threadIdx.x);
    // Publish this workgroup's worlds to k_observe, which runs beside this kernel and takes finished groups in
    // the order of this list: every wave's stores are drained by the barrier, then one lane releases at agent
    // scope (the XCDs' L2s are not coherent with each other) and appends the group.
    if (S.stepPar >= 0) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const int ticket = atomicAdd(&S.doneTickets[S.stepPar], 1);
            __hip_atomic_store(&S.doneList[S.stepPar * ngroups + ticket], (int)blockIdx.x, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#ifdef HS_PHASE_TIMING
    if (threadIdx.x == 0) for (int i = 0; i < 10; ++i) S.phaseTicks[blockIdx.x * 10 + i] += acc[i];
#endif
#undef HS_TICK
}

// Holds the stream of k_observe back until every workgroup of k_physics has started, i.e. holds a CU: the spinning
// k_observe workgroups that follow can then never keep a physics workgroup from being placed.  One wave.
__global__ void __launch_bounds__(64) k_gate(SimState S, int ngroups) {
    if (threadIdx.x != 0) return;
    for (int spin = 0; spin < (1 << 22); ++spin) {
        if (__hip_atomic_load(&S.startedCount[S.stepPar], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= ngroups) return;
        __builtin_amdgcn_s_sleep(8);
    }
    S.status[2] = 2; *S.hostFlag = 1;
}

}  // namespace hs
