// Observation kernel: collectObservationsSystem (src/sim.cpp:448-565), computeVisibilitySystem CPU
// branch (:567-605,663-708), lidarSystem (:712-759), globalPositionsDebugSystem (:895-941).
//
// One workgroup per world (a lane per ray); the world's poses, velocities and static geometry are staged once
// into LDS.  The 8 workgroups of an octet (hs_state.h) read the same contiguous blocks of the tiled columns, so
// the block index is mapped to the world such that they land on the same XCD (workgroups are dealt round-robin
// over the 8 XCDs, each with its own L2) and next to each other in time: the octet's blocks come from HBM once.
// Every (agent, entity) item writes its own 12-17 floats of an observation row: the rows of a world are adjacent
// in the exported tensors, so the world's stores fill whole cache lines in L2 before they leave for HBM (measured:
// HBM-side traffic 1.06 x the algorithmic bytes; assembling the rows in LDS first gave the same traffic and cost
// 9 % more time).
//
// Rays (A*30 lidar + A*16 visibility) are cast in two passes so that the expensive, divergent part
// runs on full waves:
//   1. one thread per ray: set the ray up, test it against the axis-aligned walls and the planes
//      (uniform loops), and run a conservative bounding-sphere cull against the <= 17 movable hulls;
//      every surviving (ray, hull) pair is appended to a pair list in LDS;
//   2. one thread per PAIR: exact ray-vs-hull test, result merged into the ray's
//      (t, body id) key with a 64-bit LDS atomic min — lexicographic (t, id) order is exactly the
//      "closest hit, ties keep the lower id" rule of the sequential trace_ray.
// The reference's GPU branch uses a 32-lane warp per agent with 17/30 lanes busy.
#pragma once
#include "hs_state.h"
#include "hs_rays.h"

namespace hs {

// Timing probes (development aid, results are wrong when set): -DHS_OBS_SKIP=<bits> leaves sections of k_observe out:
// 1 walls, 2 hull cull (no pairs), 4 exact hull tests, 8 observation rows, 16 ray results, 32 everything after staging.
#ifndef HS_OBS_SKIP
#define HS_OBS_SKIP 0
#endif

// waves per SIMD the register allocation of k_observe aims at (8 needs 48 bytes of scratch per lane: 1 % faster, but the
// spills show up as 60 % more HBM write traffic)
#ifndef HS_OBS_WAVES
#define HS_OBS_WAVES 7
#endif

constexpr int kRaysPerAgent = 46;                          // 30 lidar + 16 visibility targets
// Lane layout of the rays: the A*30 lidar rays first (ray = agent * 30 + k), the A*16 visibility rays from the next
// wave boundary on (ray = base + agent * 16 + target): a wave runs one kind of ray set-up.
constexpr int obs_vis_base(int a) { return (a * 30 + 63) / 64 * 64; }
constexpr int obs_threads(int a) { return (obs_vis_base(a) + a * 16 + 63) / 64 * 64; }       // 128, 128, 192, 192, 320, 320
constexpr int obs_max_agents(int nt) { int a = 1; while (a < kMaxAgents && obs_threads(a + 1) <= nt) ++a; return a; }
struct RayId { int agent, k; };                            // k < 30: lidar ray k; else visibility target k - 30
HSD RayId ray_id(int r, int visBase) {
    if (r < visBase) { const int i = r / 30; return {i, r - i * 30}; }
    const int rr = r - visBase;
    return {rr >> 4, 30 + (rr & 15)};
}

template <int NT>
struct ObsShared {
    static constexpr int kMaxRays = NT, kAgents = obs_max_agents(NT);
    WorldGeom g;
    float lin[kNumDSlots][3];
    float ang[kNumDSlots][3];
    int grab[kMaxAgents];
    float rayD[kMaxRays][3];
    unsigned long long rayKey[kMaxRays];
    // (ray, hull) pairs that survive the bounding-sphere cull, ray << 5 | body slot: every wave of pass 1 fills its own
    // region, sized for all its rays against every other box-shaped hull (9 boxes + 5 agents) / every ramp
    static constexpr int kWaves = NT / 64, kWavePairs = 64 * (kMaxBoxes + kMaxAgents - 1), kWaveRampPairs = 64 * kMaxRamps;
    unsigned short pairs[kWaves][kWavePairs];
    unsigned short rampPairs[kWaves][kWaveRampPairs];
    int nPairs[kWaves], nRampPairs[kWaves];
    unsigned present;                                      // bit b: body slot b exists
    // per (agent, body slot), shared by the agent's 46 rays: origin - body position, |.|^2 - bounding radius^2, and
    // the origin in the body's frame
    alignas(16) float rel[kAgents][kNumDSlots][8];
    float fwd[kAgents][3], right[kAgents][3];              // the agents' forward / right axes
};

HSD void store_posvel(float *o, V3 p, V3 e, V3 l, V3 a) {
    o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = e.x; o[4] = e.y; o[5] = e.z;
    o[6] = l.x; o[7] = l.y; o[8] = l.z; o[9] = a.x; o[10] = a.y; o[11] = a.z;
}

// Cooperative load of one world's geometry from the SoA columns into LDS.
template <int NT>
HSD void stage_world(const SimState &S, int ps, ObsShared<NT> &sh, int tid) {
    for (int i = tid; i < kNumDSlots; i += NT) sh.g.meta[i] = S.bmeta(i, ps);
    for (int i = tid; i < kNumDSlots * 3; i += NT) {
        int c = i / kNumDSlots, s = i % kNumDSlots;
        sh.g.pos[s][c] = S.bpos(c * kNumDSlots + s, ps);
        sh.lin[s][c] = S.blin(c * kNumDSlots + s, ps);
        sh.ang[s][c] = S.bang(c * kNumDSlots + s, ps);
    }
    for (int i = tid; i < kNumDSlots * 4; i += NT) {
        int c = i / kNumDSlots, s = i % kNumDSlots;
        sh.g.rot[s][c] = S.brot(c * kNumDSlots + s, ps);
    }
    // (all 36 wall rows and 3 plane rows are fetched whatever the counts are: no load waits for another one)
    for (int i = tid; i < 4 * kMaxWalls; i += NT) sh.g.wall[i % kMaxWalls][i / kMaxWalls] = S.walls(i, ps);
    for (int i = tid; i < 4 * kMaxPlanes; i += NT) sh.g.plane[i % kMaxPlanes][i / kMaxPlanes] = S.planes(i, ps);
    for (int i = tid; i < kMaxAgents; i += NT) sh.grab[i] = S.grabOther(i, ps);
    if (tid < NT / 64) { sh.nPairs[tid] = 0; sh.nRampPairs[tid] = 0; }
}

HSD unsigned long long ray_key(float t, int id) { return ((unsigned long long)__float_as_uint(t) << 32) | (unsigned)id; }
constexpr unsigned kKeyMiss = 0xffffffffu;

// NT = threads per world = obs_threads(A) (192 for the 4-agent benchmark): a lane per ray.
template <int NT>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(HS_OBS_WAVES, HS_OBS_WAVES))) k_observe(SimState S) {
    __shared__ ObsShared<NT> sh;
    const int tid = threadIdx.x;
#ifdef HS_PHASE_TIMING
    long long otk = wall_clock64();
    long long *const oacc = S.phaseTicks + phase_ticks_obs_base(S.N) + (blockIdx.x & 1023) * 16;
#define HS_OTICK(i) { const long long now_ = wall_clock64(); if ((tid & 63) == 0) atomicAdd((unsigned long long *)&oacc[i], (unsigned long long)(now_ - otk)); otk = now_; }
#else
#define HS_OTICK(i)
#endif
    // blocks b, b + 8, ..., b + 56 (same XCD under round-robin placement) take the 8 worlds of one octet
    const int blk = blockIdx.x;
    const int oct = ((blk >> 6) << 3) + (blk & 7);
    if (oct >= (S.N + kTile - 1) / kTile) return;        // (the grid covers whole groups of 8 octets)
    const int p = oct * kTile + ((blk >> 3) & 7);       // slot in the tiled columns
    // the world that lives there and its scalars (exports are indexed by world id): one load, issued beside the
    // column loads
    const int4 hdr = S.slotHdr[p];
    const int A = S.A;
    const int visBase = obs_vis_base(A);
    // (this lane's lidar angle, fetched beside the column loads as well)
    const RayId myRay = ray_id(tid, visBase);
    float lidarS = 0.f, lidarC = 0.f;
    if (tid < visBase) { lidarS = S.lidarSinCos[myRay.k]; lidarC = S.lidarSinCos[30 + myRay.k]; }
    stage_world<NT>(S, p, sh, tid);
    const int w = hdr.x;
    if (w < 0) return;                                   // (empty slot of the last octet)
    if (tid == 0) { sh.g.numWalls = hdr.y & 255; sh.g.numPlanes = (hdr.y >> 8) & 255; sh.present = 0; }
    const int counts = hdr.z;
    const int teams = hdr.w;
    const int step = hdr.y >> 16;
    __syncthreads();
    const WorldGeom &g = sh.g;
    const int nAgents = cnt_agents(counts), nBoxes = cnt_boxes(counts), nRamps = cnt_ramps(counts);
    if (HS_OBS_SKIP & 32) return;
    HS_OTICK(0)
    for (int item = tid; item < nAgents * kNumDSlots; item += NT) {
        const int i = item / kNumDSlots, b = item % kNumDSlots;
        const int m = g.meta[b];
        if (m == 0) continue;
        if (i == 0) atomicOr(&sh.present, 1u << b);
        const V3 mo = geom_pos(g, kAgentSlot0 + i) - geom_pos(g, b);
        const V3 ol = qrot(qinv(geom_rot(g, b)), mo);
        float *e = sh.rel[i][b];
        e[0] = mo.x; e[1] = mo.y; e[2] = mo.z;
        e[3] = dot(mo, mo) - obj_bound_r2(meta_obj(m));
        e[4] = ol.x; e[5] = ol.y; e[6] = ol.z;
    }
    for (int i = tid - (NT - 8); i >= 0 && i < nAgents; i += NT) {          // (the last lanes: they have no table item)
        const Q rot = geom_rot(g, kAgentSlot0 + i);
        const V3 f = qrot(rot, {0.f, 1.f, 0.f}), rt = qrot(rot, {1.f, 0.f, 0.f});
        sh.fwd[i][0] = f.x; sh.fwd[i][1] = f.y; sh.fwd[i][2] = f.z;
        sh.right[i][0] = rt.x; sh.right[i][1] = rt.y; sh.right[i][2] = rt.z;
    }
    __syncthreads();
    HS_OTICK(1)

    // ---------------- pass 1: ray setup, walls + planes, cull against the movable hulls ----------------
    for (int r = tid; r < NT; r += NT) {                       // (one trip: a lane per ray)
        const int i = myRay.agent, k = myRay.k;
        sh.rayKey[r] = ray_key(-1.f, kKeyMiss);               // "no ray" (visibility ray not cast)
        if (i >= nAgents) continue;                            // (also the padding lanes between the two kinds)
        const int slot = kAgentSlot0 + i;
        const V3 o = geom_pos(g, slot);
        const V3 fwd = {sh.fwd[i][0], sh.fwd[i][1], sh.fwd[i][2]};
        V3 d; float tmax;
        if (k < 30) {
            // lidarSystem: 30 rays in the agent's horizontal plane, t_max 200 (sim.cpp:727-738)
            const V3 right = {sh.right[i][0], sh.right[i][1], sh.right[i][2]};
            const float s = lidarS, c = lidarC;
            d = normalize(right * c + fwd * s);
            tmax = 200.f;
        } else {
            // checkVisibility (sim.cpp:586-605): FOV cone, then a segment ray to the target's origin
            const int e = k - 30;
            int tslot; bool present;
            if (e < kMaxBoxes) { tslot = kBoxSlot0 + e; present = e < nBoxes; }
            else if (e < kMaxBoxes + kMaxRamps) { tslot = kRampSlot0 + (e - kMaxBoxes); present = (e - kMaxBoxes) < nRamps; }
            else { const int jj = e - kMaxBoxes - kMaxRamps; const int j = jj < i ? jj : jj + 1; tslot = kAgentSlot0 + j; present = j < nAgents; }
            if (!present) continue;
            d = geom_pos(g, tslot) - o;
            if (dot(normalize(d), fwd) < kCosFovHalf) continue;
            tmax = 1.f;
        }
        sh.rayD[r][0] = d.x; sh.rayD[r][1] = d.y; sh.rayD[r][2] = d.z;
        HS_OTICK(2)
        // static geometry: same order and arithmetic as trace_ray
        int hit = -1; float best = tmax;
        const V3 inv = {1.f / d.x, 1.f / d.y, 1.f / d.z};
        const int nw = (HS_OBS_SKIP & 1) ? 0 : g.numWalls;
        const WallZ wz = ray_wall_z(o.z, d.z, inv.z);          // the z slab is the same for every wall
        // (a ray with an exactly zero x or y component takes the general form; decided per wave)
        if (__ballot(d.x == 0.f || d.y == 0.f) == 0) {
            WallScan ws(tmax, wz, o.x, o.y, inv);              // (t_max is 200 or 1: positive, so +1 ulp is the next float up)
            for (int q = 0; q < nw; ++q) {
                const f32x2 *wq = reinterpret_cast<const f32x2 *>(g.wall[q]);
                ws.wall(wq[0], wq[1], kHitWallBase + q);
            }
            ws.finish(tmax);
            best = ws.best; hit = ws.hit;
        } else {
            for (int q = 0; q < nw; ++q) {
                float t = ray_wall_xy(o.x - g.wall[q][0], o.y - g.wall[q][1], d, inv, g.wall[q][2], g.wall[q][3], wz);
                if (t >= 0.f && t <= best && (hit < 0 || t < best)) { best = t; hit = kHitWallBase + q; }
            }
        }
        HS_OTICK(3)
        const int np = g.numPlanes;
        for (int p = 0; p < np; ++p) {
            V3 n = {g.plane[p][0], g.plane[p][1], g.plane[p][2]};
            float dn = dot(n, d);
            if (!(dn < 0.f)) continue;
            float dist = dot(n, o) - g.plane[p][3];
            if (dist < 0.f) continue;
            float t = -dist / dn;
            if (t >= 0.f && t <= best && (hit < 0 || t < best)) { best = t; hit = kHitPlaneBase + p; }
        }
        HS_OTICK(4)
        unsigned long long key = hit < 0 ? ray_key(tmax, kKeyMiss) : ray_key(best, hit);
        // movable hulls: conservative bounding-sphere cull here, exact test in pass 2.  The loop over the bodies is
        // uniform, so the survivors of one body are a lane mask (ballot) and their pairs go to consecutive entries of
        // this wave's region of the pair list: no atomics.  The agent's own hull is left out: its centre is the ray's
        // origin (origin - centre is exactly 0), and a ray that starts inside a hull never hits it (every slab's entry
        // is negative).
        const float dd2 = dot(d, d);
        const unsigned others = __builtin_amdgcn_readfirstlane(sh.present) & ((HS_OBS_SKIP & 2) ? 0u : ~0u);
        const float *relI = &sh.rel[i][0][0];
        constexpr unsigned kRampBits = ((1u << kMaxRamps) - 1u) << kRampSlot0;
        const int wv = tid >> 6;
        int at = 0, ar = 0;                                     // (uniform: entries of this wave's regions in use)
#pragma unroll
        for (int b = 0; b < kNumDSlots; ++b) {
            if (!((others >> b) & 1u)) continue;
            const float4 e = *reinterpret_cast<const float4 *>(relI + b * 8);
            const float bb = hs_fma(e.z, d.z, hs_fma(e.y, d.y, e.x * d.x)), cc = e.w;      // (= dot(origin - centre, d))
            // culled: cc > 0 && (bb > 0 || bb * bb < dd2 * cc * 0.999f) — as lane masks (ordered comparisons 2 = ">",
            // 4 = "<"; 32 = integer "=="), combined by scalar instructions
            unsigned long long m = __builtin_amdgcn_fcmpf(cc, 0.f, 2) &
                                   (__builtin_amdgcn_fcmpf(bb, 0.f, 2) | __builtin_amdgcn_fcmpf(bb * bb, dd2 * cc * 0.999f, 4));
            if (b >= kAgentSlot0) m |= __builtin_amdgcn_sicmp(slot, b, 32);
            m = __builtin_amdgcn_read_exec() & ~m;             // the rays that keep body b
            if (m == 0) continue;
            const bool ramp = (kRampBits >> b) & 1u;
            if (__builtin_amdgcn_inverse_ballot_w64(m)) {
                const int idx = (ramp ? ar : at) + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                (ramp ? sh.rampPairs[wv] : sh.pairs[wv])[idx] = (unsigned short)((r << 5) | b);
            }
            if (ramp) ar += __builtin_popcountll(m); else at += __builtin_popcountll(m);
        }
        if ((tid & 63) == __builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) { sh.nPairs[wv] = at; sh.nRampPairs[wv] = ar; }
        sh.rayKey[r] = key;
    }
    __syncthreads();
    HS_OTICK(5)
#ifdef HS_PHASE_TIMING
    if (tid < NT / 64) { atomicAdd((unsigned long long *)&oacc[9], (unsigned long long)sh.nPairs[tid]); atomicAdd((unsigned long long *)&oacc[10], (unsigned long long)sh.nRampPairs[tid]); }
#endif
    // ---------------- pass 2: exact ray-vs-hull tests, one thread per surviving pair ----------------
    // (box-shaped hulls and ramps from separate lists: a wave runs one kind of test)
    {
        constexpr int NW = NT / 64;
        int np2 = 0, nr2 = 0;
#pragma unroll
        for (int v = 0; v < NW; ++v) { np2 += sh.nPairs[v]; nr2 += sh.nRampPairs[v]; }
        if (HS_OBS_SKIP & 4) { np2 = 0; nr2 = 0; }
        for (int p = tid; p < np2; p += NT) {
            int q = p, v = 0;                                     // p-th pair of the concatenated regions
#pragma unroll
            for (int u = 0; u < NW - 1; ++u) { const int n = sh.nPairs[u]; if (v == u && q >= n) { q -= n; v = u + 1; } }
            const int pr = sh.pairs[v][q];
            const int r = pr >> 5, b = pr & 31;
            const int i = ray_id(r, visBase).agent;
            const float *e = sh.rel[i][b];
            const V3 ol = {e[4], e[5], e[6]};
            const V3 dl = qrot(qinv(geom_rot(g, b)), {sh.rayD[r][0], sh.rayD[r][1], sh.rayD[r][2]});
            const V3 he = obj_half_extents(meta_obj(g.meta[b]));
            const float tmax = r < visBase ? 200.f : 1.f;
            const float t = __ballot(dl.x == 0.f || dl.y == 0.f || dl.z == 0.f) == 0 ? ray_box_local_nz(ol, dl, he) : ray_box_local(ol, dl, he);
            if (t >= 0.f && t <= tmax) atomicMin(&sh.rayKey[r], ray_key(t, b));
        }
        for (int p = NT - 1 - tid; p < nr2; p += NT) {                // (the last wave first: it has the fewest box pairs)
            int q = p, v = 0;
#pragma unroll
            for (int u = 0; u < NW - 1; ++u) { const int n = sh.nRampPairs[u]; if (v == u && q >= n) { q -= n; v = u + 1; } }
            const int pr = sh.rampPairs[v][q];
            const int r = pr >> 5, b = pr & 31;
            const int i = ray_id(r, visBase).agent;
            const float *e = sh.rel[i][b];
            const V3 dl = qrot(qinv(geom_rot(g, b)), {sh.rayD[r][0], sh.rayD[r][1], sh.rayD[r][2]});
            const float tmax = r < visBase ? 200.f : 1.f;
            const float t = ray_wedge_local({e[4], e[5], e[6]}, dl);
            if (t >= 0.f && t <= tmax) atomicMin(&sh.rayKey[r], ray_key(t, b));
        }
    }
    __syncthreads();
    HS_OTICK(6)
    // ---------------- pass 3: ray results -> exported columns ----------------
    for (int r = tid; r < ((HS_OBS_SKIP & 16) ? 0 : NT); r += NT) {
        const RayId rid = ray_id(r, visBase);
        const int i = rid.agent, k = rid.k;
        if (i >= nAgents) continue;
        const int row = w * A + i;
        const unsigned long long key = sh.rayKey[r];
        const unsigned id = (unsigned)(key & 0xffffffffull);
        const float t = __uint_as_float((unsigned)(key >> 32));
        if (k < 30) {
            S.xLidar[row * 30 + k] = id == kKeyMiss ? 0.f : t;
        } else {
            const int e = k - 30;
            if (e < kMaxBoxes) {
                S.xVisBoxes[row * kMaxBoxes + e] = (id == (unsigned)(kBoxSlot0 + e)) ? 1.f : 0.f;
            } else if (e < kMaxBoxes + kMaxRamps) {
                const int rr = e - kMaxBoxes;
                S.xVisRamps[row * kMaxRamps + rr] = (id == (unsigned)(kRampSlot0 + rr)) ? 1.f : 0.f;
            } else {
                const int jj = e - kMaxBoxes - kMaxRamps;
                const int j = jj < i ? jj : jj + 1;
                const float vis = (id == (unsigned)(kAgentSlot0 + j)) ? 1.f : 0.f;
                // CPU-branch side effect (sim.cpp:700-705): all writers store the same value
                if (vis != 0.f && team_agent_type(teams, i) == AGENT_SEEKER && team_agent_type(teams, j) == AGENT_HIDER)
                    S.hiderTeamReward[w] = -1.f;
                S.xVisAgents[row * (kMaxAgents - 1) + jj] = vis;
            }
        }
    }
    HS_OTICK(7)
    // ---------------- collectObservationsSystem rows + globalPositionsDebugSystem ----------------
    // items: A*16 (agent, other entity) rows first — a whole wave of the same work for 4 agents —, then the A self rows,
    // then the debug positions
    const int nRel = A * 16, nObs = (HS_OBS_SKIP & 8) ? -1 : nRel + A;
    for (int item = tid; item < nObs + 1; item += NT) {
        if (item < nObs) {
            const int i = item < nRel ? item >> 4 : item - nRel, e = item < nRel ? 1 + (item & 15) : 0;
            if (i >= nAgents) continue;
            const int row = w * A + i;
            const int slot = kAgentSlot0 + i;
            const V3 mpos = geom_pos(g, slot);
            const Q mrot = geom_rot(g, slot);
            const V3 mlin = {sh.lin[slot][0], sh.lin[slot][1], sh.lin[slot][2]};
            const V3 mang = {sh.ang[slot][0], sh.ang[slot][1], sh.ang[slot][2]};
            const Q toF = qinv(mrot);
            if (e == 0) {
                if (step <= kNumPrepSteps) S.xPrep[row] = kNumPrepSteps - step;
                float *so = S.xSelfObs + row * 13;
                store_posvel(so, mpos, quat_to_euler(mrot), qrot(toF, mlin), qrot(toF, mang));
                so[12] = sh.grab[i] >= 0 ? 1.f : 0.f;
                continue;
            }
            int tslot; float *o; int width; bool present;
            if (e <= kMaxBoxes) {
                const int b = e - 1; tslot = kBoxSlot0 + b; width = 17; present = b < nBoxes;
                o = S.xBoxObs + (row * kMaxBoxes + b) * 17;
            } else if (e <= kMaxBoxes + kMaxRamps) {
                const int r = e - 1 - kMaxBoxes; tslot = kRampSlot0 + r; width = 14; present = r < nRamps;
                o = S.xRampObs + (row * kMaxRamps + r) * 14;
            } else {
                const int jj = e - 1 - kMaxBoxes - kMaxRamps;
                const int j = jj < i ? jj : jj + 1;
                tslot = kAgentSlot0 + j; width = 14; present = j < nAgents;
                o = S.xAgentObs + (row * (kMaxAgents - 1) + jj) * 14;
            }
            if (!present) { for (int k = 0; k < width; ++k) o[k] = 0.f; continue; }
            // computeRelativePosVelObs (sim.cpp:401-420)
            const V3 x = geom_pos(g, tslot);
            const Q q = geom_rot(g, tslot);
            const V3 lin = {sh.lin[tslot][0], sh.lin[tslot][1], sh.lin[tslot][2]};
            const V3 ang = {sh.ang[tslot][0], sh.ang[tslot][1], sh.ang[tslot][2]};
            V3 p = qrot(toF, x - mpos);
            Q qr = qnormalize(qmul(toF, q));
            store_posvel(o, p, quat_to_euler(qr), qrot(toF, lin - mlin), qrot(toF, ang - mang));
            const int m = g.meta[tslot];
            if (tslot < kAgentSlot0) {
                float *lk = o + (width - 2);
                if (tslot < kRampSlot0) {
                    const bool big = meta_obj(m) == OBJ_BOX;      // boxSizes level_gen.cpp:150,191
                    o[12] = big ? 8.f : 2.f; o[13] = big ? 1.5f : 2.f; o[14] = 2.f;
                }
                // computeLockObservation (sim.cpp:422-446)
                if (meta_resp(m) != RESP_STATIC) { lk[0] = 0.f; lk[1] = 0.f; }
                else if (meta_owner(m) == OWNER_HIDER) { lk[0] = 1.f; lk[1] = 0.f; }
                else { lk[0] = 0.f; lk[1] = 1.f; }
            } else {
                const int j = tslot - kAgentSlot0;
                o[12] = team_agent_type(teams, j) == AGENT_HIDER ? 1.f : 0.f;
                o[13] = sh.grab[j] >= 0 ? 1.f : 0.f;
            }
        } else {
            // ---- globalPositionsDebugSystem, including the reference's double-increment tail
            float *gp = S.xGlobalPos + w * 34;
            for (int b = 0; b < kMaxBoxes; ++b) {
                bool on = b < nBoxes;
                gp[b * 2] = on ? g.pos[kBoxSlot0 + b][0] : 0.f; gp[b * 2 + 1] = on ? g.pos[kBoxSlot0 + b][1] : 0.f;
            }
            for (int r = 0; r < kMaxRamps; ++r) {
                bool on = r < nRamps;
                gp[18 + r * 2] = on ? g.pos[kRampSlot0 + r][0] : 0.f; gp[18 + r * 2 + 1] = on ? g.pos[kRampSlot0 + r][1] : 0.f;
            }
            float *ga = gp + 22;
            int o = 0;
            for (int h = 0; h < cnt_hiders(counts); ++h, ++o) {
                int s = kAgentSlot0 + team_hider(teams, h);
                ga[o * 2] = g.pos[s][0]; ga[o * 2 + 1] = g.pos[s][1];
            }
            for (int k = 0; k < cnt_seekers(counts); ++k, ++o) {
                int s = kAgentSlot0 + team_seeker(teams, k);
                ga[o * 2] = g.pos[s][0]; ga[o * 2 + 1] = g.pos[s][1];
            }
            for (; o < kMaxAgents; o += 2) { ga[o * 2] = 0.f; ga[o * 2 + 1] = 0.f; }
        }
    }
    HS_OTICK(8)
#undef HS_OTICK
}

}  // namespace hs
