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

constexpr int kRaysPerAgent = 46;                          // 30 lidar + 16 visibility targets
// NT threads = one lane per ray of A agents, rounded up to whole waves: the per-ray / per-agent LDS is sized by NT
constexpr int obs_max_agents(int nt) { return nt / kRaysPerAgent < kMaxAgents ? nt / kRaysPerAgent : kMaxAgents; }
constexpr int kMaxPairs = 1536;

template <int NT>
struct ObsShared {
    static constexpr int kMaxRays = NT, kAgents = obs_max_agents(NT);
    WorldGeom g;
    float lin[kNumDSlots][3];
    float ang[kNumDSlots][3];
    int grab[kMaxAgents];
    float rayO[kMaxRays][3], rayD[kMaxRays][3];
    unsigned long long rayKey[kMaxRays];
    unsigned short pairs[kMaxPairs];                       // ray << 5 | body slot
    int nPairs;
    float lidarSin[30], lidarCos[30];                      // hs_sincosf of the 30 lidar angles, once per workgroup
    // per (agent, body slot): origin - body position and |.|^2 - bounding radius^2, shared by the agent's 46 rays
    float rel[kAgents][kNumDSlots][4];
};

HSD void store_posvel(float *o, V3 p, V3 e, V3 l, V3 a) {
    o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = e.x; o[4] = e.y; o[5] = e.z;
    o[6] = l.x; o[7] = l.y; o[8] = l.z; o[9] = a.x; o[10] = a.y; o[11] = a.z;
}

// Cooperative load of one world's geometry from the SoA columns into LDS.
template <int NT>
HSD void stage_world(const SimState &S, int w, int ps, ObsShared<NT> &sh, int tid) {
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
    const int nw = S.numWalls[w], np = S.numPlanes[w];
    if (tid == 0) { sh.g.numWalls = nw; sh.g.numPlanes = np; sh.nPairs = 0; }
    for (int i = tid; i < nw * 4; i += NT) {
        int c = i / nw, k = i % nw;
        sh.g.wall[k][c] = S.walls(c * kMaxWalls + k, ps);
    }
    for (int i = tid; i < np * 4; i += NT) {
        int c = i / np, p = i % np;
        sh.g.plane[p][c] = S.planes(c * kMaxPlanes + p, ps);
    }
    for (int i = tid; i < kMaxAgents; i += NT) sh.grab[i] = S.grabOther(i, ps);
}

HSD unsigned long long ray_key(float t, int id) { return ((unsigned long long)__float_as_uint(t) << 32) | (unsigned)id; }
constexpr unsigned kKeyMiss = 0xffffffffu;

// NT = threads per world = A*46 rays rounded up to whole waves (192 for the 4-agent benchmark): every
// lane of pass 1 / pass 3 has a ray.
template <int NT>
__global__ void __launch_bounds__(NT) k_observe(SimState S) {
    __shared__ ObsShared<NT> sh;
    const int tid = threadIdx.x;
#ifdef HS_PHASE_TIMING
    long long otk = wall_clock64();
    long long *const oacc = S.phaseTicks + (size_t)10 * ((S.N + kTile - 1) / kTile) + (blockIdx.x & 1023) * 16;
#define HS_OTICK(i) { const long long now_ = wall_clock64(); if ((tid & 63) == 0) atomicAdd((unsigned long long *)&oacc[i], (unsigned long long)(now_ - otk)); otk = now_; }
#else
#define HS_OTICK(i)
#endif
    // blocks b, b + 8, ..., b + 56 (same XCD under round-robin placement) take the 8 worlds of one octet
    const int blk = blockIdx.x;
    int oct = ((blk >> 6) << 3) + (blk & 7);
    if (oct >= (S.N + kTile - 1) / kTile) return;        // (the grid covers whole groups of 8 octets)
    if (S.stepPar >= 0) {
        // Dependency schedule (hideseek.hip launch_step): `oct` counts finished octets — this workgroup takes a world
        // of the oct-th octet whose physics wave FINISHES, and waits for it: one relaxed poll loop by one lane (bounded,
        // so that a bug cannot hang the GPU), one agent-scope acquire, then the barrier.  Every physics wave is
        // resident before this kernel starts (k_gate) and never waits for anything, so the wait always ends.
        __shared__ int sh_oct;
        const int noct = (S.N + kTile - 1) / kTile;
        if (tid == 0) {
            const int *slot = &S.doneList[S.stepPar * noct + oct];
            int v = -1;
            for (int spin = 0; spin < (1 << 22); ++spin) {
                v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v >= 0) break;
                __builtin_amdgcn_s_sleep(16);
            }
            if (v < 0) { S.status[2] = 1; *S.hostFlag = 1; }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            sh_oct = v;
        }
        __syncthreads();
        oct = sh_oct;
        if (oct < 0) return;
    }
    const int p = oct * kTile + ((blk >> 3) & 7);       // slot in the tiled columns
    const int w = S.worldOfSlot[p];                      // the world that lives there (exports, per-world scalars)
    if (w < 0) return;
    const int A = S.A;
    stage_world<NT>(S, w, p, sh, tid);
    if (tid < 30) {       // lidarSystem angles (sim.cpp:727-738): the same 30 values for every agent
        float theta = 2.f * kPi * ((float)tid / 30.f) + kPi / 2.f;
        hs_sincosf(theta, &sh.lidarSin[tid], &sh.lidarCos[tid]);
    }
    const int counts = S.counts[w];
    const int teams = S.teams[w];
    const int step = S.curEpisodeStep[w];
    __syncthreads();
    const WorldGeom &g = sh.g;
    const int nAgents = cnt_agents(counts), nBoxes = cnt_boxes(counts), nRamps = cnt_ramps(counts);
    const int nRays = A * kRaysPerAgent;
    HS_OTICK(0)
    for (int item = tid; item < nAgents * kNumDSlots; item += NT) {
        const int i = item / kNumDSlots, b = item % kNumDSlots;
        const V3 mo = geom_pos(g, kAgentSlot0 + i) - geom_pos(g, b);
        const int m = g.meta[b];
        sh.rel[i][b][0] = mo.x; sh.rel[i][b][1] = mo.y; sh.rel[i][b][2] = mo.z;
        sh.rel[i][b][3] = dot(mo, mo) - obj_bound_r2(meta_obj(m));
    }
    __syncthreads();
    HS_OTICK(1)

    // ---------------- pass 1: ray setup, walls + planes, cull against the movable hulls ----------------
    for (int r = tid; r < nRays; r += NT) {
        const int i = r / kRaysPerAgent, k = r % kRaysPerAgent;
        sh.rayKey[r] = ray_key(-1.f, kKeyMiss);               // "no ray" (visibility ray not cast)
        if (i >= nAgents) continue;
        const int slot = kAgentSlot0 + i;
        const Q rot = geom_rot(g, slot);
        const V3 o = geom_pos(g, slot);
        const V3 fwd = qrot(rot, {0.f, 1.f, 0.f});
        V3 d; float tmax;
        if (k < 30) {
            // lidarSystem: 30 rays in the agent's horizontal plane, t_max 200 (sim.cpp:727-738)
            const V3 right = qrot(rot, {1.f, 0.f, 0.f});
            const float s = sh.lidarSin[k], c = sh.lidarCos[k];
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
        sh.rayO[r][0] = o.x; sh.rayO[r][1] = o.y; sh.rayO[r][2] = o.z;
        sh.rayD[r][0] = d.x; sh.rayD[r][1] = d.y; sh.rayD[r][2] = d.z;
        HS_OTICK(2)
        // static geometry: same order and arithmetic as trace_ray
        int hit = -1; float best = tmax;
        const V3 inv = {1.f / d.x, 1.f / d.y, 1.f / d.z};
        const int nw = g.numWalls;
        const WallZ wz = ray_wall_z(o.z, d.z, inv.z);          // the z slab is the same for every wall
        // (a ray with an exactly zero x or y component takes the general form; decided per wave)
        if (__ballot(d.x == 0.f || d.y == 0.f) == 0) {
            WallScan ws(tmax, wz);                              // (t_max is 200 or 1: positive, so +1 ulp is the next float up)
            for (int q = 0; q < nw; ++q)
                ws.wall(o.x - g.wall[q][0], o.y - g.wall[q][1], inv, g.wall[q][2], g.wall[q][3], kHitWallBase + q);
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
        // movable hulls: conservative cull here, exact test in pass 2
        const float dd2 = dot(d, d);
        for (int b = 0; b < kNumDSlots; ++b) {
            const int m = g.meta[b];
            if (m == 0) continue;
            const int obj = meta_obj(m);
            const V3 mo = {sh.rel[i][b][0], sh.rel[i][b][1], sh.rel[i][b][2]};
            const float bb = dot(mo, d), cc = sh.rel[i][b][3];
            if (cc > 0.f && (bb > 0.f || bb * bb < dd2 * cc * 0.999f)) continue;
            const int slotp = atomicAdd(&sh.nPairs, 1);
            if (slotp < kMaxPairs) {
                sh.pairs[slotp] = (unsigned short)((r << 5) | b);
            } else {
                // pair list full: test in place
                Q qi = qinv(geom_rot(g, b));
                V3 ol = qrot(qi, mo), dl = qrot(qi, d);
                float t = obj == OBJ_RAMP ? ray_wedge_local(ol, dl) : ray_box_local(ol, dl, obj_half_extents(obj));
                if (t >= 0.f && t <= tmax) { unsigned long long kk = ray_key(t, b); key = kk < key ? kk : key; }
            }
        }
        sh.rayKey[r] = key;
    }
    __syncthreads();
    HS_OTICK(5)
    // ---------------- pass 2: exact ray-vs-hull tests, one thread per surviving pair ----------------
    {
        const int np2 = sh.nPairs < kMaxPairs ? sh.nPairs : kMaxPairs;
        for (int p = tid; p < np2; p += NT) {
            const int pr = sh.pairs[p];
            const int r = pr >> 5, b = pr & 31;
            const int obj = meta_obj(g.meta[b]);
            const V3 o = {sh.rayO[r][0], sh.rayO[r][1], sh.rayO[r][2]};
            const V3 d = {sh.rayD[r][0], sh.rayD[r][1], sh.rayD[r][2]};
            const float tmax = (r % kRaysPerAgent) < 30 ? 200.f : 1.f;
            Q qi = qinv(geom_rot(g, b));
            V3 ol = qrot(qi, o - geom_pos(g, b)), dl = qrot(qi, d);
            float t;
            if (__ballot(obj != OBJ_RAMP && (dl.x == 0.f || dl.y == 0.f || dl.z == 0.f)) == 0)
                t = obj == OBJ_RAMP ? ray_wedge_local(ol, dl) : ray_box_local_nz(ol, dl, obj_half_extents(obj));
            else
                t = obj == OBJ_RAMP ? ray_wedge_local(ol, dl) : ray_box_local(ol, dl, obj_half_extents(obj));
            if (t >= 0.f && t <= tmax) atomicMin(&sh.rayKey[r], ray_key(t, b));
        }
    }
    __syncthreads();
    HS_OTICK(6)
    // ---------------- pass 3: ray results -> exported columns ----------------
    for (int r = tid; r < nRays; r += NT) {
        const int i = r / kRaysPerAgent, k = r % kRaysPerAgent;
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
    const int nObs = A * 17;
    for (int item = tid; item < nObs + 1; item += NT) {
        if (item < nObs) {
            const int i = item / 17, e = item % 17;
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
