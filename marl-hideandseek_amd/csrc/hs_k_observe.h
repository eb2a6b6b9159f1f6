// Observation kernel: collectObservationsSystem (src/sim.cpp:448-565), computeVisibilitySystem CPU
// branch (:567-605,663-708), lidarSystem (:712-759), globalPositionsDebugSystem (:895-941).
//
// One 256-thread workgroup per world.  The world's poses, velocities and static geometry are
// staged once into LDS (coalesced dword loads from the world-fastest SoA columns); work items are
// then spread over the threads as [A*30 lidar rays][A*16 visibility rays][A*17 relative-obs
// entities][1 debug-positions item], which keeps each 64-lane wave on one kind of item.  The
// reference's GPU branch uses a 32-lane warp per agent with 17/30 lanes busy; here a
// wave64 is filled with rays of several agents of the same world instead.
#pragma once
#include "hs_state.h"
#include "hs_rays.h"

namespace hs {

struct ObsShared {
    WorldGeom g;
    float lin[kNumDSlots][3];
    float ang[kNumDSlots][3];
    int grab[kMaxAgents];
};

HSD void store_posvel(float *o, V3 p, V3 e, V3 l, V3 a) {
    o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = e.x; o[4] = e.y; o[5] = e.z;
    o[6] = l.x; o[7] = l.y; o[8] = l.z; o[9] = a.x; o[10] = a.y; o[11] = a.z;
}

// Cooperative load of one world's geometry from the SoA columns into LDS.
template <int NT>
HSD void stage_world(const SimState &S, int w, ObsShared &sh, int tid) {
    const int N = S.N;
    for (int i = tid; i < kNumDSlots; i += NT) sh.g.meta[i] = S.bmeta[i * N + w];
    for (int i = tid; i < kNumDSlots * 3; i += NT) {
        int c = i / kNumDSlots, s = i % kNumDSlots;
        sh.g.pos[s][c] = S.bpos[(c * kNumDSlots + s) * N + w];
        sh.lin[s][c] = S.blin[(c * kNumDSlots + s) * N + w];
        sh.ang[s][c] = S.bang[(c * kNumDSlots + s) * N + w];
    }
    for (int i = tid; i < kNumDSlots * 4; i += NT) {
        int c = i / kNumDSlots, s = i % kNumDSlots;
        sh.g.rot[s][c] = S.brot[(c * kNumDSlots + s) * N + w];
    }
    const int nw = S.numWalls[w], np = S.numPlanes[w];
    if (tid == 0) { sh.g.numWalls = nw; sh.g.numPlanes = np; }
    for (int i = tid; i < nw * 4; i += NT) {
        int c = i / nw, k = i % nw;
        sh.g.wall[k][c] = S.walls[(c * kMaxWalls + k) * N + w];
    }
    for (int i = tid; i < np * 4; i += NT) {
        int c = i / np, p = i % np;
        sh.g.plane[p][c] = S.planes[(c * kMaxPlanes + p) * N + w];
    }
    for (int i = tid; i < kMaxAgents; i += NT) sh.grab[i] = S.grabOther[i * N + w];
}

// checkVisibility lambda (sim.cpp:586-605)
HSD float check_visibility(const WorldGeom &g, V3 me_pos, V3 fwd, int slot) {
    V3 to = geom_pos(g, slot) - me_pos;
    float c = dot(normalize(to), fwd);
    if (c < kCosFovHalf) return 0.f;
    float t;
    return trace_ray(g, me_pos, to, 1.f, &t) == slot ? 1.f : 0.f;
}

__global__ void __launch_bounds__(256) k_observe(SimState S) {
    __shared__ ObsShared sh;
    const int w = blockIdx.x;
    const int tid = threadIdx.x;
    const int A = S.A;
    stage_world<256>(S, w, sh, tid);
    const int counts = S.counts[w];
    const int teams = S.teams[w];
    const int step = S.curEpisodeStep[w];
    __syncthreads();
    const WorldGeom &g = sh.g;
    const int nAgents = cnt_agents(counts), nBoxes = cnt_boxes(counts), nRamps = cnt_ramps(counts);
    const int nLidar = A * 30, nVis = A * 16, nObs = A * 17;
    const int total = nLidar + nVis + nObs + 1;
    for (int item = tid; item < total; item += 256) {
        if (item < nLidar) {
            // ---- lidarSystem: 30 rays in the agent's horizontal plane, t_max 200
            const int i = item / 30, k = item % 30;
            if (i >= nAgents) continue;
            const int slot = kAgentSlot0 + i;
            const Q rot = geom_rot(g, slot);
            const V3 pos = geom_pos(g, slot);
            V3 fwd = qrot(rot, {0.f, 1.f, 0.f}), right = qrot(rot, {1.f, 0.f, 0.f});
            float theta = 2.f * kPi * ((float)k / 30.f) + kPi / 2.f;
            float s, c; hs_sincosf(theta, &s, &c);
            V3 dir = normalize(right * c + fwd * s);
            float t;
            int hit = trace_ray(g, pos, dir, 200.f, &t);
            S.xLidar[(w * A + i) * 30 + k] = hit < 0 ? 0.f : t;
        } else if (item < nLidar + nVis) {
            // ---- computeVisibilitySystem: FOV cone + segment ray to each box / ramp / other agent
            const int v = item - nLidar;
            const int i = v / 16, e = v % 16;
            if (i >= nAgents) continue;
            const int row = w * A + i;
            const int slot = kAgentSlot0 + i;
            const V3 pos = geom_pos(g, slot);
            const V3 fwd = qrot(geom_rot(g, slot), {0.f, 1.f, 0.f});
            if (e < kMaxBoxes) {
                S.xVisBoxes[row * kMaxBoxes + e] = e < nBoxes ? check_visibility(g, pos, fwd, kBoxSlot0 + e) : 0.f;
            } else if (e < kMaxBoxes + kMaxRamps) {
                const int r = e - kMaxBoxes;
                S.xVisRamps[row * kMaxRamps + r] = r < nRamps ? check_visibility(g, pos, fwd, kRampSlot0 + r) : 0.f;
            } else {
                const int jj = e - kMaxBoxes - kMaxRamps;
                const int j = jj < i ? jj : jj + 1;
                float vis = 0.f;
                if (j < nAgents) {
                    vis = check_visibility(g, pos, fwd, kAgentSlot0 + j);
                    // CPU-branch side effect (sim.cpp:700-705): all writers store the same value
                    if (team_agent_type(teams, i) == AGENT_SEEKER && vis != 0.f &&
                        team_agent_type(teams, j) == AGENT_HIDER)
                        S.hiderTeamReward[w] = -1.f;
                }
                S.xVisAgents[row * (kMaxAgents - 1) + jj] = vis;
            }
        } else if (item < nLidar + nVis + nObs) {
            // ---- collectObservationsSystem: self / box / ramp / other-agent rows
            const int v = item - nLidar - nVis;
            const int i = v / 17, e = v % 17;
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
}

}  // namespace hs
