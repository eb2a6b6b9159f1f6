// Simulator state in HBM.
//
// Internal entity-component columns are struct-of-arrays ACROSS WORLDS, tiled by kTile = 8 consecutive worlds
// (an "octet"): element (row r, world w) of a column with ROWS rows per world lives at
//     base[((w / 8) * ROWS + r) * 8 + (w % 8)]          (struct Col below; r = component * SLOTS + slot)
// so the whole working set of an octet is ONE contiguous block per column (e.g. 51 rows x 8 worlds x 4 B = 1.6 KB of
// positions) that a wave streams with full 256-byte transactions, and inside the block the 8 worlds of a row are
// adjacent: a wave whose lanes are (row, world) pairs reads and writes conflict-free.  The physics kernel gives
// one wave to an octet, the observation kernel one workgroup.  This replaces Madrona's ECS archetype tables
// (SURVEY §2 row 9): every capacity is a small compile-time bound (src/sim.hpp:39-41), so slots are fixed and
// nothing is ever compacted or sorted.
//
// Exported tensors keep the reference's contract instead (AoS, row-major, agent row =
// world*A + slot; src/mgr.cpp:1062-1331) because scripts alias them in place.
#pragma once
#include "hs_dev.h"

namespace hs {

constexpr int kTile = 8;          // worlds per octet
// A column with ROWS rows per world, tiled by octets.  Allocated for ceil(N / 8) octets.
template <typename T, int ROWS> struct Col {
    T *p;
    static constexpr int kRows = ROWS;
    HSD T &operator()(int row, int w) const { return p[((size_t)(w >> 3) * ROWS + row) * kTile + (w & 7)]; }
    HSD T *octet(int o) const { return p + (size_t)o * ROWS * kTile; }       // rows x 8 worlds, contiguous
};

struct SimState {
    int N;                 // worlds in this shard
    int A;                 // maxAgentsPerWorld = maxHiders + maxSeekers (src/mgr.cpp:684)
    uint32_t flags;        // SimFlags | extension bits
    RandKey initKey;       // rand::initKey(seed) (src/mgr.cpp:678)
    int minHiders, maxHiders, minSeekers, maxSeekers;
    int worldOffset;
    int wbeg, wcnt;        // the octet's first slot (set inside k_physics)
    // --- which world lives where.  The tiled columns are indexed by SLOT; per-world scalars and the exported tensors
    // by WORLD id.  Initially slot == world; k_balance (hs_k_balance.h) deals the worlds to the octets by contact load
    // so that the physics waves — one per octet, the launch ends with the slowest — carry about the same work.
    int *slotOfWorld;      // [N]
    int *worldOfSlot;      // [ceil(N / 8) * 8]  -1 = empty slot (padding of the last octet)
    int *loadAcc;          // [N] candidate pairs seen since the last k_balance
    int *wallHist;         // [N] bit s: body slot s had a wall / extra-plane manifold in the previous step (ordering hint only)
    // What k_observe needs to know about the world in a slot, in slot order so that no load waits for worldOfSlot:
    // {world id (-1: empty), numWalls | numPlanes << 8 | curEpisodeStep << 16, counts, teams}.  Rewritten by whoever
    // changes one of them (write_slot_hdr: reset / level generation, checkpoint load, k_balance_commit).
    int4 *slotHdr;         // [ceil(N / 8) * 8]
    // How long each physics wave took in the previous step (100 MHz ticks) and the sums over all waves of the last
    // three steps: a wave that was slower than the average raises its issue priority (hs_k_physics.h).
    int *octTicks;         // [octets]
    unsigned long long *tickSum;   // [3], indexed by stepIdx
    int stepIdx;           // launch counter mod 3
    float *lidarSinCos;    // [60] hs_sincosf of the 30 lidar angles (sim.cpp:727-738), filled by k_lidar_table at hs_create

    // --- movable bodies: 17 slots (9 boxes, 2 ramps, 6 agents)
    Col<float, 3 * kNumDSlots> bpos;       // row = component * 17 + slot
    Col<float, 4 * kNumDSlots> brot;       // w,x,y,z
    Col<float, 3 * kNumDSlots> blin;
    Col<float, 3 * kNumDSlots> bang;
    Col<int, kNumDSlots> bmeta;            // meta_pack()
    Col<float, 4 * kMaxAgents> aforce;     // ExternalForce xyz + ExternalTorque z of the agents; row = component * 6 + agent
    // --- static geometry
    Col<float, 4 * kMaxWalls> walls;       // cx, cy, hx, hy; row = component * 36 + wall
    Col<float, 4 * kMaxPlanes> planes;     // nx, ny, nz, d;  row = component * 3 + plane
    int *numWalls;         // [N]
    int *numPlanes;        // [N]
    // --- world scalars (Sim fields src/sim.hpp:326-362 and singletons :105-121)
    uint32_t *curWorldEpisode;   // [N]
    uint32_t *rngKeyA, *rngKeyB, *rngCount;   // episode RNG
    uint32_t *epKeyA, *epKeyB;   // [N] curEpisodeRNDCounter = {episode index, global world id} (sim.cpp:107-111)
    int *curEpisodeStep;         // [N]
    float *hiderTeamReward;      // [N]
    int *counts;    // [N] numHiders | numSeekers<<4 | numActiveAgents<<8 | numActiveBoxes<<12 | numActiveRamps<<16 | seekersFirst<<20
    int *teams;     // [N] hiders[3] (3 bits each, bits 0-8), seekers[3] (bits 9-17), agentType[6] (bits 18-23), agentActive[6] (bits 24-29)
    Col<int, 2> runningScores;
    // --- grab joints, one per agent slot
    Col<int, kMaxAgents> grabOther;                   // D-slot or -1
    Col<float, kGrabWords * kMaxAgents> grabData;     // row = word * 6 + agent: r2 xyz, attach2 wxyz, separation, r1 xyz, attach1 wxyz

    // --- exported columns (AoS)
    int32_t *xReset, *xPrep, *xAction, *xSelfType, *xSeed, *xDone, *xPolicy;
    float *xSelfObs, *xSelfMask, *xAgentObs, *xBoxObs, *xRampObs, *xVisAgents, *xVisBoxes, *xVisRamps;
    float *xLidar, *xReward, *xGlobalPos, *xEpisodeResult;
    int32_t *xCkptCtrl;    // [N]  CheckpointControl::trigger (sim.hpp:279-281)
    uint8_t *xCkpt;        // [N][sizeof(hs_checkpoint)]  (include/hideseek.h, sim.hpp:283-313)
    // --- contact-manifold workspace of the physics kernel (hs_k_physics.h): written by the lane that ran the convex
    // test, read by the lanes that solve the contact; one record for EVERY possible pair of a world, indexed by the pair's
    // place in the world's candidate order: [slots][136] ManDD, [slots][646] ManS.  The fast path touches the first
    // 16 / 24 places of a world at most (and usually none: its manifolds live in LDS); the rest is the spill path's.
    void *wsDD, *wsSC;
    unsigned short *spPair;   // [slots][136 + 646] pair codes of the candidates beyond the LDS capacities, at their places
    int *spInfo;              // [slots][18] totals (body-body | body-static << 16), per body: first spilled static | count << 16
    int *status;           // [4] sticky counters: body-body / body-static candidate pairs that took the spill path, -, -
                           // (include/hideseek.h hs_device_status); bumped only when something happens
    long long *phaseTicks; // [physics waves][10] accumulated per-phase ticks, then k_observe's sections and the convex tests'
                           // counters from phase_ticks_obs_base() on (HS_PHASE_TIMING builds only)
};

// phaseTicks: the physics waves' part is sized for the finer tile (4 worlds per wave)
__host__ __device__ inline size_t phase_ticks_obs_base(int N) { return (size_t)10 * ((N + kTile - 1) / kTile * 2); }
// ... and after k_observe's part: [N][8] work counters per WORLD (HS_LOAD_STUDY builds, tools/load_study.py): body-body /
// body-static candidates, accepted body-body manifolds, rounds of the body-body solver the world was pending in,
// the physics wave it lives in, and — for the first world of a wave — the solver rounds the wave ran
constexpr int kStudyWords = 8;
__host__ __device__ inline size_t phase_ticks_study_base(int N) { return phase_ticks_obs_base(N) + 16 * 1024 + 16; }

HSD int cnt_hiders(int c) { return c & 15; }
HSD int cnt_seekers(int c) { return (c >> 4) & 15; }
HSD int cnt_agents(int c) { return (c >> 8) & 15; }
HSD int cnt_boxes(int c) { return (c >> 12) & 15; }
HSD int cnt_ramps(int c) { return (c >> 16) & 15; }
HSD int cnt_seekers_first(int c) { return (c >> 20) & 1; }
HSD int cnt_pack(int nh, int ns, int na, int nb, int nr, int sf) {
    return nh | (ns << 4) | (na << 8) | (nb << 12) | (nr << 16) | (sf << 20);
}
HSD int team_hider(int t, int i) { return (t >> (3 * i)) & 7; }
HSD int team_seeker(int t, int i) { return (t >> (9 + 3 * i)) & 7; }
HSD int team_agent_type(int t, int i) { return (t >> (18 + i)) & 1; }
HSD int team_agent_active(int t, int i) { return (t >> (24 + i)) & 1; }

}  // namespace hs
