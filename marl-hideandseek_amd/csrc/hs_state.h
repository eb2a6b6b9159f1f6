// Simulator state in HBM.
//
// Internal entity-component columns are struct-of-arrays ACROSS WORLDS: element (field f, slot s,
// world w) lives at base[(f*SLOTS + s)*N + w], so a wave whose lanes walk consecutive worlds (or a
// group that loads one world's slots) issues coalesced dword loads.  This replaces Madrona's ECS
// archetype tables: every capacity is a small compile-time bound
// (src/sim.hpp:39-41), so slots are fixed and nothing is ever compacted or sorted.
//
// Exported tensors keep the reference's contract instead (AoS, row-major, agent row =
// world*A + slot; src/mgr.cpp:1062-1331) because scripts alias them in place.
#pragma once
#include "hs_dev.h"

namespace hs {

struct SimState {
    int N;                 // worlds in this shard
    int A;                 // maxAgentsPerWorld = maxHiders + maxSeekers (src/mgr.cpp:684)
    uint32_t flags;        // SimFlags | extension bits
    RandKey initKey;       // rand::initKey(seed) (src/mgr.cpp:678)
    int minHiders, maxHiders, minSeekers, maxSeekers;
    int worldOffset;
    int wbeg, wcnt;        // world range owned by a workgroup of k_physics (set inside the kernel)

    // --- movable bodies: 17 slots (9 boxes, 2 ramps, 6 agents)
    float *bpos;           // [3][17][N]
    float *brot;           // [4][17][N]  w,x,y,z
    float *blin;           // [3][17][N]
    float *bang;           // [3][17][N]
    int *bmeta;            // [17][N]     meta_pack()
    float *aforce;         // [4][6][N]   ExternalForce xyz + ExternalTorque z of the agents
    // --- static geometry
    float *walls;          // [4][36][N]  cx, cy, hx, hy
    float *planes;         // [4][3][N]   nx, ny, nz, d
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
    int *runningScores;          // [2][N]
    // --- grab joints, one per agent slot
    int *grabOther;        // [6][N]  D-slot or -1
    float *grabData;       // [kGrabWords][6][N]  r2 xyz, attach2 wxyz, separation, r1 xyz, attach1 wxyz

    // --- exported columns (AoS)
    int32_t *xReset, *xPrep, *xAction, *xSelfType, *xSeed, *xDone, *xPolicy;
    float *xSelfObs, *xSelfMask, *xAgentObs, *xBoxObs, *xRampObs, *xVisAgents, *xVisBoxes, *xVisRamps;
    float *xLidar, *xReward, *xGlobalPos, *xEpisodeResult;
    int32_t *xCkptCtrl;    // [N]  CheckpointControl::trigger (sim.hpp:279-281)
    uint8_t *xCkpt;        // [N][sizeof(hs_checkpoint)]  (include/hideseek.h, sim.hpp:283-313)
    // --- substep scratch of the physics kernel (hs_k_pipeline.h) that does not fit its LDS-resident working set,
    // all SoA across worlds
    int *gman;             // [2][17][N] (double-buffered by substep parity)     ground manifold: np | vertex ids << 4 | has-static-candidates << 30
    float *goff, *glam;    // [4][17][N]  ground manifold plane offsets / accumulated multipliers
    int *ndd, *nsc;        // [N]         candidate counts
    int *ddPair, *scPair;  // [kMaxDDCand][N] a | b << 8 ; [kMaxSCand][N] body | static << 8
    int *wflags;           // [N]         1 = world has a grab joint
    void *wsDD, *wsSC;     // contact-manifold workspace: [N][kMaxDDCand] ManDD, [N][kMaxSCand] ManS
    int *bodyList;         // [N][17] existing bodies of a workgroup's worlds, compacted once per step (local world << 5 | slot)
    int *satList, *wallList, *ddwList;   // work lists of one substep; workgroup b of k_physics uses the slice of its worlds
    int *status;           // [4] device-side conditions: dropped body-body pairs, dropped body-static pairs, -, -
                           // (include/hideseek.h hs_device_status); bumped only when something happens
    int *hostFlag;         // pinned host word (device-visible): set to 1 together with any change of status
    long long *phaseTicks; // [workgroups][10] accumulated per-phase ticks (HS_PHASE_TIMING builds only)
};

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
