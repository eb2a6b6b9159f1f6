// ORACLE — TEST INFRASTRUCTURE ONLY (see hs_ref_math.hpp header).
// Per-world state of the CPU restatement.  Mirrors `struct Sim` (src/sim.hpp:315-363) and the
// ECS columns it touches, flattened into fixed-capacity slot tables (SURVEY §7 design stance).
#pragma once
#include "hs_ref_math.hpp"
#include "hs_ref_rng.hpp"

namespace hsref {

// src/sim.hpp:39-41
constexpr int kMaxBoxes = 9;
constexpr int kMaxRamps = 2;
constexpr int kMaxAgents = 6;
// src/sim.cpp:14-17
constexpr float kDeltaT = 1.f / 30.f;
constexpr int kNumSubsteps = 4;
constexpr int kNumPrepSteps = 96;
constexpr int kEpisodeLen = 240;

// src/sim.hpp:78-88
enum SimObject : int32_t {
    OBJ_SPHERE = 0, OBJ_PLANE = 1, OBJ_CUBE = 2, OBJ_WALL = 3, OBJ_HIDER = 4, OBJ_SEEKER = 5,
    OBJ_RAMP = 6, OBJ_BOX = 7, OBJ_NONE = -1,
};
// src/sim.hpp:127-132
enum OwnerTeam : int32_t { OWNER_NONE = 0, OWNER_SEEKER = 1, OWNER_HIDER = 2, OWNER_UNOWNABLE = 3 };
// madrona::phys::ResponseType (enum values are the build's own)
enum ResponseType : int32_t { RESP_DYNAMIC = 0, RESP_KINEMATIC = 1, RESP_STATIC = 2 };
// src/sim.hpp:138-141
enum AgentType : int32_t { AGENT_SEEKER = 0, AGENT_HIDER = 1 };
// src/sim_flags.hpp:7-13
enum SimFlags : uint32_t {
    FLAG_DEFAULT = 0, FLAG_USE_FIXED_WORLD = 1, FLAG_IGNORE_EPISODE_LENGTH = 2,
    FLAG_RANDOM_FLIP_TEAMS = 4, FLAG_ZERO_AGENT_VELOCITY = 8,
};

// Slot layout of movable ("D") bodies: boxes, then ramps, then agents.
constexpr int kBoxSlot0 = 0;
constexpr int kRampSlot0 = kMaxBoxes;                 // 9
constexpr int kAgentSlot0 = kMaxBoxes + kMaxRamps;    // 11
constexpr int kNumDSlots = kAgentSlot0 + kMaxAgents;  // 17
// Static bodies: axis-aligned walls and infinite planes.
// Reference worst case is 34 walls (geo_gen.cpp:429-462; its TmpArray holds 33 and the
// overflow is only asserted in debug builds, geo_gen.cpp:24,144-147).  We hold 36.
constexpr int kMaxWalls = 36;
constexpr int kMaxPlanes = 3;
constexpr int kMaxStatics = kMaxWalls + kMaxPlanes;   // 39

struct DBody {
    int32_t objType;       // SimObject or OBJ_NONE when the slot is empty
    int32_t response;      // ResponseType
    int32_t owner;         // OwnerTeam
    V3 pos; Q rot;
    V3 lin, ang;           // Velocity
    V3 extForce, extTorque;
    // substep scratch
    V3 prevPos; Q prevRot;
};

struct WallS { float cx, cy, hx, hy; };       // z in [0, 2.5]; half extents in x/y
struct PlaneS { V3 n; float d; };             // n.p = d

struct GrabJoint {                            // PhysicsSystem::makeFixedJoint (sim.cpp:354-356)
    int32_t other;                            // D-slot of the grabbed body, -1 when none
    V3 r1, r2; Q attach1, attach2; float separation;
};

struct World {
    // --- Sim fields (sim.hpp:326-362)
    uint32_t curWorldEpisode;
    RandKey curEpisodeRNDCounter;
    RNG rng;
    int32_t numHiders, numSeekers, numActiveAgents;
    int32_t hiders[3], seekers[3];            // agent indices (0..A-1)
    int32_t numActiveBoxes, numActiveRamps;
    V3 boxSizes[kMaxBoxes];
    int32_t curEpisodeStep;
    float hiderTeamReward;
    // --- singletons (sim.hpp:105-121)
    bool seekersFirst;
    int32_t runningScores[2];
    // --- bodies
    DBody d[kNumDSlots];
    int32_t numWalls; WallS walls[kMaxWalls];
    int32_t numPlanes; PlaneS planes[kMaxPlanes];
    // --- agent interface columns that are not exported tensors
    int32_t agentType[kMaxAgents];
    int32_t agentActive[kMaxAgents];          // SimEntity != none
    GrabJoint grab[kMaxAgents];
};

}  // namespace hsref
