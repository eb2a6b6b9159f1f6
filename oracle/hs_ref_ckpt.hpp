// ORACLE — TEST INFRASTRUCTURE ONLY (see hs_ref_math.hpp header).  PARITY UNPINNED: the engine's
// JointConstraint::Fixed layout is not in the container; {attachRot1, attachRot2, separation}
// (9 floats) is assumed, which gives sizeof(Checkpoint) = 1392 as SURVEY §8b quotes.
// Checkpoint snapshot / restore: Checkpoint layout src/sim.hpp:283-313, saveCheckpointSystem
// src/sim.cpp:1046-1137, loadCheckpointSystem src/sim.cpp:956-1044.
#pragma once
#include "hs_ref_sim.hpp"

namespace hsref {

struct CkptRigidBody { float pos[3]; float rot[4]; float lin[3]; float ang[3]; };      // sim.hpp:288-292
struct CkptObject : CkptRigidBody { uint32_t team; uint8_t isLocked; uint8_t pad[3]; }; // :294-297
struct CkptAgent : CkptRigidBody {                                                       // :299-304
    int32_t grabIdx; float grabR1[3]; float grabR2[3];
    float attachRot1[4]; float attachRot2[4]; float separation;
};
struct Checkpoint {
    uint32_t episodeRNDKey[2]; int32_t runningScores[2]; int32_t episodeStep;            // :284-286
    CkptAgent agents[kMaxAgents]; CkptObject boxes[kMaxBoxes]; CkptObject ramps[kMaxRamps];
    int32_t numHiders, numSeekers, numBoxes, numRamps;                                   // :309-312
};
static_assert(sizeof(Checkpoint) == 1392, "Checkpoint layout");

static inline void ckpt_put_body(CkptRigidBody &o, const DBody &b) {
    o.pos[0] = b.pos.x; o.pos[1] = b.pos.y; o.pos[2] = b.pos.z;
    o.rot[0] = b.rot.w; o.rot[1] = b.rot.x; o.rot[2] = b.rot.y; o.rot[3] = b.rot.z;
    o.lin[0] = b.lin.x; o.lin[1] = b.lin.y; o.lin[2] = b.lin.z;
    o.ang[0] = b.ang.x; o.ang[1] = b.ang.y; o.ang[2] = b.ang.z;
}
static inline void ckpt_get_body(const CkptRigidBody &o, DBody &b) {
    b.pos = {o.pos[0], o.pos[1], o.pos[2]}; b.rot = {o.rot[0], o.rot[1], o.rot[2], o.rot[3]};
    b.lin = {o.lin[0], o.lin[1], o.lin[2]}; b.ang = {o.ang[0], o.ang[1], o.ang[2]};
}

// saveCheckpointSystem sim.cpp:1046-1137
static inline void save_checkpoint_system(const World &w, int32_t &trigger, Checkpoint &ck) {
    if (!trigger) return;
    trigger = 0;
    std::memset((void *)&ck, 0, sizeof(ck));
    ck.episodeRNDKey[0] = w.curEpisodeRNDCounter.a; ck.episodeRNDKey[1] = w.curEpisodeRNDCounter.b;
    ck.runningScores[0] = w.runningScores[0]; ck.runningScores[1] = w.runningScores[1];
    ck.episodeStep = w.curEpisodeStep;
    int cur = 0;
    auto agent = [&](int ai) {
        CkptAgent &a = ck.agents[cur++];
        ckpt_put_body(a, w.d[kAgentSlot0 + ai]);
        a.grabIdx = -1;
        const GrabJoint &g = w.grab[ai];
        if (g.other >= 0) {
            a.grabR1[0] = g.r1.x; a.grabR1[1] = g.r1.y; a.grabR1[2] = g.r1.z;
            a.grabR2[0] = g.r2.x; a.grabR2[1] = g.r2.y; a.grabR2[2] = g.r2.z;
            a.attachRot1[0] = g.attach1.w; a.attachRot1[1] = g.attach1.x; a.attachRot1[2] = g.attach1.y; a.attachRot1[3] = g.attach1.z;
            a.attachRot2[0] = g.attach2.w; a.attachRot2[1] = g.attach2.x; a.attachRot2[2] = g.attach2.y; a.attachRot2[3] = g.attach2.z;
            a.separation = g.separation;
            // boxes occupy D-slots [0, numActiveBoxes), ramps [kRampSlot0, kRampSlot0 + numActiveRamps)
            if (g.other < w.numActiveBoxes) a.grabIdx = g.other;
            else if (g.other >= kRampSlot0 && g.other < kRampSlot0 + w.numActiveRamps) a.grabIdx = g.other - kRampSlot0 + w.numActiveBoxes;
        }
    };
    ck.numHiders = w.numHiders;
    for (int i = 0; i < w.numHiders; ++i) agent(w.hiders[i]);
    ck.numSeekers = w.numSeekers;
    for (int i = 0; i < w.numSeekers; ++i) agent(w.seekers[i]);
    auto object = [&](const DBody &b, CkptObject &o) {
        ckpt_put_body(o, b);
        o.team = (uint32_t)b.owner;
        o.isLocked = b.response == RESP_STATIC ? 1 : 0;
    };
    ck.numBoxes = w.numActiveBoxes;
    for (int i = 0; i < w.numActiveBoxes; ++i) object(w.d[i], ck.boxes[i]);
    ck.numRamps = w.numActiveRamps;
    for (int i = 0; i < w.numActiveRamps; ++i) object(w.d[kRampSlot0 + i], ck.ramps[i]);
}

// loadCheckpointSystem sim.cpp:956-1044.  The trigger is left at 1 (:963 writes 1, not 0).
// Counts outside the build's capacities are clamped (the reference only asserts).
static inline void load_checkpoint_system(World &w, Exports &ex, int wi, const Config &cfg, RandKey init_key,
                                          int32_t &trigger, const Checkpoint &ck) {
    if (!trigger) return;
    trigger = 1;
    reset_environment(w, 0, init_key, false);
    w.curEpisodeRNDCounter = {ck.episodeRNDKey[0], ck.episodeRNDKey[1]};
    w.rng = RNG(rand_split_i(init_key, ck.episodeRNDKey[0], ck.episodeRNDKey[1]));
    w.runningScores[0] = ck.runningScores[0]; w.runningScores[1] = ck.runningScores[1];
    w.curEpisodeStep = ck.episodeStep;
    // "HACK, need to burn RNG state to get same result in generateEnv" (:976-980)
    w.rng.sampleI32(cfg.minHiders, cfg.maxHiders + 1);
    w.rng.sampleI32(cfg.minSeekers, cfg.maxSeekers + 1);
    RandKey lvl = w.rng.randKey();
    if ((cfg.simFlags & FLAG_USE_FIXED_WORLD) == FLAG_USE_FIXED_WORLD) lvl = {0u, 0u};
    auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    const int nh = clampi(ck.numHiders, 0, ex.A < 3 ? ex.A : 3);
    const int ns = clampi(ck.numSeekers, 0, ex.A - nh < 3 ? ex.A - nh : 3);
    generate_environment(w, ex, wi, lvl, 1, cfg.simFlags, nh, ns);
    auto object = [&](DBody &b, const CkptObject &o) {
        ckpt_get_body(o, b);
        b.owner = (int32_t)(o.team & 3u);
        b.response = o.isLocked ? RESP_STATIC : RESP_DYNAMIC;
    };
    const int nb = clampi(ck.numBoxes, 0, w.numActiveBoxes), nr = clampi(ck.numRamps, 0, w.numActiveRamps);
    for (int i = 0; i < nb; ++i) object(w.d[i], ck.boxes[i]);
    for (int i = 0; i < nr; ++i) object(w.d[kRampSlot0 + i], ck.ramps[i]);
    auto agent = [&](int ai, const CkptAgent &a) {
        ckpt_get_body(a, w.d[kAgentSlot0 + ai]);
        GrabJoint &g = w.grab[ai];
        g.other = -1;
        if (a.grabIdx >= 0 && a.grabIdx < nb + nr) {
            g.other = a.grabIdx < nb ? a.grabIdx : kRampSlot0 + (a.grabIdx - nb);
            g.r1 = {a.grabR1[0], a.grabR1[1], a.grabR1[2]}; g.r2 = {a.grabR2[0], a.grabR2[1], a.grabR2[2]};
            g.attach1 = {a.attachRot1[0], a.attachRot1[1], a.attachRot1[2], a.attachRot1[3]};
            g.attach2 = {a.attachRot2[0], a.attachRot2[1], a.attachRot2[2], a.attachRot2[3]};
            g.separation = a.separation;
        }
    };
    for (int i = 0; i < w.numHiders; ++i) agent(w.hiders[i], ck.agents[i]);
    for (int i = 0; i < w.numSeekers; ++i) agent(w.seekers[i], ck.agents[i + w.numHiders]);
}

// SaveCheckpoints / LoadCheckpoints task graphs (sim.cpp:1315-1333): the load graph re-runs the
// observation nodes for every world.
static inline void sim_save_checkpoints(Sim &s) {
    Checkpoint *ck = (Checkpoint *)s.s_ckpt.data();
    s.parallel([&s, ck](int i) { save_checkpoint_system(s.worlds[i], s.s_ckptCtrl[i], ck[i]); });
}
static inline void sim_load_checkpoints(Sim &s) {
    const Checkpoint *ck = (const Checkpoint *)s.s_ckpt.data();
    s.parallel([&s, ck](int i) {
        load_checkpoint_system(s.worlds[i], s.ex, i, s.cfg, s.initKey, s.s_ckptCtrl[i], ck[i]);
        s.observations(s.worlds[i], i);
    });
}

}  // namespace hsref
