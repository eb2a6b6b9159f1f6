// ORACLE — TEST INFRASTRUCTURE ONLY (see hs_ref_math.hpp header).
// ECS systems and task-graph order of src/sim.cpp, one world at a time.
#pragma once
#include <vector>
#include <thread>
#include <cstring>
#include "hs_ref_level.hpp"
#include "hs_ref_phys.hpp"

namespace hsref {

struct Config {                 // Manager::Config mgr.hpp:16-32 (+ sharding offset)
    int32_t numWorlds;
    uint32_t simFlags;
    uint32_t randSeed;
    int32_t minHiders, maxHiders, minSeekers, maxSeekers;
    int32_t worldOffset;        // global id of local world 0
    int32_t skipObservations;   // build-side extension for the physics-only config
};

// ---- sim.cpp:105-159
static inline void reset_environment(World &w, uint32_t world_id, RandKey init_key, bool update_rng) {
    w.curEpisodeStep = 0;
    for (int i = 0; i < kNumDSlots; ++i) clear_dbody(w.d[i]);
    w.numWalls = 0; w.numPlanes = 0;
    w.numActiveBoxes = 0; w.numActiveRamps = 0;
    for (int i = 0; i < kMaxAgents; ++i) w.grab[i].other = -1;
    w.numHiders = 0; w.numSeekers = 0; w.numActiveAgents = 0;
    if (update_rng) {
        RandKey ctr = {w.curWorldEpisode++, world_id};
        w.curEpisodeRNDCounter = ctr;
        w.rng = RNG(rand_split_i(init_key, ctr.a, ctr.b));
    }
}

// ---- resetSystem sim.cpp:172-200 (+ levelGenRandKey :161-170)
static inline void reset_system(World &w, Exports &ex, int wi, const Config &cfg, RandKey init_key) {
    int32_t level = ex.reset[wi];
    if ((cfg.simFlags & FLAG_IGNORE_EPISODE_LENGTH) != FLAG_IGNORE_EPISODE_LENGTH &&
        w.curEpisodeStep == kEpisodeLen - 1) level = 1;
    if (level != 0) {
        reset_environment(w, (uint32_t)(cfg.worldOffset + wi), init_key, true);
        ex.reset[wi] = 0;
        int32_t nh = w.rng.sampleI32(cfg.minHiders, cfg.maxHiders + 1);
        int32_t ns = w.rng.sampleI32(cfg.minSeekers, cfg.maxSeekers + 1);
        RandKey lvl = w.rng.randKey();
        if ((cfg.simFlags & FLAG_USE_FIXED_WORLD) == FLAG_USE_FIXED_WORLD) lvl = {0u, 0u};
        generate_environment(w, ex, wi, lvl, level, cfg.simFlags, nh, ns);
    } else {
        w.curEpisodeStep += 1;
    }
    w.hiderTeamReward = 1.f;
}

// ---- movementSystem / instantMovementSystem sim.cpp:202-254
static inline void movement_system(World &w, Exports &ex, int wi, const Config &cfg) {
    const bool instant = (cfg.simFlags & FLAG_ZERO_AGENT_VELOCITY) == FLAG_ZERO_AGENT_VELOCITY;
    for (int i = 0; i < ex.A; ++i) {
        if (!w.agentActive[i]) continue;
        if (w.agentType[i] == AGENT_SEEKER && w.curEpisodeStep < kNumPrepSteps - 1) continue;
        const int32_t *a = ex.action + (wi * ex.A + i) * 5;
        float fx, fy, tz;
        if (instant) { fx = 400.f * (float)(a[0] - 2); fy = 400.f * (float)(a[1] - 2); tz = 120.f * (float)(a[2] - 2); }
        else { fx = 12.f * (float)(a[0] - 5); fy = 12.f * (float)(a[1] - 5); tz = 3.f * (float)(a[2] - 5); }
        DBody &b = w.d[kAgentSlot0 + i];
        b.extForce = qrot(b.rot, {fx, fy, 0.f});
        b.extTorque = {0.f, 0.f, tz};
    }
}

// ---- actionSystem sim.cpp:270-370 (agents in interface order)
static inline void action_system(World &w, Exports &ex, int wi) {
    for (int i = 0; i < ex.A; ++i) {
        if (!w.agentActive[i]) continue;
        if (w.agentType[i] == AGENT_SEEKER && w.curEpisodeStep < kNumPrepSteps - 1) continue;
        int32_t *a = ex.action + (wi * ex.A + i) * 5;
        DBody &me = w.d[kAgentSlot0 + i];
        if (a[4] == 1) {
            float t; V3 o = me.pos + V3{0.f, 0.f, 0.5f};
            int hit = trace_ray(w, o, qrot(me.rot, {0.f, 1.f, 0.f}), 2.5f, &t);
            if (hit >= 0 && hit < kNumDSlots) {
                DBody &e = w.d[hit];
                if (e.response == RESP_STATIC) {
                    if ((w.agentType[i] == AGENT_SEEKER && e.owner == OWNER_SEEKER) ||
                        (w.agentType[i] == AGENT_HIDER && e.owner == OWNER_HIDER)) {
                        e.response = RESP_DYNAMIC; e.owner = OWNER_NONE;
                    }
                } else if (e.owner == OWNER_NONE) {
                    e.response = RESP_STATIC;
                    e.owner = w.agentType[i] == AGENT_HIDER ? OWNER_HIDER : OWNER_SEEKER;
                }
            }
        }
        if (a[3] == 1) {
            GrabJoint &g = w.grab[i];
            if (g.other >= 0) {
                g.other = -1;
            } else {
                float t; V3 o = me.pos + V3{0.f, 0.f, 0.5f};
                V3 dir = qrot(me.rot, {0.f, 1.f, 0.f});
                int hit = trace_ray(w, o, dir, 2.5f, &t);
                if (hit >= 0 && hit < kNumDSlots) {
                    DBody &e = w.d[hit];
                    if (e.owner == OWNER_NONE && e.response == RESP_DYNAMIC) {
                        V3 hit_pos = o + dir * t;
                        g.other = hit;
                        g.r1 = {0.f, 1.25f, 0.5f};
                        g.r2 = qrot(qinv(e.rot), hit_pos - e.pos);
                        g.attach1 = {1.f, 0.f, 0.f, 0.f};
                        g.attach2 = qnormalize(qmul(qinv(e.rot), me.rot));
                        g.separation = t - 1.25f;
                    }
                }
            }
        }
        a[0] = 2; a[1] = 2; a[2] = 2; a[3] = 0; a[4] = 0;
    }
}

// ---- agentZeroVelSystem sim.cpp:258-268
static inline void agent_zero_vel_system(World &w) {
    for (int i = 0; i < kMaxAgents; ++i) {
        DBody &b = w.d[kAgentSlot0 + i];
        if (b.objType == OBJ_NONE) continue;
        b.lin.x = 0.f; b.lin.y = 0.f; b.lin.z = fminf(b.lin.z, 0.f);
        b.ang = {0.f, 0.f, 0.f};
    }
}

constexpr float kCosFovHalf = 0.382683426f;   // cosf(toRadians(135/2)) sim.cpp:582,767

// ---- rewardsVisSystem sim.cpp:763-804
static inline void rewards_vis_system(World &w, int A) {
    for (int i = 0; i < A; ++i) {
        if (!w.agentActive[i] || w.agentType[i] != AGENT_SEEKER) continue;
        const DBody &s = w.d[kAgentSlot0 + i];
        V3 fwd = qrot(s.rot, {0.f, 1.f, 0.f});
        for (int h = 0; h < w.numHiders; ++h) {
            int hs = kAgentSlot0 + w.hiders[h];
            V3 to = w.d[hs].pos - s.pos;
            float c = dot(normalize(to), fwd);
            if (c < kCosFovHalf) continue;
            float t;
            if (trace_ray(w, s.pos, to, 1.f, &t) == hs) { w.hiderTeamReward = -1.f; break; }
        }
    }
}

// ---- outputRewardsDonesSystem sim.cpp:806-841
static inline void output_rewards_dones_system(World &w, Exports &ex, int wi) {
    for (int i = 0; i < ex.A; ++i) {
        if (!w.agentActive[i]) continue;
        int row = wi * ex.A + i;
        int step = w.curEpisodeStep;
        if (step == 0) ex.done[row] = 0;
        if (step < kNumPrepSteps - 1) { ex.reward[row] = 0.f; continue; }
        else if (step == kEpisodeLen - 1) ex.done[row] = 1;
        float r = w.hiderTeamReward;
        if (w.agentType[i] == AGENT_SEEKER) r *= -1.f;
        V3 p = w.d[kAgentSlot0 + i].pos;
        if (fabsf(p.x) >= 18.f || fabsf(p.y) >= 18.f) r -= 10.f;
        ex.reward[row] = r;
    }
}

// ---- updateEpisodeResultsSystem sim.cpp:843-893
static inline void update_episode_results_system(World &w, Exports &ex, int wi) {
    float *res = ex.episodeResult + wi * 2;
    int step = w.curEpisodeStep;
    if (step == 0) { res[0] = 0.f; res[1] = 0.f; w.runningScores[0] = 0; w.runningScores[1] = 0; }
    if (step >= kNumPrepSteps) {
        bool hidden = w.hiderTeamReward == 1.f;
        int win = hidden ? (w.seekersFirst ? 1 : 0) : (w.seekersFirst ? 0 : 1);
        w.runningScores[win] += 1;
    }
    if (step == kEpisodeLen - 1) {
        if (w.runningScores[0] > w.runningScores[1]) { res[0] = 1.f; res[1] = 0.f; }
        else if (w.runningScores[0] < w.runningScores[1]) { res[0] = 0.f; res[1] = 1.f; }
        else { res[0] = 0.5f; res[1] = 0.5f; }
    }
}

// ---- quatToEuler sim.cpp:372-399
static inline V3 quat_to_euler(Q q) {
    float sinr = 2.f * (q.w * q.x + q.y * q.z);
    float cosr = 1.f - 2.f * (q.x * q.x + q.y * q.y);
    float roll = hs_atan2f(sinr, cosr);
    float sinp = 2.f * (q.w * q.y - q.z * q.x);
    float pitch = fabsf(sinp) >= 1.f ? copysignf(3.14159265358979323846f / 2.f, sinp) : hs_asinf(sinp);
    float siny = 2.f * (q.w * q.z + q.x * q.y);
    float cosy = 1.f - 2.f * (q.y * q.y + q.z * q.z);
    float yaw = hs_atan2f(siny, cosy);
    return {roll, pitch, yaw};
}

// ---- computeRelativePosVelObs sim.cpp:401-420 ; writes 12 floats
static inline void rel_posvel_obs(float *o, V3 origin, Q to_frame, V3 flin, V3 fang, V3 x, Q q, V3 lin, V3 ang) {
    V3 p = qrot(to_frame, x - origin);
    Q qr = qnormalize(qmul(to_frame, q));
    V3 e = quat_to_euler(qr);
    V3 l = qrot(to_frame, lin - flin), a = qrot(to_frame, ang - fang);
    o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = e.x; o[4] = e.y; o[5] = e.z;
    o[6] = l.x; o[7] = l.y; o[8] = l.z; o[9] = a.x; o[10] = a.y; o[11] = a.z;
}
static inline void lock_obs(float *o, const DBody &b) {   // computeLockObservation sim.cpp:422-446
    if (b.response != RESP_STATIC) { o[0] = 0.f; o[1] = 0.f; }
    else if (b.owner == OWNER_HIDER) { o[0] = 1.f; o[1] = 0.f; }
    else { o[0] = 0.f; o[1] = 1.f; }
}

// ---- collectObservationsSystem sim.cpp:448-565
static inline void collect_observations_system(World &w, Exports &ex, int wi) {
    for (int i = 0; i < ex.A; ++i) {
        if (!w.agentActive[i]) continue;
        int row = wi * ex.A + i;
        int step = w.curEpisodeStep;
        if (step <= kNumPrepSteps) ex.prep[row] = kNumPrepSteps - step;
        const DBody &me = w.d[kAgentSlot0 + i];
        Q toF = qinv(me.rot);
        float *so = ex.selfObs + row * 13;
        V3 e = quat_to_euler(me.rot);
        V3 l = qrot(toF, me.lin), a = qrot(toF, me.ang);
        so[0] = me.pos.x; so[1] = me.pos.y; so[2] = me.pos.z; so[3] = e.x; so[4] = e.y; so[5] = e.z;
        so[6] = l.x; so[7] = l.y; so[8] = l.z; so[9] = a.x; so[10] = a.y; so[11] = a.z;
        so[12] = w.grab[i].other >= 0 ? 1.f : 0.f;

        float *bo = ex.boxObs + row * (kMaxBoxes * 17);
        for (int b = 0; b < kMaxBoxes; ++b) {
            float *o = bo + b * 17;
            if (b >= w.numActiveBoxes) { for (int k = 0; k < 17; ++k) o[k] = 0.f; continue; }
            const DBody &bd = w.d[kBoxSlot0 + b];
            rel_posvel_obs(o, me.pos, toF, me.lin, me.ang, bd.pos, bd.rot, bd.lin, bd.ang);
            o[12] = w.boxSizes[b].x; o[13] = w.boxSizes[b].y; o[14] = w.boxSizes[b].z;
            lock_obs(o + 15, bd);
        }
        float *ro = ex.rampObs + row * (kMaxRamps * 14);
        for (int r = 0; r < kMaxRamps; ++r) {
            float *o = ro + r * 14;
            if (r >= w.numActiveRamps) { for (int k = 0; k < 14; ++k) o[k] = 0.f; continue; }
            const DBody &bd = w.d[kRampSlot0 + r];
            rel_posvel_obs(o, me.pos, toF, me.lin, me.ang, bd.pos, bd.rot, bd.lin, bd.ang);
            lock_obs(o + 12, bd);
        }
        float *ao = ex.agentObs + row * ((kMaxAgents - 1) * 14);
        int n_other = 0;
        for (int j = 0; j < kMaxAgents; ++j) {
            if (j >= w.numActiveAgents) {
                float *o = ao + (n_other++) * 14;
                for (int k = 0; k < 14; ++k) o[k] = 0.f;
                continue;
            }
            if (j == i) continue;
            float *o = ao + (n_other++) * 14;
            const DBody &bd = w.d[kAgentSlot0 + j];
            rel_posvel_obs(o, me.pos, toF, me.lin, me.ang, bd.pos, bd.rot, bd.lin, bd.ang);
            o[12] = w.agentType[j] == AGENT_HIDER ? 1.f : 0.f;
            o[13] = w.grab[j].other >= 0 ? 1.f : 0.f;
        }
    }
}

// ---- computeVisibilitySystem, CPU branch sim.cpp:567-605, 663-708
static inline float check_visibility(const World &w, const DBody &me, V3 fwd, int slot) {
    V3 to = w.d[slot].pos - me.pos;
    float c = dot(normalize(to), fwd);
    if (c < kCosFovHalf) return 0.f;
    float t;
    return trace_ray(w, me.pos, to, 1.f, &t) == slot ? 1.f : 0.f;
}
static inline void compute_visibility_system(World &w, Exports &ex, int wi) {
    for (int i = 0; i < ex.A; ++i) {
        if (!w.agentActive[i]) continue;
        int row = wi * ex.A + i;
        const DBody &me = w.d[kAgentSlot0 + i];
        V3 fwd = qrot(me.rot, {0.f, 1.f, 0.f});
        float *vb = ex.visBoxes + row * kMaxBoxes;
        for (int b = 0; b < kMaxBoxes; ++b)
            vb[b] = b < w.numActiveBoxes ? check_visibility(w, me, fwd, kBoxSlot0 + b) : 0.f;
        float *vr = ex.visRamps + row * kMaxRamps;
        for (int r = 0; r < kMaxRamps; ++r)
            vr[r] = r < w.numActiveRamps ? check_visibility(w, me, fwd, kRampSlot0 + r) : 0.f;
        float *va = ex.visAgents + row * (kMaxAgents - 1);
        int n_other = 0;
        for (int j = 0; j < kMaxAgents; ++j) {
            if (j >= w.numActiveAgents) { va[n_other++] = 0.f; continue; }
            if (j == i) continue;
            float vis = check_visibility(w, me, fwd, kAgentSlot0 + j);
            if (w.agentType[i] == AGENT_SEEKER && vis != 0.f && w.agentType[j] == AGENT_HIDER)
                w.hiderTeamReward = -1.f;
            va[n_other++] = vis;
        }
    }
}

// ---- lidarSystem sim.cpp:712-759
static inline void lidar_system(const World &w, Exports &ex, int wi) {
    const float pi = 3.14159265358979323846f;
    for (int i = 0; i < ex.A; ++i) {
        if (!w.agentActive[i]) continue;
        const DBody &me = w.d[kAgentSlot0 + i];
        V3 fwd = qrot(me.rot, {0.f, 1.f, 0.f}), right = qrot(me.rot, {1.f, 0.f, 0.f});
        float *out = ex.lidar + (wi * ex.A + i) * 30;
        for (int k = 0; k < 30; ++k) {
            float theta = 2.f * pi * ((float)k / 30.f) + pi / 2.f;
            float s, c; hs_sincosf(theta, &s, &c);
            V3 dir = normalize(right * c + fwd * s);
            float t;
            int hit = trace_ray(w, me.pos, dir, 200.f, &t);
            out[k] = hit < 0 ? 0.f : t;
        }
    }
}

// ---- globalPositionsDebugSystem sim.cpp:895-941 (incl. the double-increment tail)
static inline void global_positions_system(const World &w, Exports &ex, int wi) {
    float *g = ex.globalPos + wi * 34;
    for (int b = 0; b < kMaxBoxes; ++b) {
        if (b >= w.numActiveBoxes) { g[b * 2] = 0.f; g[b * 2 + 1] = 0.f; continue; }
        g[b * 2] = w.d[kBoxSlot0 + b].pos.x; g[b * 2 + 1] = w.d[kBoxSlot0 + b].pos.y;
    }
    float *gr = g + 18;
    for (int r = 0; r < kMaxRamps; ++r) {
        if (r >= w.numActiveRamps) { gr[r * 2] = 0.f; gr[r * 2 + 1] = 0.f; continue; }
        gr[r * 2] = w.d[kRampSlot0 + r].pos.x; gr[r * 2 + 1] = w.d[kRampSlot0 + r].pos.y;
    }
    float *ga = g + 22;
    int o = 0;
    for (int h = 0; h < w.numHiders; ++h, ++o) {
        ga[o * 2] = w.d[kAgentSlot0 + w.hiders[h]].pos.x; ga[o * 2 + 1] = w.d[kAgentSlot0 + w.hiders[h]].pos.y;
    }
    for (int s = 0; s < w.numSeekers; ++s, ++o) {
        ga[o * 2] = w.d[kAgentSlot0 + w.seekers[s]].pos.x; ga[o * 2 + 1] = w.d[kAgentSlot0 + w.seekers[s]].pos.y;
    }
    for (; o < kMaxAgents; o += 2) { ga[o * 2] = 0.f; ga[o * 2 + 1] = 0.f; }
}

// ----------------------------------------------------------------------------------------
// Manager-level driver: Manager::init / Manager::step (mgr.cpp:861-903), graph order
// setupInitTasks / setupStepTasks (sim.cpp:1295-1313).
// ----------------------------------------------------------------------------------------
class Sim {
public:
    Config cfg; int A; RandKey initKey; int threads = 1;
    std::vector<World> worlds;
    Exports ex;
    std::vector<int32_t> s_reset, s_prep, s_action, s_selfType, s_seed, s_done, s_policy;
    std::vector<float> s_selfObs, s_selfMask, s_agentObs, s_boxObs, s_rampObs, s_visAgents,
        s_visBoxes, s_visRamps, s_lidar, s_reward, s_globalPos, s_episodeResult;
    std::vector<int32_t> s_ckptCtrl;            // CheckpointControl::trigger per world (sim.hpp:279-281)
    std::vector<uint8_t> s_ckpt;                // Checkpoint per world (hs_ref_ckpt.hpp)

    explicit Sim(const Config &c) : cfg(c) {
        A = c.maxHiders + c.maxSeekers;
        initKey = rand_init_key(c.randSeed);
        int N = c.numWorlds, R = N * A;
        worlds.resize(N);
        s_ckptCtrl.assign(N, 0); s_ckpt.assign((size_t)N * 1392, 0);
        auto mk_i = [](std::vector<int32_t> &v, size_t n) { v.assign(n, 0); return v.data(); };
        auto mk_f = [](std::vector<float> &v, size_t n) { v.assign(n, 0.f); return v.data(); };
        ex.A = A;
        ex.reset = mk_i(s_reset, N); ex.prep = mk_i(s_prep, R); ex.action = mk_i(s_action, R * 5);
        ex.selfType = mk_i(s_selfType, R); ex.seed = mk_i(s_seed, R * 2); ex.done = mk_i(s_done, R);
        ex.policy = mk_i(s_policy, R);
        ex.selfObs = mk_f(s_selfObs, R * 13); ex.selfMask = mk_f(s_selfMask, R);
        ex.agentObs = mk_f(s_agentObs, R * 5 * 14); ex.boxObs = mk_f(s_boxObs, R * 9 * 17);
        ex.rampObs = mk_f(s_rampObs, R * 2 * 14); ex.visAgents = mk_f(s_visAgents, R * 5);
        ex.visBoxes = mk_f(s_visBoxes, R * 9); ex.visRamps = mk_f(s_visRamps, R * 2);
        ex.lidar = mk_f(s_lidar, R * 30); ex.reward = mk_f(s_reward, R);
        ex.globalPos = mk_f(s_globalPos, N * 34); ex.episodeResult = mk_f(s_episodeResult, N * 2);
        for (int wi = 0; wi < N; ++wi) {     // Sim::Sim sim.cpp:1346-1408
            World &w = worlds[wi];
            std::memset((void *)&w, 0, sizeof(World));
            for (int i = 0; i < kNumDSlots; ++i) clear_dbody(w.d[i]);
            for (int i = 0; i < kMaxAgents; ++i) w.grab[i].other = -1;
            w.hiderTeamReward = 1.f;
            ex.reset[wi] = 1;
        }
    }

    void observations(World &w, int wi) {
        if (cfg.skipObservations) return;
        collect_observations_system(w, ex, wi);
        compute_visibility_system(w, ex, wi);
        lidar_system(w, ex, wi);
        global_positions_system(w, ex, wi);
    }
    void init_world(int wi) {
        World &w = worlds[wi];
        reset_system(w, ex, wi, cfg, initKey);
        observations(w, wi);
    }
    void step_world(int wi) {
        World &w = worlds[wi];
        movement_system(w, ex, wi, cfg);
        action_system(w, ex, wi);
        for (int s = 0; s < kNumSubsteps; ++s) physics_substep(w);
        if ((cfg.simFlags & FLAG_ZERO_AGENT_VELOCITY) == FLAG_ZERO_AGENT_VELOCITY) agent_zero_vel_system(w);
        rewards_vis_system(w, A);
        output_rewards_dones_system(w, ex, wi);
        update_episode_results_system(w, ex, wi);
        reset_system(w, ex, wi, cfg, initKey);
        observations(w, wi);
    }
    template <typename F> void parallel(F f) {
        int N = cfg.numWorlds, T = threads < 1 ? 1 : threads;
        if (T == 1 || N < 2 * T) { for (int i = 0; i < N; ++i) f(i); return; }
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([=]() { for (int i = (int)((int64_t)N * t / T); i < (int)((int64_t)N * (t + 1) / T); ++i) f(i); });
        for (auto &x : th) x.join();
    }
    void init() { parallel([this](int i) { init_world(i); }); }
    void step() { parallel([this](int i) { step_world(i); }); }
};

}  // namespace hsref
