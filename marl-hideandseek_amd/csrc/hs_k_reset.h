// Episode reset kernel: resetSystem (src/sim.cpp:172-200) with the level generator
// (src/level_gen.cpp, src/geo_gen.cpp) — one lane per world.
//
// The generator is serial and branchy per world (rejection sampling, wall splitting); it runs
// once per 240 steps, so it is mapped lane-per-world with the world-fastest SoA layout giving
// coalesced stores, and its divergence is accepted.  Integer decisions are gated
// by float compares, so the arithmetic below is order-for-order the oracle's.
#pragma once
#include "../../include/hideseek.h"     // hs_checkpoint
#include "hs_state.h"

namespace hs {

struct Seg { float x1, y1, x2, y2; };
struct AABB { V3 lo, hi; };

HSD Seg seg_make(float ax, float ay, float bx, float by) {     // geo_gen.cpp:60-65
    if (ax > bx || ay > by) return {bx, by, ax, ay};
    return {ax, ay, bx, by};
}
HSD void seg_resort(Seg &s) {                                  // geo_gen.cpp:71-77
    if (s.x1 > s.x2 || s.y1 > s.y2) { Seg t = s; s = {t.x2, t.y2, t.x1, t.y1}; }
}
HSD bool seg_horizontal(const Seg &s) { return fabsf(s.y1 - s.y2) < 0.000001f; }   // :67-69
HSD float seg_length(const Seg &s) { return seg_horizontal(s) ? (s.x2 - s.x1) : (s.y2 - s.y1); }

struct WallSet {                                               // geo_gen.cpp:139-175
    Seg segs[kMaxWalls]; int n;
    uint8_t horiz[kMaxWalls]; int nh;
    uint8_t vert[kMaxWalls]; int nv;
};
HSD int wallset_add(WallSet &w, Seg s) {
    if (w.n >= kMaxWalls) return w.n - 1;
    if (seg_horizontal(s)) w.horiz[w.nh++] = (uint8_t)w.n;
    else w.vert[w.nv++] = (uint8_t)w.n;
    w.segs[w.n++] = s;
    return w.n - 1;
}

// geo_gen.cpp:177-270
HSD int find_facing_wall(const WallSet &w, bool hz, int chosen_i, RNG &rng) {
    const uint8_t *list = hz ? w.horiz : w.vert;
    const int ln = hz ? w.nh : w.nv;
    const float min_len = hz ? 0.3f : 0.5f;
    const Seg c = w.segs[list[chosen_i]];
    const float c_lo = hz ? c.x1 : c.y1, c_hi = hz ? c.x2 : c.y2, c_ac = hz ? c.y1 : c.x1;
    const float c_len = seg_length(c);
    int start = chosen_i + 1 + rng.sampleI32(0, ln - 1);
    for (int i = 0; i < ln - 1; ++i) {
        int cur = (start + i) % ln;
        if (cur == chosen_i) cur = (cur + 1) % ln;
        const Seg o = w.segs[list[cur]];
        const float o_lo = hz ? o.x1 : o.y1, o_hi = hz ? o.x2 : o.y2, o_ac = hz ? o.y1 : o.x1;
        if (!(c_lo >= o_hi || c_hi <= o_lo) && c_len >= min_len && seg_length(o) >= min_len) {
            float high = fminf(c_hi, o_hi);
            float low = fmaxf(c_lo, o_lo);
            bool works = true;
            for (int j = 0; j < ln; ++j) {
                if (j == cur) continue;
                const Seg b = w.segs[list[j]];
                float bl = fmaxf(hz ? b.x1 : b.y1, low - 0.1f);
                float bh = fminf(hz ? b.x2 : b.y2, high + 0.1f);
                if (bl < bh) {
                    float v = hz ? b.y1 : b.x1;
                    float vmin = fminf(c_ac, o_ac);
                    float vmax = fmaxf(c_ac, o_ac);
                    if (v > vmin && v < vmax) { works = false; break; }
                }
            }
            if (works) return cur;
        }
    }
    return -1;
}

HSD void cut_door(WallSet &w, int wi, float door, RNG &rng) {   // geo_gen.cpp:275-307
    Seg s = w.segs[wi];
    float rat = 0.3f + rng.sampleUniform() * 0.4f;
    Seg ns;
    if (seg_horizontal(s)) {
        float low = s.x1 + door, high = s.x2 - door;
        float x = low + rat * (high - low);
        float old = s.x2;
        s.x2 = x - door * 0.5f;
        seg_resort(s);
        ns = seg_make(x + door * 0.5f, s.y1, old, s.y1);
    } else {
        float low = s.y1 + door, high = s.y2 - door;
        float y = low + rat * (high - low);
        float old = s.y2;
        s.y2 = y - door * 0.5f;
        seg_resort(s);
        ns = seg_make(s.x1, y + door * 0.5f, s.x1, old);
    }
    w.segs[wi] = s;
    wallset_add(w, ns);
}

HSD void wall_op_connect(WallSet &w, RNG &rng) {               // geo_gen.cpp:311-409
    bool hz = rng.sampleI32(0, 2) != 0;
    int ln = hz ? w.nh : w.nv;
    int wi = rng.sampleI32(0, ln);
    int oi;
    int counter = 0;
    while ((oi = find_facing_wall(w, hz, wi, rng)) == -1) {
        hz = rng.sampleI32(0, 2) != 0;
        ln = hz ? w.nh : w.nv;
        wi = rng.sampleI32(0, ln);
        if (counter++ > 4) return;
    }
    const uint8_t *list = hz ? w.horiz : w.vert;
    int fi = list[wi], si = list[oi];
    const float kDoor = 0.1f;
    Seg f = w.segs[fi], s = w.segs[si];
    if (hz) {
        float high = fminf(f.x2, s.x2);
        float low = fmaxf(f.x1, s.x1);
        if (f.y1 > s.y1) { int t = fi; fi = si; si = t; Seg ts = f; f = s; s = ts; }
        float rat = 0.4f + rng.sampleUniform() * 0.2f;
        float x = low + rat * (high - low);
        int ni = wallset_add(w, seg_make(x, f.y1, x, s.y1));
        float fold = f.x2, sold = s.x2;
        f.x2 = x; seg_resort(f);
        s.x2 = x; seg_resort(s);
        w.segs[fi] = f; w.segs[si] = s;
        wallset_add(w, seg_make(x, f.y1, fold, f.y1));
        wallset_add(w, seg_make(x, s.y1, sold, s.y1));
        cut_door(w, ni, kDoor, rng);
    } else {
        float high = fminf(f.y2, s.y2);
        float low = fmaxf(f.y1, s.y1);
        if (f.x1 > s.x1) { int t = fi; fi = si; si = t; Seg ts = f; f = s; s = ts; }
        float rat = 0.4f + rng.sampleUniform() * 0.2f;
        float y = low + rat * (high - low);
        int ni = wallset_add(w, seg_make(f.x1, y, s.x1, y));
        float fold = f.y2, sold = s.y2;
        f.y2 = y; seg_resort(f);
        s.y2 = y; seg_resort(s);
        w.segs[fi] = f; w.segs[si] = s;
        wallset_add(w, seg_make(f.x1, y, f.x1, fold));
        wallset_add(w, seg_make(s.x1, y, s.x1, sold));
        cut_door(w, ni, kDoor, rng);
    }
}

HSD void wall_op_add_door(WallSet &w, RNG &rng) {              // geo_gen.cpp:411-421
    const float door = 0.1f * 2.0f;
    int wi = rng.sampleI32(0, w.n);
    if (seg_length(w.segs[wi]) > 3.0f * door) cut_door(w, wi, door, rng);
}

HSD void make_walls(WallSet &w, RNG &rng) {                    // geo_gen.cpp:429-465, 96-137
    w.n = w.nh = w.nv = 0;
    wallset_add(w, seg_make(0.f, 0.f, 1.f, 0.f));
    wallset_add(w, seg_make(0.f, 0.f, 0.f, 1.f));
    wallset_add(w, seg_make(0.f, 1.f, 1.f, 1.f));
    wallset_add(w, seg_make(1.f, 1.f, 1.f, 0.f));
    int c0 = 1 + rng.sampleI32(0, 6);
    int c1 = 4 + rng.sampleI32(0, 3);
    int op0 = 0, op1 = 1, nsel = 2;
    do {
        int oi = rng.sampleI32(0, nsel);
        int op = oi == 0 ? op0 : op1;
        int left = op == 0 ? --c0 : --c1;
        if (left == 0) { --nsel; if (oi == 0) op0 = (nsel == 1) ? op1 : op0; }
        if (op == 0) wall_op_connect(w, rng); else wall_op_add_door(w, rng);
    } while (c0 != 0 || c1 != 0);
}

// AABB::applyTRS with unit scale (level_gen.cpp:142-143)
HSD AABB aabb_apply_tr(AABB b, V3 t, Q r) {
    M3 m = m3_from_quat(r);
    m.c0 = m.c0 * 1.f; m.c1 = m.c1 * 1.f; m.c2 = m.c2 * 1.f;
    float lo[3] = {t.x, t.y, t.z}, hi[3] = {t.x, t.y, t.z};
    const float mm[3][3] = {{m.c0.x, m.c1.x, m.c2.x}, {m.c0.y, m.c1.y, m.c2.y}, {m.c0.z, m.c1.z, m.c2.z}};
    const float bl[3] = {b.lo.x, b.lo.y, b.lo.z}, bh[3] = {b.hi.x, b.hi.y, b.hi.z};
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            float e = mm[i][j] * bl[j], f = mm[i][j] * bh[j];
            if (e < f) { lo[i] += e; hi[i] += f; } else { lo[i] += f; hi[i] += e; }
        }
    }
    return {{lo[0], lo[1], lo[2]}, {hi[0], hi[1], hi[2]}};
}
HSD AABB object_aabb(int obj) {
    if (obj == OBJ_RAMP) return {{-1.f, -2.f, -1.f}, {1.f, 1.f, 1.f}};
    if (obj == OBJ_BOX) return {{-4.f, -0.75f, -1.f}, {4.f, 0.75f, 1.f}};
    return {{-1.f, -1.f, -1.f}, {1.f, 1.f, 1.f}};
}
HSD bool aabb_overlaps(const AABB &a, const AABB &b) {
    return a.lo.x < b.hi.x && b.lo.x < a.hi.x && a.lo.y < b.hi.y && b.lo.y < a.hi.y &&
           a.lo.z < b.hi.z && b.lo.z < a.hi.z;
}

// Per-lane working copy of the world being generated.
struct GenWorld {
    int numWalls; float wcx[kMaxWalls], wcy[kMaxWalls], whx[kMaxWalls], why[kMaxWalls];
    int obj[kNumDSlots]; V3 pos[kNumDSlots]; Q rot[kNumDSlots]; int resp[kNumDSlots]; int owner[kNumDSlots];
    V3 lin[kNumDSlots];
    AABB box[kAgentSlot0];           // world AABB of every placed box / ramp (what placement_free tests against), computed once
    int numPlanes; V3 pn[kMaxPlanes]; float pd[kMaxPlanes];
    int numHiders, numSeekers, numActiveAgents, numActiveBoxes, numActiveRamps, seekersFirst;
    int hiders[3], seekers[3]; int agentType[kMaxAgents];
};

HSD void gen_put(GenWorld &g, int slot, V3 p, Q r, int obj, int resp = RESP_DYNAMIC, int owner = OWNER_NONE) {
    g.obj[slot] = obj; g.pos[slot] = p; g.rot[slot] = r; g.resp[slot] = resp; g.owner[slot] = owner;
    g.lin[slot] = {0.f, 0.f, 0.f};
    if (slot < kAgentSlot0) g.box[slot] = aabb_apply_tr(object_aabb(obj), p, r);
}
HSD void gen_agent(GenWorld &g, V3 p, Q r, int type) {          // makeAgent level_gen.cpp:12-66
    int idx = g.numActiveAgents++;
    g.agentType[idx] = type;
    gen_put(g, kAgentSlot0 + idx, p, r, type == AGENT_SEEKER ? OBJ_SEEKER : OBJ_HIDER, RESP_DYNAMIC, OWNER_UNOWNABLE);
    if (type == AGENT_SEEKER) g.seekers[g.numSeekers++] = idx; else g.hiders[g.numHiders++] = idx;
}
HSD void gen_plane(GenWorld &g, V3 n, float d) { g.pn[g.numPlanes] = n; g.pd[g.numPlanes] = d; g.numPlanes++; }

HSD bool placement_free(const GenWorld &g, const AABB &a) {     // level_gen.cpp:104-121
    for (int i = 0; i < g.numWalls; ++i) {
        AABB o = {{g.wcx[i] - g.whx[i], g.wcy[i] - g.why[i], 0.f}, {g.wcx[i] + g.whx[i], g.wcy[i] + g.why[i], 2.5f}};
        if (aabb_overlaps(a, o)) return false;
    }
    for (int s = 0; s < kAgentSlot0; ++s) {
        if (g.obj[s] == OBJ_NONE) continue;
        if (aabb_overlaps(a, g.box[s])) return false;        // (= aabb_apply_tr(object_aabb(obj), pos, rot), kept since gen_put)
    }
    return true;
}
HSD void sample_placement(const GenWorld &g, RNG &rng, int obj, V3 *pos_out, Q *rot_out) {   // :123-229
    const float lo = -18.f, diff = 18.f - (-18.f);
    int rejections = 0;
    while (true) {
        float px = lo + rng.sampleUniform() * diff;
        float py = lo + rng.sampleUniform() * diff;
        V3 pos = {px, py, 1.0f};
        float theta = rng.sampleUniform() * kPi;
        Q rot = quat_angle_axis_z(theta);
        AABB a = aabb_apply_tr(object_aabb(obj), pos, rot);
        if (placement_free(g, a) || rejections == 20) { *pos_out = pos; *rot_out = rot; return; }
        rejections++;
    }
}

// The generator's working memory for one world.  The physics kernel, whose tail is the reset, places it in the LDS its
// octet no longer needs (one per lane of the 8 that regenerate; an odd number of words apart, so the 8 never share a bank)
// — as a local variable it lives in scratch memory, and the generator's chains of dependent array accesses then cost a
// round trip to memory each (the step on which all 16 000 worlds regenerate took 2.2 ms).
struct GenScratch { GenWorld g; WallSet ws; int pad[(sizeof(GenWorld) + sizeof(WallSet)) / 4 % 2 == 0 ? 1 : 2]; };
static_assert(sizeof(GenScratch) % 8 == 4, "odd word count");

HSD void gen_training(GenWorld &g, WallSet &ws, RNG &rng, RNG &episode_rng, uint32_t flags, int num_hiders, int num_seekers) {
    int total_boxes = rng.sampleI32(3, 10);
    int num_elongated = rng.sampleI32(3, total_boxes);
    int num_cubes = total_boxes - num_elongated;
    {   // populateStaticGeometry geo_gen.cpp:467-505
        make_walls(ws, rng);
        const float mn = -18.f, range = 18.f - (-18.f);
        g.numWalls = ws.n;
        for (int i = 0; i < ws.n; ++i) {
            Seg s = ws.segs[i];
            s.x1 = mn + range * s.x1; s.y1 = mn + range * s.y1;
            s.x2 = mn + range * s.x2; s.y2 = mn + range * s.y2;
            float cx = 0.5f * (s.x1 + s.x2), cy = 0.5f * (s.y1 + s.y2);
            g.wcx[i] = cx; g.wcy[i] = cy;
            if (seg_horizontal(s)) { g.whx[i] = s.x2 - cx; g.why[i] = 0.2f; }
            else { g.whx[i] = 0.2f; g.why[i] = s.y2 - cy; }
        }
    }
    V3 p; Q r;
    for (int i = 0; i < num_elongated; ++i) { sample_placement(g, rng, OBJ_BOX, &p, &r); gen_put(g, kBoxSlot0 + i, p, r, OBJ_BOX); }
    for (int i = 0; i < num_cubes; ++i) { sample_placement(g, rng, OBJ_CUBE, &p, &r); gen_put(g, kBoxSlot0 + num_elongated + i, p, r, OBJ_CUBE); }
    g.numActiveBoxes = total_boxes;
    for (int i = 0; i < kMaxRamps; ++i) { sample_placement(g, rng, OBJ_RAMP, &p, &r); gen_put(g, kRampSlot0 + i, p, r, OBJ_RAMP); }
    g.numActiveRamps = kMaxRamps;
    bool seekers_first = episode_rng.sampleI32(0, 2) == 1;       // level_gen.cpp:232-238
    if ((flags & FLAG_RANDOM_FLIP_TEAMS) != FLAG_RANDOM_FLIP_TEAMS) seekers_first = false;
    g.seekersFirst = seekers_first ? 1 : 0;
    for (int t = 0; t < 2; ++t) {
        int type = (t == 0) == seekers_first ? AGENT_SEEKER : AGENT_HIDER;
        int cnt = type == AGENT_SEEKER ? num_seekers : num_hiders;
        for (int i = 0; i < cnt; ++i) {
            sample_placement(g, rng, type == AGENT_SEEKER ? OBJ_SEEKER : OBJ_HIDER, &p, &r);
            gen_agent(g, p, r, type);
        }
    }
    gen_plane(g, {0.f, 0.f, 1.f}, 0.f);
}

HSD void gen_debug(GenWorld &g, int level) {                    // level_gen.cpp:336-526
    const Q ident = {1.f, 0.f, 0.f, 0.f};
    const V3 up = {0.f, 0.f, 1.f};
    if (level == 2) { gen_put(g, 0, {0, 0, 5}, {0.880476236f, 0.364705205f, 0.279848129f, -0.115916893f}, OBJ_CUBE); gen_plane(g, up, 0.f); }
    else if (level == 3) { gen_put(g, 0, {0, 0, 5}, ident, OBJ_CUBE); gen_plane(g, up, 0.f); }
    else if (level == 4) { gen_put(g, 0, {0, 0, 10}, {0.923879504f, 0.f, 0.382683426f, 0.f}, OBJ_BOX); gen_plane(g, up, 0.f); }
    else if (level == 5) { gen_plane(g, up, 0.f); gen_agent(g, {0, 0, 1}, ident, AGENT_HIDER); }
    else if (level == 6) {
        gen_plane(g, up, 0.f);
        g.wcx[0] = 0.f; g.wcy[0] = 0.f; g.whx[0] = 10.f; g.why[0] = 0.2f; g.numWalls = 1;
        gen_put(g, 0, {0, -5, 1}, ident, OBJ_CUBE);
        gen_agent(g, {-15, -15, 1.5f}, {0.923879504f, 0.f, 0.f, -0.382683426f}, AGENT_HIDER);
        gen_agent(g, {-15, -10, 1.5f}, {0.923879504f, 0.f, 0.f, 0.382683426f}, AGENT_SEEKER);
    } else if (level == 7) {
        const Q rot = {0.868162751f, 0.315985411f, 0.359604806f, -0.130885437f};
        gen_put(g, 0, {0, 0, 5}, rot, OBJ_CUBE);
        gen_put(g, 1, {0, 0, 10}, rot, OBJ_CUBE);
        gen_plane(g, up, 0.f); gen_plane(g, {1.f, 0.f, 0.f}, -20.f); gen_plane(g, {-1.f, 0.f, 0.f}, -20.f);
    } else if (level == 8) {
        gen_put(g, kRampSlot0, {0, 0, 10}, {0.579227984f, 0.405579776f, 0.405579776f, 0.579227984f}, OBJ_RAMP);
        g.lin[kRampSlot0] = {0.f, 0.f, -30.f};
        gen_put(g, kRampSlot0 + 1, {-0.5f, -0.5f, 1.f}, {0.f, 0.f, 0.707106769f, -0.707106769f}, OBJ_RAMP, RESP_STATIC, OWNER_NONE);
        gen_plane(g, up, 0.f); gen_plane(g, {1.f, 0.f, 0.f}, -20.f); gen_plane(g, {-1.f, 0.f, 0.f}, -20.f);
    }
}

// Regenerate world w: resetSystem's branch (sim.cpp:185-194) when ck == nullptr, loadCheckpointSystem
// (sim.cpp:956-1044) otherwise — same level generator, but the episode key, agent counts and episode
// step come from the checkpoint, the world's episode counter is not advanced and the saved body /
// joint state is written over the generated spawn state.
template <bool LOAD>
HSD void regenerate_world(const SimState &S, int w, int level, const hs_checkpoint *ck, GenScratch &scratch) {
    const int N = S.N;
    const int ps = S.slotOfWorld[w];        // the world's slot in the tiled columns (hs_state.h)
    uint32_t ep, world_id;
    if (LOAD) {
        ep = ck->episode_key[0]; world_id = ck->episode_key[1];
    } else {
        // ---- resetEnvironment + initEpisodeRNG (sim.cpp:105-159)
        S.xReset[w] = 0;
        ep = S.curWorldEpisode[w];
        S.curWorldEpisode[w] = ep + 1;
        world_id = (uint32_t)(S.worldOffset + w);
    }
    S.epKeyA[w] = ep; S.epKeyB[w] = world_id;
    RNG erng; erng.k = threefry2x32(S.initKey, ep, world_id); erng.count = 0;
    int nh = erng.sampleI32(S.minHiders, S.maxHiders + 1);     // burned on load (sim.cpp:976-980)
    int ns = erng.sampleI32(S.minSeekers, S.maxSeekers + 1);
    if (LOAD) {        // counts outside the build's capacities are clamped (the reference only asserts)
        const int capH = S.A < 3 ? S.A : 3;
        nh = ck->num_hiders < 0 ? 0 : (ck->num_hiders > capH ? capH : ck->num_hiders);
        const int capS = S.A - nh < 3 ? S.A - nh : 3;
        ns = ck->num_seekers < 0 ? 0 : (ck->num_seekers > capS ? capS : ck->num_seekers);
    }
    RandKey lvl = erng.randKey();
    if ((S.flags & FLAG_USE_FIXED_WORLD) == FLAG_USE_FIXED_WORLD) lvl = {0u, 0u};

    GenWorld &g = scratch.g;
    g.numWalls = 0; g.numPlanes = 0;
    g.numHiders = g.numSeekers = g.numActiveAgents = g.numActiveBoxes = g.numActiveRamps = 0;
    g.seekersFirst = cnt_seekers_first(S.counts[w]);   // TeamState persists across debug levels
    for (int i = 0; i < 3; ++i) { g.hiders[i] = 0; g.seekers[i] = 0; }
    for (int i = 0; i < kMaxAgents; ++i) g.agentType[i] = team_agent_type(S.teams[w], i);
    for (int i = 0; i < kNumDSlots; ++i) {
        g.obj[i] = OBJ_NONE; g.pos[i] = {0.f, 0.f, 0.f}; g.rot[i] = {1.f, 0.f, 0.f, 0.f};
        g.resp[i] = RESP_STATIC; g.owner[i] = OWNER_NONE; g.lin[i] = {0.f, 0.f, 0.f};
    }
    RNG lrng; lrng.k = lvl; lrng.count = 0;
    if (LOAD || level == 1) gen_training(g, scratch.ws, lrng, erng, S.flags, nh, ns);
    else gen_debug(g, level);

    // ---- write back
    S.rngKeyA[w] = erng.k.a; S.rngKeyB[w] = erng.k.b; S.rngCount[w] = erng.count;
    S.curEpisodeStep[w] = LOAD ? ck->episode_step : 0;
    if (!LOAD) S.hiderTeamReward[w] = 1.f;        // resetSystem sim.cpp:199; the load graph leaves it alone
    S.numWalls[w] = g.numWalls; S.numPlanes[w] = g.numPlanes;
    for (int i = 0; i < g.numWalls; ++i) {
        S.walls(0 * kMaxWalls + i, ps) = g.wcx[i]; S.walls(1 * kMaxWalls + i, ps) = g.wcy[i];
        S.walls(2 * kMaxWalls + i, ps) = g.whx[i]; S.walls(3 * kMaxWalls + i, ps) = g.why[i];
    }
    for (int p = 0; p < g.numPlanes; ++p) {
        S.planes(0 * kMaxPlanes + p, ps) = g.pn[p].x; S.planes(1 * kMaxPlanes + p, ps) = g.pn[p].y;
        S.planes(2 * kMaxPlanes + p, ps) = g.pn[p].z; S.planes(3 * kMaxPlanes + p, ps) = g.pd[p];
    }
    for (int i = 0; i < kNumDSlots; ++i) {
        S.bmeta(i, ps) = g.obj[i] == OBJ_NONE ? 0 : meta_pack(g.obj[i], g.resp[i], g.owner[i]);
        S.bpos(0 * kNumDSlots + i, ps) = g.pos[i].x; S.bpos(1 * kNumDSlots + i, ps) = g.pos[i].y;
        S.bpos(2 * kNumDSlots + i, ps) = g.pos[i].z;
        S.brot(0 * kNumDSlots + i, ps) = g.rot[i].w; S.brot(1 * kNumDSlots + i, ps) = g.rot[i].x;
        S.brot(2 * kNumDSlots + i, ps) = g.rot[i].y; S.brot(3 * kNumDSlots + i, ps) = g.rot[i].z;
        S.blin(0 * kNumDSlots + i, ps) = g.lin[i].x; S.blin(1 * kNumDSlots + i, ps) = g.lin[i].y;
        S.blin(2 * kNumDSlots + i, ps) = g.lin[i].z;
        S.bang(0 * kNumDSlots + i, ps) = 0.f; S.bang(1 * kNumDSlots + i, ps) = 0.f;
        S.bang(2 * kNumDSlots + i, ps) = 0.f;
    }
    int teams = 0;
    for (int i = 0; i < 3; ++i) { teams |= (g.hiders[i] & 7) << (3 * i); teams |= (g.seekers[i] & 7) << (9 + 3 * i); }
    for (int i = 0; i < kMaxAgents; ++i) {
        teams |= (g.agentType[i] & 1) << (18 + i);
        if (i < g.numActiveAgents) teams |= 1 << (24 + i);
        S.grabOther(i, ps) = -1;
        for (int c = 0; c < 4; ++c) S.aforce(c * kMaxAgents + i, ps) = 0.f;
    }
    S.teams[w] = teams;
    S.counts[w] = cnt_pack(g.numHiders, g.numSeekers, g.numActiveAgents, g.numActiveBoxes, g.numActiveRamps, g.seekersFirst);
    // exported agent-interface columns (makeAgent level_gen.cpp:16-32; generateEnvironment :326-333)
    const int A = S.A;
    for (int i = 0; i < A; ++i) {
        int row = w * A + i;
        if (i < g.numActiveAgents) {
            S.xSelfType[row] = g.agentType[i];
            S.xSelfMask[row] = 1.f;
            S.xSeed[row * 2 + 0] = (int32_t)ep; S.xSeed[row * 2 + 1] = (int32_t)world_id;
            int32_t *a = S.xAction + row * 5;
            a[0] = 2; a[1] = 2; a[2] = 2; a[3] = 0; a[4] = 0;
        } else {
            S.xSelfMask[row] = 0.f;
        }
    }
    if (!LOAD) return;
    // ---- loadCheckpointSystem sim.cpp:973, 985-1043
    S.runningScores(0, ps) = ck->running_scores[0]; S.runningScores(1, ps) = ck->running_scores[1];
    auto put_body = [&](int slot, const float *b) {       // pos3 rot4 lin3 ang3
        for (int c = 0; c < 3; ++c) {
            S.bpos(c * kNumDSlots + slot, ps) = b[c];
            S.blin(c * kNumDSlots + slot, ps) = b[7 + c];
            S.bang(c * kNumDSlots + slot, ps) = b[10 + c];
        }
        for (int c = 0; c < 4; ++c) S.brot(c * kNumDSlots + slot, ps) = b[3 + c];
    };
    const int nb = ck->num_boxes < 0 ? 0 : (ck->num_boxes > g.numActiveBoxes ? g.numActiveBoxes : ck->num_boxes);
    const int nr = ck->num_ramps < 0 ? 0 : (ck->num_ramps > g.numActiveRamps ? g.numActiveRamps : ck->num_ramps);
    for (int i = 0; i < nb + nr; ++i) {
        const hs_ckpt_object &o = i < nb ? ck->boxes[i] : ck->ramps[i - nb];
        const int slot = i < nb ? i : kRampSlot0 + (i - nb);
        put_body(slot, o.pos);
        S.bmeta(slot, ps) = meta_pack(g.obj[slot], o.is_locked ? RESP_STATIC : RESP_DYNAMIC, (int)(o.team & 3u));
    }
    for (int i = 0; i < g.numHiders + g.numSeekers; ++i) {
        const hs_ckpt_agent &a = ck->agents[i];
        const int ai = i < g.numHiders ? g.hiders[i] : g.seekers[i - g.numHiders];
        put_body(kAgentSlot0 + ai, a.pos);
        if (a.grab_idx >= 0 && a.grab_idx < nb + nr) {
            S.grabOther(ai, ps) = a.grab_idx < nb ? a.grab_idx : kRampSlot0 + (a.grab_idx - nb);
            float gd[kGrabWords] = {a.grab_r2[0], a.grab_r2[1], a.grab_r2[2],
                                    a.attach_rot2[0], a.attach_rot2[1], a.attach_rot2[2], a.attach_rot2[3], a.separation,
                                    a.grab_r1[0], a.grab_r1[1], a.grab_r1[2],
                                    a.attach_rot1[0], a.attach_rot1[1], a.attach_rot1[2], a.attach_rot1[3]};
            for (int c = 0; c < kGrabWords; ++c) S.grabData(c * kMaxAgents + ai, ps) = gd[c];
        }
    }
}

// Keeps SimState::slotHdr in step with the per-world scalars it mirrors.
HSD void write_slot_hdr(const SimState &S, int w, int ps) {
    // (the step counter is unbounded under IgnoreEpisodeLength, sim.cpp:196; k_observe only asks "step <= 96" and
    // "96 - step", so the header carries it saturated at 0x7fff instead of letting it overflow the 16 bits)
    const int step = S.curEpisodeStep[w];
    S.slotHdr[ps] = make_int4(w, S.numWalls[w] | (S.numPlanes[w] << 8) | ((step < 0x7fff ? step : 0x7fff) << 16), S.counts[w], S.teams[w]);
}

// resetSystem (src/sim.cpp:172-200) for one world
HSD void reset_world(const SimState &S, int w, GenScratch &scratch) {
    int level = S.xReset[w];
    const int step = S.curEpisodeStep[w];
    if ((S.flags & FLAG_IGNORE_EPISODE_LENGTH) != FLAG_IGNORE_EPISODE_LENGTH && step == kEpisodeLen - 1) level = 1;
    if (level == 0) {
        S.curEpisodeStep[w] = step + 1;
        S.hiderTeamReward[w] = 1.f;
        write_slot_hdr(S, w, S.slotOfWorld[w]);
        return;
    }
    regenerate_world<false>(S, w, level, nullptr, scratch);
    write_slot_hdr(S, w, S.slotOfWorld[w]);
}

// Stand-alone launch: Manager::init (the Init task graph has no physics in front of the reset).  In a step the
// reset runs at the end of k_physics, by the workgroup that owns the world.
__global__ void __launch_bounds__(64) k_reset(SimState S) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= S.N) return;
    GenScratch scratch;
    reset_world(S, w, scratch);
}

// lidarSystem angles (sim.cpp:727-738): the same 30 values for every agent of every world; once per simulator.
__global__ void __launch_bounds__(64) k_lidar_table(SimState S) {
    const int k = threadIdx.x;
    if (k >= 30) return;
    const float theta = 2.f * kPi * ((float)k / 30.f) + kPi / 2.f;
    hs_sincosf(theta, &S.lidarSinCos[k], &S.lidarSinCos[30 + k]);
}

// LoadCheckpoints graph, first node (sim.cpp:1324-1329); the trigger is left at 1 as sim.cpp:963 does.
__global__ void __launch_bounds__(64) k_load_ckpt(SimState S) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= S.N) return;
    if (S.xCkptCtrl[w] == 0) return;
    S.xCkptCtrl[w] = 1;
    GenScratch scratch;
    regenerate_world<true>(S, w, 1, (const hs_checkpoint *)S.xCkpt + w, scratch);
    write_slot_hdr(S, w, S.slotOfWorld[w]);
}

// SaveCheckpoints graph (sim.cpp:1315-1322): saveCheckpointSystem sim.cpp:1046-1137.
__global__ void __launch_bounds__(64) k_save_ckpt(SimState S) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = S.N;
    if (w >= N) return;
    if (S.xCkptCtrl[w] == 0) return;
    S.xCkptCtrl[w] = 0;
    const int ps = S.slotOfWorld[w];
    hs_checkpoint *ck = (hs_checkpoint *)S.xCkpt + w;
    uint32_t *raw = (uint32_t *)ck;
    for (int i = 0; i < (int)(sizeof(hs_checkpoint) / 4); ++i) raw[i] = 0u;
    ck->episode_key[0] = S.epKeyA[w]; ck->episode_key[1] = S.epKeyB[w];
    ck->running_scores[0] = S.runningScores(0, ps); ck->running_scores[1] = S.runningScores(1, ps);
    ck->episode_step = S.curEpisodeStep[w];
    const int cnt = S.counts[w], teams = S.teams[w];
    const int nh = cnt_hiders(cnt), ns = cnt_seekers(cnt), nb = cnt_boxes(cnt), nr = cnt_ramps(cnt);
    auto get_body = [&](int slot, float *b) {
        for (int c = 0; c < 3; ++c) {
            b[c] = S.bpos(c * kNumDSlots + slot, ps);
            b[7 + c] = S.blin(c * kNumDSlots + slot, ps);
            b[10 + c] = S.bang(c * kNumDSlots + slot, ps);
        }
        for (int c = 0; c < 4; ++c) b[3 + c] = S.brot(c * kNumDSlots + slot, ps);
    };
    for (int i = 0; i < nh + ns; ++i) {
        const int ai = i < nh ? team_hider(teams, i) : team_seeker(teams, i - nh);
        hs_ckpt_agent &a = ck->agents[i];
        get_body(kAgentSlot0 + ai, a.pos);
        a.grab_idx = -1;
        const int other = S.grabOther(ai, ps);
        if (other >= 0) {
            float gd[kGrabWords];
            for (int c = 0; c < kGrabWords; ++c) gd[c] = S.grabData(c * kMaxAgents + ai, ps);
            for (int c = 0; c < 3; ++c) { a.grab_r2[c] = gd[c]; a.grab_r1[c] = gd[8 + c]; }
            for (int c = 0; c < 4; ++c) { a.attach_rot2[c] = gd[3 + c]; a.attach_rot1[c] = gd[11 + c]; }
            a.separation = gd[7];
            if (other < nb) a.grab_idx = other;
            else if (other >= kRampSlot0 && other < kRampSlot0 + nr) a.grab_idx = other - kRampSlot0 + nb;
        }
    }
    ck->num_hiders = nh; ck->num_seekers = ns;
    for (int i = 0; i < nb + nr; ++i) {
        const int slot = i < nb ? i : kRampSlot0 + (i - nb);
        hs_ckpt_object &o = i < nb ? ck->boxes[i] : ck->ramps[i - nb];
        get_body(slot, o.pos);
        const int m = S.bmeta(slot, ps);
        o.team = (uint32_t)meta_owner(m);
        o.is_locked = meta_resp(m) == RESP_STATIC ? 1 : 0;
    }
    ck->num_boxes = nb; ck->num_ramps = nr;
}

}  // namespace hs
