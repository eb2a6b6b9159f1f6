// ORACLE — TEST INFRASTRUCTURE ONLY (see hs_ref_math.hpp header).
// Episode reset: procedural walls (src/geo_gen.cpp) and entity placement (src/level_gen.cpp).
// Draw order follows SURVEY Appendix A exactly; the generator itself is the build's own
// (hs_ref_rng.hpp, parity unpinned).
#pragma once
#include "hs_ref_world.hpp"

namespace hsref {

// Batch-wide exported columns (row-major, agent row = world*A + slot).  mgr.cpp:1062-1331.
struct Exports {
    int32_t A;  // maxAgentsPerWorld
    int32_t *reset, *prep, *action, *selfType, *seed, *done, *policy;
    float *selfObs, *selfMask, *agentObs, *boxObs, *rampObs, *visAgents, *visBoxes, *visRamps;
    float *lidar, *reward, *globalPos, *episodeResult;
};

// Object-space AABBs of the collision hulls (data/*_collision.obj vertex extrema,
// SURVEY Appendix A; engine `rigidBodyAABBs` unverifiable — no padding assumed).
static inline AABB object_aabb(int32_t obj) {
    switch (obj) {
    case OBJ_WALL: return {{-1.f, -1.f, 0.f}, {1.f, 1.f, 2.5f}};
    case OBJ_RAMP: return {{-1.f, -2.f, -1.f}, {1.f, 1.f, 1.f}};
    case OBJ_BOX:  return {{-4.f, -0.75f, -1.f}, {4.f, 0.75f, 1.f}};
    default:       return {{-1.f, -1.f, -1.f}, {1.f, 1.f, 1.f}};  // cube, hider, seeker
    }
}

// ----------------------------------------------------------------------------------------
// geo_gen.cpp: wall layout on the unit square
// ----------------------------------------------------------------------------------------
struct Seg {               // geo_gen.cpp:52-87 `Wall`; invariant p1 <= p2
    float x1, y1, x2, y2;
};
static inline Seg seg_make(float ax, float ay, float bx, float by) {   // geo_gen.cpp:60-65
    if (ax > bx || ay > by) return {bx, by, ax, ay};
    return {ax, ay, bx, by};
}
static inline void seg_resort(Seg &s) {                                // geo_gen.cpp:71-77
    if (s.x1 > s.x2 || s.y1 > s.y2) { Seg t = s; s = {t.x2, t.y2, t.x1, t.y1}; }
}
static inline bool seg_horizontal(const Seg &s) {                      // geo_gen.cpp:67-69
    return fabsf(s.y1 - s.y2) < 0.000001f;
}
static inline float seg_length(const Seg &s) {                         // geo_gen.cpp:79-86
    return seg_horizontal(s) ? (s.x2 - s.x1) : (s.y2 - s.y1);
}

struct WallSet {           // geo_gen.cpp:139-175 `Walls`
    Seg segs[kMaxWalls]; int n;
    uint8_t horiz[kMaxWalls]; int nh;
    uint8_t vert[kMaxWalls]; int nv;
    bool overflow;
};
static inline int wallset_add(WallSet &w, Seg s) {                     // geo_gen.cpp:150-161
    if (w.n >= kMaxWalls) { w.overflow = true; return w.n - 1; }
    if (seg_horizontal(s)) w.horiz[w.nh++] = (uint8_t)w.n;
    else w.vert[w.nv++] = (uint8_t)w.n;
    w.segs[w.n++] = s;
    return w.n - 1;
}

// geo_gen.cpp:177-270.  `hz` selects which coordinate plays "along" vs "across".
static inline int find_facing_wall(const WallSet &w, const uint8_t *list, int ln, int chosen_i,
                                   bool hz, RNG &rng) {
    auto lo_al = [&](const Seg &s) { return hz ? s.x1 : s.y1; };
    auto hi_al = [&](const Seg &s) { return hz ? s.x2 : s.y2; };
    auto across = [&](const Seg &s) { return hz ? s.y1 : s.x1; };
    const float min_len = hz ? 0.3f : 0.5f;
    const Seg &c = w.segs[list[chosen_i]];
    int start = chosen_i + 1 + rng.sampleI32(0, ln - 1);
    for (int i = 0; i < ln - 1; ++i) {
        int cur = (start + i) % ln;
        if (cur == chosen_i) cur = (cur + 1) % ln;
        const Seg &o = w.segs[list[cur]];
        if (!(lo_al(c) >= hi_al(o) || hi_al(c) <= lo_al(o)) && seg_length(c) >= min_len &&
            seg_length(o) >= min_len) {
            float high = fminf(hi_al(c), hi_al(o));
            float low = fmaxf(lo_al(c), lo_al(o));
            bool works = true;
            for (int j = 0; j < ln; ++j) {
                if (j == cur) continue;
                const Seg &b = w.segs[list[j]];
                float bl = fmaxf(lo_al(b), low - 0.1f);
                float bh = fminf(hi_al(b), high + 0.1f);
                if (bl < bh) {
                    float v = across(b);
                    float vmin = fminf(across(c), across(o));
                    float vmax = fmaxf(across(c), across(o));
                    if (v > vmin && v < vmax) { works = false; break; }
                }
            }
            if (works) return cur;
        }
    }
    return -1;
}

// geo_gen.cpp:275-307
static inline void cut_door(WallSet &w, int wi, float door, RNG &rng) {
    Seg &s = w.segs[wi];
    float rat = 0.3f + rng.sampleUniform() * 0.4f;
    if (seg_horizontal(s)) {
        float low = s.x1 + door, high = s.x2 - door;
        float x = low + rat * (high - low);
        float old = s.x2;
        s.x2 = x - door * 0.5f;
        seg_resort(s);
        Seg ns = seg_make(x + door * 0.5f, s.y1, old, s.y1);
        wallset_add(w, ns);
    } else {
        float low = s.y1 + door, high = s.y2 - door;
        float y = low + rat * (high - low);
        float old = s.y2;
        s.y2 = y - door * 0.5f;
        seg_resort(s);
        Seg ns = seg_make(s.x1, y + door * 0.5f, s.x1, old);
        wallset_add(w, ns);
    }
}

// geo_gen.cpp:309-427
static inline void wall_op_connect(WallSet &w, RNG &rng) {
    bool hz = rng.sampleI32(0, 2) != 0;
    const uint8_t *list = hz ? w.horiz : w.vert;
    int ln = hz ? w.nh : w.nv;
    int wi = rng.sampleI32(0, ln);
    int oi;
    int counter = 0;
    while ((oi = find_facing_wall(w, list, ln, wi, hz, rng)) == -1) {
        hz = rng.sampleI32(0, 2) != 0;
        list = hz ? w.horiz : w.vert;
        ln = hz ? w.nh : w.nv;
        wi = rng.sampleI32(0, ln);
        if (counter++ > 4) return;
    }
    int fi = list[wi], si = list[oi];        // indices into segs (stable across add)
    const float kDoor = 0.1f;
    if (hz) {
        float high = fminf(w.segs[fi].x2, w.segs[si].x2);
        float low = fmaxf(w.segs[fi].x1, w.segs[si].x1);
        if (w.segs[fi].y1 > w.segs[si].y1) { int t = fi; fi = si; si = t; }
        float rat = 0.4f + rng.sampleUniform() * 0.2f;
        float x = low + rat * (high - low);
        int ni = wallset_add(w, seg_make(x, w.segs[fi].y1, x, w.segs[si].y1));
        Seg &f = w.segs[fi]; Seg &s = w.segs[si];
        float fold = f.x2, sold = s.x2;
        f.x2 = x; seg_resort(f);
        s.x2 = x; seg_resort(s);
        Seg n0 = seg_make(x, f.y1, fold, f.y1);
        Seg n1 = seg_make(x, s.y1, sold, s.y1);
        wallset_add(w, n0);
        wallset_add(w, n1);
        cut_door(w, ni, kDoor, rng);
    } else {
        float high = fminf(w.segs[fi].y2, w.segs[si].y2);
        float low = fmaxf(w.segs[fi].y1, w.segs[si].y1);
        if (w.segs[fi].x1 > w.segs[si].x1) { int t = fi; fi = si; si = t; }
        float rat = 0.4f + rng.sampleUniform() * 0.2f;
        float y = low + rat * (high - low);
        int ni = wallset_add(w, seg_make(w.segs[fi].x1, y, w.segs[si].x1, y));
        Seg &f = w.segs[fi]; Seg &s = w.segs[si];
        float fold = f.y2, sold = s.y2;
        f.y2 = y; seg_resort(f);
        s.y2 = y; seg_resort(s);
        Seg n0 = seg_make(f.x1, y, f.x1, fold);
        Seg n1 = seg_make(s.x1, y, s.x1, sold);
        wallset_add(w, n0);
        wallset_add(w, n1);
        cut_door(w, ni, kDoor, rng);
    }
}

static inline void wall_op_add_door(WallSet &w, RNG &rng) {            // geo_gen.cpp:411-421
    const float door = 0.1f * 2.0f;
    int wi = rng.sampleI32(0, w.n);
    if (seg_length(w.segs[wi]) > 3.0f * door) cut_door(w, wi, door, rng);
}

// geo_gen.cpp:429-465 (+ WallOperationSelection :96-137)
static inline void make_walls(WallSet &w, RNG &rng) {
    w.n = w.nh = w.nv = 0; w.overflow = false;
    wallset_add(w, seg_make(0.f, 0.f, 1.f, 0.f));
    wallset_add(w, seg_make(0.f, 0.f, 0.f, 1.f));
    wallset_add(w, seg_make(0.f, 1.f, 1.f, 1.f));
    wallset_add(w, seg_make(1.f, 1.f, 1.f, 0.f));
    int counts[2];
    counts[0] = 1 + rng.sampleI32(0, 6);   // connect + door
    counts[1] = 4 + rng.sampleI32(0, 3);   // door only
    int ops[2] = {0, 1};
    int nsel = 2;                          // both counts start > 0
    do {
        int oi = rng.sampleI32(0, nsel);
        int op = ops[oi];
        if (--counts[op] == 0) { --nsel; ops[oi] = ops[nsel]; }
        if (op == 0) wall_op_connect(w, rng); else wall_op_add_door(w, rng);
    } while (counts[0] != 0 || counts[1] != 0);
}

// geo_gen.cpp:467-505 (+ Walls::scale :163-174)
static inline void populate_static_geometry(World &wd, RNG &rng, float level_scale) {
    WallSet w;
    make_walls(w, rng);
    const float mn = -level_scale, range = level_scale - (-level_scale);
    wd.numWalls = w.n;
    for (int i = 0; i < w.n; ++i) {
        Seg s = w.segs[i];
        s.x1 = mn + range * s.x1; s.y1 = mn + range * s.y1;
        s.x2 = mn + range * s.x2; s.y2 = mn + range * s.y2;
        float cx = 0.5f * (s.x1 + s.x2), cy = 0.5f * (s.y1 + s.y2);
        WallS ws;
        ws.cx = cx; ws.cy = cy;
        if (seg_horizontal(s)) { ws.hx = s.x2 - cx; ws.hy = 0.2f; }
        else { ws.hx = 0.2f; ws.hy = s.y2 - cy; }
        wd.walls[i] = ws;
    }
}

// ----------------------------------------------------------------------------------------
// level_gen.cpp
// ----------------------------------------------------------------------------------------
static inline void clear_dbody(DBody &b) {
    b = DBody{};
    b.objType = OBJ_NONE; b.response = RESP_STATIC; b.owner = OWNER_NONE;
    b.rot = {1.f, 0.f, 0.f, 0.f}; b.prevRot = b.rot;
}

// makeDynObject (geo_gen.inl:5-34) into a fixed slot
static inline void make_dyn_object(World &wd, int slot, V3 pos, Q rot, int32_t obj,
                                   int32_t resp = RESP_DYNAMIC, int32_t owner = OWNER_NONE) {
    DBody &b = wd.d[slot];
    clear_dbody(b);
    b.objType = obj; b.response = resp; b.owner = owner;
    b.pos = pos; b.rot = rot;
    b.prevPos = pos; b.prevRot = rot;
}

// makeAgent (level_gen.cpp:12-66)
static inline void make_agent(World &wd, Exports &ex, int world, V3 pos, Q rot, int32_t type) {
    int idx = wd.numActiveAgents++;
    int row = world * ex.A + idx;
    wd.agentType[idx] = type;
    ex.selfType[row] = type;
    wd.agentActive[idx] = 1;
    ex.selfMask[row] = 1.f;
    ex.seed[row * 2 + 0] = (int32_t)wd.curEpisodeRNDCounter.a;
    ex.seed[row * 2 + 1] = (int32_t)wd.curEpisodeRNDCounter.b;
    int32_t *act = ex.action + row * 5;
    act[0] = 2; act[1] = 2; act[2] = 2; act[3] = 0; act[4] = 0;
    make_dyn_object(wd, kAgentSlot0 + idx, pos, rot,
                    type == AGENT_SEEKER ? OBJ_SEEKER : OBJ_HIDER, RESP_DYNAMIC, OWNER_UNOWNABLE);
    if (type == AGENT_SEEKER) wd.seekers[wd.numSeekers++] = idx;
    else wd.hiders[wd.numHiders++] = idx;
    wd.grab[idx].other = -1;
}

static inline void add_plane(World &wd, V3 n, float d) {               // makePlane :68-71
    wd.planes[wd.numPlanes++] = {n, d};
}

// checkOverlap lambda (level_gen.cpp:104-121): walls + boxes/ramps placed so far
static inline bool placement_free(const World &wd, const AABB &a) {
    for (int i = 0; i < wd.numWalls; ++i) {
        const WallS &w = wd.walls[i];
        AABB o = {{w.cx - w.hx, w.cy - w.hy, 0.f}, {w.cx + w.hx, w.cy + w.hy, 2.5f}};
        if (aabb_overlaps(a, o)) return false;
    }
    for (int s = 0; s < kAgentSlot0; ++s) {
        const DBody &b = wd.d[s];
        if (b.objType == OBJ_NONE) continue;
        AABB o = aabb_apply_trs(object_aabb(b.objType), b.pos, b.rot, {1.f, 1.f, 1.f});
        if (aabb_overlaps(a, o)) return false;
    }
    return true;
}

// The {x,y,theta} rejection loop shared by boxes, cubes, ramps and agents
// (level_gen.cpp:123-229, 266-292).
static inline void sample_placement(const World &wd, RNG &rng, int32_t obj, V3 *pos_out, Q *rot_out) {
    const float lo = -18.f, diff = 18.f - (-18.f);
    const float pi = 3.14159265358979323846f;
    int rejections = 0;
    while (true) {
        float px = lo + rng.sampleUniform() * diff;
        float py = lo + rng.sampleUniform() * diff;
        V3 pos = {px, py, 1.0f};
        float theta = rng.sampleUniform() * pi;
        Q rot = quat_angle_axis_z(theta);
        AABB a = aabb_apply_trs(object_aabb(obj), pos, rot, {1.f, 1.f, 1.f});
        if (placement_free(wd, a) || rejections == 20) { *pos_out = pos; *rot_out = rot; return; }
        rejections++;
    }
}

// generateTrainingEnvironment (level_gen.cpp:79-308)
static inline void generate_training_level(World &wd, Exports &ex, int world, RNG &rng,
                                           uint32_t flags, int num_hiders, int num_seekers) {
    int total_boxes = rng.sampleI32(3, 10);
    int num_elongated = rng.sampleI32(3, total_boxes);
    int num_cubes = total_boxes - num_elongated;
    populate_static_geometry(wd, rng, 18.f);

    V3 p; Q r;
    for (int i = 0; i < num_elongated; ++i) {
        sample_placement(wd, rng, OBJ_BOX, &p, &r);
        make_dyn_object(wd, kBoxSlot0 + i, p, r, OBJ_BOX);
        wd.boxSizes[i] = {8.f, 1.5f, 2.f};
    }
    for (int i = 0; i < num_cubes; ++i) {
        int bi = i + num_elongated;
        sample_placement(wd, rng, OBJ_CUBE, &p, &r);
        make_dyn_object(wd, kBoxSlot0 + bi, p, r, OBJ_CUBE);
        wd.boxSizes[bi] = {2.f, 2.f, 2.f};
    }
    wd.numActiveBoxes = total_boxes;
    for (int i = 0; i < kMaxRamps; ++i) {
        sample_placement(wd, rng, OBJ_RAMP, &p, &r);
        make_dyn_object(wd, kRampSlot0 + i, p, r, OBJ_RAMP);
    }
    wd.numActiveRamps = kMaxRamps;

    // episode RNG, not the level RNG (level_gen.cpp:232-238)
    bool seekers_first = wd.rng.sampleI32(0, 2) == 1;
    if ((flags & FLAG_RANDOM_FLIP_TEAMS) != FLAG_RANDOM_FLIP_TEAMS) seekers_first = false;
    wd.seekersFirst = seekers_first;

    int team_sizes[2]; int32_t team_types[2];
    if (seekers_first) {
        team_sizes[0] = num_seekers; team_types[0] = AGENT_SEEKER;
        team_sizes[1] = num_hiders;  team_types[1] = AGENT_HIDER;
    } else {
        team_sizes[0] = num_hiders;  team_types[0] = AGENT_HIDER;
        team_sizes[1] = num_seekers; team_types[1] = AGENT_SEEKER;
    }
    for (int t = 0; t < 2; ++t) {
        for (int i = 0; i < team_sizes[t]; ++i) {
            int32_t obj = team_types[t] == AGENT_SEEKER ? OBJ_SEEKER : OBJ_HIDER;
            sample_placement(wd, rng, obj, &p, &r);
            make_agent(wd, ex, world, p, r, team_types[t]);
        }
    }
    add_plane(wd, {0.f, 0.f, 1.f}, 0.f);
}

// generateDebugEnvironment (level_gen.cpp:336-526).  Quaternion constants are the float32
// roundings of the double-precision products written in the reference (tests/golden/gen.py).
static inline void generate_debug_level(World &wd, Exports &ex, int world, int level) {
    const Q ident = {1.f, 0.f, 0.f, 0.f};
    const V3 up = {0.f, 0.f, 1.f};
    switch (level) {
    case 2:
        make_dyn_object(wd, 0, {0, 0, 5}, {0.880476236f, 0.364705205f, 0.279848129f, -0.115916893f}, OBJ_CUBE);
        add_plane(wd, up, 0.f);
        break;
    case 3:
        make_dyn_object(wd, 0, {0, 0, 5}, ident, OBJ_CUBE);
        add_plane(wd, up, 0.f);
        break;
    case 4:
        make_dyn_object(wd, 0, {0, 0, 10}, {0.923879504f, 0.f, 0.382683426f, 0.f}, OBJ_BOX);
        add_plane(wd, up, 0.f);
        break;
    case 5:
        add_plane(wd, up, 0.f);
        make_agent(wd, ex, world, {0, 0, 1}, ident, AGENT_HIDER);
        break;
    case 6:
        add_plane(wd, up, 0.f);
        wd.walls[wd.numWalls++] = {0.f, 0.f, 10.f, 0.2f};
        make_dyn_object(wd, 0, {0, -5, 1}, ident, OBJ_CUBE);
        make_agent(wd, ex, world, {-15, -15, 1.5f}, {0.923879504f, 0.f, 0.f, -0.382683426f}, AGENT_HIDER);
        make_agent(wd, ex, world, {-15, -10, 1.5f}, {0.923879504f, 0.f, 0.f, 0.382683426f}, AGENT_SEEKER);
        break;
    case 7: {
        const Q rot = {0.868162751f, 0.315985411f, 0.359604806f, -0.130885437f};
        make_dyn_object(wd, 0, {0, 0, 5}, rot, OBJ_CUBE);
        make_dyn_object(wd, 1, {0, 0, 10}, rot, OBJ_CUBE);
        add_plane(wd, up, 0.f);
        add_plane(wd, {1.f, 0.f, 0.f}, -20.f);
        add_plane(wd, {-1.f, 0.f, 0.f}, -20.f);
    } break;
    case 8:
        make_dyn_object(wd, kRampSlot0, {0, 0, 10}, {0.579227984f, 0.405579776f, 0.405579776f, 0.579227984f}, OBJ_RAMP);
        wd.d[kRampSlot0].lin = {0.f, 0.f, -30.f};
        make_dyn_object(wd, kRampSlot0 + 1, {-0.5f, -0.5f, 1.f}, {0.f, 0.f, 0.707106769f, -0.707106769f},
                        OBJ_RAMP, RESP_STATIC, OWNER_NONE);
        add_plane(wd, up, 0.f);
        add_plane(wd, {1.f, 0.f, 0.f}, -20.f);
        add_plane(wd, {-1.f, 0.f, 0.f}, -20.f);
        break;
    default: break;
    }
}

// generateEnvironment (level_gen.cpp:312-334)
static inline void generate_environment(World &wd, Exports &ex, int world, RandKey level_key,
                                        int level, uint32_t flags, int num_hiders, int num_seekers) {
    RNG level_rng(level_key);
    if (level == 1) generate_training_level(wd, ex, world, level_rng, flags, num_hiders, num_seekers);
    else generate_debug_level(wd, ex, world, level);
    for (int i = wd.numActiveAgents; i < ex.A; ++i) {
        wd.agentActive[i] = 0;
        ex.selfMask[world * ex.A + i] = 0.f;
    }
}

}  // namespace hsref
