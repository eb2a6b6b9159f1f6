// ORACLE — TEST INFRASTRUCTURE ONLY (see hs_ref_math.hpp header).
// Plain C entry points so tests/ and bench.py's cpu_baseline leg can drive the CPU
// restatement through ctypes.  Export ids mirror ExportID (src/sim.hpp:45-68).
#include "hs_ref_sim.hpp"
#include "hs_ref_ckpt.hpp"
#include "hs_ref_render.hpp"

using namespace hsref;

extern "C" {

struct HsRefConfig {
    int32_t num_worlds;
    uint32_t sim_flags;
    uint32_t rand_seed;
    int32_t min_hiders, max_hiders, min_seekers, max_seekers;
    int32_t world_offset;
    int32_t skip_observations;
    int32_t threads;
};

void *hsref_create(const HsRefConfig *c) {
    if (!c || c->num_worlds <= 0) return nullptr;
    int A = c->max_hiders + c->max_seekers;
    if (A <= 0 || A > kMaxAgents || c->max_hiders > 3 || c->max_seekers > 3) return nullptr;
    Config cfg{c->num_worlds, c->sim_flags, c->rand_seed, c->min_hiders, c->max_hiders,
               c->min_seekers, c->max_seekers, c->world_offset, c->skip_observations};
    Sim *s = new Sim(cfg);
    s->threads = c->threads > 0 ? c->threads : 1;
    return s;
}
void hsref_destroy(void *p) { delete (Sim *)p; }
void hsref_init(void *p) { ((Sim *)p)->init(); }
void hsref_step(void *p) { ((Sim *)p)->step(); }
int32_t hsref_agents_per_world(void *p) { return ((Sim *)p)->A; }
void hsref_save_checkpoints(void *p) { sim_save_checkpoints(*(Sim *)p); }
void hsref_load_checkpoints(void *p) { sim_load_checkpoints(*(Sim *)p); }

// agent views of the current state: depth [N*A][H][W] f32, rgba [N*A][H][W][4] u8 (hs_ref_render.hpp)
void hsref_render(void *p, int32_t W, int32_t H, float *depth, uint8_t *rgba) { render_views(*(Sim *)p, W, H, depth, (uint32_t *)rgba); }

// ExportID order: Reset, PrepCounter, Action, SelfObs, SelfType, SelfMask, AgentObsData,
// BoxObsData, RampObsData, AgentVisMasks, BoxVisMasks, RampVisMasks, Lidar, Seed, Reward, Done,
// GlobalDebugPositions, AgentPolicy, EpisodeResult
void *hsref_tensor(void *p, int32_t id) {
    Sim *s = (Sim *)p;
    switch (id) {
    case 0: return s->ex.reset; case 1: return s->ex.prep; case 2: return s->ex.action;
    case 3: return s->ex.selfObs; case 4: return s->ex.selfType; case 5: return s->ex.selfMask;
    case 6: return s->ex.agentObs; case 7: return s->ex.boxObs; case 8: return s->ex.rampObs;
    case 9: return s->ex.visAgents; case 10: return s->ex.visBoxes; case 11: return s->ex.visRamps;
    case 12: return s->ex.lidar; case 13: return s->ex.seed; case 14: return s->ex.reward;
    case 15: return s->ex.done; case 16: return s->ex.globalPos; case 17: return s->ex.policy;
    case 18: return s->ex.episodeResult;
    case 19: return s->s_ckptCtrl.data(); case 20: return s->s_ckpt.data();
    default: return nullptr;
    }
}

// Internal-state dumps for parity tests.
// bodies: [N][17][13] = pos3 rot4(wxyz) lin3 ang3 ; meta: [N][17][3] = objType, response, owner
void hsref_dump_bodies(void *p, float *bodies, int32_t *meta) {
    Sim *s = (Sim *)p;
    for (int wi = 0; wi < s->cfg.numWorlds; ++wi) {
        const World &w = s->worlds[wi];
        for (int i = 0; i < kNumDSlots; ++i) {
            const DBody &b = w.d[i];
            float *o = bodies + ((size_t)wi * kNumDSlots + i) * 13;
            o[0] = b.pos.x; o[1] = b.pos.y; o[2] = b.pos.z;
            o[3] = b.rot.w; o[4] = b.rot.x; o[5] = b.rot.y; o[6] = b.rot.z;
            o[7] = b.lin.x; o[8] = b.lin.y; o[9] = b.lin.z;
            o[10] = b.ang.x; o[11] = b.ang.y; o[12] = b.ang.z;
            int32_t *m = meta + ((size_t)wi * kNumDSlots + i) * 3;
            m[0] = b.objType; m[1] = b.response; m[2] = b.owner;
        }
    }
}
// walls: [N][36][4] = cx, cy, hx, hy ; world_info: [N][8] = numWalls, numPlanes, numActiveBoxes,
// numActiveRamps, numHiders, numSeekers, curEpisodeStep, seekersFirst
void hsref_dump_walls(void *p, float *walls, int32_t *info) {
    Sim *s = (Sim *)p;
    for (int wi = 0; wi < s->cfg.numWorlds; ++wi) {
        const World &w = s->worlds[wi];
        for (int k = 0; k < kMaxWalls; ++k) {
            float *o = walls + ((size_t)wi * kMaxWalls + k) * 4;
            if (k < w.numWalls) { o[0] = w.walls[k].cx; o[1] = w.walls[k].cy; o[2] = w.walls[k].hx; o[3] = w.walls[k].hy; }
            else { o[0] = o[1] = o[2] = o[3] = 0.f; }
        }
        int32_t *m = info + (size_t)wi * 8;
        m[0] = w.numWalls; m[1] = w.numPlanes; m[2] = w.numActiveBoxes; m[3] = w.numActiveRamps;
        m[4] = w.numHiders; m[5] = w.numSeekers; m[6] = w.curEpisodeStep; m[7] = w.seekersFirst ? 1 : 0;
    }
}

// Unit-test hooks
void hsref_threefry(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t *out) {
    RandKey r = threefry2x32({k0, k1}, c0, c1); out[0] = r.a; out[1] = r.b;
}
void hsref_rng_draws(uint32_t k0, uint32_t k1, int32_t n, uint32_t *bits_out) {
    RNG r(RandKey{k0, k1});
    for (int i = 0; i < n; ++i) bits_out[i] = r.bits32();
}
int32_t hsref_sample_i32(uint32_t k0, uint32_t k1, int32_t a, int32_t b) { RNG r(RandKey{k0, k1}); return r.sampleI32(a, b); }
void hsref_sincos(float x, float *s, float *c) { hs_sincosf(x, s, c); }
float hsref_atan2(float y, float x) { return hs_atan2f(y, x); }
float hsref_asin(float x) { return hs_asinf(x); }
void hsref_quat_to_euler(const float *q, float *out) {
    V3 e = quat_to_euler({q[0], q[1], q[2], q[3]}); out[0] = e.x; out[1] = e.y; out[2] = e.z;
}
// ray vs a single body placed at pos/rot: returns entry t or -1
float hsref_ray_body(int32_t obj, const float *pos, const float *rot, const float *o, const float *d) {
    Q q = {rot[0], rot[1], rot[2], rot[3]};
    Q qi = qinv(q);
    V3 ol = qrot(qi, V3{o[0] - pos[0], o[1] - pos[1], o[2] - pos[2]});
    V3 dl = qrot(qi, V3{d[0], d[1], d[2]});
    return obj == OBJ_RAMP ? ray_wedge_local(ol, dl) : ray_box_local(ol, dl, obj_half_extents(obj));
}
// convex-convex collision of two bodies; returns np, writes n[3], pA[4][3], pB[4][3]
int32_t hsref_collide(int32_t objA, const float *posA, const float *rotA, int32_t objB, const float *posB,
                      const float *rotB, float *n, float *pA, float *pB) {
    Hull a, b;
    hull_from_body(a, objA, {posA[0], posA[1], posA[2]}, {rotA[0], rotA[1], rotA[2], rotA[3]});
    hull_from_body(b, objB, {posB[0], posB[1], posB[2]}, {rotB[0], rotB[1], rotB[2], rotB[3]});
    RawManifold m;
    if (!collide_hulls(a, b, m)) return 0;
    n[0] = m.n.x; n[1] = m.n.y; n[2] = m.n.z;
    for (int i = 0; i < m.np; ++i) {
        pA[i * 3] = m.pA[i].x; pA[i * 3 + 1] = m.pA[i].y; pA[i * 3 + 2] = m.pA[i].z;
        pB[i * 3] = m.pB[i].x; pB[i * 3 + 1] = m.pB[i].y; pB[i * 3 + 2] = m.pB[i].z;
    }
    return m.np;
}

// The oracle's own hull tables for one SimObject, as its collision code sees them (identity pose; a unit wall for
// OBJ_WALL), plus the object-space AABB the level generator uses: tests/test_oracle_hulls.py pins them to
// tests/golden/hulls.npz, i.e. to data/*_collision.obj of the reference.
// verts [8][3], faces [6][4] (-1 padded), counts {nv, nf, ne, ned}, normals [6][3], edges [12][3] = v0 v1 dir, aabb [6]
void hsref_hull_tables(int32_t obj, float *verts, int32_t *faces, int32_t *counts, float *normals, int32_t *edges, float *aabb) {
    Hull h;
    if (obj == OBJ_WALL) { WallS w = {0.f, 0.f, 1.f, 1.f}; hull_from_wall(h, w); }
    else hull_from_body(h, obj, {0.f, 0.f, 0.f}, {1.f, 0.f, 0.f, 0.f});
    counts[0] = h.nv; counts[1] = h.nf; counts[2] = h.ne; counts[3] = h.ned;
    for (int i = 0; i < h.nv; ++i) { verts[i * 3] = h.v[i].x; verts[i * 3 + 1] = h.v[i].y; verts[i * 3 + 2] = h.v[i].z; }
    for (int f = 0; f < h.nf; ++f) {
        for (int k = 0; k < 4; ++k) faces[f * 4 + k] = k < h.fcnt[f] ? h.fidx[f][k] : -1;
        normals[f * 3] = h.fn[f].x; normals[f * 3 + 1] = h.fn[f].y; normals[f * 3 + 2] = h.fn[f].z;
    }
    for (int e = 0; e < h.ne; ++e) { edges[e * 3] = h.e0[e]; edges[e * 3 + 1] = h.e1[e]; edges[e * 3 + 2] = h.edir[e]; }
    AABB a = object_aabb(obj);
    aabb[0] = a.lo.x; aabb[1] = a.lo.y; aabb[2] = a.lo.z; aabb[3] = a.hi.x; aabb[4] = a.hi.y; aabb[5] = a.hi.z;
}

// inverse mass, static / dynamic friction, inverse inertia (object frame) of one SimObject as the solver uses them:
// tests/test_oracle_hulls.py pins the first three and the zeroed inertia axes to tests/golden/object_table.json (mgr.cpp:441-588)
void hsref_object_params(int32_t obj, float *out) {
    V3 i = obj_inv_inertia(obj);
    out[0] = obj_inv_mass(obj); out[1] = obj_mu_s(obj); out[2] = obj_mu_d(obj); out[3] = i.x; out[4] = i.y; out[5] = i.z;
}

}  // extern "C"
