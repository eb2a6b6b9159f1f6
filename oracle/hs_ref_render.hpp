// ORACLE — TEST INFRASTRUCTURE ONLY (see hs_ref_math.hpp header).  PARITY UNPINNED.
// CPU restatement of the agent-view depth / RGB images (Manager::depthTensor / rgbTensor, src/mgr.cpp:1241-1263).
// Madrona's batch renderer is absent from the reference snapshot; first-party source fixes the camera
// (RenderingSystem::attachEntityToView(agent_iface, 100.f, 0.001f, 0.5f * math::up), src/sim.cpp:1400-1403, the
// interface entity taking its agent's pose in updateCameraSystem, :943-954), the base colours per object type
// (src/mgr.cpp:621-647) and the directional light (src/mgr.cpp:657-659).  The image itself
// is this build's own definition (DESIGN.md "Engine decisions"): a ray cast per pixel through trace_ray,
// depth = view-space depth of the closest hit (0: none), colour = base x (0.3 + 0.7 max(0, n . toLight)).
#pragma once
#include "hs_ref_sim.hpp"

namespace hsref {

constexpr float kTanHalfFov = 1.19175359259421f;      // tan(100 degrees / 2)
constexpr float kCamUp = 0.5f, kCamNear = 0.001f, kCamFar = 1000.f;

static inline V3 render_base_colour(int obj, int hit) {
    if (hit >= kHitPlaneBase) return {0.5f, 0.3f, 0.3f};
    if (hit >= kHitWallBase) return {0.4f, 0.4f, 0.4f};
    if (obj == OBJ_CUBE) return {1.f, 0.1f, 0.1f};
    if (obj == OBJ_HIDER) return {1.f, 1.f, 1.f};
    if (obj == OBJ_SEEKER) return {1.f, 0.3f, 0.3f};         // stands in for the red face texture
    if (obj == OBJ_RAMP) return {191.f / 255.f, 108.f / 255.f, 10.f / 255.f};
    if (obj == OBJ_BOX) return {12.f / 255.f, 144.f / 255.f, 150.f / 255.f};
    return {0.4f, 0.4f, 0.4f};
}
static inline V3 box_face_normal(V3 q, V3 e) {
    const float dx = fabsf(q.x) - e.x, dy = fabsf(q.y) - e.y, dz = fabsf(q.z) - e.z;
    if (dx >= dy && dx >= dz) return {q.x < 0.f ? -1.f : 1.f, 0.f, 0.f};
    if (dy >= dz) return {0.f, q.y < 0.f ? -1.f : 1.f, 0.f};
    return {0.f, 0.f, q.z < 0.f ? -1.f : 1.f};
}
static inline V3 wedge_face_normal(V3 q) {
    const float off[5] = {1.f, 1.f, 0.277350098f, 1.f, 1.f};
    int bf = 0; float bd = -3.0e38f;
    for (int f = 0; f < 5; ++f) {
        const float d = dot(V3{kWedgeFN[f][0], kWedgeFN[f][1], kWedgeFN[f][2]}, q) - off[f];
        if (d > bd) { bd = d; bf = f; }
    }
    return {kWedgeFN[bf][0], kWedgeFN[bf][1], kWedgeFN[bf][2]};
}
static inline V3 hit_normal(const World &w, int hit, V3 p) {
    if (hit >= kHitPlaneBase) return w.planes[hit - kHitPlaneBase].n;
    if (hit >= kHitWallBase) {
        const WallS &ws = w.walls[hit - kHitWallBase];
        return box_face_normal({p.x - ws.cx, p.y - ws.cy, p.z - 1.25f}, {ws.hx, ws.hy, 1.25f});
    }
    const DBody &b = w.d[hit];
    const V3 q = qrot(qinv(b.rot), p - b.pos);
    return qrot(b.rot, b.objType == OBJ_RAMP ? wedge_face_normal(q) : box_face_normal(q, obj_half_extents(b.objType)));
}
static inline uint32_t render_shade(V3 base, V3 n) {
    const V3 toLight = {-0.408248290f, -0.408248290f, 0.816496581f};
    const float lam = fmaxf(dot(n, toLight), 0.f);
    const float k = 0.3f + 0.7f * lam;
    const float r = fminf(base.x * k, 1.f), g = fminf(base.y * k, 1.f), b = fminf(base.z * k, 1.f);
    return (uint32_t)(r * 255.f + 0.5f) | ((uint32_t)(g * 255.f + 0.5f) << 8) | ((uint32_t)(b * 255.f + 0.5f) << 16) | 0xff000000u;
}

// depth [N*A][H][W], rgba [N*A][H][W] packed little-endian r, g, b, a
static inline void render_views(const Sim &s, int W, int H, float *depth, uint32_t *rgba) {
    const int A = s.A;
    for (int wi = 0; wi < s.cfg.numWorlds; ++wi) {
        const World &w = s.worlds[wi];
        for (int a = 0; a < A; ++a) {
            float *dv = depth + ((size_t)wi * A + a) * (size_t)(W * H);
            uint32_t *cv = rgba + ((size_t)wi * A + a) * (size_t)(W * H);
            if (!w.agentActive[a] || w.d[kAgentSlot0 + a].objType == OBJ_NONE) {
                for (int i = 0; i < W * H; ++i) { dv[i] = 0.f; cv[i] = 0u; }
                continue;
            }
            const DBody &me = w.d[kAgentSlot0 + a];
            const V3 fwd = qrot(me.rot, {0.f, 1.f, 0.f}), right = qrot(me.rot, {1.f, 0.f, 0.f}), up = qrot(me.rot, {0.f, 0.f, 1.f});
            const V3 o = me.pos + V3{0.f, 0.f, kCamUp};
            const float aspect = (float)W / (float)H;
            for (int py = 0; py < H; ++py)
                for (int px = 0; px < W; ++px) {
                    const float u = ((((float)px + 0.5f) / (float)W) * 2.f - 1.f) * (kTanHalfFov * aspect);
                    const float v = (1.f - (((float)py + 0.5f) / (float)H) * 2.f) * kTanHalfFov;
                    const V3 d = (fwd + right * u) + up * v;
                    float t;
                    const int hit = trace_ray(w, o, d, kCamFar, &t);
                    const int i = py * W + px;
                    if (hit < 0 || t < kCamNear) { dv[i] = 0.f; cv[i] = 0xff000000u; continue; }
                    const V3 p = o + d * t;
                    const int obj = hit < kNumDSlots ? w.d[hit].objType : OBJ_NONE;
                    dv[i] = t;
                    cv[i] = render_shade(render_base_colour(obj, hit), hit_normal(w, hit, p));
                }
        }
    }
}

}  // namespace hsref
