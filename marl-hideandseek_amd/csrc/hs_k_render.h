// Agent-view depth / RGB: the batch renderer outputs of Manager::depthTensor / rgbTensor (src/mgr.cpp:1241-1263).
//
// The reference hands these to Madrona's batch renderer (engine source absent).  What first-party
// source fixes is the camera — RenderingSystem::attachEntityToView(agent, 100 degrees vertical field of view,
// z-near 0.001, offset 0.5 up; src/sim.cpp:1400-1403) — the base colours per object type and one directional light
// (src/mgr.cpp:621-660).  This kernel is a ray caster over the same flat per-world geometry the lidar uses
// (hs_rays.h trace_ray): one workgroup per view, a lane per pixel column, closest hit per pixel.
//   depth  view-space depth of the hit (distance along the camera's forward axis), 0 where nothing is hit
//   rgb    base colour of the hit object x (0.3 ambient + 0.7 Lambert term of the light), alpha 255; black sky
// The reference's textures (floor grid, the agents' faces) are not reproduced; the seeker's red face texture is stood
// in for by a red tint so that the teams stay distinguishable.  Rendering is opt-in (HS_FLAG_EXT_RENDER or
// hs_render): with the reference scripts' arguments the tensors stay allocated and unwritten, as before.
#pragma once
#include "hs_state.h"
#include "hs_rays.h"

namespace hs {

constexpr float kTanHalfFov = 1.19175359259421f;      // tan(100 degrees / 2)
constexpr float kCamUp = 0.5f;                        // camera offset above the agent's origin
constexpr float kCamNear = 0.001f, kCamFar = 1000.f;
constexpr int kRenderThreads = 256;

HSD V3 render_base_colour(int obj, int hit) {
    if (hit >= kHitPlaneBase) return {0.5f, 0.3f, 0.3f};                 // material 3 (floor)
    if (hit >= kHitWallBase) return {0.4f, 0.4f, 0.4f};                  // material 0
    if (obj == OBJ_CUBE) return {1.f, 0.1f, 0.1f};                       // material 1
    if (obj == OBJ_HIDER) return {1.f, 1.f, 1.f};                        // material 2 (white, smile texture)
    if (obj == OBJ_SEEKER) return {1.f, 0.3f, 0.3f};                     // material 7 (white, red-smile texture)
    if (obj == OBJ_RAMP) return {191.f / 255.f, 108.f / 255.f, 10.f / 255.f};   // material 4
    if (obj == OBJ_BOX) return {12.f / 255.f, 144.f / 255.f, 150.f / 255.f};    // material 5
    return {0.4f, 0.4f, 0.4f};
}

// Outward normal of the face of an origin-centred box (half extents e) that the surface point q lies on: the axis
// along which q is closest to (or farthest beyond) its face; ties keep the lower axis.
HSD V3 box_face_normal(V3 q, V3 e) {
    const float dx = fabsf(q.x) - e.x, dy = fabsf(q.y) - e.y, dz = fabsf(q.z) - e.z;
    if (dx >= dy && dx >= dz) return {q.x < 0.f ? -1.f : 1.f, 0.f, 0.f};
    if (dy >= dz) return {0.f, q.y < 0.f ? -1.f : 1.f, 0.f};
    return {0.f, 0.f, q.z < 0.f ? -1.f : 1.f};
}
HSD V3 wedge_face_normal(V3 q) {
    const float fn[5][3] = {{0, 0, -1}, {0, 1, 0}, {0, -0.554700196f, 0.832050294f}, {1, 0, 0}, {-1, 0, 0}};
    const float off[5] = {1.f, 1.f, 0.277350098f, 1.f, 1.f};
    int bf = 0; float bd = -3.0e38f;
#pragma unroll
    for (int f = 0; f < 5; ++f) {
        const float d = dot(V3{fn[f][0], fn[f][1], fn[f][2]}, q) - off[f];
        if (d > bd) { bd = d; bf = f; }
    }
    return {fn[bf][0], fn[bf][1], fn[bf][2]};
}

template <class G>
HSD V3 hit_normal(const G &g, int hit, V3 p) {
    if (hit >= kHitPlaneBase) { const int k = hit - kHitPlaneBase; return {g.g_plane(k, 0), g.g_plane(k, 1), g.g_plane(k, 2)}; }
    if (hit >= kHitWallBase) {
        const int k = hit - kHitWallBase;
        return box_face_normal({p.x - g.g_wall(k, 0), p.y - g.g_wall(k, 1), p.z - 1.25f}, {g.g_wall(k, 2), g.g_wall(k, 3), 1.25f});
    }
    const int obj = meta_obj(g.g_meta(hit));
    const Q r = g.g_rot(hit);
    const V3 q = qrot(qinv(r), p - g.g_pos(hit));
    return qrot(r, obj == OBJ_RAMP ? wedge_face_normal(q) : box_face_normal(q, obj_half_extents(obj)));
}

HSD unsigned render_shade(V3 base, V3 n) {
    // light travels along (1, 1, -2) (src/mgr.cpp:657-659): the Lambert term is n . (-l), l normalised
    const V3 toLight = {-0.408248290f, -0.408248290f, 0.816496581f};
    const float lam = fmaxf(dot(n, toLight), 0.f);
    const float k = 0.3f + 0.7f * lam;
    const float r = fminf(base.x * k, 1.f), gch = fminf(base.y * k, 1.f), b = fminf(base.z * k, 1.f);
    return (unsigned)(r * 255.f + 0.5f) | ((unsigned)(gch * 255.f + 0.5f) << 8) | ((unsigned)(b * 255.f + 0.5f) << 16) | 0xff000000u;
}

// One pixel of agent `a`'s view of the world `g`: writes depth and packed RGBA.
template <class G>
HSD void render_pixel(const G &g, int aslot, int px, int py, int W, int H, float *depth, unsigned *rgba) {
    const Q rot = g.g_rot(aslot);
    const V3 fwd = qrot(rot, {0.f, 1.f, 0.f}), right = qrot(rot, {1.f, 0.f, 0.f}), up = qrot(rot, {0.f, 0.f, 1.f});
    const V3 o = g.g_pos(aslot) + V3{0.f, 0.f, kCamUp};
    const float aspect = (float)W / (float)H;
    const float u = ((((float)px + 0.5f) / (float)W) * 2.f - 1.f) * (kTanHalfFov * aspect);
    const float v = (1.f - (((float)py + 0.5f) / (float)H) * 2.f) * kTanHalfFov;
    const V3 d = (fwd + right * u) + up * v;            // forward component 1: t is the view-space depth
    float t;
    const int hit = trace_ray(g, o, d, kCamFar, &t);
    if (hit < 0 || t < kCamNear) { *depth = 0.f; *rgba = 0xff000000u; return; }
    const V3 p = o + d * t;
    const int obj = hit < kNumDSlots ? meta_obj(g.g_meta(hit)) : OBJ_NONE;
    *depth = t;
    *rgba = render_shade(render_base_colour(obj, hit), hit_normal(g, hit, p));
}

// One workgroup per view (world slot, agent); views of inactive agents are zero-filled.
__global__ void __launch_bounds__(kRenderThreads) k_render(SimState S, float *depth, unsigned *rgba, int W, int H) {
    __shared__ WorldGeom g;
    const int tid = threadIdx.x;
    const int A_ = S.A;
    const int ps = blockIdx.x / A_, agent = blockIdx.x % A_;        // slot of the tiled columns, agent index
    const int w = S.worldOfSlot[ps];
    if (w < 0) return;                                              // (padding slot of the last octet)
    for (int i = tid; i < kNumDSlots; i += kRenderThreads) g.meta[i] = S.bmeta(i, ps);
    for (int i = tid; i < kNumDSlots * 3; i += kRenderThreads) g.pos[i % kNumDSlots][i / kNumDSlots] = S.bpos(i, ps);
    for (int i = tid; i < kNumDSlots * 4; i += kRenderThreads) g.rot[i % kNumDSlots][i / kNumDSlots] = S.brot(i, ps);
    for (int i = tid; i < 4 * kMaxWalls; i += kRenderThreads) g.wall[i % kMaxWalls][i / kMaxWalls] = S.walls(i, ps);
    for (int i = tid; i < 4 * kMaxPlanes; i += kRenderThreads) g.plane[i % kMaxPlanes][i / kMaxPlanes] = S.planes(i, ps);
    if (tid == 0) { g.numWalls = S.numWalls[w]; g.numPlanes = S.numPlanes[w]; }
    __syncthreads();
    const size_t view = (size_t)w * A_ + agent;
    float *dv = depth + view * (size_t)(W * H);
    unsigned *cv = rgba + view * (size_t)(W * H);
    const bool active = team_agent_active(S.teams[w], agent) != 0 && g.meta[kAgentSlot0 + agent] != 0;
    for (int i = tid; i < W * H; i += kRenderThreads) {
        if (!active) { dv[i] = 0.f; cv[i] = 0u; continue; }
        float d; unsigned c;
        render_pixel(g, kAgentSlot0 + agent, i % W, i / W, W, H, &d, &c);
        dv[i] = d; cv[i] = c;
    }
}

}  // namespace hs
