// Agent-view depth / RGB: the batch renderer outputs of Manager::depthTensor / rgbTensor (src/mgr.cpp:1241-1263).
//
// The reference hands these to Madrona's batch renderer (engine source absent).  What first-party
// source fixes is the camera — RenderingSystem::attachEntityToView(agent interface, 100 degrees vertical field of
// view, z-near 0.001, offset 0.5 up; src/sim.cpp:1400-1403), the interface entity taking its agent's pose in
// updateCameraSystem (:943-954) — the base colours per object type and one directional light (src/mgr.cpp:621-660).
// This kernel is a ray caster over the same flat per-world geometry the lidar uses (hs_rays.h): one workgroup per
// view, a lane per pixel, closest hit per pixel.
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

// The camera of one view and, per movable hull, what all rays of the view share (as k_observe's per-agent table):
// camera origin - hull centre, its squared length minus the bounding radius, the origin in the hull's frame.
struct RenderView {
    float fwd[3], right[3], up[3], o[3];
    alignas(16) float rel[kNumDSlots][8];
    unsigned others;                     // bit b: hull b exists, is not the viewer's own and may be in view
    // Walls that may be in view, ascending, each with a lower bound of the view depth of any point on it.  For a camera
    // that only yaws (every agent: inverse inertia x, y = 0) a wall wholly behind the camera or wholly outside the
    // horizontal field of view is left out, and a wall farther than the depth at which a ray leaves the walls' height
    // range [0, 2.5] is skipped per wave — conservative tests with a centimetre of margin, so no hit is ever lost.
    int nWalls;
    unsigned char wallId[kMaxWalls];
    float wallNear[kMaxWalls];
};

// One pixel: the closest hit over walls, planes and hulls under trace_ray's rule (closest entry with 0 <= t <= far; ties
// keep the lower id — hulls < walls < planes, the order trace_ray visits them in), evaluated in an order that suits the
// wave: the walls by the branch-free scan of hs_rays.h, the hulls behind the conservative bounding-sphere cull with the
// origin-side terms taken from the view's table.  Same expressions as trace_ray for every candidate, so the same bits.
HSD void render_pixel(const WorldGeom &g, const RenderView &vw, int px, int py, int W, int H, float *depth, unsigned *rgba) {
    const V3 fwd = {vw.fwd[0], vw.fwd[1], vw.fwd[2]}, right = {vw.right[0], vw.right[1], vw.right[2]}, up = {vw.up[0], vw.up[1], vw.up[2]};
    const V3 o = {vw.o[0], vw.o[1], vw.o[2]};
    const float aspect = (float)W / (float)H;
    const float u = ((((float)px + 0.5f) / (float)W) * 2.f - 1.f) * (kTanHalfFov * aspect);
    const float v = (1.f - (((float)py + 0.5f) / (float)H) * 2.f) * kTanHalfFov;
    const V3 d = (fwd + right * u) + up * v;            // forward component 1: t is the view-space depth
    const float tmax = kCamFar;
    int hit = -1; float best = tmax;
    // walls
    const V3 inv = {1.f / d.x, 1.f / d.y, 1.f / d.z};
    const WallZ wz = ray_wall_z(o.z, d.z, inv.z);
    const int nw = vw.nWalls;
    if (__ballot(d.x == 0.f || d.y == 0.f) == 0) {
        WallScan ws(tmax, wz, o.x, o.y, inv);
        for (int k = 0; k < nw; ++k) {
            if (__ballot(wz.tf >= vw.wallNear[k]) == 0ull) continue;      // every ray of the wave has left [0, 2.5] before it
            const int q = vw.wallId[k];
            const f32x2 *wq = reinterpret_cast<const f32x2 *>(g.wall[q]);
            ws.wall(wq[0], wq[1], kHitWallBase + q);
        }
        ws.finish(tmax);
        best = ws.best; hit = ws.hit;
    } else {
        for (int k = 0; k < nw; ++k) {
            const int q = vw.wallId[k];
            const float t = ray_wall_xy(o.x - g.wall[q][0], o.y - g.wall[q][1], d, inv, g.wall[q][2], g.wall[q][3], wz);
            if (t >= 0.f && t <= best && (hit < 0 || t < best)) { best = t; hit = kHitWallBase + q; }
        }
    }
    // planes
    const int np = g.numPlanes;
    for (int p = 0; p < np; ++p) {
        const V3 n = {g.plane[p][0], g.plane[p][1], g.plane[p][2]};
        const float dn = dot(n, d);
        if (!(dn < 0.f)) continue;
        const float dist = dot(n, o) - g.plane[p][3];
        if (dist < 0.f) continue;
        const float t = -dist / dn;
        if (t >= 0.f && t <= best && (hit < 0 || t < best)) { best = t; hit = kHitPlaneBase + p; }
    }
    // hulls (lower ids than the static geometry: a hull wins a tie)
    const float dd2 = dot(d, d);
    const unsigned others = __builtin_amdgcn_readfirstlane(vw.others);
#pragma unroll 1
    for (int b = 0; b < kNumDSlots; ++b) {
        if (!((others >> b) & 1u)) continue;
        const float4 e = *reinterpret_cast<const float4 *>(vw.rel[b]);
        const float bb = (e.x * d.x + e.y * d.y) + e.z * d.z, cc = e.w;
        const bool culled = cc > 0.f && (bb > 0.f || bb * bb < dd2 * cc * 0.999f);
        if (__ballot(!culled) == 0ull) continue;
        if (!culled) {
            const int obj = meta_obj(g.meta[b]);
            const V3 ol = {vw.rel[b][4], vw.rel[b][5], vw.rel[b][6]};
            const V3 dl = qrot(qinv(geom_rot(g, b)), d);
            const float t = obj == OBJ_RAMP ? ray_wedge_local(ol, dl) : ray_box_local(ol, dl, obj_half_extents(obj));
            if (t >= 0.f && t <= tmax && (t < best || (t == best && b < hit) || hit < 0)) { best = t; hit = b; }
        }
    }
    if (hit < 0 || best < kCamNear) { *depth = 0.f; *rgba = 0xff000000u; return; }
    const V3 p = o + d * best;
    const int obj = hit < kNumDSlots ? meta_obj(g.meta[hit]) : OBJ_NONE;
    *depth = best;
    *rgba = render_shade(render_base_colour(obj, hit), hit_normal(g, hit, p));
}

// One workgroup per view (world slot, agent); views of inactive agents are zero-filled.
__global__ void __launch_bounds__(kRenderThreads) k_render(SimState S, float *depth, unsigned *rgba, int W, int H) {
    __shared__ WorldGeom g;
    __shared__ RenderView vw;
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
    const int aslot = kAgentSlot0 + agent;
    const size_t view = (size_t)w * A_ + agent;
    float *dv = depth + view * (size_t)(W * H);
    unsigned *cv = rgba + view * (size_t)(W * H);
    const bool active = team_agent_active(S.teams[w], agent) != 0 && g.meta[aslot] != 0;
    if (!active) {
        for (int i = tid; i < W * H; i += kRenderThreads) { dv[i] = 0.f; cv[i] = 0u; }
        return;
    }
    // the view's camera, per-hull table and wall list
    const V3 o = geom_pos(g, aslot) + V3{0.f, 0.f, kCamUp};
    const Q crot = geom_rot(g, aslot);
    const V3 cf = qrot(crot, {0.f, 1.f, 0.f}), cr = qrot(crot, {1.f, 0.f, 0.f}), cu = qrot(crot, {0.f, 0.f, 1.f});
    // horizontal culls only for a camera that yaws and nothing else (then a ray's x, y direction is fwd + u right)
    const bool yawOnly = cu.x == 0.f && cu.y == 0.f && cf.z == 0.f && cr.z == 0.f;
    const float umax = kTanHalfFov * ((float)W / (float)H);             // |u| of every pixel centre is below this
    const float kMargin = 0.01f;
    if (tid < 64) {
        const int m = tid < kNumDSlots ? g.meta[tid] : 0;
        bool keep = m != 0 && tid != aslot;
        if (keep) {
            const V3 mo = o - geom_pos(g, tid);
            const V3 ol = qrot(qinv(geom_rot(g, tid)), mo);
            float *e = vw.rel[tid];
            const float r2 = obj_bound_r2(meta_obj(m));
            e[0] = mo.x; e[1] = mo.y; e[2] = mo.z; e[3] = dot(mo, mo) - r2;
            e[4] = ol.x; e[5] = ol.y; e[6] = ol.z; e[7] = 0.f;
            if (yawOnly) {
                // bounding sphere against the two vertical side planes of the view and the plane through the camera
                const float R = sqrtf(r2) + kMargin, side = sqrtf(1.f + umax * umax);
                const float f = -(mo.x * cf.x + mo.y * cf.y), r = -(mo.x * cr.x + mo.y * cr.y);
                if (f < -R || r - umax * f > R * side || -r - umax * f > R * side) keep = false;
            }
        }
        const unsigned long long pm = __ballot(keep);
        if (tid == 0) vw.others = (unsigned)pm;
    } else if (tid < 128) {
        const int q = tid - 64;
        bool keep = q < g.numWalls;
        float nearest = -3.0e38f;
        if (keep && yawOnly) {
            float fmin = 3.0e38f, fmax = -3.0e38f; bool allRight = true, allLeft = true;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float px_ = (g.wall[q][0] + ((c & 1) ? g.wall[q][2] : -g.wall[q][2])) - o.x;
                const float py_ = (g.wall[q][1] + ((c & 2) ? g.wall[q][3] : -g.wall[q][3])) - o.y;
                const float f = px_ * cf.x + py_ * cf.y, r = px_ * cr.x + py_ * cr.y;
                fmin = fminf(fmin, f); fmax = fmaxf(fmax, f);
                allRight = allRight && (r - umax * f > kMargin * (1.f + umax));
                allLeft = allLeft && (-r - umax * f > kMargin * (1.f + umax));
            }
            if (fmax < -kMargin || allRight || allLeft) keep = false;
            nearest = fmin - kMargin;
        }
        const unsigned long long km = __ballot(keep);
        if (keep) {
            const int k = __builtin_amdgcn_mbcnt_hi((unsigned)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km, 0u));
            vw.wallId[k] = (unsigned char)q; vw.wallNear[k] = nearest;
        }
        if (tid == 64) vw.nWalls = __builtin_popcountll(km);
    } else if (tid == 128) {
        vw.fwd[0] = cf.x; vw.fwd[1] = cf.y; vw.fwd[2] = cf.z; vw.right[0] = cr.x; vw.right[1] = cr.y; vw.right[2] = cr.z;
        vw.up[0] = cu.x; vw.up[1] = cu.y; vw.up[2] = cu.z; vw.o[0] = o.x; vw.o[1] = o.y; vw.o[2] = o.z;
    }
    __syncthreads();
    for (int i = tid; i < W * H; i += kRenderThreads) {
        float d; unsigned c;
        render_pixel(g, vw, i % W, i / W, W, H, &d, &c);
        dv[i] = d; cv[i] = c;
    }
}

}  // namespace hs
