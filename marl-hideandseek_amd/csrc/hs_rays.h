// Ray casting against one world's geometry staged in LDS — the build's replacement for
// madrona::phys::broadphase::BVH::traceRay (call sites src/sim.cpp:288,331,602,738,797).
//
// A world holds at most 17 movable hulls, 36 axis-aligned walls and 3 planes, so the "BVH" is a
// flat list walked by every lane; all lanes of a workgroup read the same LDS words (broadcast, no
// bank conflicts).  Semantics (DESIGN.md "Engine decisions"): closest front-face entry with
// 0 <= t <= t_max in units of |d|; a ray starting inside a hull does not hit that hull; ties keep
// the lower body id.  Body ids: 0..16 movable slots, 100+k walls, 200+p planes, -1 miss.
#pragma once
#include "hs_dev.h"

namespace hs {

constexpr int kHitWallBase = 100;
constexpr int kHitPlaneBase = 200;

struct WorldGeom {
    int meta[kNumDSlots];
    float pos[kNumDSlots][3];
    float rot[kNumDSlots][4];
    int numWalls, numPlanes;
    float wall[kMaxWalls][4];     // cx, cy, hx, hy
    float plane[kMaxPlanes][4];   // nx, ny, nz, d
    // the geometry-view interface of trace_ray (the physics kernel has a second implementation over its
    // LDS-resident columns, hs_k_pipeline.h ResGeom)
    HSD int g_meta(int i) const { return meta[i]; }
    HSD V3 g_pos(int i) const { return {pos[i][0], pos[i][1], pos[i][2]}; }
    HSD Q g_rot(int i) const { return {rot[i][0], rot[i][1], rot[i][2], rot[i][3]}; }
    HSD int g_num_walls() const { return numWalls; }
    HSD float g_wall(int k, int c) const { return wall[k][c]; }
    HSD int g_num_planes() const { return numPlanes; }
    HSD float g_plane(int p, int c) const { return plane[p][c]; }
};

HSD V3 geom_pos(const WorldGeom &g, int i) { return {g.pos[i][0], g.pos[i][1], g.pos[i][2]}; }
HSD Q geom_rot(const WorldGeom &g, int i) { return {g.rot[i][0], g.rot[i][1], g.rot[i][2], g.rot[i][3]}; }

HSD float ray_box_local(V3 o, V3 d, V3 e) {
    float tn = -3.0e38f, tf = 3.0e38f;
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z}, ee[3] = {e.x, e.y, e.z};
    bool miss = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (dd[k] == 0.f) { if (oo[k] < -ee[k] || oo[k] > ee[k]) miss = true; continue; }
        // slab in centre / extent form: entry = -o/d - e/|d|, exit = -o/d + e/|d| (no near/far swap)
        float inv = 1.f / dd[k];
        const float r = ee[k] * fabsf(inv);                 // entry = -o/d - e/|d|, exit = -o/d + e/|d|, each one fused multiply-add
        tn = fmaxf(tn, hs_fma(-oo[k], inv, -r)); tf = fminf(tf, hs_fma(-oo[k], inv, r));
    }
    if (miss || tn > tf || tn < 0.f) return -1.f;
    return tn;
}

HSD float ray_wedge_local(V3 o, V3 d) {
    float tn = -3.0e38f, tf = 3.0e38f;
    const float fn[5][3] = {{0, 0, -1}, {0, 1, 0}, {0, -0.554700196f, 0.832050294f}, {1, 0, 0}, {-1, 0, 0}};
    const float off[5] = {1.f, 1.f, 0.277350098f, 1.f, 1.f};
    bool miss = false;
#pragma unroll
    for (int f = 0; f < 5; ++f) {
        V3 n = {fn[f][0], fn[f][1], fn[f][2]};
        float dist = dot(n, o) - off[f];
        float dn = dot(n, d);
        if (dn == 0.f) { if (dist > 0.f) miss = true; continue; }
        float t = -dist / dn;
        if (dn < 0.f) tn = fmaxf(tn, t); else tf = fminf(tf, t);
    }
    if (miss || tn > tf || tn < 0.f) return -1.f;
    return tn;
}

// Axis-aligned wall box with the per-ray reciprocal direction hoisted out of the wall loop (the same
// quotients 1/d[k] the generic slab test computes, so results are bit-identical).
HSD float ray_wall(V3 o, V3 d, V3 inv, V3 e) {
    float tn = -3.0e38f, tf = 3.0e38f;
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z}, ee[3] = {e.x, e.y, e.z}, ii[3] = {inv.x, inv.y, inv.z};
    bool miss = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (dd[k] == 0.f) { if (oo[k] < -ee[k] || oo[k] > ee[k]) miss = true; continue; }
        const float r = ee[k] * fabsf(ii[k]);                 // entry = -o/d - e/|d|, exit = -o/d + e/|d|, each one fused multiply-add
        tn = fmaxf(tn, hs_fma(-oo[k], ii[k], -r)); tf = fminf(tf, hs_fma(-oo[k], ii[k], r));
    }
    if (miss || tn > tf || tn < 0.f) return -1.f;
    return tn;
}

// The walls all span z in [0, 2.5], so a ray's z slab is the same for every wall: ray_wall_z evaluates it once per
// ray and ray_wall_xy folds the x / y slabs of one wall in.  max / min are exact and commutative, so the result is
// bit-identical to ray_wall's.
struct WallZ { float tn, tf; bool miss; };
HSD WallZ ray_wall_z(float oz, float dz, float invz) {
    WallZ z = {-3.0e38f, 3.0e38f, false};
    const float o = oz - 1.25f, e = 1.25f;
    if (dz == 0.f) { z.miss = o < -e || o > e; return z; }
    const float r = e * fabsf(invz);                 // entry = -o/d - e/|d|, exit = -o/d + e/|d|, each one fused multiply-add
    z.tn = fmaxf(z.tn, hs_fma(-o, invz, -r)); z.tf = fminf(z.tf, hs_fma(-o, invz, r));
    return z;
}
HSD float ray_wall_xy(float ox, float oy, V3 d, V3 inv, float ex, float ey, WallZ z) {
    float tn = -3.0e38f, tf = 3.0e38f;
    bool miss = z.miss;
    if (d.x == 0.f) { if (ox < -ex || ox > ex) miss = true; }
    else {
        const float r = ex * fabsf(inv.x);                 // entry = -o/d - e/|d|, exit = -o/d + e/|d|, each one fused multiply-add
        tn = fmaxf(tn, hs_fma(-ox, inv.x, -r)); tf = fminf(tf, hs_fma(-ox, inv.x, r));
    }
    if (d.y == 0.f) { if (oy < -ey || oy > ey) miss = true; }
    else {
        const float r = ey * fabsf(inv.y);                 // entry = -o/d - e/|d|, exit = -o/d + e/|d|, each one fused multiply-add
        tn = fmaxf(tn, hs_fma(-oy, inv.y, -r)); tf = fminf(tf, hs_fma(-oy, inv.y, r));
    }
    tn = fmaxf(tn, z.tn); tf = fminf(tf, z.tf);
    if (miss || tn > tf || tn < 0.f) return -1.f;
    return tn;
}

// The wall loop for rays with d.x != 0 and d.y != 0 (decided once per wave), fused with trace_ray's "closest hit, ties
// keep the lower id" update and free of branches.  Bit-identical to ray_wall_xy + that update for such rays:
//  * the z slab's bounds lie within +-3.0e38, so they absorb the initial clamps of the x / y slabs (max / min ignore
//    a NaN operand either way), and tn, tf are never NaN: the rejections can be written as positive comparisons;
//  * "t <= best while nothing was hit, t < best afterwards" is "t < best" throughout when best starts one ulp above
//    t_max (wall_best0; WallScan::finish puts t_max back when no wall was hit);
//  * a ray whose z slab misses gets tn = 3.0e38, which is never below best.
//  The x and y slabs are evaluated as one two-component vector (gfx950 has packed f32 multiply / add: each component is
//  the same IEEE operation as the scalar form).
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct WallScan {
    float best; int hit; float zn, zf;
    f32x2 o, inv, ainv;
    HSD WallScan(float tmax, WallZ z, float ox, float oy, V3 iv)
        : best(__uint_as_float(__float_as_uint(tmax) + 1u)), hit(-1), zn(z.miss ? 3.0e38f : z.tn), zf(z.tf) {
        o = {ox, oy}; inv = {iv.x, iv.y}; ainv = {fabsf(iv.x), fabsf(iv.y)};
    }
    // wall = (cx, cy, hx, hy)
    HSD void wall(f32x2 centre, f32x2 half, int id) {
        const f32x2 m = -(o - centre), r = half * ainv;
        const f32x2 lo = __builtin_elementwise_fma(m, inv, -r), hi = __builtin_elementwise_fma(m, inv, r);
        const float tn = fmaxf(fmaxf(lo.x, lo.y), zn);
        const float tf = fminf(fminf(hi.x, hi.y), zf);
        const bool ok = (tn <= tf) & (tn >= 0.f) & (tn < best);
        best = ok ? tn : best;
        hit = ok ? id : hit;
    }
    HSD void finish(float tmax) { if (hit < 0) best = tmax; }
};

// ray_box_local for a direction without zero components (decided once per wave).
HSD float ray_box_local_nz(V3 o, V3 d, V3 e) {
    const float ix = 1.f / d.x, iy = 1.f / d.y, iz = 1.f / d.z;
    const float rx = e.x * fabsf(ix), ry = e.y * fabsf(iy), rz = e.z * fabsf(iz);
    const float tn = fmaxf(fmaxf(fmaxf(-3.0e38f, hs_fma(-o.x, ix, -rx)), hs_fma(-o.y, iy, -ry)), hs_fma(-o.z, iz, -rz));
    const float tf = fminf(fminf(fminf(3.0e38f, hs_fma(-o.x, ix, rx)), hs_fma(-o.y, iy, ry)), hs_fma(-o.z, iz, rz));
    if (tn > tf || tn < 0.f) return -1.f;
    return tn;
}

// Squared bounding-sphere radius of a movable hull about its origin, inflated by 2% so that the
// conservative pre-test below can never reject a ray the exact test would accept.
HSD float obj_bound_r2(int obj) {
    return obj == OBJ_BOX ? 17.5625f * 1.02f : (obj == OBJ_RAMP ? 6.f * 1.02f : 3.f * 1.02f);
}

template <class G>
HSD int trace_ray(const G &g, V3 o, V3 d, float tmax, float *t_out) {
    int hit = -1; float best = tmax;
    const float dd2 = dot(d, d);
    for (int i = 0; i < kNumDSlots; ++i) {
        int m = g.g_meta(i);
        if (m == 0) continue;
        int obj = meta_obj(m);
        // conservative cull: the ray misses the hull's bounding sphere, or the sphere lies behind the origin
        const V3 mo = o - g.g_pos(i);
        const float b = dot(mo, d), cc = dot(mo, mo) - obj_bound_r2(obj);
        if (cc > 0.f && (b > 0.f || b * b < dd2 * cc * 0.999f)) continue;
        Q qi = qinv(g.g_rot(i));
        V3 ol = qrot(qi, mo), dl = qrot(qi, d);
        float t = obj == OBJ_RAMP ? ray_wedge_local(ol, dl) : ray_box_local(ol, dl, obj_half_extents(obj));
        if (t >= 0.f && t <= best && (hit < 0 || t < best)) { best = t; hit = i; }
    }
    const V3 inv = {1.f / d.x, 1.f / d.y, 1.f / d.z};
    const int nw = g.g_num_walls();
    for (int k = 0; k < nw; ++k) {
        V3 ol = {o.x - g.g_wall(k, 0), o.y - g.g_wall(k, 1), o.z - 1.25f};
        float t = ray_wall(ol, d, inv, {g.g_wall(k, 2), g.g_wall(k, 3), 1.25f});
        if (t >= 0.f && t <= best && (hit < 0 || t < best)) { best = t; hit = kHitWallBase + k; }
    }
    const int np = g.g_num_planes();
    for (int p = 0; p < np; ++p) {
        V3 n = {g.g_plane(p, 0), g.g_plane(p, 1), g.g_plane(p, 2)};
        float dn = dot(n, d);
        if (!(dn < 0.f)) continue;
        float dist = dot(n, o) - g.g_plane(p, 3);
        if (dist < 0.f) continue;
        float t = -dist / dn;
        if (t >= 0.f && t <= best && (hit < 0 || t < best)) { best = t; hit = kHitPlaneBase + p; }
    }
    *t_out = best;
    return hit;
}

}  // namespace hs
