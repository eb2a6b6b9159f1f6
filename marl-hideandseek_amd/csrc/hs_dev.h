// Device-side math, RNG and constants for the HIP kernels (gfx950).
//
// Float discipline: this translation unit is compiled with -ffp-contract=off and without
// fast-math.  Every expression here is written in a fixed association order; the parity tests
// compare the kernels' results with the CPU oracle bit for bit, so do not "simplify" arithmetic.
//
// The engine pieces the reference takes from Madrona (vector/quaternion math, RNG) are absent
// from the reference snapshot; see DESIGN.md "Engine decisions" for what is chosen here.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HSD __device__ __forceinline__

// The lane index as the physics kernel reads it: an OPAQUE copy of threadIdx.x.  k_physics is one function of ~60 000
// instructions whose phases each derive a dozen lane constants (world = lane / 8, LDS offsets of its rows, ...).  Given
// the plain threadIdx.x the compiler computes them all once at the top of the kernel and keeps them alive through the
// substep loop — some forty registers of loop invariants in a kernel that sits at the 256-register budget of two waves
// per SIMD (12 spilled dwords, 52 B of scratch per lane).  An empty asm makes every call a value of its own: each phase
// re-derives its constants (a few integer instructions) and nothing lives across phases: 223 registers, no spill, no
// scratch.
__device__ __forceinline__ int hs_lane() { int x = threadIdx.x; asm volatile("" : "+v"(x)); return x; }

namespace hs {

// ---- capacities and constants: src/sim.hpp:39-41, src/sim.cpp:14-17 ----
constexpr int kMaxBoxes = 9;
constexpr int kMaxRamps = 2;
constexpr int kMaxAgents = 6;
constexpr int kBoxSlot0 = 0;
constexpr int kRampSlot0 = 9;
constexpr int kAgentSlot0 = 11;
constexpr int kNumDSlots = 17;
constexpr int kMaxWalls = 36;
constexpr int kMaxPlanes = 3;
constexpr int kNumPrepSteps = 96;
constexpr int kEpisodeLen = 240;
constexpr int kNumSubsteps = 4;          // setupPhysicsStepTasks(..., 4, XPBD)  src/sim.cpp:1162-1163
constexpr float kSubstepH = (1.f / 30.f) / 4.f;
constexpr float kInvSubstepH = 120.f;
constexpr float kGravityZ = -9.8f;
constexpr float kMaxDepenVel = 3.f;
// Candidate-pair capacities per world per substep (DESIGN.md "Engine decisions"); -DHS_MAX_DD_CAND / -DHS_MAX_S_CAND
// shrink them for the overflow test (tests/test_gpu_status.py), nothing else overrides them.
#ifndef HS_MAX_DD_CAND
#define HS_MAX_DD_CAND 16
#endif
#ifndef HS_MAX_S_CAND
#define HS_MAX_S_CAND 24
#endif
constexpr int kMaxDDCand = HS_MAX_DD_CAND;
constexpr int kGrabWords = 15;      // grab-joint record: r2 3, attach2 4, separation 1, r1 3, attach1 4
constexpr int kMaxSCand = HS_MAX_S_CAND;
constexpr float kCosFovHalf = 0.382683426f;
constexpr float kPi = 3.14159265358979323846f;

// SimObject (src/sim.hpp:78-88)
enum : int { OBJ_SPHERE = 0, OBJ_PLANE = 1, OBJ_CUBE = 2, OBJ_WALL = 3, OBJ_HIDER = 4, OBJ_SEEKER = 5,
             OBJ_RAMP = 6, OBJ_BOX = 7, OBJ_NONE = -1 };
enum : int { OWNER_NONE = 0, OWNER_SEEKER = 1, OWNER_HIDER = 2, OWNER_UNOWNABLE = 3 };
enum : int { RESP_DYNAMIC = 0, RESP_KINEMATIC = 1, RESP_STATIC = 2 };
enum : int { AGENT_SEEKER = 0, AGENT_HIDER = 1 };
enum : uint32_t { FLAG_USE_FIXED_WORLD = 1, FLAG_IGNORE_EPISODE_LENGTH = 2, FLAG_RANDOM_FLIP_TEAMS = 4,
                  FLAG_ZERO_AGENT_VELOCITY = 8, FLAG_EXT_SKIP_OBSERVATIONS = 1u << 16, FLAG_EXT_RENDER = 1u << 17 };

// body meta word: (objType+1) | response<<8 | owner<<16 ; 0 == empty slot
HSD int meta_pack(int obj, int resp, int owner) { return (obj + 1) | (resp << 8) | (owner << 16); }
HSD int meta_obj(int m) { return (m & 0xff) - 1; }
HSD int meta_resp(int m) { return (m >> 8) & 0xff; }
HSD int meta_owner(int m) { return (m >> 16) & 0xff; }

struct V3 { float x, y, z; };
struct Q { float w, x, y, z; };

HSD V3 v3(float x, float y, float z) { return V3{x, y, z}; }
HSD V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
HSD V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
HSD V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
HSD V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
HSD V3 mulc(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
// (fused multiply-adds, written out one by one: the CPU restatement the tests compare with has the same ones in the same places)
HSD float hs_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
HSD float dot(V3 a, V3 b) { return hs_fma(a.z, b.z, hs_fma(a.y, b.y, a.x * b.x)); }
HSD V3 cross(V3 a, V3 b) { return {hs_fma(a.y, b.z, -(a.z * b.y)), hs_fma(a.z, b.x, -(a.x * b.z)), hs_fma(a.x, b.y, -(a.y * b.x))}; }
// a + b * s and a - b * s, each component one fused multiply-add
HSD V3 madd(V3 a, V3 b, float s) { return {hs_fma(b.x, s, a.x), hs_fma(b.y, s, a.y), hs_fma(b.z, s, a.z)}; }
HSD V3 nmadd(V3 a, V3 b, float s) { return {hs_fma(-b.x, s, a.x), hs_fma(-b.y, s, a.y), hs_fma(-b.z, s, a.z)}; }
// a . b + c and a x b + c with every product fused
HSD float dot_add(V3 a, V3 b, float c) { return hs_fma(a.z, b.z, hs_fma(a.y, b.y, hs_fma(a.x, b.x, c))); }
HSD V3 cross_add(V3 a, V3 b, V3 c) {
    return {hs_fma(a.y, b.z, hs_fma(-a.z, b.y, c.x)), hs_fma(a.z, b.x, hs_fma(-a.x, b.z, c.y)), hs_fma(a.x, b.y, hs_fma(-a.y, b.x, c.z))};
}
HSD float len2(V3 a) { return dot(a, a); }
HSD float len(V3 a) { return sqrtf(dot(a, a)); }
HSD V3 normalize(V3 a) { float inv = 1.f / len(a); return a * inv; }
HSD V3 vsel(bool c, V3 a, V3 b) { return {c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z}; }

HSD Q qmul(Q a, Q b) {
    return {hs_fma(-a.z, b.z, hs_fma(-a.y, b.y, hs_fma(-a.x, b.x, a.w * b.w))),
            hs_fma(-a.z, b.y, hs_fma(a.y, b.z, hs_fma(a.x, b.w, a.w * b.x))),
            hs_fma(a.z, b.x, hs_fma(a.y, b.w, hs_fma(-a.x, b.z, a.w * b.y))),
            hs_fma(a.z, b.w, hs_fma(-a.y, b.x, hs_fma(a.x, b.y, a.w * b.z)))};
}
HSD Q qinv(Q q) { return {q.w, -q.x, -q.y, -q.z}; }
HSD Q qnormalize(Q q) {
    float n2 = hs_fma(q.z, q.z, hs_fma(q.y, q.y, hs_fma(q.x, q.x, q.w * q.w)));
    float inv = 1.f / sqrtf(n2);
    return {q.w * inv, q.x * inv, q.y * inv, q.z * inv};
}
HSD V3 qrot(Q q, V3 v) {
    V3 p = {q.x, q.y, q.z};
    float s = q.w;
    float d2 = 2.f * dot(p, v);
    float s2 = 2.f * s;
    float k = hs_fma(s2, s, -1.f);
    V3 c = cross(p, v);
    return {hs_fma(s2, c.x, hs_fma(d2, p.x, k * v.x)), hs_fma(s2, c.y, hs_fma(d2, p.y, k * v.y)), hs_fma(s2, c.z, hs_fma(d2, p.z, k * v.z))};
}

// sin/cos by Cody-Waite pi/2 reduction + minimax polynomials; atan2/asin likewise.  These replace
// libm/ocml so that host oracle and device agree exactly.
HSD void hs_sincosf(float x, float *s_out, float *c_out) {
    const float two_over_pi = 0.63661977236758134308f;
    const float pio2_hi = 1.5707962512969970703125f;
    const float pio2_lo = 7.54978995489188216e-8f;
    float kf = x * two_over_pi;
    int k = (int)(kf + (kf >= 0.f ? 0.5f : -0.5f));
    float fk = (float)k;
    float r = (x - fk * pio2_hi) - fk * pio2_lo;
    float z = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
               - 0.5f * z + 1.f;
    int q = k & 3;
    float s = (q == 0) ? sp : (q == 1) ? cp : (q == 2) ? -sp : -cp;
    float c = (q == 0) ? cp : (q == 1) ? -sp : (q == 2) ? -cp : sp;
    *s_out = s; *c_out = c;
}
HSD float hs_atanf(float xin) {
    float sign = xin < 0.f ? -1.f : 1.f;
    float x = fabsf(xin);
    float y;
    // ranges: x > tan(3pi/8): pi/2 + atan(-(1/x));  x > tan(pi/8): pi/4 + atan((x-1)/(x+1));  else atan(x).  One division
    // serves both reductions: (-1)/x is the same correctly rounded quotient as -(1/x).
    const bool big = x > 2.414213562373095f, mid = x > 0.4142135623730950f;
    y = big ? 1.5707963267948966f : (mid ? 0.7853981633974483f : 0.f);
    if (mid) { const float num = big ? -1.f : x - 1.f, den = big ? x : x + 1.f; x = num / den; }
    float z = x * x;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x);
    return sign * y;
}
HSD float hs_atan2f(float y, float x) {
    if (x == 0.f) {
        if (y > 0.f) return 0.5f * kPi;
        if (y < 0.f) return -0.5f * kPi;
        return 0.f;
    }
    float a = hs_atanf(y / x);
    if (x < 0.f) { a = (y >= 0.f) ? a + kPi : a - kPi; }
    return a;
}
HSD float hs_asinf(float xin) {
    float sign = xin < 0.f ? -1.f : 1.f;
    float a = fabsf(xin);
    float z, x;
    bool flag = a > 0.5f;
    if (flag) { z = 0.5f * (1.f - a); x = sqrtf(z); }
    else { x = a; z = x * x; }
    float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z
               + 1.6666752422e-1f) * z * x + x;
    if (flag) { p = p + p; p = 1.5707963267948966f - p; }
    return sign * p;
}
HSD Q quat_angle_axis_z(float angle) {
    float s, c;
    hs_sincosf(angle * 0.5f, &s, &c);
    return {c, 0.f, 0.f, s};
}

struct M3 { V3 c0, c1, c2; };
HSD M3 m3_from_quat(Q q) {
    float y2 = q.y * q.y, z2 = q.z * q.z;
    float xy = q.x * q.y, xz = q.x * q.z, yz = q.y * q.z;
    M3 m;
    m.c0 = {hs_fma(-2.f, hs_fma(q.y, q.y, z2), 1.f), 2.f * hs_fma(q.w, q.z, xy), 2.f * hs_fma(-q.w, q.y, xz)};
    m.c1 = {2.f * hs_fma(-q.w, q.z, xy), hs_fma(-2.f, hs_fma(q.x, q.x, z2), 1.f), 2.f * hs_fma(q.w, q.x, yz)};
    m.c2 = {2.f * hs_fma(q.w, q.y, xz), 2.f * hs_fma(-q.w, q.x, yz), hs_fma(-2.f, hs_fma(q.x, q.x, y2), 1.f)};
    return m;
}

// quatToEuler (src/sim.cpp:372-399)
HSD V3 quat_to_euler(Q q) {
    float sinr = 2.f * (q.w * q.x + q.y * q.z);
    float cosr = 1.f - 2.f * (q.x * q.x + q.y * q.y);
    float roll = hs_atan2f(sinr, cosr);
    float sinp = 2.f * (q.w * q.y - q.z * q.x);
    float pitch = fabsf(sinp) >= 1.f ? copysignf(kPi / 2.f, sinp) : hs_asinf(sinp);
    float siny = 2.f * (q.w * q.z + q.x * q.y);
    float cosy = 1.f - 2.f * (q.y * q.y + q.z * q.z);
    float yaw = hs_atan2f(siny, cosy);
    return {roll, pitch, yaw};
}

// ---- counter-based RNG: Threefry-2x32-20 (call sites src/sim.cpp:105-114,163,187-190) ----
struct RandKey { uint32_t a, b; };
HSD uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
HSD RandKey threefry2x32(RandKey key, uint32_t c0, uint32_t c1) {
    const uint32_t ks0 = key.a, ks1 = key.b, ks2 = 0x1BD11BDAu ^ key.a ^ key.b;
    uint32_t x0 = c0 + ks0, x1 = c1 + ks1;
#define HS_TF_R(r) { x0 += x1; x1 = rotl32(x1, r); x1 ^= x0; }
    HS_TF_R(13) HS_TF_R(15) HS_TF_R(26) HS_TF_R(6)
    x0 += ks1; x1 += ks2 + 1u;
    HS_TF_R(17) HS_TF_R(29) HS_TF_R(16) HS_TF_R(24)
    x0 += ks2; x1 += ks0 + 2u;
    HS_TF_R(13) HS_TF_R(15) HS_TF_R(26) HS_TF_R(6)
    x0 += ks0; x1 += ks1 + 3u;
    HS_TF_R(17) HS_TF_R(29) HS_TF_R(16) HS_TF_R(24)
    x0 += ks1; x1 += ks2 + 4u;
    HS_TF_R(13) HS_TF_R(15) HS_TF_R(26) HS_TF_R(6)
    x0 += ks2; x1 += ks0 + 5u;
#undef HS_TF_R
    return {x0, x1};
}
struct RNG {
    RandKey k; uint32_t count;
    HSD RandKey advance() { return threefry2x32(k, count++, 0u); }
    HSD uint32_t bits32() { RandKey s = advance(); return s.a ^ s.b; }
    HSD int32_t sampleI32(int32_t a, int32_t b) {
        uint32_t range = (uint32_t)(b - a);
        uint32_t v = (uint32_t)(((uint64_t)bits32() * (uint64_t)range) >> 32);
        return a + (int32_t)v;
    }
    HSD float sampleUniform() { return (float)(bits32() >> 8) * (1.f / 16777216.f); }
    HSD RandKey randKey() { return advance(); }
};

// ---- object tables: src/mgr.cpp:476-559,577-584 ----
HSD float obj_inv_mass(int o) { return (o == OBJ_CUBE || o == OBJ_RAMP || o == OBJ_BOX) ? 0.5f : ((o == OBJ_HIDER || o == OBJ_SEEKER || o == OBJ_SPHERE) ? 1.f : 0.f); }
HSD float obj_mu_s(int o) { return o == OBJ_PLANE ? 2.f : 0.5f; }
HSD float obj_mu_d(int o) {
    return (o == OBJ_PLANE || o == OBJ_CUBE || o == OBJ_WALL) ? 2.f
         : (o == OBJ_HIDER || o == OBJ_SEEKER) ? 16.f : (o == OBJ_RAMP) ? 1.f : (o == OBJ_BOX) ? 4.f : 0.5f;
}
HSD V3 obj_inv_inertia(int o) {
    if (o == OBJ_CUBE) return {0.75f, 0.75f, 0.75f};
    if (o == OBJ_BOX) return {0.96f, 0.088235294f, 0.090566038f};
    if (o == OBJ_RAMP) return {0.692307692f, 0.9f, 0.6f};
    if (o == OBJ_HIDER || o == OBJ_SEEKER) return {0.f, 0.f, 1.5f};
    return {0.f, 0.f, 0.f};
}
// 1 / inverse inertia per axis (0 where the inverse is 0): the same IEEE quotients integrate would compute, folded
// at compile time
HSD V3 obj_inertia(int o) {
    if (o == OBJ_CUBE) return {1.f / 0.75f, 1.f / 0.75f, 1.f / 0.75f};
    if (o == OBJ_BOX) return {1.f / 0.96f, 1.f / 0.088235294f, 1.f / 0.090566038f};
    if (o == OBJ_RAMP) return {1.f / 0.692307692f, 1.f / 0.9f, 1.f / 0.6f};
    if (o == OBJ_HIDER || o == OBJ_SEEKER) return {0.f, 0.f, 1.f / 1.5f};
    return {0.f, 0.f, 0.f};
}
HSD V3 obj_half_extents(int o) { return o == OBJ_BOX ? V3{4.f, 0.75f, 1.f} : V3{1.f, 1.f, 1.f}; }

}  // namespace hs
