// ORACLE — TEST INFRASTRUCTURE ONLY (see hs_ref_math.hpp header).
//
// PARITY UNPINNED: madrona::RNG / madrona::rand (rand::initKey mgr.cpp:678, rand::split_i
// sim.cpp:112,970, RNG::sampleI32 / sampleUniform / randKey — call sites SURVEY §8a-R) are
// not in the reference snapshot.  Chosen generator (documented in DESIGN.md): counter-based
// Threefry-2x32 with 20 rounds (Salmon et al., SC'11), integer-only sampling paths.
#pragma once
#include <cstdint>

namespace hsref {

struct RandKey { uint32_t a, b; };

static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

static inline RandKey threefry2x32(RandKey key, uint32_t c0, uint32_t c1) {
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

// rand::initKey(seed)  (mgr.cpp:678)
static inline RandKey rand_init_key(uint32_t seed) { return {seed, 0u}; }
// rand::split_i(key, idx, idx_upper)  (sim.cpp:112-113)
static inline RandKey rand_split_i(RandKey k, uint32_t idx, uint32_t idx_upper) {
    return threefry2x32(k, idx, idx_upper);
}

// madrona::RNG: a key plus a draw counter; every draw derives a fresh sub-key.
struct RNG {
    RandKey k;
    uint32_t count;
    RNG() : k{0, 0}, count(0) {}
    explicit RNG(RandKey key) : k(key), count(0) {}
    RandKey advance() { return rand_split_i(k, count++, 0u); }
    uint32_t bits32() { RandKey s = advance(); return s.a ^ s.b; }
    // half-open [a, b); an empty range (level_gen.cpp:87-88 when total==3) returns a.
    int32_t sampleI32(int32_t a, int32_t b) {
        uint32_t range = (uint32_t)(b - a);
        uint32_t v = (uint32_t)(((uint64_t)bits32() * (uint64_t)range) >> 32);
        return a + (int32_t)v;
    }
    // uniform in [0,1): top 24 bits * 2^-24 (exact in fp32)
    float sampleUniform() { return (float)(bits32() >> 8) * (1.f / 16777216.f); }
    RandKey randKey() { return advance(); }
};

}  // namespace hsref
