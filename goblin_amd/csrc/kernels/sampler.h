// Sample dimensions for one camera sample.
//
// REPLAY  : read the caller's Sample record (same float layout Sampler::
//           requestSamples fills: GoblinSampler.cpp:108-197, Sample::allocateQuota
//           :35-58).
// NATIVE  : counter-based generator with the reference's stratification law.
//           For pattern P, stratum i, pixel p the sub-stratum that camera sample
//           k receives is perm_{key(p,P,i)}(k) -- a keyed bijection on [0, spp)
//           standing in for the reference's shuffle of the stratified column
//           (GoblinSampler.cpp:139-155) -- and the jitter is a hash of (key, k).
//           Image samples are not permuted (sample k sits in sub-cell k, :130-131,
//           :160-163).  Integer hashing only, so the CPU oracle reproduces every
//           draw bit for bit (oracle/goblin_oracle.cpp restates this definition;
//           neither side includes the other).
#pragma once
#include <stdint.h>

#include "vecmath.h"

__device__ __forceinline__ uint32_t nat_mix(uint32_t a, uint32_t b) {
    uint32_t h = (a ^ 0x9E3779B9u) * 0x85EBCA6Bu;
    h ^= b + 0x7F4A7C15u + (h << 6) + (h >> 2);
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= h >> 15;
    h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}

__device__ __forceinline__ float nat_u01(uint32_t h) { return static_cast<float>(h >> 8) * (1.0f / 16777216.0f); }

// keyed bijection on [0, n): cycle-walked xor / odd-multiply / xorshift rounds on the next power of two
__device__ __forceinline__ uint32_t nat_permute(uint32_t i, uint32_t n, uint32_t key) {
    if (n <= 1) return 0;
    uint32_t w = n - 1;
    w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16;
    uint32_t k1 = nat_mix(key, 0x3C6EF372u) | 1u, k2 = nat_mix(key, 0xDAA66D2Bu) | 1u;
    do {
        i ^= key & w;
        i = (i * k1) & w;
        i ^= i >> 3;
        i ^= (key >> 11) & w;
        i = (i * k2) & w;
        i ^= i >> 5;
        i = (i * 0x2C1B3C6Du) & w;
        i ^= i >> 2;
    } while (i >= n);
    return i;
}

struct SampleSource {
    const float* rec;    // replay: this sample's record
    uint32_t pixel_key;  // native: nat_mix(seed_key, pixel index in the FULL sample window)
    uint32_t k;          // sample index within the pixel
    int spp, root;

    __device__ __forceinline__ uint32_t key(uint32_t pattern, uint32_t stratum) const {
        return nat_mix(nat_mix(pixel_key, pattern), stratum);
    }
    // 1D pattern `pattern` (n = 1 stratum on the path-tracer slots)
    __device__ __forceinline__ float native_1d(uint32_t pattern) const {
        uint32_t ky = key(2u + pattern, 0u);
        uint32_t j = nat_permute(k, static_cast<uint32_t>(spp), ky);
        float strata = 1.0f / 1.0f;
        float sub = strata / spp;
        float off = j + nat_u01(nat_mix(ky, k));
        return 0u * strata + off * sub;
    }
    // slot `slot` of an n-strata 1D pattern whose slots are consumed together with other patterns' (the BSSRDF block):
    // the slot's stratum is a keyed bijection of it per camera sample, standing in for the reference's in-pattern shuffle
    // (GoblinSampler.cpp:171-196) -- without it slot j of every pattern would sit in stratum j
    __device__ __forceinline__ uint32_t slot_stratum(uint32_t pattern_id, uint32_t n, uint32_t slot) const {
        return n > 1u ? nat_permute(slot, n, nat_mix(key(pattern_id, 0x5bd1e995u), k)) : slot;
    }
    __device__ __forceinline__ float native_1d_n(uint32_t pattern, uint32_t n, uint32_t slot) const {
        uint32_t st = slot_stratum(2u + pattern, n, slot);
        uint32_t ky = key(2u + pattern, st);
        uint32_t j = nat_permute(k, static_cast<uint32_t>(spp), ky);
        float strata = 1.0f / static_cast<float>(n);
        float sub = strata / spp;
        float off = j + nat_u01(nat_mix(ky, k));
        return st * strata + off * sub;
    }
    __device__ __forceinline__ void native_2d_slot(uint32_t pattern_id, uint32_t n, uint32_t slot, float* u, float* v) const {
        native_2d(pattern_id, n, slot_stratum(pattern_id, n, slot), true, u, v);
    }
    // 2D pattern: stratum i of an n-point pattern (n a perfect square), sub-cell grid root x root
    __device__ __forceinline__ void native_2d(uint32_t pattern_id, uint32_t n, uint32_t i, bool permute, float* u, float* v) const {
        uint32_t ky = key(pattern_id, i);
        uint32_t p = permute ? nat_permute(k, static_cast<uint32_t>(spp), ky) : k;
        int r = static_cast<int>(sqrtf(static_cast<float>(n)));
        float strata = 1.0f / r;
        float sub = strata / root;
        int ux = static_cast<int>(i) % r, uy = static_cast<int>(i) / r;
        int px = static_cast<int>(p) % root, py = static_cast<int>(p) / root;
        float xo = px + nat_u01(nat_mix(ky, 2u * k));
        float yo = py + nat_u01(nat_mix(ky, 2u * k + 1u));
        *u = ux * strata + xo * sub;
        *v = uy * strata + yo * sub;
    }
};
