// hrx_fr.h — F::from(u64) for F = halo2curves bn256::Fr, the field the reference's circuits are instantiated with
// (src/lib.rs:896, examples/regex.rs:9), as 4 little-endian u64 limbs in Montgomery form.
//
// halo2curves (pulled in through halo2-base v0.2.2, Cargo.toml:12-15; not vendored in the reference) represents an
// element x as the limbs of x * R mod r with R = 2^256, and `From<u64>` is `Fr([v,0,0,0]) * R2` = v * R mod r:
//   r  = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001     (BN254 scalar field)
//   R  = 2^256 mod r  = [ac96341c4ffffffb, 36fc76959f60cd29, 666ea36f7879462e, 0e0a77c19a07df2f]  (= Fr::one()'s limbs)
// Here v * R mod r is computed directly: q' = floor(v * K / 2^64) with K = floor(R * 2^64 / r) underestimates
// q = floor(v * R / r) by at most 2, so v * R - q' * r needs at most two corrective subtractions.  Plain C++ (unsigned
// __int128) so that hipcc compiles it for the kernel and g++/hipcc for the host-side known-answer tests.
#pragma once
#include <stdint.h>

#include "hrx_lane.h"

namespace hrx {

constexpr uint64_t kFrModulus[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
constexpr uint64_t kFrR[4] = {0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full};
constexpr uint64_t kFrK = 0x4a47462623a04a7aull;  // floor(R * 2^64 / r)

// out = v * R mod r (Montgomery form of v); canonical = true: out = v (the plain little-endian integer)
HRX_HD void fr_from_u64(uint64_t v, uint64_t (&out)[4], bool canonical = false) {
    if (canonical || v == 0) { out[0] = v; out[1] = out[2] = out[3] = 0; return; }
    typedef unsigned __int128 u128;
    // p = v * R (5 limbs), s = q' * r (5 limbs)
    const uint64_t q = (uint64_t)(((u128)v * kFrK) >> 64);
    uint64_t p[5], s[5];
    u128 c = 0, d = 0;
    for (int i = 0; i < 4; ++i) {
        c += (u128)v * kFrR[i];
        p[i] = (uint64_t)c;
        c >>= 64;
        d += (u128)q * kFrModulus[i];
        s[i] = (uint64_t)d;
        d >>= 64;
    }
    p[4] = (uint64_t)c;
    s[4] = (uint64_t)d;
    // t = p - s  (0 <= t < 3r)
    uint64_t t[5];
    uint64_t borrow = 0;
    for (int i = 0; i < 5; ++i) {
        const u128 x = (u128)p[i] - s[i] - borrow;
        t[i] = (uint64_t)x;
        borrow = (uint64_t)(x >> 64) & 1u;
    }
    for (int round = 0; round < 2; ++round) {  // t -= r while t >= r
        uint64_t u[5];
        uint64_t b = 0;
        for (int i = 0; i < 5; ++i) {
            const u128 x = (u128)t[i] - (i < 4 ? kFrModulus[i] : 0) - b;
            u[i] = (uint64_t)x;
            b = (uint64_t)(x >> 64) & 1u;
        }
        if (!b) { t[0] = u[0]; t[1] = u[1]; t[2] = u[2]; t[3] = u[3]; t[4] = u[4]; }
    }
    out[0] = t[0]; out[1] = t[1]; out[2] = t[2]; out[3] = t[3];
}

// The same for v < 2^32 on 32-bit limbs (every value of the compact witness is 16 bits or less): v_mad_u64_u32 chains,
// ~4x fewer instructions than the 64-bit form.  K32 = floor(R * 2^32 / r); q' underestimates q by at most 2.
constexpr uint32_t kFrR32[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
constexpr uint32_t kFrModulus32[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
constexpr uint32_t kFrK32 = 0x4a474626u;

HRX_HD void fr_from_u32(uint32_t v, uint32_t (&out)[8], bool canonical = false) {
    if (canonical || v == 0) {
        out[0] = v;
        for (int i = 1; i < 8; ++i) out[i] = 0;
        return;
    }
    const uint32_t q = (uint32_t)(((uint64_t)v * kFrK32) >> 32);
    uint32_t t[9];
    uint64_t c = 0, d = 0;
    uint32_t borrow = 0;
    for (int i = 0; i < 8; ++i) {   // t = v * R - q * r, limb by limb
        c += (uint64_t)v * kFrR32[i];
        d += (uint64_t)q * kFrModulus32[i];
        const uint64_t x = (uint64_t)(uint32_t)c - (uint32_t)d - borrow;
        t[i] = (uint32_t)x;
        borrow = (uint32_t)(x >> 32) & 1u;
        c >>= 32;
        d >>= 32;
    }
    t[8] = (uint32_t)c - (uint32_t)d - borrow;
    for (int round = 0; round < 2; ++round) {  // t -= r while t >= r
        uint32_t u[9], b = 0;
        for (int i = 0; i < 9; ++i) {
            const uint64_t x = (uint64_t)t[i] - (i < 8 ? kFrModulus32[i] : 0u) - b;
            u[i] = (uint32_t)x;
            b = (uint32_t)(x >> 32) & 1u;
        }
        if (!b)
            for (int i = 0; i < 9; ++i) t[i] = u[i];
    }
    for (int i = 0; i < 8; ++i) out[i] = t[i];
}

}  // namespace hrx