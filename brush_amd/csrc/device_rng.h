// device_rng.h — the stochastic half of SplatTrainer::step on the device.
//
// The reference draws the mean noise from burn's GPU PRNG (`Tensor::random([N,3], Normal(0,1))`,
// brush-train/src/train.rs:395-399) and the background jitter from `rand::rng()` (train.rs:896-908);
// neither stream is reproducible outside burn, so only the DISTRIBUTIONS are contractual.
// Here both come from one counter-based generator, Philox-4x32-10 (Salmon et al., "Parallel random numbers:
// as easy as 1, 2, 3", SC'11 — the published round function and Weyl key schedule, checked against the
// Random123 known-answer vectors in tests/test_abi.py), keyed by (seed) and counted by (splat, step):
//   * no generator state in HBM, nothing to advance: a sample is a pure function of (seed, step, splat),
//   * data-parallel ranks holding identical replicas draw identical noise without exchanging anything,
//   * a splat that is not visible (97 % of them at the bench workload) costs no generator work at all.
#pragma once
#include <stdint.h>

#include "device_math.h"

namespace bh {

struct Philox4 { uint32_t x, y, z, w; };

// Philox-4x32-10: ten rounds of (two 32x32->64 multiplies, xor with the key), key bumped by the Weyl
// constants between rounds.  Integer-only, so host and device agree bit for bit.
__host__ __device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}

// 23 random bits -> u in (0,1): k + 0.5 is exact in f32 for k < 2^23, so neither 0 nor 1 can occur.
__host__ __device__ inline float unit_open(uint32_t bits) { return ((float)(bits >> 9) + 0.5f) * (1.0f / 8388608.0f); }

// stream ids (counter word 3): one per use, so no two consumers ever share a counter
constexpr uint32_t RNG_STREAM_MEAN_NOISE = 0x4D4E0001u;
constexpr uint32_t RNG_STREAM_BACKGROUND = 0x42470002u;

// Three independent N(0,1) samples for splat `i` at train step `step` (Box-Muller on two pairs of uniforms;
// the fourth normal is dropped).  v_sin_f32 / v_cos_f32 take their argument in revolutions: sin(2 pi u).
BH_DEV void normal3(uint64_t seed, uint32_t step, uint32_t i, float& n0, float& n1, float& n2) {
    const Philox4 r = philox4x32_10(i, step, 0u, RNG_STREAM_MEAN_NOISE, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float ra = __builtin_sqrtf(-2.0f * bh_logf(unit_open(r.x)));
    const float rb = __builtin_sqrtf(-2.0f * bh_logf(unit_open(r.z)));
    const float ua = unit_open(r.y), ub = unit_open(r.w);
    n0 = ra * __builtin_amdgcn_cosf(ua);
    n1 = ra * __builtin_amdgcn_sinf(ua);
    n2 = rb * __builtin_amdgcn_cosf(ub);
}

// train.rs:389-391: (1 - sigmoid(raw_opac))^150 clamped to [0,1], times the visibility gate.  visible is a
// 0/1 flag; a data-parallel caller hands in the SUM over its views -> min(v, 1).
BH_DEV float mean_noise_gate(float raw_opac, float visible) {
    const float inv_opac = 1.0f - sigmoid(raw_opac);
    // x^150 = x^128 * x^16 * x^4 * x^2
    const float x2 = inv_opac * inv_opac, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8, x32 = x16 * x16, x64 = x32 * x32, x128 = x64 * x64;
    return clampf(x128 * x16 * x4 * x2, 0.0f, 1.0f) * __builtin_fminf(visible, 1.0f);
}

}  // namespace bh
