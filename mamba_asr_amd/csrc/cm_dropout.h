// cm_dropout.h — the counter-based dropout stream shared by every kernel that applies or re-derives a dropout mask.
// The reference's semantics are torch.nn.Dropout's: an independent Bernoulli(1 - p) keep decision per element, survivors scaled by
// 1 / (1 - p) (reference modules/Conmamba.py:597-617).  Here the decision for element e of a tensor is a pure function of
// (seed, e): kernels of the training forward apply it without storing a mask, and the backward kernels re-derive it.
//   * elements are taken in groups of 8 consecutive indices (group = e / 8): four 32-bit words per group, 16 bits per element;
//   * keep <=> the element's 16 bits >= round(p * 65536); the scale uses that quantised probability, so E[dropout(x)] = x exactly.
// Cost: five murmur-finaliser rounds per 8 elements (the per-element three-round hash of round 3's first version made the
// element-wise training kernels VALU-bound: 52 us for 167 MB).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

__device__ __forceinline__ uint32_t cm_hash32(uint32_t x) {           // murmur3 finaliser
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t cm_drop_thresh(float p) {
    const float t = p * 65536.0f + 0.5f;
    return p <= 0.f ? 0u : (t >= 65535.f ? 65535u : (uint32_t)t);
}
__host__ __device__ __forceinline__ float cm_drop_scale(float p) {
    return 1.0f / (1.0f - (float)cm_drop_thresh(p) * (1.0f / 65536.0f));
}
// effective seed of a launch: the caller's seed, plus (optionally) a device word read when the kernel runs, so that a captured
// hipGraph draws fresh decisions at every replay (cm_ffn_elem_args.seed_epoch); forward and backward of one replay see the same word
__device__ __forceinline__ uint64_t cm_drop_seed(uint64_t seed, const uint64_t *epoch) {
    return epoch ? seed + *epoch * 0x9E3779B97F4A7C15ull : seed;
}
__device__ __forceinline__ uint32_t cm_drop_base(uint64_t seed, uint64_t group) {
    return cm_hash32((uint32_t)group ^ (uint32_t)seed) + (uint32_t)(group >> 32) * 0x85EBCA6Bu + (uint32_t)(seed >> 32);
}
// word j (0..3) of a group: elements 2 j (low half) and 2 j + 1 (high half)
__device__ __forceinline__ uint32_t cm_drop_word(uint32_t base, int j) { return cm_hash32(base + (uint32_t)j * 0x9E3779B9u); }
// keep flags of the 8 elements of a group as a byte
__device__ __forceinline__ uint32_t cm_drop_keep8(uint64_t seed, uint64_t group, uint32_t thresh) {
    const uint32_t base = cm_drop_base(seed, group);
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t w = cm_drop_word(base, j);
        m |= ((w & 0xffffu) >= thresh ? 1u : 0u) << (2 * j);
        m |= ((w >> 16) >= thresh ? 1u : 0u) << (2 * j + 1);
    }
    return m;
}
// keep flags (4 bits) of elements 4 half .. 4 half + 3 of a group
__device__ __forceinline__ uint32_t cm_drop_keep4(uint64_t seed, uint64_t group, int half, uint32_t thresh) {
    const uint32_t base = cm_drop_base(seed, group);
    const uint32_t w0 = cm_drop_word(base, 2 * half), w1 = cm_drop_word(base, 2 * half + 1);
    return ((w0 & 0xffffu) >= thresh ? 1u : 0u) | ((w0 >> 16) >= thresh ? 2u : 0u) | ((w1 & 0xffffu) >= thresh ? 4u : 0u) |
           ((w1 >> 16) >= thresh ? 8u : 0u);
}
