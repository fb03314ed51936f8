// cm_common.h — shared host/device helpers for libconmamba_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/conmamba_hip.h"

// ---------------------------------------------------------------------------------------
// host side: error reporting
// ---------------------------------------------------------------------------------------
void cm_set_error(const char *fmt, ...);

#define CM_REQUIRE(cond, code, ...)            \
    do {                                       \
        if (!(cond)) {                         \
            cm_set_error(__VA_ARGS__);         \
            return (code);                     \
        }                                      \
    } while (0)

static inline int cm_launch_status(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        cm_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return CM_OK;
}

// Ablation switch: exists in the ablation build only (-DCM_ABLATE, `make ablate`); the product library sees a constant 0, so
// every `cm_debug_get() == k` branch and the kernel variants behind it compile away and no process-global state is left.
#ifdef CM_ABLATE
extern "C" int cm_debug_get();
#else
static constexpr int cm_debug_get() { return 0; }
#endif

static inline bool cm_aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// ---------------------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------------------
#define CM_LOG2E 1.4426950408889634f
#define CM_LN2   0.6931471805599453f

typedef uint16_t cm_bf16_raw;   // storage type of bf16
typedef uint16_t cm_f16_raw;

template <typename T> struct cm_elem;   // io element traits
template <> struct cm_elem<float> {
    static constexpr int kVec = 4;   // elements per 16-byte vector
    static __device__ __forceinline__ float load(const float *p) { return *p; }
    static __device__ __forceinline__ void store(float *p, float v) { *p = v; }
};
struct cm_bf16 { uint16_t bits; };
struct cm_f16 { uint16_t bits; };
template <> struct cm_elem<cm_bf16> {
    static constexpr int kVec = 8;
    static __device__ __forceinline__ float from_bits(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
    static __device__ __forceinline__ uint16_t to_bits(float v) {
        __bf16 b = static_cast<__bf16>(v);       // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
        return __builtin_bit_cast(uint16_t, b);
    }
    static __device__ __forceinline__ float load(const cm_bf16 *p) { return from_bits(p->bits); }
    static __device__ __forceinline__ void store(cm_bf16 *p, float v) { p->bits = to_bits(v); }
};
template <> struct cm_elem<cm_f16> {
    static constexpr int kVec = 8;
    static __device__ __forceinline__ float from_bits(uint16_t b) { return static_cast<float>(__builtin_bit_cast(_Float16, b)); }
    static __device__ __forceinline__ uint16_t to_bits(float v) {
        _Float16 h = static_cast<_Float16>(v);
        return __builtin_bit_cast(uint16_t, h);
    }
    static __device__ __forceinline__ float load(const cm_f16 *p) { return from_bits(p->bits); }
    static __device__ __forceinline__ void store(cm_f16 *p, float v) { p->bits = to_bits(v); }
};

__device__ __forceinline__ float cm_bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float cm_bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t cm_pack_bf16(float lo, float hi) {      // one v_cvt_pk_bf16_f32 (round to nearest even)
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef __bf16 v2b __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v2f{lo, hi}, v2b));
}

// fast transcendental building blocks (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp)
__device__ __forceinline__ float cm_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float cm_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float cm_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float cm_sigmoid(float x) { return cm_rcp(1.0f + cm_exp2(-CM_LOG2E * x)); }
// GELU, erf form (torch.nn.GELU default; reference Conmamba.py FFN / convolution module activations):
// 0.5 x erfc(-x / sqrt 2) with erfc from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute, no cancellation on
// the negative side) -- ~14 issue slots instead of the ~45 of libm's erff, which made GELU the pace-setter of the
// kernels that end in it.
__device__ __forceinline__ float cm_gelu(float x) {
    const float s = fabsf(x) * 0.70710678118654752f;
    const float t = cm_rcp(fmaf(0.3275911f, s, 1.0f));
    float poly = fmaf(t, 1.061405429f, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = poly * t * cm_exp2(-CM_LOG2E * s * s);        // erfc(|x| / sqrt 2)
    return 0.5f * x * (x >= 0.f ? 2.0f - e : e);
}

// GELU for results that are rounded to bf16 right away (the FFN's hidden activations): x * sigmoid(x * P(x^2)) with
// P = a + b x^2 + c x^4 fitted (minimax on [-8, 8], scipy Nelder-Mead) to the erf form: max |error| 2.6e-5 absolute,
// i.e. below half a bf16 ulp wherever |GELU| > 0.013 and far below the error the reference's own bf16 rounding of the
// pre-activation introduces (4e-3 |x|).  7 plain + 2 transcendental issue slots instead of 16 + 2: in cm_ffn_fused the
// erf form's VALU work exceeded the MFMA work (profiles/r01: SQ_ACTIVE_INST_VALU 46 % vs MFMA busy 28 %).
__device__ __forceinline__ float cm_gelu_bf16(float x) {
    const float x2 = fminf(x * x, 64.0f);                       // beyond |x| = 8 the polynomial would turn over
    float q = fmaf(x2, 7.03033577e-04f * CM_LOG2E, -7.40112920e-02f * CM_LOG2E);
    q = fmaf(x2, q, -1.59501577f * CM_LOG2E);                   // -log2(e) * P(x^2)
    return x * cm_rcp(1.0f + cm_exp2(x * q));
}

// The same GELU on two values at once, rounded to a bf16 pair: the polynomial and the products are v_pk_*_f32 (two
// results per issue slot).  In cm_ffn_fused (64k rows) 99-100 us instead of 108-110 us with the scalar form.
__device__ __forceinline__ uint32_t cm_gelu_bf16_pack2(float a, float b) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef __bf16 v2b __attribute__((ext_vector_type(2)));
    const v2f x = {a, b};
    v2f x2 = x * x;
    x2 = v2f{fminf(x2.x, 64.0f), fminf(x2.y, 64.0f)};
    v2f q = __builtin_elementwise_fma(x2, v2f{7.03033577e-04f * CM_LOG2E, 7.03033577e-04f * CM_LOG2E},
                                      v2f{-7.40112920e-02f * CM_LOG2E, -7.40112920e-02f * CM_LOG2E});
    q = __builtin_elementwise_fma(x2, q, v2f{-1.59501577f * CM_LOG2E, -1.59501577f * CM_LOG2E});
    const v2f e = x * q;
    const v2f d = v2f{cm_exp2(e.x), cm_exp2(e.y)} + v2f{1.0f, 1.0f};
    const v2f r = x * v2f{cm_rcp(d.x), cm_rcp(d.y)};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(r, v2b));
}

// dropout(GELU(.)) of two bf16-valued pre-activations, rounded to a bf16 pair: the training forward (cm_ffn_fused) and the backward's
// recomputation (cm_bias_act_dropout_bwd) both call THIS, so the weight-gradient GEMM sees the activations the forward used.
__device__ __forceinline__ uint32_t cm_gelu_drop_bf16_pack2(float a, float b, uint32_t keep_a, uint32_t keep_b, float scale) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef __bf16 v2b __attribute__((ext_vector_type(2)));
    const v2f x = {a, b};
    v2f x2 = x * x;
    x2 = v2f{fminf(x2.x, 64.0f), fminf(x2.y, 64.0f)};
    v2f q = __builtin_elementwise_fma(x2, v2f{7.03033577e-04f * CM_LOG2E, 7.03033577e-04f * CM_LOG2E},
                                      v2f{-7.40112920e-02f * CM_LOG2E, -7.40112920e-02f * CM_LOG2E});
    q = __builtin_elementwise_fma(x2, q, v2f{-1.59501577f * CM_LOG2E, -1.59501577f * CM_LOG2E});
    const v2f e = x * q;
    const v2f d = v2f{cm_exp2(e.x), cm_exp2(e.y)} + v2f{1.0f, 1.0f};
    const v2f m = {keep_a ? scale : 0.f, keep_b ? scale : 0.f};
    const v2f r = (x * v2f{cm_rcp(d.x), cm_rcp(d.y)}) * m;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(r, v2b));
}

typedef float cm_v2f __attribute__((ext_vector_type(2)));
// the same derivative AND dropout(GELU(x)) as cm_gelu_drop_bf16_pack2 gives it (same operations in the same order: same bits), from
// one exponential and one reciprocal per element
__device__ __forceinline__ cm_v2f cm_gelu_grad_and_act_bf16_2(cm_v2f x, uint32_t keep_a, uint32_t keep_b, float scale, uint32_t &act) {
    typedef __bf16 v2b __attribute__((ext_vector_type(2)));
    constexpr float c2 = 7.03033577e-04f * CM_LOG2E, c1 = -7.40112920e-02f * CM_LOG2E, c0 = -1.59501577f * CM_LOG2E;
    cm_v2f x2 = x * x;
    x2 = cm_v2f{fminf(x2.x, 64.0f), fminf(x2.y, 64.0f)};
    cm_v2f q = __builtin_elementwise_fma(x2, cm_v2f{c2, c2}, cm_v2f{c1, c1});
    q = __builtin_elementwise_fma(x2, q, cm_v2f{c0, c0});
    const cm_v2f t2 = __builtin_elementwise_fma(x2, cm_v2f{2.f * c2, 2.f * c2}, cm_v2f{c1, c1});
    const cm_v2f de = __builtin_elementwise_fma(x2 + x2, t2, q);
    const cm_v2f e = x * q;
    const cm_v2f d = cm_v2f{cm_exp2(e.x), cm_exp2(e.y)} + cm_v2f{1.0f, 1.0f};
    const cm_v2f s = {cm_rcp(d.x), cm_rcp(d.y)};
    const cm_v2f m = {keep_a ? scale : 0.f, keep_b ? scale : 0.f};
    const cm_v2f r = (x * s) * m;
    act = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, v2b));
    const cm_v2f ss = __builtin_elementwise_fma(-s, s, s);
    return __builtin_elementwise_fma(x * ss, de * cm_v2f{-CM_LN2, -CM_LN2}, s);
}
// d/dx of 0.5 x erfc(-x / sqrt 2)
__device__ __forceinline__ float gelu_grad(float x) {
    const float cdf = cm_gelu(x) / (x == 0.f ? 1.f : x);            // Phi(x) for x != 0
    const float phi = 0.3989422804014327f * cm_exp2(-0.5f * CM_LOG2E * x * x);
    return (x == 0.f ? 0.5f : cdf) + x * phi;
}


// softplus, beta=1, threshold=20 (torch default; reference selective_scan_interface.py:112).
// For x < -15, log1p(e^x) == e^x to fp32 precision; using it avoids the 1+tiny cancellation.
__device__ __forceinline__ float cm_softplus(float x) {
    float ex = cm_exp2(CM_LOG2E * x);
    float sp = CM_LN2 * cm_log2(1.0f + ex);
    sp = x < -15.0f ? ex : sp;
    return x > 20.0f ? x : sp;
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. it makes every wave wait for its
// outstanding global STORES (and prefetch loads) at each barrier; kernels that barrier inside a streaming loop use this.
__device__ __forceinline__ void cm_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// DPP helpers (full-rate cross-lane moves inside a row of 16 lanes)
template <int CTRL> __device__ __forceinline__ float cm_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
#define CM_DPP_QUAD(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
#define CM_DPP_ROW_HALF_MIRROR 0x141
#define CM_DPP_ROW_MIRROR 0x140

// sum over groups of S adjacent lanes (S in {1,2,4,8,16}); every lane of the group gets the sum
template <int S> __device__ __forceinline__ float cm_group_sum(float v) {
    if constexpr (S >= 2) v += cm_dpp<CM_DPP_QUAD(1, 0, 3, 2)>(v);
    if constexpr (S >= 4) v += cm_dpp<CM_DPP_QUAD(2, 3, 0, 1)>(v);
    if constexpr (S >= 8) v += cm_dpp<CM_DPP_ROW_HALF_MIRROR>(v);
    if constexpr (S >= 16) v += cm_dpp<CM_DPP_ROW_MIRROR>(v);
    return v;
}

// broadcast from lane K of each group of Q adjacent lanes (Q in {1,2,4})
template <int Q, int K> __device__ __forceinline__ float cm_group_bcast(float v) {
    if constexpr (Q == 1) return v;
    else if constexpr (Q == 2) return cm_dpp<CM_DPP_QUAD(K, K, 2 + K, 2 + K)>(v);
    else return cm_dpp<CM_DPP_QUAD(K, K, K, K)>(v);
}
