// scan_rows_common.h — helpers shared by the row-group channels-last scan kernels (scan_rows_fwd.hip, scan_rows_bwd.hip).
#pragma once
#include "cm_common.h"
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// softplus (beta 1, threshold 20; reference selective_scan_interface.py:112) as max(x, 0) + log1p(e^-|x|): two
// transcendentals + four plain slots and no selects.  Beyond x = 20 the log term is exactly 0 in fp32 (the reference's
// threshold branch); for very negative x the absolute error is that of 1 + e^x (6e-8).
__device__ __forceinline__ float softplus_rows(float x) {
    const float e = cm_exp2(-CM_LOG2E * fabsf(x));
    return fmaf(CM_LN2, cm_log2(1.0f + e), fmaxf(x, 0.f));
}

constexpr int TB = 16;        // steps per block (= MFMA M)
constexpr int XS = 52;        // floats per staged x_dbl row (48 + pad: conflict-free fragment reads)

template <typename IO> __device__ __forceinline__ float ld_io(const IO *p) { return cm_elem<IO>::load(p); }

__device__ __forceinline__ void unpack_store(float *dst, const uint4 v, float) {          // 4 fp32
    *reinterpret_cast<uint4 *>(dst) = v;
}
__device__ __forceinline__ void unpack_store(float *dst, const uint4 v, cm_bf16) {        // 8 bf16 -> 8 fp32
    *reinterpret_cast<f32x4 *>(dst) = f32x4{cm_bf16_lo(v.x), cm_bf16_hi(v.x), cm_bf16_lo(v.y), cm_bf16_hi(v.y)};
    *reinterpret_cast<f32x4 *>(dst + 4) = f32x4{cm_bf16_lo(v.z), cm_bf16_hi(v.z), cm_bf16_lo(v.w), cm_bf16_hi(v.w)};
}

// raw buffer descriptor over [base, base + bytes): loads past the end return 0 and stores past the end are dropped,
// which is how the ragged last block (steps >= seqlen) is handled without per-row compares
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, int64_t bytes) {
    const int n = bytes > 0x7fffffff ? 0x7fffffff : (int)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, n, 0x00020000);
}
__device__ __forceinline__ void st_io(const __amdgpu_buffer_rsrc_t r, int voff, int soff, float v, cm_bf16) {
    __builtin_amdgcn_raw_buffer_store_b16(cm_elem<cm_bf16>::to_bits(v), r, voff, soff, 0);
}
__device__ __forceinline__ void st_io(const __amdgpu_buffer_rsrc_t r, int voff, int soff, float v, float) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}

}  // namespace
