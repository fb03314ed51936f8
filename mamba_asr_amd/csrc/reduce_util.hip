// reduce_util.hip — small deterministic reductions of the training step.
//   cm_sum_leading   out[j] = sum_b in[b][j]: the per-utterance weight-gradient products (batched GEMM over the batch axis: a
//                    K = batch * time GEMM with a 256 x 1024 output fills 64 workgroups; the batched form fills the chip) are
//                    folded here, fp32 accumulation in a fixed order, fp32 (the parameter's dtype) or bf16 out.  Replaces
//                    torch's generic reduce kernel (13-14 us per call, 288 calls per 32 x 40 s micro-batch) + the cast that
//                    followed it.
#include "cm_common.h"

namespace {

template <typename IN, typename OUT>
__global__ __launch_bounds__(256) void sum_leading_kernel(const IN *__restrict__ in, OUT *__restrict__ out, const int nb, const int64_t n) {
    constexpr int V = cm_elem<IN>::kVec;                       // elements per 16-byte vector
    const int64_t j = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V;
    if (j >= n) return;
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    const IN *p = in + j;
#pragma unroll 8
    for (int b = 0; b < nb; ++b) {
        const uint4 v = *reinterpret_cast<const uint4 *>(p + (int64_t)b * n);
        if constexpr (sizeof(IN) == 2) {
            acc[0] += cm_bf16_lo(v.x), acc[1] += cm_bf16_hi(v.x), acc[2] += cm_bf16_lo(v.y), acc[3] += cm_bf16_hi(v.y);
            acc[4] += cm_bf16_lo(v.z), acc[5] += cm_bf16_hi(v.z), acc[6] += cm_bf16_lo(v.w), acc[7] += cm_bf16_hi(v.w);
        } else {
            acc[0] += __uint_as_float(v.x), acc[1] += __uint_as_float(v.y), acc[2] += __uint_as_float(v.z), acc[3] += __uint_as_float(v.w);
        }
    }
#pragma unroll
    for (int k = 0; k < V; ++k) cm_elem<OUT>::store(out + j + k, acc[k]);
}

}  // namespace

extern "C" int cm_sum_leading(const void *in, void *out, int32_t nbatch, int64_t n, int32_t in_dtype, int32_t out_dtype, void *stream) {
    CM_REQUIRE(in && out && nbatch > 0 && n > 0, CM_EINVAL, "sum_leading: bad sizes or NULL tensor");
    CM_REQUIRE((in_dtype == CM_BF16 || in_dtype == CM_F32) && (out_dtype == CM_BF16 || out_dtype == CM_F32), CM_EUNSUPPORTED,
               "sum_leading: dtypes must be f32 or bf16");
    const int v = in_dtype == CM_BF16 ? 8 : 4;
    CM_REQUIRE(n % v == 0 && cm_aligned(in, 16) && cm_aligned(out, 16), CM_EALIGN, "sum_leading: n must be a multiple of %d, tensors 16-byte aligned", v);
    const int64_t threads = n / v;
    const dim3 grid((unsigned)((threads + 255) / 256));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (in_dtype == CM_BF16) {
        if (out_dtype == CM_F32) hipLaunchKernelGGL((sum_leading_kernel<cm_bf16, float>), grid, dim3(256), 0, st, (const cm_bf16 *)in, (float *)out, nbatch, n);
        else hipLaunchKernelGGL((sum_leading_kernel<cm_bf16, cm_bf16>), grid, dim3(256), 0, st, (const cm_bf16 *)in, (cm_bf16 *)out, nbatch, n);
    } else {
        if (out_dtype == CM_F32) hipLaunchKernelGGL((sum_leading_kernel<float, float>), grid, dim3(256), 0, st, (const float *)in, (float *)out, nbatch, n);
        else hipLaunchKernelGGL((sum_leading_kernel<float, cm_bf16>), grid, dim3(256), 0, st, (const float *)in, (cm_bf16 *)out, nbatch, n);
    }
    return cm_launch_status("cm_sum_leading");
}
