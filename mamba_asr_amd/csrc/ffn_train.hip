// ffn_train.hip — the element-wise stages of a feed-forward module's TRAINING step on (rows, features) tensors, one kernel per
// stage and direction (the reference leaves them to torch: Linear -> activation -> Dropout -> Linear -> Dropout -> 0.5 x + residual,
// reference modules/Conmamba.py:597-617, 638-648: eight element-wise launches forward, ten backward per module and micro-batch):
//   cm_bias_act_dropout_fwd   y = dropout(act(a + bias))                    (a: GEMM output without bias; act: none | GELU: the erf form
//                             for fp32 results, cm_common.h's x sigmoid(x P(x^2)) form where the result is rounded to bf16)
//                             or, with `res`:  y = res + alpha * dropout(a + bias)   (fp32 residual stream)
//   cm_bias_act_dropout_bwd   da = alpha * dy * mask / (1 - p) * act'(a + bias);  dbias = column sums of da, through per-workgroup
//                             partial rows + a fixed-order second pass (deterministic)
// Dropout decisions are cm_dropout.h's function of (seed, element index): any Bernoulli(1 - p) mask is the reference's semantics
// (torch.nn.Dropout); the host draws the seed from torch's seeded generator.  The forward stores the mask (one byte per element)
// only when asked to; the backward reads a stored mask or, without one, re-derives the decisions from the seed (the training
// forward of cm_ffn_fused stores none).
#include "cm_common.h"
#include "cm_dropout.h"

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

// d/dx of the bf16 GELU form x sigmoid(x P(x^2)) (cm_gelu_bf16): s + x s (1 - s) (P + 2 x^2 P'), on two values; for results
// that are rounded to bf16 (|error| against the erf form's derivative < 2e-4)
__device__ __forceinline__ v2f gelu_grad_bf16_2(v2f x) {
    constexpr float c2 = 7.03033577e-04f * CM_LOG2E, c1 = -7.40112920e-02f * CM_LOG2E, c0 = -1.59501577f * CM_LOG2E;
    v2f x2 = x * x;
    x2 = v2f{fminf(x2.x, 64.0f), fminf(x2.y, 64.0f)};
    const v2f t = __builtin_elementwise_fma(x2, v2f{c2, c2}, v2f{c1, c1});
    const v2f q = __builtin_elementwise_fma(x2, t, v2f{c0, c0});                        // e = x q, s = 1 / (1 + 2^e)
    const v2f t2 = __builtin_elementwise_fma(x2, v2f{2.f * c2, 2.f * c2}, v2f{c1, c1});
    const v2f de = __builtin_elementwise_fma(x2 + x2, t2, q);                           // de/dx
    const v2f e = x * q;
    const v2f d = v2f{cm_exp2(e.x), cm_exp2(e.y)} + v2f{1.0f, 1.0f};
    const v2f s = {cm_rcp(d.x), cm_rcp(d.y)};
    const v2f ss = __builtin_elementwise_fma(-s, s, s);                                 // s (1 - s)
    return __builtin_elementwise_fma(x * ss, de * v2f{-CM_LN2, -CM_LN2}, s);
}

template <typename T> __device__ __forceinline__ void ld_vec(const T *p, float *f);
template <> __device__ __forceinline__ void ld_vec<cm_bf16>(const cm_bf16 *p, float *f) {
    const uint4 v = *reinterpret_cast<const uint4 *>(p);
    f[0] = cm_bf16_lo(v.x), f[1] = cm_bf16_hi(v.x), f[2] = cm_bf16_lo(v.y), f[3] = cm_bf16_hi(v.y);
    f[4] = cm_bf16_lo(v.z), f[5] = cm_bf16_hi(v.z), f[6] = cm_bf16_lo(v.w), f[7] = cm_bf16_hi(v.w);
}
template <> __device__ __forceinline__ void ld_vec<float>(const float *p, float *f) {
    const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
    f[0] = a.x, f[1] = a.y, f[2] = a.z, f[3] = a.w, f[4] = b.x, f[5] = b.y, f[6] = b.z, f[7] = b.w;
}
template <typename T> __device__ __forceinline__ void st_vec(T *p, const float *f);
template <> __device__ __forceinline__ void st_vec<cm_bf16>(cm_bf16 *p, const float *f) {
    *reinterpret_cast<uint4 *>(p) = make_uint4(cm_pack_bf16(f[0], f[1]), cm_pack_bf16(f[2], f[3]), cm_pack_bf16(f[4], f[5]), cm_pack_bf16(f[6], f[7]));
}
template <> __device__ __forceinline__ void st_vec<float>(float *p, const float *f) {
    *reinterpret_cast<float4 *>(p) = make_float4(f[0], f[1], f[2], f[3]);
    *reinterpret_cast<float4 *>(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
}

// one thread = 8 consecutive features of one row
template <typename AT, typename YT>
__global__ __launch_bounds__(256) void bias_act_dropout_fwd_kernel(const cm_ffn_elem_args p) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int vpr = p.dim / 8;
    if (v >= p.rows * vpr) return;
    const int64_t e0 = v * 8;
    const int c = (int)(v % vpr) * 8;
    float a[8], y[8], bs8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    ld_vec<AT>(reinterpret_cast<const AT *>(p.a) + e0, a);
    if (p.bias) ld_vec<float>(p.bias + c, bs8);                      // two 16-byte loads (per-element loads: 8 extra vector-memory instructions per thread)
    const bool drop = p.p > 0.f;
    const uint64_t seed = cm_drop_seed(p.seed, p.seed_epoch);
    const float scale = drop ? cm_drop_scale(p.p) : 1.f;
    const uint32_t keep8 = drop ? cm_drop_keep8(seed, (uint64_t)v, cm_drop_thresh(p.p)) : 0xffu;
    uint32_t mlo = 0, mhi = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float t = a[k] + bs8[k];
        if (p.act == 1) t = sizeof(YT) == 2 ? cm_gelu_bf16(t) : cm_gelu(t);
        const bool keep = (keep8 >> k) & 1u;
        (k < 4 ? mlo : mhi) |= (keep ? 1u : 0u) << (8 * (k & 3));
        y[k] = keep ? t * scale : 0.f;
    }
    if (drop && p.mask) *reinterpret_cast<uint2 *>(p.mask + e0) = make_uint2(mlo, mhi);
    if (p.res) {                                                     // y = res + alpha * dropout(a + bias), fp32 stream
        float r[8];
        ld_vec<float>(p.res + e0, r);
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] = fmaf(p.alpha, y[k], r[k]);
    }
    st_vec<YT>(reinterpret_cast<YT *>(p.y) + e0, y);
}

// workgroup = RP rows per pass x (dim / 8) threads per row, walking `rows_per_wg` rows; column sums of da in registers -> LDS -> one
// partial row per workgroup
template <typename AT, typename DYT>
__global__ __launch_bounds__(256) void bias_act_dropout_bwd_kernel(const cm_ffn_elem_args p, const int rows_per_wg) {
    __shared__ float red[256 * 8];
    const int vpr = p.dim / 8, rp = 256 / vpr;                      // threads per row, rows per pass
    const int tr = threadIdx.x / vpr, tc = threadIdx.x % vpr;
    const bool live = tr < rp;
    const int c = tc * 8;
    const bool drop = p.p > 0.f;
    const uint64_t seed = cm_drop_seed(p.seed, p.seed_epoch);
    const float dscale = drop ? cm_drop_scale(p.p) : 1.f, scale = p.alpha * dscale;
    const uint32_t thresh = cm_drop_thresh(p.p);
    float bs[8], acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bs[k] = p.bias ? p.bias[live ? c + k : 0] : 0.f, acc[k] = 0.f;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
    if (live) {
        for (int64_t r = r0 + tr; r < r0 + rows_per_wg && r < p.rows; r += rp) {
            const int64_t e0 = r * p.dim + c;
            float dy[8], da[8];
            ld_vec<DYT>(reinterpret_cast<const DYT *>(p.dy) + e0, dy);
            uint32_t keep8 = 0xffu;
            if (drop) {
                if (p.mask) {
                    const uint2 m = *reinterpret_cast<const uint2 *>(p.mask + e0);
#pragma unroll
                    for (int k = 0; k < 8; ++k) keep8 = (keep8 & ~(1u << k)) | ((((k < 4 ? m.x : m.y) >> (8 * (k & 3))) & 1u) << k);
                } else keep8 = cm_drop_keep8(seed, (uint64_t)e0 >> 3, thresh);
            }
            float a[8];
            if (p.act == 1) {
                ld_vec<AT>(reinterpret_cast<const AT *>(p.a) + e0, a);
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] += bs[k];
            }
            if constexpr (sizeof(AT) == 2) {
                if (p.act == 1) {
                    uint32_t o[4];
#pragma unroll
                    for (int k = 0; k < 8; k += 2) {
                        const uint32_t ka = (keep8 >> k) & 1u, kb = (keep8 >> (k + 1)) & 1u;
                        // with act_out: also the activation the forward fed to the next GEMM, from the same exponential
                        const v2f gg = p.act_out ? cm_gelu_grad_and_act_bf16_2(v2f{a[k], a[k + 1]}, ka, kb, dscale, o[k / 2])
                                                 : gelu_grad_bf16_2(v2f{a[k], a[k + 1]});
                        da[k] = ka ? dy[k] * scale * gg.x : 0.f;
                        da[k + 1] = kb ? dy[k + 1] * scale * gg.y : 0.f;
                    }
                    if (p.act_out) *reinterpret_cast<uint4 *>(reinterpret_cast<cm_bf16 *>(p.act_out) + e0) = make_uint4(o[0], o[1], o[2], o[3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) da[k] = ((keep8 >> k) & 1u) ? dy[k] * scale : 0.f;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float g = ((keep8 >> k) & 1u) ? dy[k] * scale : 0.f;
                    if (p.act == 1) g *= gelu_grad(a[k]);
                    da[k] = g;
                }
            }
            st_vec<AT>(reinterpret_cast<AT *>(p.da) + e0, da);
            if (p.dbias_part) {
                // the bias gradient sums what the GEMMs see: the stored (rounded) da
                float q[8];
                if constexpr (sizeof(AT) == 2) {
#pragma unroll
                    for (int k = 0; k < 8; k += 2) {
                        const uint32_t w = cm_pack_bf16(da[k], da[k + 1]);
                        q[k] = cm_bf16_lo(w), q[k + 1] = cm_bf16_hi(w);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) q[k] = da[k];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += q[k];
            }
        }
    }
    if (!p.dbias_part) return;
#pragma unroll
    for (int k = 0; k < 8; ++k) red[k * 256 + threadIdx.x] = acc[k];   // [k][thread]: consecutive threads on consecutive banks (was 8-way conflicts)
    __syncthreads();
    if (live && tr == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float s = 0.f;
            for (int j = 0; j < rp; ++j) s += red[k * 256 + j * vpr + tc];
            p.dbias_part[(int64_t)blockIdx.x * p.dim + c + k] = s;
        }
    }
}

// GLU over the feature axis (the convolution module's bottleneck, reference modules/Conmamba.py:268-274, 441): a (rows, 2 dim) ->
// y (rows, dim) = (a[:, :dim] + b[:dim]) * sigmoid(a[:, dim:] + b[dim:])
template <typename AT>
__global__ __launch_bounds__(256) void bias_glu_fwd_kernel(const cm_ffn_elem_args p) {
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int vpr = p.dim / 8;
    if (v >= p.rows * vpr) return;
    const int64_t r = v / vpr;
    const int c = (int)(v % vpr) * 8;
    float a1[8], a2[8], y[8], b1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, b2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const AT *ar = reinterpret_cast<const AT *>(p.a) + r * 2 * p.dim;
    ld_vec<AT>(ar + c, a1);
    ld_vec<AT>(ar + p.dim + c, a2);
    if (p.bias) ld_vec<float>(p.bias + c, b1), ld_vec<float>(p.bias + p.dim + c, b2);   // 16-byte loads (were 16 four-byte loads per thread)
#pragma unroll
    for (int k = 0; k < 8; ++k) y[k] = (a1[k] + b1[k]) * cm_sigmoid(a2[k] + b2[k]);
    st_vec<AT>(reinterpret_cast<AT *>(p.y) + r * p.dim + c, y);
}

// da (rows, 2 dim): value half dy * sig, gate half dy * (a1 + b1) * sig * (1 - sig); dbias (2 dim) partial rows as above
template <typename AT>
__global__ __launch_bounds__(256) void bias_glu_bwd_kernel(const cm_ffn_elem_args p, const int rows_per_wg) {
    __shared__ float red[256 * 16];
    const int vpr = p.dim / 8, rp = 256 / vpr;
    const int tr = threadIdx.x / vpr, tc = threadIdx.x % vpr;
    const bool live = tr < rp;
    const int c = tc * 8;
    float b1[8], b2[8], acc[16];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        b1[k] = p.bias ? p.bias[live ? c + k : 0] : 0.f;
        b2[k] = p.bias ? p.bias[live ? p.dim + c + k : 0] : 0.f;
        acc[k] = acc[8 + k] = 0.f;
    }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
    if (live) {
        for (int64_t r = r0 + tr; r < r0 + rows_per_wg && r < p.rows; r += rp) {
            float a1[8], a2[8], dy[8], d1[8], d2[8];
            const AT *ar = reinterpret_cast<const AT *>(p.a) + r * 2 * p.dim;
            ld_vec<AT>(ar + c, a1);
            ld_vec<AT>(ar + p.dim + c, a2);
            ld_vec<AT>(reinterpret_cast<const AT *>(p.dy) + r * p.dim + c, dy);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float sg = cm_sigmoid(a2[k] + b2[k]);
                d1[k] = dy[k] * sg;
                d2[k] = dy[k] * (a1[k] + b1[k]) * sg * (1.f - sg);
            }
            AT *dr = reinterpret_cast<AT *>(p.da) + r * 2 * p.dim;
            st_vec<AT>(dr + c, d1);
            st_vec<AT>(dr + p.dim + c, d2);
            if (p.dbias_part) {
                if constexpr (sizeof(AT) == 2) {                     // sum what the GEMMs see: the rounded values
#pragma unroll
                    for (int k = 0; k < 8; k += 2) {
                        const uint32_t w1 = cm_pack_bf16(d1[k], d1[k + 1]), w2 = cm_pack_bf16(d2[k], d2[k + 1]);
                        d1[k] = cm_bf16_lo(w1), d1[k + 1] = cm_bf16_hi(w1), d2[k] = cm_bf16_lo(w2), d2[k + 1] = cm_bf16_hi(w2);
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += d1[k], acc[8 + k] += d2[k];
            }
        }
    }
    if (!p.dbias_part) return;
#pragma unroll
    for (int k = 0; k < 16; ++k) red[k * 256 + threadIdx.x] = acc[k];  // [k][thread] (PMC: 86 % of this kernel's LDS cycles were bank conflicts)
    __syncthreads();
    if (live && tr == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float s = 0.f;
            for (int j = 0; j < rp; ++j) s += red[k * 256 + j * vpr + tc];
            p.dbias_part[(int64_t)blockIdx.x * 2 * p.dim + (k < 8 ? c + k : p.dim + c + k - 8)] = s;
        }
    }
}

// dbias[c] += sum over the partial rows, fixed order: 32 columns x 32 row groups per workgroup
__global__ __launch_bounds__(1024) void colsum_partials_kernel(const float *__restrict__ part, const int nrow, const int dim, float *__restrict__ out, const int overwrite) {
    __shared__ float red[32][32];           // 32 row groups: 8 left each of the few workgroups walking 250 dependent loads at 32 k rows (14 us)
    const int col = blockIdx.x * 32 + (threadIdx.x & 31), grp = threadIdx.x >> 5;
    float s = 0.f;
    if (col < dim) {
#pragma unroll 8
        for (int b = grp; b < nrow; b += 32) s += part[(int64_t)b * dim + col];
    }
    red[grp][threadIdx.x & 31] = s;
    __syncthreads();
    if (grp == 0 && col < dim) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += red[k][threadIdx.x & 31];
        out[col] = overwrite ? t : out[col] + t;
    }
}

constexpr int ROWS_PER_WG = 16;     // 64: 500 workgroups at 32 k rows = two waves per SIMD, each walking 32 dependent row passes (75 us for 268 MB)

int check(const cm_ffn_elem_args &a, const char *what) {
    CM_REQUIRE(a.rows > 0 && a.dim > 0, CM_EINVAL, "%s: bad sizes", what);
    CM_REQUIRE(a.dim % 8 == 0 && a.dim <= 2048, CM_EUNSUPPORTED, "%s: dim %d must be a multiple of 8, at most 2048", what, a.dim);
    CM_REQUIRE(a.io_dtype == CM_BF16 || a.io_dtype == CM_F32, CM_EUNSUPPORTED, "%s: io dtype %d unsupported", what, a.io_dtype);
    CM_REQUIRE(a.act >= 0 && a.act <= 2, CM_EUNSUPPORTED, "%s: act %d (0 none, 1 GELU, 2 GLU)", what, a.act);
    CM_REQUIRE(a.act != 2 || (!a.mask && !a.res && a.dim <= 1024), CM_EUNSUPPORTED, "%s: GLU takes no dropout / residual, dim <= 1024", what);
    CM_REQUIRE(a.p >= 0.f && a.p < 1.f && (!a.mask || (a.p > 0.f && cm_aligned(a.mask, 8))), CM_EINVAL,
               "%s: dropout needs 0 <= p < 1; a mask needs p > 0 and 8-byte alignment", what);
    CM_REQUIRE(!a.act_out || (a.act == 1 && a.io_dtype == CM_BF16 && cm_aligned(a.act_out, 16)), CM_EUNSUPPORTED,
               "%s: act_out (the recomputed activation) is the bf16 GELU form's, 16-byte aligned", what);
    return CM_OK;
}

}  // namespace

extern "C" int64_t cm_bias_act_dropout_bwd_workspace_floats(int64_t rows, int32_t dim) {
    if (rows <= 0 || dim <= 0) return 0;
    return ((rows + ROWS_PER_WG - 1) / ROWS_PER_WG) * 2 * dim;      // 2 dim columns: the GLU form's bias gradient
}

extern "C" int cm_bias_act_dropout_fwd(const cm_ffn_elem_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "bias_act_dropout_fwd: args is NULL");
    const cm_ffn_elem_args &a = *args;
    if (int rc = check(a, "bias_act_dropout_fwd")) return rc;
    CM_REQUIRE(a.a && a.y && cm_aligned(a.a, 16) && cm_aligned(a.y, 16) && (!a.res || cm_aligned(a.res, 16)) && (!a.bias || cm_aligned(a.bias, 16)), CM_EALIGN,
               "bias_act_dropout_fwd: a / y (/ res, bias) must be non-NULL and 16-byte aligned");
    const int64_t threads = a.rows * (a.dim / 8);
    const dim3 grid((unsigned)((threads + 255) / 256));
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    if (a.act == 2) {
        if (a.io_dtype == CM_BF16) hipLaunchKernelGGL((bias_glu_fwd_kernel<cm_bf16>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((bias_glu_fwd_kernel<float>), grid, dim3(256), 0, st, a);
        return cm_launch_status("cm_bias_act_dropout_fwd(glu)");
    }
    const bool yf = a.res != nullptr || a.io_dtype == CM_F32;        // the residual form writes the fp32 stream
    if (a.io_dtype == CM_BF16) {
        if (yf) hipLaunchKernelGGL((bias_act_dropout_fwd_kernel<cm_bf16, float>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((bias_act_dropout_fwd_kernel<cm_bf16, cm_bf16>), grid, dim3(256), 0, st, a);
    } else hipLaunchKernelGGL((bias_act_dropout_fwd_kernel<float, float>), grid, dim3(256), 0, st, a);
    return cm_launch_status("cm_bias_act_dropout_fwd");
}

extern "C" int cm_bias_act_dropout_bwd(const cm_ffn_elem_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "bias_act_dropout_bwd: args is NULL");
    const cm_ffn_elem_args &a = *args;
    if (int rc = check(a, "bias_act_dropout_bwd")) return rc;
    CM_REQUIRE(a.dy && a.da && cm_aligned(a.dy, 16) && cm_aligned(a.da, 16) && (a.act == 0 || (a.a && cm_aligned(a.a, 16))), CM_EALIGN,
               "bias_act_dropout_bwd: dy / da (/ a with an activation) must be non-NULL and 16-byte aligned");
    CM_REQUIRE(!a.dbias || a.dbias_part, CM_EINVAL, "bias_act_dropout_bwd: dbias needs the partial-row workspace");
    const int nwg = (int)((a.rows + ROWS_PER_WG - 1) / ROWS_PER_WG);
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    const bool dyf = a.dy_f32 != 0 || a.io_dtype == CM_F32;
    if (a.act == 2) {
        CM_REQUIRE(!a.dy_f32 || a.io_dtype == CM_F32, CM_EUNSUPPORTED, "bias_act_dropout_bwd: the GLU form takes dy in the I/O dtype");
        if (a.io_dtype == CM_BF16) hipLaunchKernelGGL((bias_glu_bwd_kernel<cm_bf16>), dim3(nwg), dim3(256), 0, st, a, ROWS_PER_WG);
        else hipLaunchKernelGGL((bias_glu_bwd_kernel<float>), dim3(nwg), dim3(256), 0, st, a, ROWS_PER_WG);
        if (int rc = cm_launch_status("cm_bias_act_dropout_bwd(glu)")) return rc;
        if (a.dbias) {
            hipLaunchKernelGGL(colsum_partials_kernel, dim3((2 * a.dim + 31) / 32), dim3(1024), 0, st, a.dbias_part, nwg, 2 * a.dim, a.dbias, a.overwrite);
            return cm_launch_status("cm_bias_act_dropout_bwd(glu reduce)");
        }
        return CM_OK;
    }
    if (a.io_dtype == CM_BF16) {
        if (dyf) hipLaunchKernelGGL((bias_act_dropout_bwd_kernel<cm_bf16, float>), dim3(nwg), dim3(256), 0, st, a, ROWS_PER_WG);
        else hipLaunchKernelGGL((bias_act_dropout_bwd_kernel<cm_bf16, cm_bf16>), dim3(nwg), dim3(256), 0, st, a, ROWS_PER_WG);
    } else hipLaunchKernelGGL((bias_act_dropout_bwd_kernel<float, float>), dim3(nwg), dim3(256), 0, st, a, ROWS_PER_WG);
    if (int rc = cm_launch_status("cm_bias_act_dropout_bwd")) return rc;
    if (a.dbias) {
        hipLaunchKernelGGL(colsum_partials_kernel, dim3((a.dim + 31) / 32), dim3(1024), 0, st, a.dbias_part, nwg, a.dim, a.dbias, a.overwrite);
        return cm_launch_status("cm_bias_act_dropout_bwd(reduce)");
    }
    return CM_OK;
}
