// conv_cl_bwd.hip — backward of the channels-last causal depthwise conv + SiLU of BOTH BiMamba directions in one pass
// (contract: cm_conv_cl_bwd; the reference runs causal_conv1d_cuda.causal_conv1d_bwd once per direction on (B, E, T) tensors,
// selective_scan_interface.py:286-288, and adds the two dxz of bimamba.py:223-248 through autograd).
//   forward (cm_conv_cl_fwd / cm_conv_xproj):  u_f[t] = silu(b_f + sum_k w_f[k] x[t - 3 + k]),  u_b[t] = silu(b_b + sum_k w_b[k] x[t + 3 - k])
//   here: dpre = du silu'(pre) with pre recomputed from x;  dx[t] = sum_k w_f[k] dpre_f[t + 3 - k] + sum_k w_b[k] dpre_b[t - 3 + k];
//         dw_f[k] = sum_t dpre_f[t] x[t - 3 + k], dw_b[k] = sum_t dpre_b[t] x[t + 3 - k], db = sum_t dpre;  dz = dz_f + dz_b.
// One thread = one 32-bit word of channels (2 x bf16 or 1 x fp32) and CH consecutive chunks of TC steps; all rows of a chunk
// are requested before the first use.  Tap / bias gradients: per-thread sums over its CH * TC steps -> workspace -> fixed-order
// second pass (deterministic).
#include "cm_common.h"

namespace {

constexpr int W = 4, TC = 8, CH = 8;
constexpr int NSLOT = 2 * (W + 1);            // per channel: dw_f[4], db_f, dw_b[4], db_b

__device__ __forceinline__ float silu_grad(float p) {
    const float s = cm_sigmoid(p);
    return s * fmaf(p, 1.f - s, 1.f);
}

template <typename IO> struct word_io;
template <> struct word_io<cm_bf16> {
    static constexpr int N = 2;
    static __device__ __forceinline__ float get(uint32_t w, int j) { return j ? cm_bf16_hi(w) : cm_bf16_lo(w); }
    static __device__ __forceinline__ uint32_t pack(const float *v) { return cm_pack_bf16(v[0], v[1]); }
};
template <> struct word_io<float> {
    static constexpr int N = 1;
    static __device__ __forceinline__ float get(uint32_t w, int) { return __uint_as_float(w); }
    static __device__ __forceinline__ uint32_t pack(const float *v) { return __float_as_uint(v[0]); }
};

template <typename IO, bool TWO>
__global__ __launch_bounds__(256) void conv_cl_bwd_kernel(const cm_conv_cl_bwd_args p, const int vpr, const int ngroup) {
    using WIO = word_io<IO>;
    constexpr int N = WIO::N;
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int vec = (int)(v % vpr), grp = (int)((v / vpr) % ngroup), b = (int)(v / ((int64_t)vpr * ngroup));
    if (b >= p.batch) return;
    const int c0 = vec * N, T = p.seqlen;
    auto ptr = [&](const void *base, int64_t bs) { return base ? reinterpret_cast<const IO *>(base) + (int64_t)b * bs + c0 : nullptr; };
    const IO *x = ptr(p.x, p.x_bs), *gf = ptr(p.du_f, p.duf_bs), *gb = TWO ? ptr(p.du_b, p.dub_bs) : nullptr;
    const IO *zf = ptr(p.dz_f, p.dzf_bs), *zb = TWO ? ptr(p.dz_b, p.dzb_bs) : nullptr;
    IO *dx = const_cast<IO *>(ptr(p.dx, p.dx_bs)), *dz = const_cast<IO *>(ptr(p.dz, p.dz_bs));
    auto ld = [&](const IO *base, int64_t ts, int t) -> uint32_t {
        return (base && t >= 0 && t < T) ? *reinterpret_cast<const uint32_t *>(base + (int64_t)t * ts) : 0u;
    };
    float wf[N][W], wb[N][W], bf[N], bb[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
#pragma unroll
        for (int k = 0; k < W; ++k) {
            wf[j][k] = p.weight_f[(c0 + j) * W + k];
            wb[j][k] = TWO ? p.weight_b[(c0 + j) * W + k] : 0.f;
        }
        bf[j] = p.bias_f ? p.bias_f[c0 + j] : 0.f;
        bb[j] = (TWO && p.bias_b) ? p.bias_b[c0 + j] : 0.f;
    }
    float acc[N][NSLOT];
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) acc[j][s] = 0.f;

    for (int ci = 0; ci < CH; ++ci) {
        const int t0 = (grp * CH + ci) * TC;
        if (t0 >= T) break;
        uint32_t xr[TC + 6], gfr[TC + 3], gbr[TC + 3], zfr[TC], zbr[TC];
#pragma unroll
        for (int r = 0; r < TC + 6; ++r) xr[r] = ld(x, p.x_ts, t0 - 3 + r);
#pragma unroll
        for (int r = 0; r < TC + 3; ++r) {
            gfr[r] = ld(gf, p.duf_ts, t0 + r);
            gbr[r] = TWO ? ld(gb, p.dub_ts, t0 - 3 + r) : 0u;
        }
        if (dz) {
#pragma unroll
            for (int r = 0; r < TC; ++r) {
                zfr[r] = ld(zf, p.dzf_ts, t0 + r);
                zbr[r] = TWO ? ld(zb, p.dzb_ts, t0 + r) : 0u;
            }
        }
        float dxo[TC][N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float xv[TC + 6], dpf[TC + 3], dpb[TC + 3];
#pragma unroll
            for (int r = 0; r < TC + 6; ++r) xv[r] = WIO::get(xr[r], j);
#pragma unroll
            for (int i = 0; i < TC + 3; ++i) {
                float pf = bf[j];
#pragma unroll
                for (int k = 0; k < W; ++k) pf = fmaf(wf[j][k], xv[i + k], pf);                  // step t0 + i: x[t - 3 + k]
                dpf[i] = WIO::get(gfr[i], j) * silu_grad(pf);
                if (TWO) {
                    float pb = bb[j];
#pragma unroll
                    for (int k = 0; k < W; ++k) pb = fmaf(wb[j][k], xv[i + 3 - k], pb);          // step t0 - 3 + i: x[s + 3 - k]
                    dpb[i] = WIO::get(gbr[i], j) * silu_grad(pb);
                } else dpb[i] = 0.f;
            }
#pragma unroll
            for (int m = 0; m < TC; ++m) {
                float d = 0.f;
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    d = fmaf(wf[j][k], dpf[m + 3 - k], d);
                    if (TWO) d = fmaf(wb[j][k], dpb[m + k], d);
                }
                dxo[m][j] = d;
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    acc[j][k] = fmaf(dpf[m], xv[m + k], acc[j][k]);
                    if (TWO) acc[j][W + 1 + k] = fmaf(dpb[m + 3], xv[m + 6 - k], acc[j][W + 1 + k]);
                }
                acc[j][W] += dpf[m];
                if (TWO) acc[j][2 * W + 1] += dpb[m + 3];
            }
        }
#pragma unroll
        for (int m = 0; m < TC; ++m) {
            if (t0 + m < T) {
                *reinterpret_cast<uint32_t *>(dx + (int64_t)(t0 + m) * p.dx_ts) = WIO::pack(dxo[m]);
                if (dz) {
                    float s[N];
#pragma unroll
                    for (int j = 0; j < N; ++j) s[j] = WIO::get(zfr[m], j) + (TWO ? WIO::get(zbr[m], j) : 0.f);
                    *reinterpret_cast<uint32_t *>(dz + (int64_t)(t0 + m) * p.dz_ts) = WIO::pack(s);
                }
            }
        }
    }
    float *ws = p.workspace + (((int64_t)b * ngroup + grp) * p.dim + c0) * NSLOT;
#pragma unroll
    for (int j = 0; j < N; ++j)
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) ws[j * NSLOT + s] = acc[j][s];
}

// fixed-order sum over (batch, time group), ACCUMULATED into the caller's fp32 gradients: 32 columns x 8 row groups per workgroup
__global__ __launch_bounds__(256) void conv_cl_bwd_reduce_kernel(const cm_conv_cl_bwd_args p, const int ngroup) {
    __shared__ float red[8][32];
    const int i = blockIdx.x * 32 + (threadIdx.x & 31), rg = threadIdx.x >> 5;
    const int ncol = p.dim * NSLOT;
    const int64_t n = (int64_t)p.batch * ngroup;
    float part = 0.f;
    if (i < ncol) {
#pragma unroll 8
        for (int64_t r = rg; r < n; r += 8) part += p.workspace[r * ncol + i];
    }
    red[rg][threadIdx.x & 31] = part;
    __syncthreads();
    if (rg != 0 || i >= ncol) return;
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) a += red[k][threadIdx.x & 31];
    const int c = i / NSLOT, s = i % NSLOT;
    if (s < W) p.dweight_f[c * W + s] += a;
    else if (s == W) { if (p.dbias_f) p.dbias_f[c] += a; }
    else if (s < 2 * W + 1) { if (p.dweight_b) p.dweight_b[c * W + s - W - 1] += a; }
    else if (p.dbias_b) p.dbias_b[c] += a;
}

inline int groups_for(int seqlen) { return (seqlen + TC * CH - 1) / (TC * CH); }

}  // namespace

extern "C" int64_t cm_conv_cl_bwd_workspace_floats(int32_t batch, int32_t seqlen, int32_t dim) {
    if (batch <= 0 || seqlen <= 0 || dim <= 0) return 0;
    return (int64_t)batch * groups_for(seqlen) * dim * NSLOT;
}

extern "C" int cm_conv_cl_bwd(const cm_conv_cl_bwd_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "conv_cl_bwd: args is NULL");
    const cm_conv_cl_bwd_args &a = *args;
    CM_REQUIRE(a.batch > 0 && a.seqlen > 0 && a.dim > 0, CM_EINVAL, "conv_cl_bwd: bad sizes");
    CM_REQUIRE(a.width == W, CM_EUNSUPPORTED, "conv_cl_bwd: width %d unsupported (4 only)", a.width);
    CM_REQUIRE(a.x && a.weight_f && a.du_f && a.dx && a.dweight_f, CM_EINVAL, "conv_cl_bwd: x / weight_f / du_f / dx / dweight_f must be non-NULL");
    const bool two = a.du_b != nullptr;
    CM_REQUIRE(!two || (a.weight_b && a.dweight_b), CM_EINVAL, "conv_cl_bwd: du_b given without weight_b / dweight_b");
    CM_REQUIRE(!a.dz || a.dz_f, CM_EINVAL, "conv_cl_bwd: dz given without dz_f");
    CM_REQUIRE(a.io_dtype == CM_BF16 || a.io_dtype == CM_F32, CM_EUNSUPPORTED, "conv_cl_bwd: unsupported dtype %d", a.io_dtype);
    const int n = a.io_dtype == CM_F32 ? 1 : 2;
    auto ok = [&](const void *ptr, int64_t bs, int64_t ts) { return !ptr || (cm_aligned(ptr, 4) && bs % n == 0 && ts % n == 0); };
    CM_REQUIRE(a.dim % n == 0 && ok(a.x, a.x_bs, a.x_ts) && ok(a.du_f, a.duf_bs, a.duf_ts) && ok(a.du_b, a.dub_bs, a.dub_ts) &&
                   ok(a.dz_f, a.dzf_bs, a.dzf_ts) && ok(a.dz_b, a.dzb_bs, a.dzb_ts) && ok(a.dx, a.dx_bs, a.dx_ts) && ok(a.dz, a.dz_bs, a.dz_ts),
               CM_EALIGN, "conv_cl_bwd: dim and strides must be multiples of %d elements, pointers 4-byte aligned", n);
    const int ngroup = groups_for(a.seqlen);
    CM_REQUIRE(a.workspace && a.workspace_floats >= cm_conv_cl_bwd_workspace_floats(a.batch, a.seqlen, a.dim), CM_EINVAL,
               "conv_cl_bwd: needs a workspace of cm_conv_cl_bwd_workspace_floats() floats");
    const int vpr = a.dim / n;
    const int64_t threads = (int64_t)a.batch * ngroup * vpr;
    CM_REQUIRE((threads + 255) / 256 <= 2147483647LL, CM_EINVAL, "conv_cl_bwd: problem too large");
    const dim3 grid((unsigned)((threads + 255) / 256));
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    if (a.io_dtype == CM_BF16) {
        if (two) hipLaunchKernelGGL((conv_cl_bwd_kernel<cm_bf16, true>), grid, dim3(256), 0, st, a, vpr, ngroup);
        else hipLaunchKernelGGL((conv_cl_bwd_kernel<cm_bf16, false>), grid, dim3(256), 0, st, a, vpr, ngroup);
    } else {
        if (two) hipLaunchKernelGGL((conv_cl_bwd_kernel<float, true>), grid, dim3(256), 0, st, a, vpr, ngroup);
        else hipLaunchKernelGGL((conv_cl_bwd_kernel<float, false>), grid, dim3(256), 0, st, a, vpr, ngroup);
    }
    if (int rc = cm_launch_status("cm_conv_cl_bwd")) return rc;
    hipLaunchKernelGGL(conv_cl_bwd_reduce_kernel, dim3((unsigned)((a.dim * NSLOT + 31) / 32)), dim3(256), 0, st, a, ngroup);
    return cm_launch_status("cm_conv_cl_bwd(reduce)");
}
