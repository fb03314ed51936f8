// dwconv_cl.hip — depthwise Conv1d over time on CHANNELS-LAST rows (batch, seqlen, dim), forward and backward
// (contracts: cm_dwconv_cl_fwd / cm_dwconv_cl_bwd in include/conmamba_hip.h; the ConvolutionModule's depthwise stage,
// reference modules/Conmamba.py:271-284, 443).  With it the module-API ConvolutionModule never leaves the
// (batch, time, channel) layout: the pointwise conv is a GEMM on rows, GLU runs over the last axis, and the four
// transposing copies per layer (forward + backward) around the (batch, dim, seqlen) kernels disappear.
//   thread = channel, a workgroup walks RUN consecutive tiles of TT steps of one utterance with a TT + K - 1 register
//   window per tile (rows are coalesced 2-byte-per-lane reads);
//   backward: dx = the same walk over dy with flipped taps; the tap gradients are accumulated in registers over the
//   workgroup's tiles, written as per-workgroup partials and summed by a second tiny kernel (deterministic).
#include "cm_common.h"

namespace {

constexpr int KMAX = 32;
constexpr int TT = 16;            // outputs per tile per thread
constexpr int RUN = 4;            // tiles per workgroup

template <typename IO>
__device__ __forceinline__ float ld_row(const IO *base, int64_t ts, int t, int T, int c) {
    return (t >= 0 && t < T) ? cm_elem<IO>::load(base + (int64_t)t * ts + c) : 0.f;
}

template <typename IO>
__global__ __launch_bounds__(256) void dwconv_cl_fwd_kernel(const cm_dwconv_cl_args p, int nrun) {
    const int T = p.seqlen, K = p.ksize, D = p.dim;
    const int b = blockIdx.x / nrun, r0 = (blockIdx.x % nrun) * RUN * TT;
    const IO *x = reinterpret_cast<const IO *>(p.x) + (int64_t)b * p.x_bs;
    IO *y = reinterpret_cast<IO *>(p.y) + (int64_t)b * p.y_bs;
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float w[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) w[k] = k < K ? p.weight[c * K + k] : 0.f;
        const float bias = p.bias ? p.bias[c] : 0.f;
        for (int t0 = r0; t0 < min(r0 + RUN * TT, T); t0 += TT) {
            float win[TT + KMAX - 1];
#pragma unroll
            for (int i = 0; i < TT + KMAX - 1; ++i) win[i] = ld_row(x, p.x_ts, t0 + i - p.pad_left, T, c);
#pragma unroll
            for (int j = 0; j < TT; ++j) {
                float acc = bias;
#pragma unroll
                for (int k = 0; k < KMAX; ++k) acc = fmaf(w[k], win[j + k], acc);
                if (t0 + j < T) cm_elem<IO>::store(y + (int64_t)(t0 + j) * p.y_ts + c, acc);
            }
        }
    }
}

template <typename IO>
__global__ __launch_bounds__(256) void dwconv_cl_bwd_kernel(const cm_dwconv_cl_args p, int nrun) {
    const int T = p.seqlen, K = p.ksize, D = p.dim;
    const int b = blockIdx.x / nrun, r0 = (blockIdx.x % nrun) * RUN * TT;
    const IO *x = reinterpret_cast<const IO *>(p.x) + (int64_t)b * p.x_bs;
    const IO *dy = reinterpret_cast<const IO *>(p.dy) + (int64_t)b * p.dy_bs;
    IO *dx = reinterpret_cast<IO *>(p.dx) + (int64_t)b * p.dx_bs;
    float *part = p.partial + (int64_t)blockIdx.x * D * (KMAX + 1);       // [workgroup][channel][KMAX + 1]
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float wr[KMAX];                                                  // flipped taps
#pragma unroll
        for (int k = 0; k < KMAX; ++k) wr[k] = k < K ? p.weight[c * K + (K - 1 - k)] : 0.f;
        float dw[KMAX], db = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) dw[k] = 0.f;
        for (int t0 = r0; t0 < min(r0 + RUN * TT, T); t0 += TT) {
            float win[TT + KMAX - 1];
            // dx[s] = sum_k w[K-1-k] * dy[s + k - (K-1-pad_left)]
#pragma unroll
            for (int i = 0; i < TT + KMAX - 1; ++i) win[i] = ld_row(dy, p.dy_ts, t0 + i - (K - 1 - p.pad_left), T, c);
#pragma unroll
            for (int j = 0; j < TT; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < KMAX; ++k) acc = fmaf(wr[k], win[j + k], acc);
                if (t0 + j < T) cm_elem<IO>::store(dx + (int64_t)(t0 + j) * p.dx_ts + c, acc);
            }
            // dw[k] += sum_j dy[t0+j] * x[t0+j+k-pad_left];  db += sum_j dy[t0+j]
            float g[TT];
#pragma unroll
            for (int j = 0; j < TT; ++j) { g[j] = ld_row(dy, p.dy_ts, t0 + j, T, c); db += g[j]; }   // (L1 hits; a run-time index
                                                                                                       // into win[] would go to scratch)
#pragma unroll
            for (int i = 0; i < TT + KMAX - 1; ++i) win[i] = ld_row(x, p.x_ts, t0 + i - p.pad_left, T, c);
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
#pragma unroll
                for (int j = 0; j < TT; ++j) dw[k] = fmaf(g[j], win[j + k], dw[k]);
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) part[c * (KMAX + 1) + k] = dw[k];
        part[c * (KMAX + 1) + KMAX] = db;
    }
}

// dweight[c][k] += sum over workgroups of partial[wg][c][k]; dbias[c] += ... (fixed order: deterministic).
// A workgroup owns 32 consecutive (c, k) columns; its 8 thread groups walk the partial rows 8 apart with 8 loads in
// flight each, then add up through LDS (one thread per column walking all rows serially took 120 us for 512 rows).
__global__ __launch_bounds__(256) void dwconv_cl_reduce_kernel(const cm_dwconv_cl_args p, int nwg) {
    __shared__ float red[8][32];
    const int n = p.dim * (KMAX + 1);
    const int i = blockIdx.x * 32 + (threadIdx.x & 31), grp = threadIdx.x >> 5;          // i = c * (KMAX + 1) + k
    float s = 0.f;
    if (i < n) {
#pragma unroll 8
        for (int wg = grp; wg < nwg; wg += 8) s += p.partial[(int64_t)wg * n + i];
    }
    red[grp][threadIdx.x & 31] = s;
    __syncthreads();
    if (grp != 0 || i >= n) return;
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) t += red[g][threadIdx.x & 31];
    const int c = i / (KMAX + 1), k = i % (KMAX + 1);
    if (k < p.ksize) p.dweight[c * p.ksize + k] += t;
    else if (k == KMAX && p.dbias) p.dbias[c] += t;
}

int check(const cm_dwconv_cl_args &a, const char *what) {
    CM_REQUIRE(a.batch > 0 && a.dim > 0 && a.seqlen > 0 && a.x && a.weight, CM_EINVAL, "%s: bad sizes or NULL tensor", what);
    CM_REQUIRE(a.ksize >= 1 && a.ksize <= KMAX && a.pad_left >= 0 && a.pad_left < a.ksize, CM_EUNSUPPORTED,
               "%s: kernel size %d / left padding %d unsupported (1..32, 0..k-1)", what, a.ksize, a.pad_left);
    CM_REQUIRE(a.io_dtype == CM_BF16 || a.io_dtype == CM_F32, CM_EUNSUPPORTED, "%s: unsupported dtype %d", what, a.io_dtype);
    CM_REQUIRE(a.pad_left <= KMAX - 1 && a.ksize - 1 - a.pad_left <= KMAX - 1, CM_EUNSUPPORTED, "%s: padding out of range", what);
    return CM_OK;
}

}  // namespace

extern "C" int64_t cm_dwconv_cl_workspace_floats(int32_t batch, int32_t seqlen, int32_t dim) {
    const int64_t nrun = (seqlen + RUN * TT - 1) / (RUN * TT);
    return (int64_t)batch * nrun * dim * (KMAX + 1);
}

extern "C" int cm_dwconv_cl_fwd(const cm_dwconv_cl_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "dwconv_cl_fwd: args is NULL");
    const cm_dwconv_cl_args &a = *args;
    if (int rc = check(a, "dwconv_cl_fwd")) return rc;
    CM_REQUIRE(a.y != nullptr, CM_EINVAL, "dwconv_cl_fwd: y must be non-NULL");
    const int nrun = (a.seqlen + RUN * TT - 1) / (RUN * TT);
    CM_REQUIRE((int64_t)a.batch * nrun <= 2147483647LL, CM_EINVAL, "dwconv_cl_fwd: grid too large");
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    dim3 grid((unsigned)(a.batch * nrun)), block(a.dim >= 256 ? 256 : ((a.dim + 63) / 64) * 64);
    if (a.io_dtype == CM_BF16) hipLaunchKernelGGL(dwconv_cl_fwd_kernel<cm_bf16>, grid, block, 0, st, a, nrun);
    else hipLaunchKernelGGL(dwconv_cl_fwd_kernel<float>, grid, block, 0, st, a, nrun);
    return cm_launch_status("cm_dwconv_cl_fwd");
}

extern "C" int cm_dwconv_cl_bwd(const cm_dwconv_cl_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "dwconv_cl_bwd: args is NULL");
    const cm_dwconv_cl_args &a = *args;
    if (int rc = check(a, "dwconv_cl_bwd")) return rc;
    CM_REQUIRE(a.dy && a.dx && a.dweight && a.partial, CM_EINVAL, "dwconv_cl_bwd: dy / dx / dweight / partial must be non-NULL");
    const int nrun = (a.seqlen + RUN * TT - 1) / (RUN * TT);
    const int64_t nwg = (int64_t)a.batch * nrun;
    CM_REQUIRE(nwg <= 2147483647LL, CM_EINVAL, "dwconv_cl_bwd: grid too large");
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    dim3 grid((unsigned)nwg), block(a.dim >= 256 ? 256 : ((a.dim + 63) / 64) * 64);
    if (a.io_dtype == CM_BF16) hipLaunchKernelGGL(dwconv_cl_bwd_kernel<cm_bf16>, grid, block, 0, st, a, nrun);
    else hipLaunchKernelGGL(dwconv_cl_bwd_kernel<float>, grid, block, 0, st, a, nrun);
    const int n = a.dim * (KMAX + 1);
    hipLaunchKernelGGL(dwconv_cl_reduce_kernel, dim3((n + 31) / 32), dim3(256), 0, st, a, (int)nwg);
    return cm_launch_status("cm_dwconv_cl_bwd");
}
