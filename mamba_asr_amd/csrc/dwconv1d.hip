// dwconv1d.hip — depthwise Conv1d over time, forward and backward, time-contiguous (batch, dim, seqlen) layout
// (contracts: cm_dwconv1d_fwd / cm_dwconv1d_bwd in include/conmamba_hip.h; the ConvolutionModule's depthwise stage,
// reference modules/Conmamba.py:271-284 (definition) and :443 (call): kernel 31, stride 1, zero padding, groups = dim).
//
// The module API (training path) ran this stage in the vendor library, which picks naive kernels for it on gfx950 and an
// im2col + GEMM per utterance for the weight gradient: 23 ms of a 181 ms ConMamba-large training step (32 x 40 s).
//   forward : one workgroup per (batch, channel) row: the row (+ K-1 zeros of padding) is staged in LDS as fp32, each
//             thread produces 8 consecutive outputs from a 8+K-1 register window;
//   backward: one workgroup per CHANNEL walks the batch: dx is the same computation on dy with flipped taps; the tap
//             gradients dw[k] = sum_{b,t} dy[t] x[t+k-pad] are accumulated per thread in registers over the whole batch
//             walk and reduced once (wave shuffles + LDS) -- no atomics, deterministic.
#include "cm_common.h"

namespace {

constexpr int KMAX = 32;          // taps held in registers (kernel sizes up to 32)
constexpr int NT = 256;
constexpr int TV = 8;             // outputs per thread per pass

template <typename IO>
__device__ __forceinline__ void stage_row(float *dst, const IO *src, int T, int lead, int total) {
    // dst[i] = src[i - lead] for lead <= i < lead + T, 0 elsewhere (i < total)
    for (int i = threadIdx.x; i < total; i += NT) {
        const int t = i - lead;
        dst[i] = (t >= 0 && t < T) ? cm_elem<IO>::load(src + t) : 0.f;
    }
}

template <typename IO>
__global__ __launch_bounds__(NT) void dwconv1d_fwd_kernel(const cm_dwconv1d_args p) {
    extern __shared__ float xs[];                                 // [T + KMAX]
    const int T = p.seqlen, K = p.ksize;
    const int c = blockIdx.x % p.dim, b = blockIdx.x / p.dim;
    const IO *x = reinterpret_cast<const IO *>(p.x) + (int64_t)b * p.x_bs + (int64_t)c * p.x_ds;
    IO *y = reinterpret_cast<IO *>(p.y) + (int64_t)b * p.y_bs + (int64_t)c * p.y_ds;
    stage_row(xs, x, T, p.pad_left, T + KMAX + TV);
    float w[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) w[k] = k < K ? p.weight[c * K + k] : 0.f;
    const float bias = p.bias ? p.bias[c] : 0.f;
    __syncthreads();
    for (int t0 = threadIdx.x * TV; t0 < T; t0 += NT * TV) {
        float win[TV + KMAX - 1];
#pragma unroll
        for (int i = 0; i < TV + KMAX - 1; ++i) win[i] = xs[t0 + i];
#pragma unroll
        for (int j = 0; j < TV; ++j) {
            float acc = bias;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) acc = fmaf(w[k], win[j + k], acc);
            if (t0 + j < T) cm_elem<IO>::store(y + t0 + j, acc);
        }
    }
}

template <typename IO>
__global__ __launch_bounds__(NT) void dwconv1d_bwd_kernel(const cm_dwconv1d_args p) {
    extern __shared__ float sm[];
    const int T = p.seqlen, K = p.ksize;
    const int row = T + KMAX + TV;
    float *xs = sm, *ds = sm + row;                               // x padded by pad_left, dy padded by K-1-pad_left
    float *red = ds + row;                                        // [4 waves][KMAX + 1]
    const int c = blockIdx.x;
    float w[KMAX], wr[KMAX];                                      // taps and flipped taps (for dx)
#pragma unroll
    for (int k = 0; k < KMAX; ++k) w[k] = k < K ? p.weight[c * K + k] : 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) wr[k] = k < K ? p.weight[c * K + (K - 1 - k)] : 0.f;
    float dw[KMAX], db = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) dw[k] = 0.f;
    for (int b = 0; b < p.batch; ++b) {
        const IO *x = reinterpret_cast<const IO *>(p.x) + (int64_t)b * p.x_bs + (int64_t)c * p.x_ds;
        const IO *dy = reinterpret_cast<const IO *>(p.dy) + (int64_t)b * p.dy_bs + (int64_t)c * p.dy_ds;
        IO *dx = reinterpret_cast<IO *>(p.dx) + (int64_t)b * p.dx_bs + (int64_t)c * p.dx_ds;
        __syncthreads();                                          // previous row is consumed
        stage_row(xs, x, T, p.pad_left, row);
        stage_row(ds, dy, T, K - 1 - p.pad_left, row);
        __syncthreads();
        for (int t0 = threadIdx.x * TV; t0 < T; t0 += NT * TV) {
            float win[TV + KMAX - 1];
            // dx[s] = sum_k w[K-1-k] * dys[s + k]
#pragma unroll
            for (int i = 0; i < TV + KMAX - 1; ++i) win[i] = ds[t0 + i];
#pragma unroll
            for (int j = 0; j < TV; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < KMAX; ++k) acc = fmaf(wr[k], win[j + k], acc);
                if (t0 + j < T) cm_elem<IO>::store(dx + t0 + j, acc);
            }
            // dw[k] += sum_j dy[t0+j] * xs[t0+j+k];  db += sum_j dy[t0+j]      (dy[t] = ds[t + K-1-pad_left])
            float g[TV];
#pragma unroll
            for (int j = 0; j < TV; ++j) { g[j] = ds[t0 + j + (K - 1 - p.pad_left)]; db += g[j]; }
#pragma unroll
            for (int i = 0; i < TV + KMAX - 1; ++i) win[i] = xs[t0 + i];
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
#pragma unroll
                for (int j = 0; j < TV; ++j) dw[k] = fmaf(g[j], win[j + k], dw[k]);
        }
    }
    // one reduction for the whole batch walk: wave shuffles, then the four waves through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto wsum = [](float v) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        return v;
    };
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const float s = wsum(dw[k]);
        if (lane == 0) red[wave * (KMAX + 1) + k] = s;
    }
    {
        const float s = wsum(db);
        if (lane == 0) red[wave * (KMAX + 1) + KMAX] = s;
    }
    __syncthreads();
    if (threadIdx.x <= KMAX) {
        const int k = threadIdx.x;
        const float s = (red[k] + red[(KMAX + 1) + k]) + (red[2 * (KMAX + 1) + k] + red[3 * (KMAX + 1) + k]);
        if (k < K) p.dweight[c * K + k] += s;                     // accumulated into (caller zero-initialises)
        if (k == KMAX && p.dbias) p.dbias[c] += s;
    }
}

int check(const cm_dwconv1d_args &a, const char *what, bool bwd) {
    CM_REQUIRE(a.batch > 0 && a.dim > 0 && a.seqlen > 0 && a.x && a.weight, CM_EINVAL, "%s: bad sizes or NULL tensor", what);
    CM_REQUIRE(a.ksize >= 1 && a.ksize <= KMAX && a.pad_left >= 0 && a.pad_left < a.ksize, CM_EUNSUPPORTED,
               "%s: kernel size %d / left padding %d unsupported (1..32, 0..k-1)", what, a.ksize, a.pad_left);
    CM_REQUIRE(a.io_dtype == CM_BF16 || a.io_dtype == CM_F32, CM_EUNSUPPORTED, "%s: unsupported dtype %d", what, a.io_dtype);
    CM_REQUIRE((size_t)(a.seqlen + KMAX + TV) * 4 * (bwd ? 2 : 1) + 1024 <= 150 * 1024, CM_EUNSUPPORTED, "%s: seqlen %d too long for the LDS row",
               what, a.seqlen);
    if (bwd) CM_REQUIRE(a.dy && a.dx && a.dweight, CM_EINVAL, "%s: dy / dx / dweight must be non-NULL", what);
    else CM_REQUIRE(a.y != nullptr, CM_EINVAL, "%s: y must be non-NULL", what);
    return CM_OK;
}

template <typename K>
int set_lds(K kern, size_t smem, const char *what) {
    if (smem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) { cm_set_error("%s: LDS attribute failed: %s", what, hipGetErrorString(e)); return (int)e; }
    }
    return CM_OK;
}

}  // namespace

extern "C" int cm_dwconv1d_fwd(const cm_dwconv1d_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "dwconv1d_fwd: args is NULL");
    const cm_dwconv1d_args &a = *args;
    if (int rc = check(a, "dwconv1d_fwd", false)) return rc;
    CM_REQUIRE((int64_t)a.batch * a.dim <= 2147483647LL, CM_EINVAL, "dwconv1d_fwd: too many rows");
    const size_t smem = (size_t)(a.seqlen + KMAX + TV) * 4;
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    dim3 grid((unsigned)(a.batch * a.dim));
    if (a.io_dtype == CM_BF16) {
        if (int rc = set_lds(dwconv1d_fwd_kernel<cm_bf16>, smem, "dwconv1d_fwd")) return rc;
        hipLaunchKernelGGL(dwconv1d_fwd_kernel<cm_bf16>, grid, dim3(NT), smem, st, a);
    } else {
        if (int rc = set_lds(dwconv1d_fwd_kernel<float>, smem, "dwconv1d_fwd")) return rc;
        hipLaunchKernelGGL(dwconv1d_fwd_kernel<float>, grid, dim3(NT), smem, st, a);
    }
    return cm_launch_status("cm_dwconv1d_fwd");
}

extern "C" int cm_dwconv1d_bwd(const cm_dwconv1d_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "dwconv1d_bwd: args is NULL");
    const cm_dwconv1d_args &a = *args;
    if (int rc = check(a, "dwconv1d_bwd", true)) return rc;
    const size_t smem = (size_t)(a.seqlen + KMAX + TV) * 4 * 2 + 4 * (KMAX + 1) * 4;
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    dim3 grid((unsigned)a.dim);
    if (a.io_dtype == CM_BF16) {
        if (int rc = set_lds(dwconv1d_bwd_kernel<cm_bf16>, smem, "dwconv1d_bwd")) return rc;
        hipLaunchKernelGGL(dwconv1d_bwd_kernel<cm_bf16>, grid, dim3(NT), smem, st, a);
    } else {
        if (int rc = set_lds(dwconv1d_bwd_kernel<float>, smem, "dwconv1d_bwd")) return rc;
        hipLaunchKernelGGL(dwconv1d_bwd_kernel<float>, grid, dim3(NT), smem, st, a);
    }
    return cm_launch_status("cm_dwconv1d_bwd");
}
