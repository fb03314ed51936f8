// causal_conv1d.hip — depthwise causal conv1d (+SiLU) forward/backward for gfx950.
//
// Contract: cm_causal_conv1d_fwd / _bwd in include/conmamba_hip.h (replace
// causal_conv1d_cuda.causal_conv1d_fwd / _bwd, reference selective_scan_interface.py:182, 286;
// op definition: reference modules/mamba/bimamba.py:83-91, 278-279).
//
// HBM-bound streaming op.  Layout (batch, dim, seqlen), time contiguous: one thread owns one
// 16-byte vector of consecutive timesteps of one (batch, channel) row, so a wave reads/writes
// 1 KiB of contiguous row bytes per instruction; the (width-1)-element halo comes from the
// neighbouring vector (an L1/L2 hit).  Backward: one workgroup per row, dweight/dbias reduced
// wave -> LDS -> one fp32 atomic per (row, tap).
#include "cm_common.h"

namespace {

constexpr int kMaxW = 4;

template <typename IO>
__device__ __forceinline__ float ld(const IO *row, int t, int L) {
    return (t >= 0 && t < L) ? cm_elem<IO>::load(row + t) : 0.f;
}

// ------------------------------------------------------------------------------- forward
template <typename IO, int W, bool REV, bool VECOK>
__global__ __launch_bounds__(256) void conv_fwd_kernel(const cm_conv_args p) {
    constexpr int VEC = cm_elem<IO>::kVec;
    const int L = p.seqlen;
    const int vpr = (L + VEC - 1) / VEC;
    const int64_t total = (int64_t)p.batch * p.dim * vpr;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int tv = (int)(idx % vpr);
        const int64_t r = idx / vpr;
        const int d = (int)(r % p.dim);
        const int b = (int)(r / p.dim);
        const IO *xrow = reinterpret_cast<const IO *>(p.x) + (int64_t)b * p.x_bs + (int64_t)d * p.x_ds;
        IO *yrow = reinterpret_cast<IO *>(p.y) + (int64_t)b * p.y_bs + (int64_t)d * p.y_ds;
        const int t0 = tv * VEC;
        float w[W];
#pragma unroll
        for (int k = 0; k < W; ++k) w[k] = p.weight[d * W + k];
        const float bias = p.bias ? p.bias[d] : 0.f;
        // window of inputs covering outputs t0..t0+VEC-1: causal needs [t0-(W-1), t0+VEC), anti-causal [t0, t0+VEC+W-1)
        float xin[VEC + W - 1];
        const int base = REV ? t0 : t0 - (W - 1);
        if constexpr (VECOK) {
            alignas(16) IO tmp[VEC];
            *reinterpret_cast<uint4 *>(tmp) = *reinterpret_cast<const uint4 *>(xrow + t0);
#pragma unroll
            for (int j = 0; j < VEC; ++j) xin[(REV ? 0 : W - 1) + j] = cm_elem<IO>::load(&tmp[j]);
#pragma unroll
            for (int j = 0; j < W - 1; ++j) {
                const int pos = REV ? VEC + j : j;
                xin[pos] = ld(xrow, base + pos, L);
            }
        } else {
#pragma unroll
            for (int j = 0; j < VEC + W - 1; ++j) xin[j] = ld(xrow, base + j, L);
        }
        alignas(16) IO outv[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float acc = bias;
#pragma unroll
            for (int k = 0; k < W; ++k) {
                // causal: tap k multiplies x[t-(W-1)+k] -> xin[j+k]; anti-causal: x[t+(W-1)-k] -> xin[j+W-1-k]
                acc = fmaf(w[k], xin[REV ? j + W - 1 - k : j + k], acc);
            }
            if (p.silu) acc *= cm_sigmoid(acc);
            cm_elem<IO>::store(&outv[j], acc);
        }
        if constexpr (VECOK) {
            *reinterpret_cast<uint4 *>(yrow + t0) = *reinterpret_cast<const uint4 *>(outv);
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                if (t0 + j < L) yrow[t0 + j] = outv[j];
        }
    }
}

// ------------------------------------------------------------------------------ backward
// grid = (dim, batch); one workgroup walks one row.
template <typename IO, int W, bool REV>
__global__ __launch_bounds__(256) void conv_bwd_kernel(const cm_conv_args p) {
    constexpr int VEC = 4;   // timesteps per thread per iteration (element-wise access: alignment-free)
    const int L = p.seqlen;
    const int d = blockIdx.x, b = blockIdx.y;
    const IO *xrow = reinterpret_cast<const IO *>(p.x) + (int64_t)b * p.x_bs + (int64_t)d * p.x_ds;
    const IO *gyrow = reinterpret_cast<const IO *>(p.dy) + (int64_t)b * p.dy_bs + (int64_t)d * p.dy_ds;
    IO *dxrow = reinterpret_cast<IO *>(p.dx) + (int64_t)b * p.dx_bs + (int64_t)d * p.dx_ds;
    float w[W];
#pragma unroll
    for (int k = 0; k < W; ++k) w[k] = p.weight[d * W + k];
    const float bias = p.bias ? p.bias[d] : 0.f;
    float dw[W], db = 0.f;
#pragma unroll
    for (int k = 0; k < W; ++k) dw[k] = 0.f;

    // Work in a "causal frame": for REV the time axis is mirrored (s = L-1-t), which turns the
    // anti-causal conv into the causal one; X(s), G(s) read through the mirror.
    auto X = [&](int s) { return ld(xrow, REV ? L - 1 - s : s, L); };
    auto G = [&](int s) { return ld(gyrow, REV ? L - 1 - s : s, L); };

    for (int s0 = threadIdx.x * VEC; s0 < L; s0 += blockDim.x * VEC) {
        // inputs x[s0-(W-1) .. s0+VEC+W-2], pre-activation grads for s0 .. s0+VEC+W-2
        float xin[VEC + 2 * (W - 1)];
#pragma unroll
        for (int j = 0; j < VEC + 2 * (W - 1); ++j) xin[j] = X(s0 - (W - 1) + j);
        float gp[VEC + W - 1];
#pragma unroll
        for (int j = 0; j < VEC + W - 1; ++j) {
            const int s = s0 + j;
            float g = G(s);
            if (p.silu) {
                float pre = bias;
#pragma unroll
                for (int k = 0; k < W; ++k) pre = fmaf(w[k], xin[j + k], pre);
                const float sg = cm_sigmoid(pre);
                g *= sg * (1.f + pre * (1.f - sg));
            }
            gp[j] = (s < L) ? g : 0.f;
        }
        // dx[s] = sum_k w[k] * gp[s + (W-1) - k];   dweight[k] += gp[s] * x[s-(W-1)+k] for owned s
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < W; ++k) acc = fmaf(w[k], gp[j + W - 1 - k], acc);
            const int s = s0 + j;
            if (s < L) {
                cm_elem<IO>::store(dxrow + (REV ? L - 1 - s : s), acc);
                db += gp[j];
#pragma unroll
                for (int k = 0; k < W; ++k) dw[k] = fmaf(gp[j], xin[j + k], dw[k]);
            }
        }
    }
    // block reduction of dw[W], db
    __shared__ float red[4][W + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k <= W; ++k) {
        float v = k < W ? dw[k < W ? k : 0] : db;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x <= W) {
        const int k = threadIdx.x;
        const float v = red[0][k] + red[1][k] + red[2][k] + red[3][k];
        if (p.workspace) p.workspace[((int64_t)b * p.dim + d) * (W + 1) + k] = v;      // summed over the batch by conv_bwd_reduce_kernel
        else if (k < W) atomicAdd(p.dweight + d * W + k, v);
        else if (p.dbias) atomicAdd(p.dbias + d, v);
    }
}

// deterministic path: dweight[d][k] += sum_b partial[b][d][k] in batch order (k == W: dbias)
template <int W>
__global__ __launch_bounds__(256) void conv_bwd_reduce_kernel(const cm_conv_args p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.dim * (W + 1)) return;
    const int d = i / (W + 1), k = i % (W + 1);
    float acc = 0.f;
    for (int b = 0; b < p.batch; ++b) acc += p.workspace[((int64_t)b * p.dim + d) * (W + 1) + k];
    if (k < W) p.dweight[d * W + k] += acc;
    else if (p.dbias) p.dbias[d] += acc;
}

template <typename IO, int W>
int launch_fwd(const cm_conv_args &a, bool vecok) {
    constexpr int VEC = cm_elem<IO>::kVec;
    const int64_t total = (int64_t)a.batch * a.dim * ((a.seqlen + VEC - 1) / VEC);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, st, a);
        return cm_launch_status("cm_causal_conv1d_fwd");
    };
    if (a.reverse_time) return vecok ? go(conv_fwd_kernel<IO, W, true, true>) : go(conv_fwd_kernel<IO, W, true, false>);
    return vecok ? go(conv_fwd_kernel<IO, W, false, true>) : go(conv_fwd_kernel<IO, W, false, false>);
}

template <typename IO, int W>
int launch_bwd(const cm_conv_args &a) {
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    dim3 grid(a.dim, a.batch);
    if (a.reverse_time) hipLaunchKernelGGL((conv_bwd_kernel<IO, W, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_bwd_kernel<IO, W, false>), grid, dim3(256), 0, st, a);
    if (a.workspace)
        hipLaunchKernelGGL((conv_bwd_reduce_kernel<W>), dim3((a.dim * (W + 1) + 255) / 256), dim3(256), 0, st, a);
    return cm_launch_status("cm_causal_conv1d_bwd");
}

template <typename IO>
int by_width(const cm_conv_args &a, bool bwd, bool vecok) {
    switch (a.width) {
        case 2: return bwd ? launch_bwd<IO, 2>(a) : launch_fwd<IO, 2>(a, vecok);
        case 3: return bwd ? launch_bwd<IO, 3>(a) : launch_fwd<IO, 3>(a, vecok);
        case 4: return bwd ? launch_bwd<IO, 4>(a) : launch_fwd<IO, 4>(a, vecok);
        default:
            cm_set_error("causal_conv1d: width %d unsupported (2..%d)", a.width, kMaxW);
            return CM_EUNSUPPORTED;
    }
}

int check_common(const cm_conv_args &a, const char *who) {
    CM_REQUIRE(a.batch > 0 && a.dim > 0 && a.seqlen > 0, CM_EINVAL, "%s: bad sizes batch=%d dim=%d seqlen=%d", who,
               a.batch, a.dim, a.seqlen);
    CM_REQUIRE(a.batch <= 65535, CM_EINVAL, "%s: batch %d exceeds the grid limit 65535", who, a.batch);
    CM_REQUIRE(a.x && a.weight, CM_EINVAL, "%s: x/weight must be non-NULL", who);
    return CM_OK;
}

int by_dtype(const cm_conv_args &a, bool bwd, bool vecok) {
    switch (a.io_dtype) {
        case CM_F32: return by_width<float>(a, bwd, vecok);
        case CM_BF16: return by_width<cm_bf16>(a, bwd, vecok);
        case CM_F16: return by_width<cm_f16>(a, bwd, vecok);
        default:
            cm_set_error("causal_conv1d: unsupported io dtype %d", a.io_dtype);
            return CM_EUNSUPPORTED;
    }
}

}  // namespace

extern "C" int cm_causal_conv1d_fwd(const cm_conv_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "causal_conv1d_fwd: args is NULL");
    const cm_conv_args &a = *args;
    if (int rc = check_common(a, "causal_conv1d_fwd")) return rc;
    CM_REQUIRE(a.y != nullptr, CM_EINVAL, "causal_conv1d_fwd: y is NULL");
    const int vec = a.io_dtype == CM_F32 ? 4 : 8;
    const bool vecok = a.seqlen % vec == 0 && cm_aligned(a.x, 16) && cm_aligned(a.y, 16) && a.x_bs % vec == 0 &&
                       a.x_ds % vec == 0 && a.y_bs % vec == 0 && a.y_ds % vec == 0;
    return by_dtype(a, false, vecok);
}

extern "C" int cm_causal_conv1d_bwd(const cm_conv_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "causal_conv1d_bwd: args is NULL");
    const cm_conv_args &a = *args;
    if (int rc = check_common(a, "causal_conv1d_bwd")) return rc;
    CM_REQUIRE(a.dy && a.dx && a.dweight, CM_EINVAL, "causal_conv1d_bwd: dy/dx/dweight must be non-NULL");
    return by_dtype(a, true, false);
}
