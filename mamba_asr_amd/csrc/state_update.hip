// state_update.hip — single-step (decode-time) Mamba updates for gfx950: the O(1)-per-token path of
// bimamba.Mamba.step (reference modules/mamba/bimamba.py:320-365).  Contracts: cm_causal_conv1d_update and
// cm_selective_state_update in include/conmamba_hip.h; they stand where the reference binds
// causal_conv1d.causal_conv1d_update (K5, bimamba.py:24, 337-343) and
// mamba_ssm.ops.triton.selective_state_update (K6, bimamba.py:29, 360-362).
// One thread per (batch, channel); the recurrent states (fp32) are read and written once, everything else is a few
// hundred bytes per thread: these are latency-bound kernels whose job is to be ONE launch each.
#include "cm_common.h"

namespace {

template <typename IO>
__global__ __launch_bounds__(256) void conv_update_kernel(const cm_conv_update_args p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)p.batch * p.dim) return;
    const int c = (int)(i % p.dim);
    const int W = p.width;
    float *st = p.conv_state + i * W;                             // (batch, dim, width): oldest sample first
    const float xn = cm_elem<IO>::load(reinterpret_cast<const IO *>(p.x) + i);
    float acc = p.bias ? p.bias[c] : 0.f;
    for (int k = 0; k + 1 < W; ++k) {
        const float v = st[k + 1];                                // shift left by one ...
        st[k] = v;
        acc = fmaf(p.weight[c * W + k], v, acc);
    }
    st[W - 1] = xn;                                               // ... and append the new sample
    acc = fmaf(p.weight[c * W + W - 1], xn, acc);
    if (p.silu) acc *= cm_sigmoid(acc);
    cm_elem<IO>::store(reinterpret_cast<IO *>(p.out) + i, acc);
}

template <typename IO>
__global__ __launch_bounds__(256) void state_update_kernel(const cm_state_update_args p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)p.batch * p.dim) return;
    const int c = (int)(i % p.dim), b = (int)(i / p.dim);
    const int N = p.dstate;
    const IO *xs = reinterpret_cast<const IO *>(p.x), *dts = reinterpret_cast<const IO *>(p.dt);
    const IO *Bs = reinterpret_cast<const IO *>(p.B) + (int64_t)b * N, *Cs = reinterpret_cast<const IO *>(p.C) + (int64_t)b * N;
    const float x = cm_elem<IO>::load(xs + i);
    float dt = cm_elem<IO>::load(dts + i) + (p.dt_bias ? p.dt_bias[c] : 0.f);
    if (p.dt_softplus) dt = cm_softplus(dt);
    float *st = p.state + i * N;                                  // (batch, dim, dstate) fp32, updated in place
    const float dtx = dt * x;
    float y = 0.f;
    for (int n = 0; n < N; ++n) {
        const float h = fmaf(cm_exp2(dt * p.A[c * N + n] * CM_LOG2E), st[n], dtx * cm_elem<IO>::load(Bs + n));
        st[n] = h;
        y = fmaf(h, cm_elem<IO>::load(Cs + n), y);
    }
    if (p.D) y = fmaf(p.D[c], x, y);
    if (p.z) {
        const float z = cm_elem<IO>::load(reinterpret_cast<const IO *>(p.z) + i);
        y *= z * cm_sigmoid(z);
    }
    cm_elem<IO>::store(reinterpret_cast<IO *>(p.out) + i, y);
}

}  // namespace

extern "C" int cm_causal_conv1d_update(const cm_conv_update_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "causal_conv1d_update: args is NULL");
    const cm_conv_update_args &a = *args;
    CM_REQUIRE(a.batch > 0 && a.dim > 0 && a.width >= 1 && a.width <= 64 && a.x && a.conv_state && a.weight && a.out, CM_EINVAL,
               "causal_conv1d_update: bad sizes or NULL tensor");
    const int64_t n = (int64_t)a.batch * a.dim;
    dim3 grid((unsigned)((n + 255) / 256));
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    if (a.io_dtype == CM_BF16) hipLaunchKernelGGL(conv_update_kernel<cm_bf16>, grid, dim3(256), 0, st, a);
    else if (a.io_dtype == CM_F32) hipLaunchKernelGGL(conv_update_kernel<float>, grid, dim3(256), 0, st, a);
    else { cm_set_error("causal_conv1d_update: unsupported dtype %d", a.io_dtype); return CM_EUNSUPPORTED; }
    return cm_launch_status("cm_causal_conv1d_update");
}

extern "C" int cm_selective_state_update(const cm_state_update_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "selective_state_update: args is NULL");
    const cm_state_update_args &a = *args;
    CM_REQUIRE(a.batch > 0 && a.dim > 0 && a.dstate >= 1 && a.dstate <= 256 && a.state && a.x && a.dt && a.A && a.B && a.C && a.out,
               CM_EINVAL, "selective_state_update: bad sizes or NULL tensor");
    const int64_t n = (int64_t)a.batch * a.dim;
    dim3 grid((unsigned)((n + 255) / 256));
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    if (a.io_dtype == CM_BF16) hipLaunchKernelGGL(state_update_kernel<cm_bf16>, grid, dim3(256), 0, st, a);
    else if (a.io_dtype == CM_F32) hipLaunchKernelGGL(state_update_kernel<float>, grid, dim3(256), 0, st, a);
    else { cm_set_error("selective_state_update: unsupported dtype %d", a.io_dtype); return CM_EUNSUPPORTED; }
    return cm_launch_status("cm_selective_state_update");
}
