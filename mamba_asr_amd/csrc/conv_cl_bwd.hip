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
#include <type_traits>

namespace {

constexpr int W = 4, TC = 4, CH = 8;   // measured at 32 x 1000 x 512 (us, reduce included): TC 4 x CH 8 69-72, TC 8 x CH 4 74, TC 4 x CH 4 82
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

// grid: x = 256-thread slices of the channel words, y = (sequence, time group): sequence and group are workgroup-uniform, so
// every tensor is one raw buffer descriptor (rows outside [0, seqlen) read 0 / are dropped) and a row access is ONE 32-bit
// offset register advanced by the row stride -- as flat 64-bit addresses the 52 rows of a chunk held 104 address registers
// (176-256 VGPRs, one or two waves per SIMD, each waiting on its own loads: 106 us for 230 MB).
template <typename IO, bool TWO>
__global__ __launch_bounds__(256) void conv_cl_bwd_kernel(const cm_conv_cl_bwd_args p, const int vpr, const int ngroup) {
    using WIO = word_io<IO>;
    constexpr int N = WIO::N, S = (int)sizeof(IO);
    const int vec = blockIdx.x * 256 + threadIdx.x, grp = blockIdx.y % ngroup, b = blockIdx.y / ngroup;
    if (vec >= vpr) return;
    const int c0 = vec * N, T = p.seqlen;
    auto rs = [&](const void *base, int64_t bs, int64_t ts) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<IO *>(reinterpret_cast<const IO *>(base) + (int64_t)b * bs), 0,
                                                 base ? (int)(((int64_t)(T - 1) * ts + p.dim) * S) : 0, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t xr_ = rs(p.x, p.x_bs, p.x_ts), gfr_ = rs(p.du_f, p.duf_bs, p.duf_ts), gbr_ = rs(TWO ? p.du_b : nullptr, p.dub_bs, p.dub_ts),
                                 zfr_ = rs(p.dz_f, p.dzf_bs, p.dzf_ts), zbr_ = rs(TWO ? p.dz_b : nullptr, p.dzb_bs, p.dzb_ts),
                                 dxr_ = rs(p.dx, p.dx_bs, p.dx_ts), dzr_ = rs(p.dz, p.dz_bs, p.dz_ts);
    const bool dz = p.dz != nullptr;
    // a row's offset is workgroup-uniform (scalar registers) + this thread's word offset; rows outside [0, seqlen) are zeroed by a
    // uniform select (the scalar offset is not part of the descriptor's range check)
    const int vo = c0 * S;
    // EDGE: the chunk touches rows outside [0, seqlen) (first / last chunks of a sequence only): per-row uniform selects; the interior
    // chunks carry none of that scalar state (as per-row flags it spilled scalar registers into ~90 vector registers)
    auto ld = [&](auto edge, const __amdgpu_buffer_rsrc_t r, int64_t ts, int t) -> uint32_t {
        if constexpr (decltype(edge)::value) {
            const bool in = t >= 0 && t < T;
            const uint32_t v = __builtin_amdgcn_raw_buffer_load_b32(r, vo, (int)((int64_t)(in ? t : 0) * ts * S), 0);
            return in ? v : 0u;
        } else return __builtin_amdgcn_raw_buffer_load_b32(r, vo, (int)((int64_t)t * ts * S), 0);
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

    auto chunk = [&](auto edge, const int t0) {
        uint32_t xr[TC + 6], gfr[TC + 3], gbr[TC + 3], zfr[TC], zbr[TC];
#pragma unroll
        for (int r = 0; r < TC + 6; ++r) xr[r] = ld(edge, xr_, p.x_ts, t0 - 3 + r);
#pragma unroll
        for (int r = 0; r < TC + 3; ++r) {
            gfr[r] = ld(edge, gfr_, p.duf_ts, t0 + r);
            gbr[r] = TWO ? ld(edge, gbr_, p.dub_ts, t0 - 3 + r) : 0u;
        }
        if (dz) {
#pragma unroll
            for (int r = 0; r < TC; ++r) {
                zfr[r] = ld(edge, zfr_, p.dzf_ts, t0 + r);
                zbr[r] = TWO ? ld(edge, zbr_, p.dzb_ts, t0 + r) : 0u;
            }
        }
        // streaming order over the chunk's TC + 3 pre-activation gradients: only 4-deep windows of them are live at a time (the
        // array form kept every row of the chunk in registers: ~200 VGPRs, two waves per SIMD waiting on their own loads)
        float dxo[TC][N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float dpf[4] = {0.f, 0.f, 0.f, 0.f}, dpb[4] = {0.f, 0.f, 0.f, 0.f};      // [i - 3 .. i]
#pragma unroll
            for (int i = 0; i < TC + 3; ++i) {
                float pf = bf[j], pb = bb[j];
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    pf = fmaf(wf[j][k], WIO::get(xr[i + k], j), pf);                          // step t0 + i: x[t - 3 + k]
                    if (TWO) pb = fmaf(wb[j][k], WIO::get(xr[i + 3 - k], j), pb);              // step t0 - 3 + i: x[s + 3 - k]
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) dpf[k] = dpf[k + 1], dpb[k] = dpb[k + 1];
                dpf[3] = WIO::get(gfr[i], j) * silu_grad(pf);
                dpb[3] = TWO ? WIO::get(gbr[i], j) * silu_grad(pb) : 0.f;
                if (i >= 3) {
                    const int m = i - 3;                                                        // dpf[k] = step m + k, dpb[k] = step m - 3 + k
                    float d = 0.f;
#pragma unroll
                    for (int k = 0; k < W; ++k) {
                        d = fmaf(wf[j][k], dpf[3 - k], d);
                        if (TWO) d = fmaf(wb[j][k], dpb[k], d);
                    }
                    dxo[m][j] = d;
#pragma unroll
                    for (int k = 0; k < W; ++k) {
                        acc[j][k] = fmaf(dpf[0], WIO::get(xr[m + k], j), acc[j][k]);
                        if (TWO) acc[j][W + 1 + k] = fmaf(dpb[3], WIO::get(xr[m + 6 - k], j), acc[j][W + 1 + k]);
                    }
                    acc[j][W] += dpf[0];
                    if (TWO) acc[j][2 * W + 1] += dpb[3];
                }
            }
            __builtin_amdgcn_sched_barrier(0);                       // one channel at a time: interleaved, the two channels' windows double the registers
        }
#pragma unroll
        for (int m = 0; m < TC; ++m) {
            if (!decltype(edge)::value || t0 + m < T) {
                __builtin_amdgcn_raw_buffer_store_b32(WIO::pack(dxo[m]), dxr_, vo, (int)((int64_t)(t0 + m) * p.dx_ts * S), 0);
                if (dz) {
                    float s[N];
#pragma unroll
                    for (int j = 0; j < N; ++j) s[j] = WIO::get(zfr[m], j) + (TWO ? WIO::get(zbr[m], j) : 0.f);
                    __builtin_amdgcn_raw_buffer_store_b32(WIO::pack(s), dzr_, vo, (int)((int64_t)(t0 + m) * p.dz_ts * S), 0);
                }
            }
        }
        };
#pragma unroll 1
    for (int ci = 0; ci < CH; ++ci) {
        const int t0 = (grp * CH + ci) * TC;
        if (t0 >= T) break;
        if (t0 >= 3 && t0 + TC + 3 <= T) chunk(std::false_type{}, t0);
        else chunk(std::true_type{}, t0);
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
    auto put = [&](float *dst) { *dst = p.overwrite ? a : *dst + a; };
    if (s < W) put(p.dweight_f + c * W + s);
    else if (s == W) { if (p.dbias_f) put(p.dbias_f + c); }
    else if (s < 2 * W + 1) { if (p.dweight_b) put(p.dweight_b + c * W + s - W - 1); }
    else if (p.dbias_b) put(p.dbias_b + c);
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
    CM_REQUIRE((int64_t)a.batch * ngroup <= 65535 * 32768LL && ((int64_t)(a.seqlen - 1) * 8192 + a.dim) * 4 < 2147483647LL, CM_EINVAL, "conv_cl_bwd: problem too large");
    const dim3 grid((unsigned)((vpr + 255) / 256), (unsigned)(a.batch * ngroup));
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
