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

// ------------------------------------------------------------------------------------------------------------------
// Tiled kernels (round 3): the thread-per-channel kernels above read 2 bytes per lane and hold a 47-row register window
// (fwd 64 us, bwd 155 us per 32 x 1000 x 256 bf16 call: 0.3-0.8 TB/s).  Here a workgroup stages a tile of 32 steps + halo of
// 256 channels in LDS with 16-byte loads, a thread owns (channel pair, half of the steps), and the taps are v_pk_fma_f32 on
// the pair, accumulated BY INPUT ROW (each staged row is read once per use side): the layout of the inference path's
// dwconv_rows_kernel.  Backward = the same correlation over dy with flipped taps for dx, then the tap gradients by x row
// with the 16 dy values of the thread in registers; partial rows per workgroup as before (fixed-order second pass).
// ------------------------------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int TT2 = 32, CB = 256, HS = 16;          // steps per tile, channels per workgroup, steps per thread
constexpr int ROWS2 = TT2 + KMAX - 1;               // staged rows per tile

template <typename IO> __device__ __forceinline__ v2f ld_pair(const unsigned char *tile, int row, int pair) {
    if constexpr (sizeof(IO) == 2) {
        const uint32_t w = *reinterpret_cast<const uint32_t *>(tile + (row * CB + 2 * pair) * 2);
        return v2f{cm_bf16_lo(w), cm_bf16_hi(w)};
    } else {
        const float2 f = *reinterpret_cast<const float2 *>(tile + (row * CB + 2 * pair) * 4);
        return v2f{f.x, f.y};
    }
}
template <typename IO> __device__ __forceinline__ void st_pair(IO *dst, v2f v) {
    if constexpr (sizeof(IO) == 2) *reinterpret_cast<uint32_t *>(dst) = cm_pack_bf16(v.x, v.y);
    else *reinterpret_cast<float2 *>(dst) = make_float2(v.x, v.y);
}

// rows [t_first, t_first + ROWS2) x channels [c0, c0 + CB) of src -> tile (zero outside the sequence / past dim)
template <typename IO>
__device__ __forceinline__ void stage_tile(unsigned char *tile, const IO *src, int64_t ts, int t_first, int T, int c0, int D) {
    constexpr int VEC = cm_elem<IO>::kVec, CPR = CB / VEC, NPC = (ROWS2 * CPR + 255) / 256;
    // every piece of the thread is requested before the first is waited for (as a rolled loop: load, s_waitcnt vmcnt(0), ds_write per
    // piece -- NPC dependent round trips per tile)
    uint4 v[NPC];
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
        const int i = threadIdx.x + 256 * k, r = i / CPR, ch = c0 + (i % CPR) * VEC, t = t_first + r;
        v[k] = make_uint4(0u, 0u, 0u, 0u);
        if (i < ROWS2 * CPR && t >= 0 && t < T && ch < D) v[k] = *reinterpret_cast<const uint4 *>(src + (int64_t)t * ts + ch);
    }
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
        const int i = threadIdx.x + 256 * k;
        if (i < ROWS2 * CPR) *reinterpret_cast<uint4 *>(tile + (size_t)i * 16) = v[k];
    }
}

// the workgroup's taps, (channels c0 .. c0 + CB) x K floats contiguous in memory, through LDS: coalesced loads instead of 62
// stride-K loads per thread (measured: those loads, not the arithmetic, set the first version's 43 / 71 us).  -> w[k] of the
// thread's channel pair, flipped when asked.  Uses the tile region: call before staging, it ends with a barrier.
__device__ __forceinline__ void load_taps(float *scratch, const float *weight, int c0, int D, int K, int pair, bool flip, v2f (&w)[KMAX]) {
    const int n = min(CB, D - c0) * K;
    {
        constexpr int NW = CB * KMAX / 256;                           // the same for the taps: all requests first
        float wv[NW];
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int i = threadIdx.x + 256 * k;
            wv[k] = i < n ? weight[(int64_t)c0 * K + i] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int i = threadIdx.x + 256 * k;
            if (i < n) scratch[i] = wv[k];
        }
    }
    __syncthreads();
    const bool ok = c0 + 2 * pair < D;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int kk = flip ? K - 1 - k : k;
        w[k] = (ok && k < K) ? v2f{scratch[(2 * pair) * K + kk], scratch[(2 * pair + 1) * K + kk]} : v2f{0.f, 0.f};
    }
    __syncthreads();
}

// acc[j] += sum_k w[k] * tile[row0 + j + k], j < HS, accumulated by input row
template <typename IO>
__device__ __forceinline__ void corr16(const unsigned char *tile, int row0, int pair, const v2f (&w)[KMAX], v2f (&acc)[HS]) {
#pragma unroll
    for (int r = 0; r < HS + KMAX - 1; ++r) {
        const v2f xv = ld_pair<IO>(tile, row0 + r, pair);
#pragma unroll
        for (int j = 0; j < HS; ++j) {
            const int k = r - j;
            if (k >= 0 && k < KMAX) acc[j] = __builtin_elementwise_fma(w[k], xv, acc[j]);
        }
    }
}

template <typename IO>
__global__ __launch_bounds__(256) void dwconv_cl_fwd_tiled_kernel(const cm_dwconv_cl_args p, int ntile) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int T = p.seqlen, K = p.ksize, D = p.dim;
    const int b = blockIdx.x / ntile, t0 = (blockIdx.x % ntile) * TT2, c0 = blockIdx.y * CB;
    const int pair = threadIdx.x & 127, hf = threadIdx.x >> 7, c = c0 + 2 * pair;
    const IO *x = reinterpret_cast<const IO *>(p.x) + (int64_t)b * p.x_bs;
    IO *y = reinterpret_cast<IO *>(p.y) + (int64_t)b * p.y_bs;
    const bool ok = c < D;
    v2f w[KMAX], acc[HS];
    load_taps(reinterpret_cast<float *>(lds), p.weight, c0, D, K, pair, false, w);
    stage_tile<IO>(lds, x, p.x_ts, t0 - p.pad_left, T, c0, D);
    const v2f bias = (ok && p.bias) ? v2f{p.bias[c], p.bias[c + 1]} : v2f{0.f, 0.f};
#pragma unroll
    for (int j = 0; j < HS; ++j) acc[j] = bias;
    __syncthreads();
    corr16<IO>(lds, HS * hf, pair, w, acc);
    if (!ok) return;
#pragma unroll
    for (int j = 0; j < HS; ++j) {
        const int t = t0 + HS * hf + j;
        if (t < T) st_pair<IO>(y + (int64_t)t * p.y_ts + c, acc[j]);
    }
}

template <typename IO>
__global__ __launch_bounds__(256) void dwconv_cl_bwd_tiled_kernel(const cm_dwconv_cl_args p, int nrun) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int TILE = ROWS2 * CB * (int)sizeof(IO);
    unsigned char *xt = lds, *gt = lds + TILE;
    const int T = p.seqlen, K = p.ksize, D = p.dim;
    const int b = blockIdx.x / nrun, run = blockIdx.x % nrun, c0 = blockIdx.y * CB;
    const int pair = threadIdx.x & 127, hf = threadIdx.x >> 7, c = c0 + 2 * pair;
    const IO *x = reinterpret_cast<const IO *>(p.x) + (int64_t)b * p.x_bs;
    const IO *dy = reinterpret_cast<const IO *>(p.dy) + (int64_t)b * p.dy_bs;
    IO *dx = reinterpret_cast<IO *>(p.dx) + (int64_t)b * p.dx_bs;
    const bool ok = c < D;
    const int padr = K - 1 - p.pad_left;                                  // halo in front of the dy tile
    v2f dw[KMAX], db = {0.f, 0.f}, wr[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) dw[k] = v2f{0.f, 0.f};
    load_taps(reinterpret_cast<float *>(lds), p.weight, c0, D, K, pair, true, wr);          // flipped taps for dx
    for (int ti = 0; ti < RUN * TT / TT2; ++ti) {
        const int t0 = run * RUN * TT + ti * TT2;
        if (t0 >= T) break;
        __syncthreads();                                                  // the previous tile's readers are done
        stage_tile<IO>(xt, x, p.x_ts, t0 - p.pad_left, T, c0, D);
        stage_tile<IO>(gt, dy, p.dy_ts, t0 - padr, T, c0, D);
        __syncthreads();
        {   // dx[s] = sum_k' w[K-1-k'] dy[s + k' - (K-1-pad_left)]
            v2f acc[HS];
#pragma unroll
            for (int j = 0; j < HS; ++j) acc[j] = v2f{0.f, 0.f};
            corr16<IO>(gt, HS * hf, pair, wr, acc);
            if (ok) {
#pragma unroll
                for (int j = 0; j < HS; ++j) {
                    const int t = t0 + HS * hf + j;
                    if (t < T) st_pair<IO>(dx + (int64_t)t * p.dx_ts + c, acc[j]);
                }
            }
        }
        {   // dw[k] += sum_j dy[t0 + 16 hf + j] x[t0 + 16 hf + j + k - pad_left], by x row
            v2f g[HS];
#pragma unroll
            for (int j = 0; j < HS; ++j) { g[j] = ld_pair<IO>(gt, HS * hf + j + padr, pair); db += g[j]; }
#pragma unroll
            for (int r = 0; r < HS + KMAX - 1; ++r) {
                const v2f xv = ld_pair<IO>(xt, HS * hf + r, pair);
#pragma unroll
                for (int j = 0; j < HS; ++j) {
                    const int k = r - j;
                    if (k >= 0 && k < KMAX) dw[k] = __builtin_elementwise_fma(g[j], xv, dw[k]);
                }
            }
        }
    }
    // the two step-halves of a channel pair meet through LDS; one partial row per (sequence, run)
    __syncthreads();
    float *red = reinterpret_cast<float *>(lds);
    if (hf == 1) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) *reinterpret_cast<float2 *>(red + (pair * (KMAX + 1) + k) * 2) = make_float2(dw[k].x, dw[k].y);
        *reinterpret_cast<float2 *>(red + (pair * (KMAX + 1) + KMAX) * 2) = make_float2(db.x, db.y);
    }
    __syncthreads();
    if (hf == 0 && ok) {
        float *part = p.partial + (int64_t)blockIdx.x * D * (KMAX + 1);
#pragma unroll
        for (int k = 0; k <= KMAX; ++k) {
            const float2 o = *reinterpret_cast<const float2 *>(red + (pair * (KMAX + 1) + k) * 2);
            const v2f m = k < KMAX ? dw[k] : db;
            part[c * (KMAX + 1) + k] = m.x + o.x;
            part[(c + 1) * (KMAX + 1) + k] = m.y + o.y;
        }
    }
}

inline bool tiled_ok(const cm_dwconv_cl_args &a) {
    const int vec = a.io_dtype == CM_BF16 ? 8 : 4;
    auto al = [&](const void *ptr, int64_t bs, int64_t ts) { return !ptr || (cm_aligned(ptr, 16) && bs % vec == 0 && ts % vec == 0); };
    return a.dim % vec == 0 && al(a.x, a.x_bs, a.x_ts) && al(a.y, a.y_bs, a.y_ts) && al(a.dy, a.dy_bs, a.dy_ts) && al(a.dx, a.dx_bs, a.dx_ts);
}

template <typename K> int set_lds(K kern, size_t smem, const char *what) {
    if (smem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) { cm_set_error("%s: LDS attribute failed: %s", what, hipGetErrorString(e)); return (int)e; }
    }
    return CM_OK;
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
    if (k < p.ksize) p.dweight[c * p.ksize + k] = p.overwrite ? t : p.dweight[c * p.ksize + k] + t;
    else if (k == KMAX && p.dbias) p.dbias[c] = p.overwrite ? t : p.dbias[c] + t;
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
    if (tiled_ok(a)) {
        const int ntile = (a.seqlen + TT2 - 1) / TT2;
        const dim3 tg((unsigned)(a.batch * ntile), (unsigned)((a.dim + CB - 1) / CB));
        const size_t smem = (size_t)ROWS2 * CB * (a.io_dtype == CM_BF16 ? 2 : 4);
        if (a.io_dtype == CM_BF16) {
            if (int rc = set_lds(dwconv_cl_fwd_tiled_kernel<cm_bf16>, smem, "dwconv_cl_fwd")) return rc;
            hipLaunchKernelGGL(dwconv_cl_fwd_tiled_kernel<cm_bf16>, tg, dim3(256), smem, st, a, ntile);
        } else {
            if (int rc = set_lds(dwconv_cl_fwd_tiled_kernel<float>, smem, "dwconv_cl_fwd")) return rc;
            hipLaunchKernelGGL(dwconv_cl_fwd_tiled_kernel<float>, tg, dim3(256), smem, st, a, ntile);
        }
        return cm_launch_status("cm_dwconv_cl_fwd(tiled)");
    }
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
    if (tiled_ok(a)) {
        const dim3 tg((unsigned)nwg, (unsigned)((a.dim + CB - 1) / CB));
        const size_t smem = (size_t)2 * ROWS2 * CB * (a.io_dtype == CM_BF16 ? 2 : 4);
        if (a.io_dtype == CM_BF16) {
            if (int rc = set_lds(dwconv_cl_bwd_tiled_kernel<cm_bf16>, smem, "dwconv_cl_bwd")) return rc;
            hipLaunchKernelGGL(dwconv_cl_bwd_tiled_kernel<cm_bf16>, tg, dim3(256), smem, st, a, nrun);
        } else {
            if (int rc = set_lds(dwconv_cl_bwd_tiled_kernel<float>, smem, "dwconv_cl_bwd")) return rc;
            hipLaunchKernelGGL(dwconv_cl_bwd_tiled_kernel<float>, tg, dim3(256), smem, st, a, nrun);
        }
    } else if (a.io_dtype == CM_BF16) hipLaunchKernelGGL(dwconv_cl_bwd_kernel<cm_bf16>, grid, block, 0, st, a, nrun);
    else hipLaunchKernelGGL(dwconv_cl_bwd_kernel<float>, grid, block, 0, st, a, nrun);
    const int n = a.dim * (KMAX + 1);
    hipLaunchKernelGGL(dwconv_cl_reduce_kernel, dim3((n + 31) / 32), dim3(256), 0, st, a, (int)nwg);
    return cm_launch_status("cm_dwconv_cl_bwd");
}
