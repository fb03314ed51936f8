// layernorm_train.hip — LayerNorm over the last axis with saved statistics, forward and backward, for the training path
// (contracts: cm_layernorm_fwd / cm_layernorm_bwd in include/conmamba_hip.h).  The reference builds every one of the
// encoder layer's five normalisations from torch's LayerNorm (modules/Conmamba.py:262, :287, :597-620 through
// speechbrain's wrapper; eps 1e-5, final norm 1e-6 at :687); under autocast they run in fp32 whatever the input type.
//
// In the training step (32 x 40 s, 110 LayerNorms forward and backward) the vendor kernels moved 1.7 TB/s: 38 us forward,
// 56 + 36 us backward per call on (32000, 256) rows -- 12 % of the step.  Here:
//   forward : 16 lanes per row when dim <= 256 (4 rows per wave instruction, statistics = in-lane adds + 4 DPP steps),
//             64 lanes per row up to dim 1024; every lane has up to four 16-byte loads in flight; mean and rstd are
//             saved for the backward (fp32, one pair per row);
//   backward: same mapping; dx = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma; the gamma / beta gradients are
//             accumulated per lane in registers over all the rows a wave walks, reduced once per workgroup (shuffles +
//             LDS) into a (workgroups, 2, dim) partial array, and summed by a second kernel in a fixed order
//             (deterministic, no atomics).
#include "cm_common.h"

namespace {

template <int LPR> constexpr int nthreads() { return LPR == 64 ? 256 : 512; }   // 4 or 8 waves (the backward's LDS column sums fit either way)
constexpr int MAXV = 4;                 // 16-byte column groups per lane
constexpr int BWD_BLOCKS = 512;         // most workgroups of a backward = rows of the partial array (sizes the workspace)
constexpr int BWD_BLOCKS_ROWS = 256;    // the narrow-row backward's grid: one 8-wave workgroup per CU (two 4-wave ones when a row takes a whole wave)

template <typename T> __device__ __forceinline__ float4 ld4(const T *p);
template <> __device__ __forceinline__ float4 ld4<float>(const float *p) { return *reinterpret_cast<const float4 *>(p); }
template <> __device__ __forceinline__ float4 ld4<cm_bf16>(const cm_bf16 *p) {
    const uint2 w = *reinterpret_cast<const uint2 *>(p);
    return make_float4(cm_bf16_lo(w.x), cm_bf16_hi(w.x), cm_bf16_lo(w.y), cm_bf16_hi(w.y));
}
__device__ __forceinline__ void st4(float *p, const float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ void st4(cm_bf16 *p, const float4 v) {
    uint2 w;
    w.x = (uint32_t)cm_elem<cm_bf16>::to_bits(v.x) | ((uint32_t)cm_elem<cm_bf16>::to_bits(v.y) << 16);
    w.y = (uint32_t)cm_elem<cm_bf16>::to_bits(v.z) | ((uint32_t)cm_elem<cm_bf16>::to_bits(v.w) << 16);
    *reinterpret_cast<uint2 *>(p) = w;
}

// Optional epilogue (the front end's Conv2d blocks: LayerNorm over (freq, channel) -> LeakyReLU -> Dropout2d, speechbrain
// ConvolutionFrontEnd; reference hparams/CTC/conmamba_large.yaml:187-194): y = leaky(LN(x)) * chan_mask[row / mask_rows][col % mask_c].
// As three torch kernels behind the LayerNorm the block's fp32 activations made four extra trips through HBM forward and backward.
// mrow: the mask row of this LayerNorm row (chan_mask + (row / mask_rows) * mask_c, computed ONCE per row: a 64-bit division per
// 16-byte group made the wide forward 1.8x slower than without the epilogue), mc: the group's channel offset c % mask_c (fixed per thread)
__device__ __forceinline__ float4 ln_epilogue(const cm_layernorm_args &p, float4 y, const float *mrow, int mc) {
    if (p.act == 1) {
        y.x = y.x > 0.f ? y.x : y.x * p.act_slope; y.y = y.y > 0.f ? y.y : y.y * p.act_slope;
        y.z = y.z > 0.f ? y.z : y.z * p.act_slope; y.w = y.w > 0.f ? y.w : y.w * p.act_slope;
    }
    if (mrow) {
        const float4 m = *reinterpret_cast<const float4 *>(mrow + mc);
        y.x *= m.x, y.y *= m.y, y.z *= m.z, y.w *= m.w;
    }
    return y;
}
// the gradient arriving at LN's output through that epilogue: d * mask * leaky'(z), z = xhat * gamma + beta recomputed
__device__ __forceinline__ float4 ln_epilogue_grad(const cm_layernorm_args &p, float4 d, float4 xh, float4 g, const float *mrow, int mc, int c) {
    if (mrow) {
        const float4 m = *reinterpret_cast<const float4 *>(mrow + mc);
        d.x *= m.x, d.y *= m.y, d.z *= m.z, d.w *= m.w;
    }
    if (p.act == 1) {
        const float4 b = *reinterpret_cast<const float4 *>(p.beta + c);
        d.x *= fmaf(xh.x, g.x, b.x) > 0.f ? 1.f : p.act_slope; d.y *= fmaf(xh.y, g.y, b.y) > 0.f ? 1.f : p.act_slope;
        d.z *= fmaf(xh.z, g.z, b.z) > 0.f ? 1.f : p.act_slope; d.w *= fmaf(xh.w, g.w, b.w) > 0.f ? 1.f : p.act_slope;
    }
    return d;
}
__device__ __forceinline__ const float *ln_mask_row(const cm_layernorm_args &p, int64_t row) {
    return p.chan_mask ? p.chan_mask + (int64_t)((uint32_t)row / (uint32_t)p.mask_rows) * p.mask_c : nullptr;
}

template <int LPR> __device__ __forceinline__ float row_sum(float v) {
    v = cm_group_sum<16>(v);
    if constexpr (LPR == 64) {
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
    }
    return v;
}

// LPR lanes per row; lane lr of a row owns the 4-column groups lr + LPR*i, i < MAXV (those below dim)
template <int LPR, typename XT, typename YT>
__global__ __launch_bounds__(nthreads<LPR>()) void ln_fwd_kernel(const cm_layernorm_args p) {
    constexpr int RPW = 64 / LPR, NT = nthreads<LPR>();
    const int lane = threadIdx.x & 63, lr = lane % LPR, sub = lane / LPR;
    const int64_t wave = (int64_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * (NT / 64);
    const int dim = p.dim;
    const float inv = 1.0f / dim;
    const XT *x = reinterpret_cast<const XT *>(p.x);
    YT *y = reinterpret_cast<YT *>(p.y);
    bool on[MAXV];
    int mcol[MAXV];
    float4 g[MAXV], b[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = 4 * (lr + LPR * i);
        mcol[i] = p.chan_mask ? c % p.mask_c : 0;
        on[i] = c < dim;
        g[i] = on[i] ? *reinterpret_cast<const float4 *>(p.gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        b[i] = on[i] ? *reinterpret_cast<const float4 *>(p.beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int64_t r0 = wave * RPW; r0 < p.rows; r0 += nwaves * RPW) {
        const int64_t row = r0 + sub;
        const bool ok = row < p.rows;
        const int64_t rc = ok ? row : p.rows - 1;
        float4 v[MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            v[i] = on[i] ? ld4<XT>(x + rc * dim + 4 * (lr + LPR * i)) : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        const float mean = row_sum<LPR>(s) * inv;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            if (on[i]) {
                v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
                q = fmaf(v[i].x, v[i].x, fmaf(v[i].y, v[i].y, fmaf(v[i].z, v[i].z, fmaf(v[i].w, v[i].w, q))));
            }
        }
        const float rstd = rsqrtf(row_sum<LPR>(q) * inv + p.eps);
        if (ok) {
            const float *mrow = ln_mask_row(p, row);
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                if (on[i])
                    st4(y + row * dim + 4 * (lr + LPR * i),
                        ln_epilogue(p, make_float4(fmaf(v[i].x * rstd, g[i].x, b[i].x), fmaf(v[i].y * rstd, g[i].y, b[i].y),
                                                   fmaf(v[i].z * rstd, g[i].z, b[i].z), fmaf(v[i].w * rstd, g[i].w, b[i].w)), mrow, mcol[i]));
            }
            if (lr == 0 && p.mean) { p.mean[row] = mean; p.rstd[row] = rstd; }
        }
    }
}

template <int LPR, typename XT, typename YT, bool PIPE = true>
__global__ __launch_bounds__(nthreads<LPR>()) void ln_bwd_kernel(const cm_layernorm_args p) {
    constexpr int RPW = 64 / LPR, NT = nthreads<LPR>();
    __shared__ float red[NT / 64][2][MAXV * 4 * LPR];             // per-wave column sums (after the in-wave reduction)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lr = lane % LPR, sub = lane / LPR;
    const int64_t wave = (int64_t)blockIdx.x * (NT / 64) + wv, nwaves = (int64_t)gridDim.x * (NT / 64);
    const int dim = p.dim;
    const float inv = 1.0f / dim;
    const XT *x = reinterpret_cast<const XT *>(p.x);
    const YT *dy = reinterpret_cast<const YT *>(p.dy);
    XT *dx = reinterpret_cast<XT *>(p.dx);
    bool on[MAXV];
    int mcol[MAXV];
    float4 g[MAXV], dg[MAXV], db[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = 4 * (lr + LPR * i);
        mcol[i] = p.chan_mask ? c % p.mask_c : 0;
        on[i] = c < dim;
        g[i] = on[i] ? *reinterpret_cast<const float4 *>(p.gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        dg[i] = db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // Everything a row group needs (x, dy, the residual branch's gradient, the statistics) is requested in one go, and the
    // next group's before this one is worked on: the first version asked for dres only after the row reductions (two dependent
    // round trips per group) and had nothing in flight while it computed -- 3.4 TB/s on (32000, 256) rows.
    struct RowIn { float4 x[MAXV], d[MAXV], r[MAXV]; float mean, rstd; };
    auto request = [&](const int64_t r0, RowIn &in) {
        const int64_t rc = min(r0 + sub, p.rows - 1);
        in.mean = p.mean[rc], in.rstd = p.rstd[rc];
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            in.x[i] = in.d[i] = in.r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (on[i]) {
                in.x[i] = ld4<XT>(x + rc * dim + 4 * (lr + LPR * i));
                in.d[i] = ld4<YT>(dy + rc * dim + 4 * (lr + LPR * i));
                if (p.dres && dx) in.r[i] = *reinterpret_cast<const float4 *>(p.dres + rc * dim + 4 * (lr + LPR * i));
            }
        }
    };
    const int64_t rstep = nwaves * RPW;
    RowIn cur, nxt;
    int64_t r0 = wave * RPW;
    if (PIPE && r0 < p.rows) request(r0, cur);
    for (; r0 < p.rows; r0 += rstep) {
        if constexpr (!PIPE) request(r0, cur);
        else if (r0 + rstep < p.rows) request(r0 + rstep, nxt);     // wave-uniform
        const int64_t row = r0 + sub;
        const bool ok = row < p.rows;
        const int64_t rc = ok ? row : p.rows - 1;
        const float mean = cur.mean, rstd = cur.rstd;
        const float *mrow = ln_mask_row(p, rc);
        float4 xh[MAXV], gy[MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
            xh[i] = d;
            if (on[i]) {
                const float4 v = cur.x[i];
                d = cur.d[i];
                if (!ok) d = make_float4(0.f, 0.f, 0.f, 0.f);     // rows past the end add nothing to the column sums
                xh[i] = make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
                if (p.act | (p.chan_mask != nullptr)) d = ln_epilogue_grad(p, d, xh[i], g[i], mrow, mcol[i], 4 * (lr + LPR * i));
            }
            gy[i] = make_float4(d.x * g[i].x, d.y * g[i].y, d.z * g[i].z, d.w * g[i].w);
            s1 += (gy[i].x + gy[i].y) + (gy[i].z + gy[i].w);
            s2 = fmaf(gy[i].x, xh[i].x, fmaf(gy[i].y, xh[i].y, fmaf(gy[i].z, xh[i].z, fmaf(gy[i].w, xh[i].w, s2))));
            dg[i].x = fmaf(d.x, xh[i].x, dg[i].x); dg[i].y = fmaf(d.y, xh[i].y, dg[i].y);
            dg[i].z = fmaf(d.z, xh[i].z, dg[i].z); dg[i].w = fmaf(d.w, xh[i].w, dg[i].w);
            db[i].x += d.x; db[i].y += d.y; db[i].z += d.z; db[i].w += d.w;
        }
        const float c1 = row_sum<LPR>(s1) * inv, c2 = row_sum<LPR>(s2) * inv;
        if (ok && dx) {
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                if (on[i]) {
                    float4 o = make_float4(rstd * (gy[i].x - c1 - xh[i].x * c2), rstd * (gy[i].y - c1 - xh[i].y * c2),
                                           rstd * (gy[i].z - c1 - xh[i].z * c2), rstd * (gy[i].w - c1 - xh[i].w * c2));
                    const float4 r = cur.r[i];                     // + the residual branch's gradient (pre-norm block), or zeros
                    o.x += r.x, o.y += r.y, o.z += r.z, o.w += r.w;
                    st4(dx + row * dim + 4 * (lr + LPR * i), o);
                }
            }
        }
        if constexpr (PIPE) cur = nxt;
    }
    // column sums: over the row groups of a wave (shuffles), over the waves (LDS), one partial row per workgroup
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        float *a[8] = {&dg[i].x, &dg[i].y, &dg[i].z, &dg[i].w, &db[i].x, &db[i].y, &db[i].z, &db[i].w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = *a[k];
            if constexpr (RPW >= 2) v += __shfl_xor(v, 16, 64);
            if constexpr (RPW >= 4) v += __shfl_xor(v, 32, 64);
            *a[k] = v;
        }
        if (sub == 0) {
            st4(&red[wv][0][4 * (lr + LPR * i)], dg[i]);
            st4(&red[wv][1][4 * (lr + LPR * i)], db[i]);
        }
    }
    __syncthreads();
    float *part = p.workspace + (int64_t)blockIdx.x * 2 * dim;
    for (int j = threadIdx.x; j < 2 * dim; j += NT) {
        const int which = j / dim, c = j % dim;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) s += red[w][which][c];
        part[j] = s;
    }
}

// ---- rows wider than 1024 (the CNN front end's LayerNorm over (frequency, channel): 40 x 64 = 2560): one workgroup of 256
// threads per row, thread t owns the 4-column groups t + 256 i; row statistics are block reductions (one barrier each,
// alternating scratch slots), the column sums stay in the thread's registers across the rows its workgroup walks.
__device__ __forceinline__ float block_sum(float v, float (*sh)[4], int &phase) {
    v = cm_group_sum<16>(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    float *slot = sh[phase & 1];
    phase ^= 1;
    if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = v;
    cm_lds_barrier();                                              // LDS only: __syncthreads() would also drain the next row's loads (vmcnt)
    return (slot[0] + slot[1]) + (slot[2] + slot[3]);
}

template <typename XT, typename YT>
__global__ __launch_bounds__(256) void ln_fwd_wide_kernel(const cm_layernorm_args p) {
    __shared__ float sh[2][4];
    int phase = 0;
    const int dim = p.dim, t = threadIdx.x;
    const float inv = 1.0f / dim;
    const XT *x = reinterpret_cast<const XT *>(p.x);
    YT *y = reinterpret_cast<YT *>(p.y);
    bool on[MAXV];
    int mcol[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) on[i] = 4 * (t + 256 * i) < dim, mcol[i] = p.chan_mask ? (4 * (t + 256 * i)) % p.mask_c : 0;
    // gamma / beta live in registers for all rows, and a row's mask values are loaded BEFORE the next row is requested: the
    // memory counter is in order, so a parameter load issued behind the prefetch made its wait drain the prefetch
    float4 gw[MAXV], bw[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        gw[i] = on[i] ? *reinterpret_cast<const float4 *>(p.gamma + 4 * (t + 256 * i)) : make_float4(0.f, 0.f, 0.f, 0.f);
        bw[i] = on[i] ? *reinterpret_cast<const float4 *>(p.beta + 4 * (t + 256 * i)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 nv[MAXV];
    auto request = [&](int64_t row) {
#pragma unroll
        for (int i = 0; i < MAXV; ++i) nv[i] = on[i] ? ld4<XT>(x + row * dim + 4 * (t + 256 * i)) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    if ((int64_t)blockIdx.x < p.rows) request(blockIdx.x);
    for (int64_t row = blockIdx.x; row < p.rows; row += gridDim.x) {
        float4 v[MAXV], mk[MAXV];
        float s = 0.f;
        const float *mrow = ln_mask_row(p, row);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            v[i] = nv[i];
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            mk[i] = (mrow && on[i]) ? *reinterpret_cast<const float4 *>(mrow + mcol[i]) : make_float4(1.f, 1.f, 1.f, 1.f);
        }
        if (row + gridDim.x < p.rows) request(row + gridDim.x);   // the next row travels under this row's two reductions
        const float mean = block_sum(s, sh, phase) * inv;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            if (on[i]) {
                v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
                q = fmaf(v[i].x, v[i].x, fmaf(v[i].y, v[i].y, fmaf(v[i].z, v[i].z, fmaf(v[i].w, v[i].w, q))));
            }
        }
        const float rstd = rsqrtf(block_sum(q, sh, phase) * inv + p.eps);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            if (on[i]) {
                const int c = 4 * (t + 256 * i);
                const float4 g = gw[i], b = bw[i];
                float4 o = make_float4(fmaf(v[i].x * rstd, g.x, b.x), fmaf(v[i].y * rstd, g.y, b.y), fmaf(v[i].z * rstd, g.z, b.z), fmaf(v[i].w * rstd, g.w, b.w));
                o = ln_epilogue(p, o, nullptr, 0);                // activation
                st4(y + row * dim + c, make_float4(o.x * mk[i].x, o.y * mk[i].y, o.z * mk[i].z, o.w * mk[i].w));
            }
        }
        if (t == 0 && p.mean) { p.mean[row] = mean; p.rstd[row] = rstd; }
    }
}

template <typename XT, typename YT>
__global__ __launch_bounds__(256) void ln_bwd_wide_kernel(const cm_layernorm_args p) {
    __shared__ float sh[2][4];
    int phase = 0;
    const int dim = p.dim, t = threadIdx.x;
    const float inv = 1.0f / dim;
    const XT *x = reinterpret_cast<const XT *>(p.x);
    const YT *dy = reinterpret_cast<const YT *>(p.dy);
    XT *dx = reinterpret_cast<XT *>(p.dx);
    bool on[MAXV];
    int mcol[MAXV];
    float4 dg[MAXV], db[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        on[i] = 4 * (t + 256 * i) < dim;
        mcol[i] = p.chan_mask ? (4 * (t + 256 * i)) % p.mask_c : 0;
        dg[i] = db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // the NEXT row's x / dy are requested before this row's reductions: with 512 workgroups (two per CU) walking 125 rows each, a row's
    // load -> sum -> barrier -> sum -> barrier -> store chain ran at 1.8 TB/s
    float4 gw[MAXV], bw[MAXV];                                     // gamma (and beta for the activation) in registers for all rows
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        gw[i] = on[i] ? *reinterpret_cast<const float4 *>(p.gamma + 4 * (t + 256 * i)) : make_float4(0.f, 0.f, 0.f, 0.f);
        bw[i] = (on[i] && p.act == 1) ? *reinterpret_cast<const float4 *>(p.beta + 4 * (t + 256 * i)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 nx[MAXV], nd[MAXV];
    float nmean = 0.f, nrstd = 0.f;
    auto request = [&](int64_t row) {
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            if (on[i]) {
                const int c = 4 * (t + 256 * i);
                nx[i] = ld4<XT>(x + row * dim + c);
                nd[i] = ld4<YT>(dy + row * dim + c);
            }
        }
        nmean = p.mean[row], nrstd = p.rstd[row];
    };
    if ((int64_t)blockIdx.x < p.rows) request(blockIdx.x);
    for (int64_t row = blockIdx.x; row < p.rows; row += gridDim.x) {
        const float mean = nmean, rstd = nrstd;
        float4 cx[MAXV], cd[MAXV];
#pragma unroll
        for (int i = 0; i < MAXV; ++i) cx[i] = nx[i], cd[i] = nd[i];
        const float *mrow = ln_mask_row(p, row);
        float4 mk[MAXV];
#pragma unroll
        for (int i = 0; i < MAXV; ++i) mk[i] = (mrow && on[i]) ? *reinterpret_cast<const float4 *>(mrow + mcol[i]) : make_float4(1.f, 1.f, 1.f, 1.f);
        if (row + gridDim.x < p.rows) request(row + gridDim.x);  // behind the mask loads: the memory counter is in order
        float4 xh[MAXV], gy[MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            xh[i] = gy[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (on[i]) {
                const int c = 4 * (t + 256 * i);
                const float4 v = cx[i];
                float4 d = cd[i];
                const float4 g = gw[i];
                xh[i] = make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
                d.x *= mk[i].x, d.y *= mk[i].y, d.z *= mk[i].z, d.w *= mk[i].w;
                if (p.act == 1) {
                    const float4 b = bw[i];
                    d.x *= fmaf(xh[i].x, g.x, b.x) > 0.f ? 1.f : p.act_slope; d.y *= fmaf(xh[i].y, g.y, b.y) > 0.f ? 1.f : p.act_slope;
                    d.z *= fmaf(xh[i].z, g.z, b.z) > 0.f ? 1.f : p.act_slope; d.w *= fmaf(xh[i].w, g.w, b.w) > 0.f ? 1.f : p.act_slope;
                }
                (void)c;
                gy[i] = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
                dg[i].x = fmaf(d.x, xh[i].x, dg[i].x); dg[i].y = fmaf(d.y, xh[i].y, dg[i].y);
                dg[i].z = fmaf(d.z, xh[i].z, dg[i].z); dg[i].w = fmaf(d.w, xh[i].w, dg[i].w);
                db[i].x += d.x; db[i].y += d.y; db[i].z += d.z; db[i].w += d.w;
            }
            s1 += (gy[i].x + gy[i].y) + (gy[i].z + gy[i].w);
            s2 = fmaf(gy[i].x, xh[i].x, fmaf(gy[i].y, xh[i].y, fmaf(gy[i].z, xh[i].z, fmaf(gy[i].w, xh[i].w, s2))));
        }
        const float c1 = block_sum(s1, sh, phase) * inv, c2 = block_sum(s2, sh, phase) * inv;
        if (dx) {
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                if (on[i])
                    st4(dx + row * dim + 4 * (t + 256 * i),
                        make_float4(rstd * (gy[i].x - c1 - xh[i].x * c2), rstd * (gy[i].y - c1 - xh[i].y * c2),
                                    rstd * (gy[i].z - c1 - xh[i].z * c2), rstd * (gy[i].w - c1 - xh[i].w * c2)));
            }
        }
    }
    float *part = p.workspace + (int64_t)blockIdx.x * 2 * dim;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (on[i]) {
            st4(part + 4 * (t + 256 * i), dg[i]);
            st4(part + dim + 4 * (t + 256 * i), db[i]);
        }
    }
}

// dgamma | dbeta (2*dim columns) = sum over the partial rows, fixed order: 32 columns x 8 row groups per workgroup
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float *__restrict__ part, int nblk, int dim,
                                                           float *__restrict__ dgamma, float *__restrict__ dbeta) {
    __shared__ float red[8][32];
    const int col = blockIdx.x * 32 + (threadIdx.x & 31), grp = threadIdx.x >> 5;
    float s = 0.f;
    if (col < 2 * dim) {
#pragma unroll 8
        for (int b = grp; b < nblk; b += 8) s += part[(int64_t)b * 2 * dim + col];     // unrolled: 8 loads in flight per thread
    }
    red[grp][threadIdx.x & 31] = s;
    __syncthreads();
    if (grp == 0 && col < 2 * dim) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x & 31];
        if (col < dim) dgamma[col] = t;
        else dbeta[col - dim] = t;
    }
}

int check(const cm_layernorm_args &a, const char *what) {
    CM_REQUIRE(a.rows > 0 && a.dim > 0 && a.x && a.gamma, CM_EINVAL, "%s: bad sizes or NULL tensor", what);
    CM_REQUIRE(a.dim % 4 == 0 && a.dim <= 4096, CM_EUNSUPPORTED, "%s: dim must be a multiple of 4, at most 4096 (got %d)", what, a.dim);
    CM_REQUIRE((a.x_dtype == CM_F32 || a.x_dtype == CM_BF16) && (a.y_dtype == CM_F32 || a.y_dtype == CM_BF16), CM_EUNSUPPORTED,
               "%s: dtypes must be f32 or bf16", what);
    CM_REQUIRE(cm_aligned(a.x, 16) && cm_aligned(a.gamma, 16), CM_EALIGN, "%s: tensors must be 16-byte aligned", what);
    return CM_OK;
}

int grid_for(int64_t rows, int rpw, int nt, int cap) {
    const int64_t need = (rows + (int64_t)rpw * (nt / 64) - 1) / ((int64_t)rpw * (nt / 64));
    return (int)(need < cap ? need : cap);
}

template <typename XT, typename YT>
int launch_wide(const cm_layernorm_args &a, bool bwd) {
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    if (!bwd) {
        hipLaunchKernelGGL((ln_fwd_wide_kernel<XT, YT>), dim3((unsigned)(a.rows < 2048 ? a.rows : 2048)), dim3(256), 0, st, a);
        return cm_launch_status("cm_layernorm_fwd");
    }
    const int nblk = (int)(a.rows < BWD_BLOCKS ? a.rows : BWD_BLOCKS);
    hipLaunchKernelGGL((ln_bwd_wide_kernel<XT, YT>), dim3(nblk), dim3(256), 0, st, a);
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((2 * a.dim + 31) / 32), dim3(256), 0, st, a.workspace, nblk, a.dim, a.dgamma, a.dbeta);
    return cm_launch_status("cm_layernorm_bwd");
}

int dispatch_wide(const cm_layernorm_args &a, bool bwd) {
    const bool xb = a.x_dtype == CM_BF16, yb = a.y_dtype == CM_BF16;
    if (xb) return yb ? launch_wide<cm_bf16, cm_bf16>(a, bwd) : launch_wide<cm_bf16, float>(a, bwd);
    return yb ? launch_wide<float, cm_bf16>(a, bwd) : launch_wide<float, float>(a, bwd);
}

template <template <int, typename, typename> class Launch>
int dispatch(const cm_layernorm_args &a) {
    const bool narrow = a.dim <= 256, xb = a.x_dtype == CM_BF16, yb = a.y_dtype == CM_BF16;
    if (narrow) {
        if (xb) return yb ? Launch<16, cm_bf16, cm_bf16>::run(a) : Launch<16, cm_bf16, float>::run(a);
        return yb ? Launch<16, float, cm_bf16>::run(a) : Launch<16, float, float>::run(a);
    }
    if (xb) return yb ? Launch<64, cm_bf16, cm_bf16>::run(a) : Launch<64, cm_bf16, float>::run(a);
    return yb ? Launch<64, float, cm_bf16>::run(a) : Launch<64, float, float>::run(a);
}

template <int LPR, typename XT, typename YT> struct FwdLaunch {
    static int run(const cm_layernorm_args &a) {
        hipLaunchKernelGGL((ln_fwd_kernel<LPR, XT, YT>), dim3(grid_for(a.rows, 64 / LPR, nthreads<LPR>(), 2048)), dim3(nthreads<LPR>()), 0,
                           reinterpret_cast<hipStream_t>(a.stream), a);
        return cm_launch_status("cm_layernorm_fwd");
    }
};
template <int LPR, typename XT, typename YT> struct BwdLaunch {
    static int run(const cm_layernorm_args &a) {
        // one workgroup per CU: with two row groups in flight per wave the kernel holds ~200 VGPRs, so only one 8-wave workgroup
        // fits a CU anyway and 512 workgroups ran as two rounds (profiles/r03/ln_bwd_rows.txt: 25.5 -> 22.9 us, reduce 5.1 -> 4.1)
        int nblk = grid_for(a.rows, 64 / LPR, nthreads<LPR>(), BWD_BLOCKS_ROWS * 512 / nthreads<LPR>());   // 8 waves per CU either way
        hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
#ifdef CM_ABLATE
        if (cm_debug_get() == 50 || cm_debug_get() == 51) nblk = grid_for(a.rows, 64 / LPR, nthreads<LPR>(), BWD_BLOCKS);   // A/B: two rounds
        if (cm_debug_get() == 50) {                                  // A/B: and no cross-iteration prefetch
            hipLaunchKernelGGL((ln_bwd_kernel<LPR, XT, YT, false>), dim3(nblk), dim3(nthreads<LPR>()), 0, st, a);
            hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((2 * a.dim + 31) / 32), dim3(256), 0, st, a.workspace, nblk, a.dim, a.dgamma, a.dbeta);
            return cm_launch_status("cm_layernorm_bwd");
        }
#endif
        hipLaunchKernelGGL((ln_bwd_kernel<LPR, XT, YT>), dim3(nblk), dim3(nthreads<LPR>()), 0, st, a);
        hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((2 * a.dim + 31) / 32), dim3(256), 0, st, a.workspace, nblk, a.dim, a.dgamma, a.dbeta);
        return cm_launch_status("cm_layernorm_bwd");
    }
};

}  // namespace

extern "C" int64_t cm_layernorm_bwd_workspace_floats(int64_t rows, int32_t dim) {
    (void)rows;
    return (int64_t)BWD_BLOCKS * 2 * dim;
}

extern "C" int cm_layernorm_fwd(const cm_layernorm_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "layernorm_fwd: args is NULL");
    const cm_layernorm_args &a = *args;
    if (int rc = check(a, "layernorm_fwd")) return rc;
    CM_REQUIRE(a.beta && a.y && (!a.mean == !a.rstd), CM_EINVAL, "layernorm_fwd: beta / y must be non-NULL, mean and rstd given together");
    CM_REQUIRE(cm_aligned(a.beta, 16) && cm_aligned(a.y, 16), CM_EALIGN, "layernorm_fwd: tensors must be 16-byte aligned");
    return a.dim > 1024 ? dispatch_wide(a, false) : dispatch<FwdLaunch>(a);
}

extern "C" int cm_layernorm_bwd(const cm_layernorm_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "layernorm_bwd: args is NULL");
    const cm_layernorm_args &a = *args;
    if (int rc = check(a, "layernorm_bwd")) return rc;
    CM_REQUIRE(a.dy && a.mean && a.rstd && a.dgamma && a.dbeta && a.workspace, CM_EINVAL,
               "layernorm_bwd: dy / mean / rstd / dgamma / dbeta / workspace must be non-NULL");
    CM_REQUIRE(cm_aligned(a.dy, 16) && (!a.dx || cm_aligned(a.dx, 16)), CM_EALIGN, "layernorm_bwd: tensors must be 16-byte aligned");
    CM_REQUIRE(!a.dres || (a.dx && a.x_dtype == CM_F32 && a.dim <= 1024 && cm_aligned(a.dres, 16)), CM_EUNSUPPORTED,
               "layernorm_bwd: dres needs dx, fp32 x, dim <= 1024 and 16-byte alignment");
    return a.dim > 1024 ? dispatch_wide(a, true) : dispatch<BwdLaunch>(a);
}
