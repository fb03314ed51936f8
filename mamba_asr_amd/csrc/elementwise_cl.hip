// elementwise_cl.hip — channels-last streaming kernels of the fused ConMamba layer (gfx950):
//   cm_conv_cl_fwd         depthwise causal conv + SiLU for both BiMamba directions in one pass
//   cm_add_layernorm       residual add + one or two LayerNorms
//   cm_glu_dwconv_ln_gelu  GLU -> depthwise conv k (31) -> LayerNorm -> GELU of the convolution module
// All are HBM-bound: 16-byte vectors along the contiguous channel axis, every input read once.
#include "cm_common.h"


namespace {

template <typename IO> struct vec8 {                  // 16-byte vector of IO elements <-> floats
    static constexpr int N = cm_elem<IO>::kVec;
    static __device__ __forceinline__ void load(const IO *p, float (&f)[N]) {
        alignas(16) IO tmp[N];
        *reinterpret_cast<uint4 *>(tmp) = *reinterpret_cast<const uint4 *>(p);
#pragma unroll
        for (int j = 0; j < N; ++j) f[j] = cm_elem<IO>::load(&tmp[j]);
    }
    static __device__ __forceinline__ void store(IO *p, const float (&f)[N]) {
        alignas(16) IO tmp[N];
#pragma unroll
        for (int j = 0; j < N; ++j) cm_elem<IO>::store(&tmp[j], f[j]);
        *reinterpret_cast<uint4 *>(p) = *reinterpret_cast<const uint4 *>(tmp);
    }
};

__device__ __forceinline__ float gelu_erf(float x) { return cm_gelu(x); }

// ------------------------------------------------------------------------------------------------
// conv, both directions.  One thread owns one 16-byte channel vector and walks a run of TC steps with a
// rolling window of W rows: the newest W rows x[s-W+1..s] give y_fwd[s] and y_bwd[s-W+1].
// ------------------------------------------------------------------------------------------------
template <typename IO, int W, int TC>
__global__ __launch_bounds__(256) void conv_cl_kernel(const cm_conv_cl_args p, int vpr) {
    // one thread = one 32-bit word of channels (2 x bf16 or 1 x fp32) x TC consecutive steps: ~80 VGPRs, so 6+ waves
    // per SIMD stay resident (the 8-channel/16-byte version needed 192 VGPRs and ran at 2 waves per SIMD)
    constexpr int N = 4 / (int)sizeof(IO);
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nchunk = (p.seqlen + TC - 1) / TC;
    const int vec = (int)(v % vpr), chunk = (int)((v / vpr) % nchunk), b = (int)(v / ((int64_t)vpr * nchunk));
    if (b >= p.batch) return;
    const int c0 = vec * N;
    const bool two = p.y_bwd != nullptr;
    const IO *x = reinterpret_cast<const IO *>(p.x) + (int64_t)b * p.x_bs + c0;
    IO *yf = reinterpret_cast<IO *>(p.y_fwd) + (int64_t)b * p.yf_bs + c0;
    IO *yb = two ? reinterpret_cast<IO *>(p.y_bwd) + (int64_t)b * p.yb_bs + c0 : nullptr;
    const int t0 = chunk * TC;
    uint32_t raw[TC + 2 * (W - 1)];                               // rows t0-(W-1) .. t0+TC-1+(W-1), all loads issued first
#pragma unroll
    for (int r = 0; r < TC + 2 * (W - 1); ++r) {
        const int s = t0 - (W - 1) + r;
        raw[r] = (s >= 0 && s < p.seqlen) ? *reinterpret_cast<const uint32_t *>(x + (int64_t)s * p.x_ts) : 0u;
    }
    float wf[N][W], wb[N][W], bf[N], bb[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
#pragma unroll
        for (int k = 0; k < W; ++k) {
            wf[j][k] = p.weight_f[(c0 + j) * W + k];
            wb[j][k] = two ? p.weight_b[(c0 + j) * W + k] : 0.f;
        }
        bf[j] = p.bias_f ? p.bias_f[c0 + j] : 0.f;
        bb[j] = (two && p.bias_b) ? p.bias_b[c0 + j] : 0.f;
    }
    auto elem = [&](int r, int j) -> float {
        if constexpr (sizeof(IO) == 4) return __uint_as_float(raw[r]);
        else return cm_elem<IO>::from_bits((uint16_t)(raw[r] >> (16 * j)));
    };
    auto pack = [&](const float (&o)[N]) -> uint32_t {
        if constexpr (sizeof(IO) == 4) return __float_as_uint(o[0]);
        else return (uint32_t)cm_elem<IO>::to_bits(o[0]) | ((uint32_t)cm_elem<IO>::to_bits(o[N - 1]) << 16);
    };
#pragma unroll
    for (int i = 0; i < TC; ++i) {
        const int t = t0 + i;
        if (t >= p.seqlen) break;
        float of[N], ob[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float af = bf[j], ab = bb[j];
#pragma unroll
            for (int k = 0; k < W; ++k) {
                af = fmaf(wf[j][k], elem(i + k, j), af);                       // x[t-(W-1)+k]
                ab = fmaf(wb[j][k], elem(i + 2 * (W - 1) - k, j), ab);         // x[t+(W-1)-k]
            }
            of[j] = p.silu ? af * cm_sigmoid(af) : af;
            ob[j] = p.silu ? ab * cm_sigmoid(ab) : ab;
        }
        *reinterpret_cast<uint32_t *>(yf + (int64_t)t * p.yf_ts) = pack(of);
        if (two) *reinterpret_cast<uint32_t *>(yb + (int64_t)t * p.yb_ts) = pack(ob);
    }
}

// ------------------------------------------------------------------------------------------------
// add + LayerNorm(s): one wave per row, float4 per lane per 256 columns.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <typename YT, typename OT, int NV>          // NV = float4 groups per lane (dim <= NV*256)
__global__ __launch_bounds__(256) void add_ln_kernel(const cm_add_ln_args p) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const int D = p.dim;
    float r[NV][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            float4 xv = {0.f, 0.f, 0.f, 0.f};
            if (p.x) xv = *reinterpret_cast<const float4 *>(p.x + row * D + c);
            r[i][0] = xv.x; r[i][1] = xv.y; r[i][2] = xv.z; r[i][3] = xv.w;
            if (p.y) {
                const YT *yp = reinterpret_cast<const YT *>(p.y) + row * D + c;
#pragma unroll
                for (int j = 0; j < 4; ++j) r[i][j] = fmaf(p.alpha, cm_elem<YT>::load(yp + j), r[i][j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) r[i][j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) s += r[i][j];
    }
    auto layer_norm = [&](const float *g, const float *bta, float eps) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sum += r[i][j];
        const float mean = wave_sum(sum) / D;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dlt = (c < D) ? r[i][j] - mean : 0.f;
                sq += dlt * dlt;
            }
        }
        const float rstd = rsqrtf(wave_sum(sq) / D + eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < D) {
#pragma unroll
                for (int j = 0; j < 4; ++j) r[i][j] = (r[i][j] - mean) * rstd * g[c + j] + bta[c + j];
            }
        }
    };
    (void)s;
    if (p.g1) layer_norm(p.g1, p.b1, p.eps1);
    if (p.x_out) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < D) *reinterpret_cast<float4 *>(p.x_out + row * D + c) = make_float4(r[i][0], r[i][1], r[i][2], r[i][3]);
        }
    }
    if (p.out) {
        if (p.g2) layer_norm(p.g2, p.b2, p.eps2);
        OT *op = reinterpret_cast<OT *>(p.out) + row * D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < D) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = r[i][j];
                    if (p.out_act == 1) v = v > 0.f ? v : 0.01f * v;
                    cm_elem<OT>::store(op + c + j, v);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// GLU -> depthwise conv (K taps, same padding) -> LayerNorm -> GELU.
// Workgroup = one batch row x TT output steps; thread = channel.  Phase 1: GLU of the TT+K-1 input rows
// into LDS (each input element read once from HBM).  Phase 2: thread c convolves its channel along time
// from LDS (conflict-free: consecutive threads, consecutive addresses) and writes the result back to LDS.
// Phase 3: one wave per output row does LayerNorm + GELU and the coalesced store.
// ------------------------------------------------------------------------------------------------
template <typename IO, int K, int TT>
__global__ __launch_bounds__(256) void glu_dwconv_kernel(const cm_glu_dwconv_args p) {
    constexpr int tt = TT;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int D = p.dim, T = p.seqlen;
    const int b = blockIdx.y, t0 = blockIdx.x * tt;
    const int nin = tt + K - 1;                                   // input rows t0-K/2 .. t0+tt-1+K/2
    const int CS = D + 16;                                        // co row stride: rows of one wave land in different banks
    float *co = sm;                                               // [tt][CS] fp32 conv outputs
    IO *g = reinterpret_cast<IO *>(sm + (size_t)tt * CS);         // [nin][D] GLU outputs in the I/O dtype
    const bool pre = p.glu_done != 0;                              // input already gated: D-wide rows, plain copy
    const int IW = pre ? D : 2 * D;
    const IO *in = reinterpret_cast<const IO *>(p.in) + (int64_t)b * T * IW;
    // phase 1: GLU rows -> LDS  (a = in[:, :D], gate = in[:, D:])
    constexpr int NV = cm_elem<IO>::kVec;
    if (pre && D % NV == 0) {
        for (int idx = threadIdx.x; idx < nin * (D / NV); idx += blockDim.x) {
            const int r = idx / (D / NV), c = (idx % (D / NV)) * NV;
            const int t = t0 - K / 2 + r;
            uint4 v4 = {0u, 0u, 0u, 0u};
            if (t >= 0 && t < T) v4 = *reinterpret_cast<const uint4 *>(in + (int64_t)t * D + c);
            *reinterpret_cast<uint4 *>(g + r * D + c) = v4;
        }
    } else if (D % NV == 0) {
        for (int idx = threadIdx.x; idx < nin * (D / NV); idx += blockDim.x) {
            const int r = idx / (D / NV), c = (idx % (D / NV)) * NV;
            const int t = t0 - K / 2 + r;
            float av[NV], gv[NV];
            if (t >= 0 && t < T) {
                const IO *row = in + (int64_t)t * 2 * D;
                vec8<IO>::load(row + c, av);
                vec8<IO>::load(row + D + c, gv);
#pragma unroll
                for (int j = 0; j < NV; ++j) av[j] *= cm_sigmoid(gv[j]);
            } else {
#pragma unroll
                for (int j = 0; j < NV; ++j) av[j] = 0.f;
            }
            vec8<IO>::store(g + r * D + c, av);
        }
    } else {
        for (int idx = threadIdx.x; idx < nin * D; idx += blockDim.x) {
            const int r = idx / D, c = idx % D;
            const int t = t0 - K / 2 + r;
            float v0 = 0.f;
            if (t >= 0 && t < T) {
                const IO *row = in + (int64_t)t * IW;
                v0 = pre ? cm_elem<IO>::load(row + c) : cm_elem<IO>::load(row + c) * cm_sigmoid(cm_elem<IO>::load(row + D + c));
            }
            cm_elem<IO>::store(g + r * D + c, v0);
        }
    }
    __syncthreads();
    // phase 2: depthwise conv along time
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float w[K];
#pragma unroll
        for (int k = 0; k < K; ++k) w[k] = p.weight_t ? p.weight_t[k * D + c] : p.weight[c * K + k];   // (K, D): coalesced
        const float bias = p.bias ? p.bias[c] : 0.f;
        float col[TT + K - 1];                                     // this channel's GLU outputs over the tile (+halo)
#pragma unroll
        for (int r = 0; r < TT + K - 1; ++r) col[r] = cm_elem<IO>::load(g + r * D + c);
#pragma unroll
        for (int r = 0; r < TT; ++r) {
            float acc = bias;
#pragma unroll
            for (int k = 0; k < K; ++k) acc = fmaf(w[k], col[r + k], acc);
            co[r * CS + c] = acc;
        }
    }
    __syncthreads();
    // phase 3: LayerNorm + GELU per row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    IO *out = reinterpret_cast<IO *>(p.out) + (int64_t)b * T * D;
    if (D == 256) {
        // a row of 16 lanes per LayerNorm row (16 values per lane): statistics by in-lane adds + 4 DPP steps, four rows
        // per wave instruction, 8/16-byte stores that are contiguous over the 16 lanes
        const int l15 = lane & 15, lq = lane >> 4;
        for (int r = wave * 4 + lq; r < tt; r += 16) {
            const int t = t0 + r;
            float4 v[4];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = *reinterpret_cast<const float4 *>(co + r * CS + 4 * (l15 + 16 * i));
                s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            }
            const float mean = cm_group_sum<16>(s) * (1.f / 256);
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
                sq = fmaf(v[i].x, v[i].x, fmaf(v[i].y, v[i].y, fmaf(v[i].z, v[i].z, fmaf(v[i].w, v[i].w, sq))));
            }
            const float rstd = rsqrtf(cm_group_sum<16>(sq) * (1.f / 256) + p.eps);
            if (t < T) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = 4 * (l15 + 16 * i);
                    const float4 gm = *reinterpret_cast<const float4 *>(p.ln_g + c);
                    const float4 bt = *reinterpret_cast<const float4 *>(p.ln_b + c);
                    // outputs rounded to bf16 take the 7-slot GELU (cm_gelu_bf16: 2.6e-5 from the erf form), two at a time; fp32 outputs the erf form
                    const float y0 = fmaf(v[i].x * rstd, gm.x, bt.x), y1 = fmaf(v[i].y * rstd, gm.y, bt.y);
                    const float y2 = fmaf(v[i].z * rstd, gm.z, bt.z), y3 = fmaf(v[i].w * rstd, gm.w, bt.w);
                    if constexpr (sizeof(IO) == 2) {
                        *reinterpret_cast<uint2 *>(out + (int64_t)t * D + c) = make_uint2(cm_gelu_bf16_pack2(y0, y1), cm_gelu_bf16_pack2(y2, y3));
                    } else {
                        const float o[4] = {gelu_erf(y0), gelu_erf(y1), gelu_erf(y2), gelu_erf(y3)};
#pragma unroll
                        for (int j = 0; j < 4; ++j) cm_elem<IO>::store(out + (int64_t)t * D + c + j, o[j]);
                    }
                }
            }
        }
        return;
    }
    for (int r = wave; r < tt; r += 4) {
        const int t = t0 + r;
        if (t >= T) break;
        float s = 0.f;
        for (int c = lane; c < D; c += 64) s += co[r * CS + c];
        const float mean = wave_sum(s) / D;
        float sq = 0.f;
        for (int c = lane; c < D; c += 64) { const float dl = co[r * CS + c] - mean; sq += dl * dl; }
        const float rstd = rsqrtf(wave_sum(sq) / D + p.eps);
        for (int c = lane; c < D; c += 64) {
            const float v = (co[r * CS + c] - mean) * rstd * p.ln_g[c] + p.ln_b[c];
            cm_elem<IO>::store(out + (int64_t)t * D + c, gelu_erf(v));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same module for bf16 rows with dim % 64 == 0 (d_model 256 / 512: the fused encoder route), re-tiled:
//   * 32 output steps per workgroup: 62 input rows staged per 32 outputs (the 16-step tile above stages 46 per 16);
//   * thread = (channel PAIR, half of the tile's steps): one 4-byte LDS read feeds two channels and the 31 x 16 taps run
//     as v_pk_fma_f32, accumulating by INPUT row (one staged value live at a time; 16 x 2 accumulators + 31 x 2 taps
//     in registers -- the gather form above keeps a 46-row window per thread);
//   * the fp32 conv tile overlays the staged bf16 rows (outputs wait in registers across a barrier), so a workgroup holds
//     (32 x (dim + 16)) x 4 bytes of LDS: four workgroups per CU at dim 256;
//   * LayerNorm + GELU on rows of 16 lanes (dim / 16 values per lane, DPP sums), 8-byte stores.
// ------------------------------------------------------------------------------------------------
typedef float f32x2_t __attribute__((ext_vector_type(2)));

//   * LIN (dim 256): the convolution module's closing Linear (reference Conmamba.py:156-158, dim -> dim) runs on the tile
//     before it leaves the CU -- activations to LDS (bf16, 528-byte rows), v_mfma_f32_16x16x32_bf16 with the weight rows as
//     the A operand straight from their packed image in L2 (cm_ffn_pack_weights; ring filled during the LayerNorm phase),
//     wave = 64 output features x 32 tokens, bias added in fp32.  It replaces a library GEMM that re-read the rows.
typedef __attribute__((ext_vector_type(8))) __bf16 dw_bf16x8;
typedef __attribute__((ext_vector_type(4))) float dw_f32x4;

template <int K, int TT, int DV, bool LIN = false>
__global__ __launch_bounds__(256, DV == 4 ? 4 : 2) void dwconv_rows_kernel(const cm_glu_dwconv_args p) {
    constexpr int D = DV * 64, NIN = TT + K - 1, TH = TT / 2, CS = D + 16, NIT = (D / 2 + 127) / 128;
    constexpr int XS = D + 8;                                     // LIN: activation row stride in bf16 elements
    static_assert(!LIN || (D == 256 && TT == 32), "the Linear epilogue is built for dim 256, 32-step tiles");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    uint16_t *g = reinterpret_cast<uint16_t *>(sm);               // [NIN][D] staged (gated) rows, bf16
    float *co = sm;                                               // [TT][CS] conv outputs, overlays g after phase 2
    uint16_t *xt = reinterpret_cast<uint16_t *>(sm);              // LIN: [TT][XS] activations, overlays co after phase 3
    const int T = p.seqlen, tid = threadIdx.x;
    const int b = blockIdx.y, t0 = blockIdx.x * TT;
    const bool pre = p.glu_done != 0;
    const int IW = pre ? D : 2 * D;
    const uint16_t *in = reinterpret_cast<const uint16_t *>(p.in) + (int64_t)b * T * IW;
    // phase 1: rows t0 - K/2 .. t0 + TT - 1 + K/2 -> LDS, zero outside the sequence ('same' padding).
    // Already-gated rows (the encoder's call: cm_ln_pw_glu did the GLU): every 16-byte piece of the thread is requested before the
    // first one is waited for.  As a rolled loop (below, kept for the GLU form) the compiler emitted load -> s_waitcnt vmcnt(0) ->
    // ds_write per piece: eight dependent round trips per thread, "39 % of the workgroup's life waiting for its rows" in the stamps.
    constexpr int NCH = NIN * (D / 8), NPC = (NCH + 255) / 256;
    if (pre) {
        uint4 v[NPC];
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int idx = tid + 256 * i, r = idx / (D / 8), c = (idx % (D / 8)) * 8;
            const int t = t0 - K / 2 + r;
            v[i] = uint4{0u, 0u, 0u, 0u};
            if (idx < NCH && t >= 0 && t < T) v[i] = *reinterpret_cast<const uint4 *>(in + (int64_t)t * IW + c);
        }
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int idx = tid + 256 * i, r = idx / (D / 8), c = (idx % (D / 8)) * 8;
            if (idx < NCH) *reinterpret_cast<uint4 *>(g + r * D + c) = v[i];
        }
    } else
    for (int idx = tid; idx < NIN * (D / 8); idx += 256) {
        const int r = idx / (D / 8), c = (idx % (D / 8)) * 8;
        const int t = t0 - K / 2 + r;
        uint4 v = {0u, 0u, 0u, 0u};
        if (t >= 0 && t < T) {
            v = *reinterpret_cast<const uint4 *>(in + (int64_t)t * IW + c);
            if (!pre) {                                           // GLU: a * sigmoid(gate), gate = the row's second half
                const uint4 q = *reinterpret_cast<const uint4 *>(in + (int64_t)t * IW + D + c);
                const uint32_t av[4] = {v.x, v.y, v.z, v.w}, gv[4] = {q.x, q.y, q.z, q.w};
                uint32_t o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    o[j] = cm_pack_bf16(cm_bf16_lo(av[j]) * cm_sigmoid(cm_bf16_lo(gv[j])), cm_bf16_hi(av[j]) * cm_sigmoid(cm_bf16_hi(gv[j])));
                v = uint4{o[0], o[1], o[2], o[3]};
            }
        }
        *reinterpret_cast<uint4 *>(g + r * D + c) = v;
    }
    __syncthreads();
    // phase 2: depthwise conv, two channels x TH steps per thread, accumulated by input row
    const int rh = tid >> 7;                                      // wave-uniform
    f32x2_t acc[NIT][TH];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int cp = (tid & 127) + 128 * it;
        const int c = 2 * (cp < D / 2 ? cp : 0);
        f32x2_t w[K];
#pragma unroll
        for (int k = 0; k < K; ++k)
            w[k] = p.weight_t ? *reinterpret_cast<const f32x2_t *>(p.weight_t + k * D + c) : f32x2_t{p.weight[c * K + k], p.weight[(c + 1) * K + k]};
        const f32x2_t bias = p.bias ? *reinterpret_cast<const f32x2_t *>(p.bias + c) : f32x2_t{0.f, 0.f};
#pragma unroll
        for (int r = 0; r < TH; ++r) acc[it][r] = bias;
        const uint16_t *col = g + (rh * TH) * D + c;
#pragma unroll
        for (int j = 0; j < TH + K - 1; ++j) {                    // input row rh*TH + j feeds outputs r = j - k, 0 <= k < K
            const uint32_t u = *reinterpret_cast<const uint32_t *>(col + j * D);
            const f32x2_t x2 = {cm_bf16_lo(u), cm_bf16_hi(u)};
#pragma unroll
            for (int r = (j - (K - 1) > 0 ? j - (K - 1) : 0); r <= (j < TH - 1 ? j : TH - 1); ++r)
                acc[it][r] = __builtin_elementwise_fma(w[j - r], x2, acc[it][r]);
        }
    }
    __syncthreads();                                              // every thread is done reading g
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int cp = (tid & 127) + 128 * it;
        if (cp < D / 2) {
#pragma unroll
            for (int r = 0; r < TH; ++r) *reinterpret_cast<f32x2_t *>(co + (rh * TH + r) * CS + 2 * cp) = acc[it][r];
        }
    }
    __syncthreads();
    // phase 3: LayerNorm + GELU, a row of 16 lanes per output row
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, lq = lane >> 4;
    uint16_t *out = reinterpret_cast<uint16_t *>(p.out) + (int64_t)b * T * D;
    // LIN: weight fragments of the wave's 64 output features, k-steps 0..PF-1 in flight under the LayerNorm phase
    constexpr int PF = 2;                                         // 2-deep: 128 VGPRs, four workgroups per CU
    dw_bf16x8 wq[LIN ? PF : 1][4];
    uint2 held[LIN ? TT / 16 : 1][DV];                            // LIN: a lane's activations wait here until every wave has read co
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.lin_w), 0, LIN ? D * D * 2 : 0, 0x00020000);
    auto wload = [&](int ks, dw_bf16x8(&dst)[4]) {                 // fragment (16-row band, k-tile) = 1 KB at (band * D/32 + ks) * 1024
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
            dst[mb] = __builtin_bit_cast(dw_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, lane * 16, ((wave * 4 + mb) * (D / 32) + ks) * 1024, 0));
    };
    if constexpr (LIN) {
#pragma unroll
        for (int s2 = 0; s2 < PF; ++s2) wload(s2, wq[s2]);
    }
#pragma unroll
    for (int it = 0; it < TT / 16; ++it) {
        const int r = wave * 4 + lq + 16 * it;
        const int t = t0 + r;
        float4 v[DV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < DV; ++i) {
            v[i] = *reinterpret_cast<const float4 *>(co + r * CS + 4 * (l15 + 16 * i));
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        const float mean = cm_group_sum<16>(s) * (1.f / D);
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < DV; ++i) {
            v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
            sq = fmaf(v[i].x, v[i].x, fmaf(v[i].y, v[i].y, fmaf(v[i].z, v[i].z, fmaf(v[i].w, v[i].w, sq))));
        }
        const float rstd = rsqrtf(cm_group_sum<16>(sq) * (1.f / D) + p.eps);
        if (LIN || t < T) {
#pragma unroll
            for (int i = 0; i < DV; ++i) {
                const int c = 4 * (l15 + 16 * i);
                const float4 gm = *reinterpret_cast<const float4 *>(p.ln_g + c);
                const float4 bt = *reinterpret_cast<const float4 *>(p.ln_b + c);
                const float y0 = fmaf(v[i].x * rstd, gm.x, bt.x), y1 = fmaf(v[i].y * rstd, gm.y, bt.y);
                const float y2 = fmaf(v[i].z * rstd, gm.z, bt.z), y3 = fmaf(v[i].w * rstd, gm.w, bt.w);
                const uint2 pk = make_uint2(cm_gelu_bf16_pack2(y0, y1), cm_gelu_bf16_pack2(y2, y3));
                if constexpr (LIN) held[it][i] = pk;             // rows past the sequence: finite, never stored
                else *reinterpret_cast<uint2 *>(out + (int64_t)t * D + c) = pk;
            }
        }
    }
    if constexpr (LIN) {
        // phase 4: out[t][f] = sum_c W[f][c] act[t][c] + bias[f].  Lane holds token nb*16 + l15, features wave*64 + mb*16 + lq*4 + j.
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");             // LDS only: the weight ring stays in flight
#pragma unroll
        for (int it = 0; it < TT / 16; ++it)
#pragma unroll
            for (int i = 0; i < DV; ++i)
                *reinterpret_cast<uint2 *>(xt + (wave * 4 + lq + 16 * it) * XS + 4 * (l15 + 16 * i)) = held[it][i];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        dw_f32x4 acc[4][2];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[mb][nb] = dw_f32x4{0.f, 0.f, 0.f, 0.f};
        const uint16_t *frag = xt + l15 * XS + lq * 8;
#pragma unroll
        for (int ks = 0; ks < D / 32; ++ks) {
            dw_bf16x8 tok[2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) tok[nb] = *reinterpret_cast<const dw_bf16x8 *>(frag + nb * 16 * XS + ks * 32);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[ks % PF][mb], tok[nb], acc[mb][nb], 0, 0, 0);
            if (ks + PF < D / 32) wload(ks + PF, wq[ks % PF]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // out through LDS (the activation tile is dead once every wave has its fragments): whole 512-byte rows per wave
        // instruction instead of 8-byte pieces of 16 rows
        const int f0 = wave * 64 + lq * 4;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const float4 bv = *reinterpret_cast<const float4 *>(p.lin_b + f0 + mb * 16);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                *reinterpret_cast<uint2 *>(xt + (nb * 16 + l15) * XS + f0 + mb * 16) =
                    make_uint2(cm_pack_bf16(acc[mb][nb][0] + bv.x, acc[mb][nb][1] + bv.y), cm_pack_bf16(acc[mb][nb][2] + bv.z, acc[mb][nb][3] + bv.w));
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int i = 0; i < TT * 32 / 256; ++i) {
            const int idx = tid + 256 * i, row = idx >> 5, chunk = idx & 31;
            const uint4 v = *reinterpret_cast<const uint4 *>(xt + row * XS + chunk * 8);
            if (t0 + row < T) *reinterpret_cast<uint4 *>(out + (int64_t)(t0 + row) * D + chunk * 8) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// CNN block 1: conv 3x3 stride 2 (1 input channel) + LayerNorm(F1*C) + LeakyReLU, one workgroup per padded
// output time row.  The three input rows live in LDS; each thread produces F1*C/256 outputs, the row statistics
// are a block reduction, and the row (plus its reflected frequency border) is written once.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

template <typename OT, int MAXPT>
__global__ __launch_bounds__(256) void cnn_block1_kernel(const cm_cnn_block1_args p, int T1, int F1, int rpw) {
    __shared__ float rows[2][3][128 + 2];              // input rows 2*t1-1 .. 2*t1+1 with reflect padding in F, double-buffered
    __shared__ float red[2][2][4];
    __shared__ float wsh[128 * 9 + 128];               // conv taps [C][9] and bias [C] (C <= 128)
    const int C = p.C, F = p.F, T = p.T;
    const int P = p.pad_out;
    const int b = blockIdx.y;
    const int ntp = T1 + 2 * P;
    const int tp0 = blockIdx.x * rpw, tp1 = min(tp0 + rpw, ntp);
    const float *feats = p.feats + (int64_t)b * T * F;
    const int n = F1 * C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < C * 9; i += blockDim.x) wsh[i] = p.weight[i];
    for (int i = threadIdx.x; i < C; i += blockDim.x) wsh[C * 9 + i] = p.bias ? p.bias[i] : 0.f;
    // LayerNorm parameters of this thread's outputs stay in registers for all rows of the workgroup
    float lg[MAXPT], lb[MAXPT];
#pragma unroll
    for (int i = 0; i < MAXPT / 2; ++i) {
        const int o = 2 * threadIdx.x + i * 512;
        lg[2 * i] = o < n ? p.ln_g[o] : 0.f; lg[2 * i + 1] = o < n ? p.ln_g[o + 1] : 0.f;
        lb[2 * i] = o < n ? p.ln_b[o] : 0.f; lb[2 * i + 1] = o < n ? p.ln_b[o + 1] : 0.f;
    }
    // o = 2*tid + 512*i: when C divides 512 the channel pair (c, c+1) is the same for every i -> taps in registers
    const bool reg_taps = (512 % C) == 0;
    const int c_fix = (2 * threadIdx.x) % C;
    float wr0[9], wr1[9], br0 = 0.f, br1 = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { wr0[k] = p.weight[c_fix * 9 + k]; wr1[k] = p.weight[(c_fix + 1) * 9 + k]; }
    if (p.bias) { br0 = p.bias[c_fix]; br1 = p.bias[c_fix + 1]; }
    auto load_rows = [&](int tp, int buf) {
        const int t1 = reflect_idx(tp - P, T1);        // source output row (reflect border of the padded output)
        for (int i = threadIdx.x; i < 3 * (F + 2); i += blockDim.x) {
            const int r = i / (F + 2), f = i % (F + 2);
            const int tin = reflect_idx(2 * t1 + r - 1, T), fin = reflect_idx(f - 1, F);
            rows[buf][r][f] = feats[(int64_t)tin * F + fin];
        }
    };
    auto store2 = [&](OT *dst, float y0, float y1) {
        if constexpr (sizeof(OT) == 2) {
            *reinterpret_cast<uint32_t *>(dst) = (uint32_t)cm_elem<OT>::to_bits(y0) | ((uint32_t)cm_elem<OT>::to_bits(y1) << 16);
        } else {
            *reinterpret_cast<float2 *>(dst) = make_float2(y0, y1);
        }
    };
    load_rows(tp0, 0);
    __syncthreads();
    for (int tp = tp0; tp < tp1; ++tp) {
        const int buf = (tp - tp0) & 1;
        if (tp + 1 < tp1) load_rows(tp + 1, buf ^ 1);  // next row's inputs while this row is computed
        float v[MAXPT];
        float s = 0.f, sq = 0.f;
        // thread owns channel PAIRS: o = 2*tid + 512*i -> (f1, c, c+1); C is even, so a pair never straddles f1
#pragma unroll
        for (int i = 0; i < MAXPT / 2; ++i) {
            const int o = 2 * threadIdx.x + i * 512;
            v[2 * i] = v[2 * i + 1] = 0.f;
            if (o < n) {
                const int f1 = o / C, c = o % C;
                float a0, a1;
                if (reg_taps) {
                    a0 = br0; a1 = br1;
#pragma unroll
                    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                        for (int df = 0; df < 3; ++df) {
                            const float xin = rows[buf][dt][2 * f1 + df];
                            a0 = fmaf(wr0[dt * 3 + df], xin, a0);
                            a1 = fmaf(wr1[dt * 3 + df], xin, a1);
                        }
                } else {
                    a0 = wsh[C * 9 + c]; a1 = wsh[C * 9 + c + 1];
#pragma unroll
                    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                        for (int df = 0; df < 3; ++df) {
                            const float xin = rows[buf][dt][2 * f1 + df];
                            a0 = fmaf(wsh[c * 9 + dt * 3 + df], xin, a0);
                            a1 = fmaf(wsh[(c + 1) * 9 + dt * 3 + df], xin, a1);
                        }
                }
                v[2 * i] = a0;
                v[2 * i + 1] = a1;
                s += a0 + a1;
                sq += a0 * a0 + a1 * a1;
            }
        }
        // one block reduction for both moments (variance from E[x^2] - mean^2 in fp32: |x| = O(1) after input normalisation)
        s = wave_sum(s);
        sq = wave_sum(sq);
        if (lane == 0) { red[buf][0][wave] = s; red[buf][1][wave] = sq; }
        __syncthreads();                               // also orders next row's LDS inputs
        const float mean = (red[buf][0][0] + red[buf][0][1] + red[buf][0][2] + red[buf][0][3]) / n;
        const float ex2 = (red[buf][1][0] + red[buf][1][1] + red[buf][1][2] + red[buf][1][3]) / n;
        const float rstd = rsqrtf(fmaxf(ex2 - mean * mean, 0.f) + p.eps);
        OT *out = reinterpret_cast<OT *>(p.out) + ((int64_t)b * ntp + tp) * (F1 + 2 * P) * C;
#pragma unroll
        for (int i = 0; i < MAXPT / 2; ++i) {
            const int o = 2 * threadIdx.x + i * 512;
            if (o < n) {
                const int f1 = o / C, c = o % C;
                float y0 = (v[2 * i] - mean) * rstd * lg[2 * i] + lb[2 * i];
                float y1 = (v[2 * i + 1] - mean) * rstd * lg[2 * i + 1] + lb[2 * i + 1];
                y0 = y0 > 0.f ? y0 : p.slope * y0;
                y1 = y1 > 0.f ? y1 : p.slope * y1;
                store2(out + (f1 + P) * C + c, y0, y1);
                if (P) {                               // reflected frequency border: fp = 0 <- f1 = 1, fp = F1+1 <- f1 = F1-2
                    if (f1 == 1) store2(out + c, y0, y1);
                    if (f1 == F1 - 2) store2(out + (F1 + 1) * C + c, y0, y1);
                }
            }
        }
    }
}

}  // namespace

extern "C" int cm_cnn_block1(const cm_cnn_block1_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "cnn_block1: args is NULL");
    const cm_cnn_block1_args &a = *args;
    CM_REQUIRE(a.batch > 0 && a.T > 1 && a.F > 1 && a.C > 0 && a.feats && a.weight && a.ln_g && a.ln_b && a.out, CM_EINVAL,
               "cnn_block1: bad sizes or NULL tensor");
    CM_REQUIRE(a.F <= 128 && a.C <= 128 && a.C % 2 == 0, CM_EUNSUPPORTED, "cnn_block1: F %d / C %d unsupported (<= 128, C even)", a.F, a.C);
    CM_REQUIRE(a.pad_out == 0 || a.pad_out == 1, CM_EINVAL, "cnn_block1: pad_out must be 0 or 1");
    const int T1 = (a.T + 1) / 2, F1 = (a.F + 1) / 2;
    CM_REQUIRE(F1 * a.C <= 16 * 256 && T1 >= 3 && F1 >= 3, CM_EUNSUPPORTED, "cnn_block1: F1*C = %d unsupported (<= 4096)", F1 * a.C);
    const int rpw = 8;                                              // output rows per workgroup
    dim3 grid((T1 + 2 * a.pad_out + rpw - 1) / rpw, a.batch);
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    const int pt = (F1 * a.C + 255) / 256;
    auto launch = [&](auto kern) { hipLaunchKernelGGL(kern, grid, dim3(256), 0, st, a, T1, F1, rpw); };
    if (a.io_dtype == CM_BF16) { if (pt <= 10) launch(cnn_block1_kernel<cm_bf16, 10>); else launch(cnn_block1_kernel<cm_bf16, 16>); }
    else if (a.io_dtype == CM_F32) { if (pt <= 10) launch(cnn_block1_kernel<float, 10>); else launch(cnn_block1_kernel<float, 16>); }
    else { cm_set_error("cnn_block1: unsupported dtype %d", a.io_dtype); return CM_EUNSUPPORTED; }
    return cm_launch_status("cm_cnn_block1");
}

extern "C" int cm_conv_cl_fwd(const cm_conv_cl_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "conv_cl_fwd: args is NULL");
    const cm_conv_cl_args &a = *args;
    CM_REQUIRE(a.batch > 0 && a.seqlen > 0 && a.dim > 0, CM_EINVAL, "conv_cl_fwd: bad sizes");
    CM_REQUIRE(a.width == 4, CM_EUNSUPPORTED, "conv_cl_fwd: width %d unsupported (4 only)", a.width);
    CM_REQUIRE(a.x && a.weight_f && a.y_fwd, CM_EINVAL, "conv_cl_fwd: x/weight_f/y_fwd must be non-NULL");
    CM_REQUIRE(!a.y_bwd || a.weight_b, CM_EINVAL, "conv_cl_fwd: y_bwd given without weight_b");
    const int n = a.io_dtype == CM_F32 ? 1 : 2;                    // channels per 32-bit word
    auto ok = [&](const void *ptr, int64_t bs, int64_t ts) { return !ptr || (cm_aligned(ptr, 4) && bs % n == 0 && ts % n == 0); };
    CM_REQUIRE(a.dim % n == 0 && ok(a.x, a.x_bs, a.x_ts) && ok(a.y_fwd, a.yf_bs, a.yf_ts) && ok(a.y_bwd, a.yb_bs, a.yb_ts),
               CM_EALIGN, "conv_cl_fwd: dim and strides must be multiples of %d elements, pointers 4-byte aligned", n);
    const int vpr = a.dim / n;
    constexpr int tc = 8;
    const int64_t threads = (int64_t)a.batch * ((a.seqlen + tc - 1) / tc) * vpr;
    CM_REQUIRE((threads + 255) / 256 <= 2147483647LL, CM_EINVAL, "conv_cl_fwd: problem too large");
    dim3 grid((unsigned)((threads + 255) / 256));
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    if (a.io_dtype == CM_BF16) hipLaunchKernelGGL((conv_cl_kernel<cm_bf16, 4, tc>), grid, dim3(256), 0, st, a, vpr);
    else if (a.io_dtype == CM_F32) hipLaunchKernelGGL((conv_cl_kernel<float, 4, tc>), grid, dim3(256), 0, st, a, vpr);
    else { cm_set_error("conv_cl_fwd: unsupported dtype %d", a.io_dtype); return CM_EUNSUPPORTED; }
    return cm_launch_status("cm_conv_cl_fwd");
}

extern "C" int cm_add_layernorm(const cm_add_ln_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "add_layernorm: args is NULL");
    const cm_add_ln_args &a = *args;
    CM_REQUIRE(a.rows > 0 && a.dim > 0 && (a.x || a.y), CM_EINVAL, "add_layernorm: bad sizes / x and y both NULL");
    CM_REQUIRE(a.dim % 4 == 0 && a.dim <= 1024, CM_EUNSUPPORTED, "add_layernorm: dim %d unsupported (multiple of 4, <= 1024)", a.dim);
    CM_REQUIRE((!a.g1 || a.b1) && (!a.g2 || a.b2), CM_EINVAL, "add_layernorm: LayerNorm weight without bias");
    dim3 grid((unsigned)((a.rows + 3) / 4));
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    const int nv = (a.dim + 255) / 256;
    const int key = (a.y_dtype == CM_BF16 ? 0 : 1) * 2 + (a.out_dtype == CM_BF16 ? 0 : 1);
#define CM_LN(YT, OT)                                                                                   \
    switch (nv) {                                                                                       \
        case 1: hipLaunchKernelGGL((add_ln_kernel<YT, OT, 1>), grid, dim3(256), 0, st, a); break;       \
        case 2: hipLaunchKernelGGL((add_ln_kernel<YT, OT, 2>), grid, dim3(256), 0, st, a); break;       \
        default: hipLaunchKernelGGL((add_ln_kernel<YT, OT, 4>), grid, dim3(256), 0, st, a); break;      \
    }
    switch (key) {
        case 0: CM_LN(cm_bf16, cm_bf16) break;
        case 1: CM_LN(cm_bf16, float) break;
        case 2: CM_LN(float, cm_bf16) break;
        default: CM_LN(float, float) break;
    }
#undef CM_LN
    return cm_launch_status("cm_add_layernorm");
}

extern "C" int cm_glu_dwconv_ln_gelu(const cm_glu_dwconv_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "glu_dwconv: args is NULL");
    const cm_glu_dwconv_args &a = *args;
    CM_REQUIRE(a.batch > 0 && a.seqlen > 0 && a.dim > 0 && a.in && a.out && a.weight && a.ln_g && a.ln_b, CM_EINVAL,
               "glu_dwconv: bad sizes or NULL tensor");
    CM_REQUIRE(a.ksize == 31, CM_EUNSUPPORTED, "glu_dwconv: kernel size %d unsupported (31 only)", a.ksize);
    CM_REQUIRE(a.dim % 2 == 0, CM_EUNSUPPORTED, "glu_dwconv: dim must be even");
    hipStream_t st0 = reinterpret_cast<hipStream_t>(a.stream);
    if (a.io_dtype == CM_BF16 && (a.dim == 256 || a.dim == 512) && a.variant != 1 && cm_aligned(a.in, 16) && cm_aligned(a.out, 16) &&
        (!a.weight_t || cm_aligned(a.weight_t, 8)) && (!a.bias || cm_aligned(a.bias, 8)) && cm_aligned(a.ln_g, 16) && cm_aligned(a.ln_b, 16)) {
        constexpr int TT = 32;
        const dim3 grid((a.seqlen + TT - 1) / TT, a.batch);
        const bool lin = a.lin_w != nullptr;
        CM_REQUIRE(!lin || (a.dim == 256 && a.lin_b && cm_aligned(a.lin_w, 16) && cm_aligned(a.lin_b, 16)), CM_EUNSUPPORTED,
                   "glu_dwconv: the Linear epilogue needs dim 256, lin_b, 16-byte aligned tensors");
        const size_t lds_co = (size_t)TT * (a.dim + 16) * 4, lds_g = (size_t)(TT + 30) * a.dim * 2;
        const size_t smem = lds_co > lds_g ? lds_co : lds_g;
        auto run = [&](auto kern) -> int {
            if (smem > 48 * 1024) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
                if (e != hipSuccess) { cm_set_error("glu_dwconv: LDS attribute failed: %s", hipGetErrorString(e)); return (int)e; }
            }
            hipLaunchKernelGGL(kern, grid, dim3(256), smem, st0, a);
            return cm_launch_status("cm_glu_dwconv_ln_gelu(rows)");
        };
        if (lin) return run(dwconv_rows_kernel<31, TT, 4, true>);
        return a.dim == 256 ? run(dwconv_rows_kernel<31, TT, 4>) : run(dwconv_rows_kernel<31, TT, 8>);
    }
    CM_REQUIRE(!a.lin_w, CM_EUNSUPPORTED, "glu_dwconv: the Linear epilogue is built for bf16 rows of dim 256 (16-byte aligned) only");
    // time tile: LDS holds (tt + K - 1 + tt) rows of dim floats; keep it under 64 KB so 2 workgroups share a CU
    const size_t el = a.io_dtype == CM_F32 ? 4 : 2;
    auto lds_bytes = [&](int t) { return (size_t)t * (a.dim + 16) * 4 + (size_t)(t + 30) * a.dim * el; };
    int tt = 16;
    while (tt > 4 && lds_bytes(tt) > 64 * 1024) tt /= 2;
    const size_t smem = lds_bytes(tt);
    dim3 grid((a.seqlen + tt - 1) / tt, a.batch);
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    auto go = [&](auto kern) -> int {
        if (smem > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) { cm_set_error("glu_dwconv: LDS attribute failed: %s", hipGetErrorString(e)); return (int)e; }
        }
        hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, a);
        return cm_launch_status("cm_glu_dwconv_ln_gelu");
    };
    if (a.io_dtype == CM_BF16) {
        if (tt == 16) return go(glu_dwconv_kernel<cm_bf16, 31, 16>);
        if (tt == 8) return go(glu_dwconv_kernel<cm_bf16, 31, 8>);
        return go(glu_dwconv_kernel<cm_bf16, 31, 4>);
    }
    if (a.io_dtype == CM_F32) {
        if (tt == 16) return go(glu_dwconv_kernel<float, 31, 16>);
        if (tt == 8) return go(glu_dwconv_kernel<float, 31, 8>);
        return go(glu_dwconv_kernel<float, 31, 4>);
    }
    cm_set_error("glu_dwconv: unsupported dtype %d", a.io_dtype);
    return CM_EUNSUPPORTED;
}
