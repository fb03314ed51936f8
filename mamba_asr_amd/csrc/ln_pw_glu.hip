// ln_pw_glu.hip — the seam between the BiMamba mixer and the convolution module as ONE kernel (contract: cm_ln_pw_glu
// in include/conmamba_hip.h; reference modules/Conmamba.py:639-640 and :441-443):
//     x <- x + alpha * y            (residual add of the mixer output, fp32 stream)
//     h  = LayerNorm(x)             (convolution module's first LayerNorm)
//     pw = h @ W^T + b              (pointwise Conv1d D -> 2D, kernel 1)
//     g  = pw[:, :D] * sigmoid(pw[:, D:])        (GLU over the channel axis)
// Separately this was cm_add_layernorm (x read + written, h written), a library GEMM (h read, 2D-wide pw written) and
// the GLU inside cm_glu_dwconv_ln_gelu (2D-wide pw read with its 30-row halo, sigmoid recomputed per overlapping tile).
// Here a workgroup owns 64 tokens: the normalised tokens sit in LDS (bf16, rows padded to 528 B), each wave computes
// 64 features x 64 tokens of BOTH halves with v_mfma_f32_16x16x32_bf16 (weights as the A operand straight from their
// packed image in L2, as in cm_ffn_fused), so a lane holds matching (value, gate) pairs and the GLU happens in
// registers; only x (once in, once out) and the D-wide bf16 g touch HBM.
#include "cm_common.h"


namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int D = 256;        // d_model
constexpr int TOK = 64;       // tokens per workgroup
constexpr int XS = 264;       // LDS row stride in bf16 elements (528 bytes)

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ uint32_t pack2(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}

// OCC: workgroups per CU the register budget is cut for; PF: weight-fragment ring depth (k-steps in flight)
template <int OCC, int PF>
__global__ __launch_bounds__(256, OCC) void ln_pw_glu_kernel(const cm_ln_pw_glu_args p) {
    __shared__ __attribute__((aligned(16))) uint16_t xn[TOK * XS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int t0 = blockIdx.x * TOK, M = p.rows;
    const uint16_t *yv = reinterpret_cast<const uint16_t *>(p.y);

    // weight stream: W (2D, D) packed in 16-row x 32-column fragment images (cm_ffn_pack_weights).  The wave's 64 value
    // features (and their 64 gates, rows D + ...) are processed in two passes of 32: step s of pass ps = half * 8 + k-tile
    // (accumulators: 2 halves x 2 bands x 4 token tiles = 64 VGPRs instead of 128 -> 3-4 workgroups per CU, which this
    // memory-phase-dominated kernel needs more than it needs MFMA density)
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w), 0, 2 * D * D * 2, 0x00020000);
    const int vl = lane * 16;
    auto wload = [&](int ps, int s, bf16x8(&dst)[2]) {
        const int half = s >> 3, ks = s & 7;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
            dst[mb] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                     wr, vl, (((half * D + wave * 64) / 16 + 2 * ps + mb) * (D / 32) + ks) * 1024, 0));
    };
    bf16x8 wq[PF][2];
#pragma unroll
    for (int s = 0; s < PF; ++s) wload(0, s, wq[s]);

    // ---- phase 0: x <- x + alpha*y (written back), xn = LayerNorm(x) in bf16.  Wave w owns tokens 16w .. 16w+15, four per
    // round; a row of 16 lanes holds one token (16 floats per lane): statistics = in-lane adds + four DPP steps.
    {
        float4 v[4][4];
#pragma unroll
        for (int rd = 0; rd < 4; ++rd) {
            const int tok = min(t0 + wave * 16 + rd * 4 + lq, M - 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = (l15 + 16 * i) * 4;
                float4 x4 = *reinterpret_cast<const float4 *>(p.x + (int64_t)tok * D + col);
                if (yv) {
                    const uint2 a = *reinterpret_cast<const uint2 *>(yv + (int64_t)tok * D + col);
                    x4.x = fmaf(p.alpha, cm_bf16_lo(a.x), x4.x);
                    x4.y = fmaf(p.alpha, cm_bf16_hi(a.x), x4.y);
                    x4.z = fmaf(p.alpha, cm_bf16_lo(a.y), x4.z);
                    x4.w = fmaf(p.alpha, cm_bf16_hi(a.y), x4.w);
                }
                v[rd][i] = x4;
            }
        }
#pragma unroll
        for (int rd = 0; rd < 4; ++rd) {
            const int tok = t0 + wave * 16 + rd * 4 + lq;
            if (p.x_out && tok < M) {
#pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<float4 *>(p.x_out + (int64_t)tok * D + (l15 + 16 * i) * 4) = v[rd][i];
            }
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) s += (v[rd][i].x + v[rd][i].y) + (v[rd][i].z + v[rd][i].w);
            const float mean = cm_group_sum<16>(s) * (1.f / D);
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[rd][i].x -= mean; v[rd][i].y -= mean; v[rd][i].z -= mean; v[rd][i].w -= mean;
                q = fmaf(v[rd][i].x, v[rd][i].x, fmaf(v[rd][i].y, v[rd][i].y, fmaf(v[rd][i].z, v[rd][i].z, fmaf(v[rd][i].w, v[rd][i].w, q))));
            }
            const float rstd = rsqrtf(cm_group_sum<16>(q) * (1.f / D) + p.eps);
            uint16_t *dst = xn + (wave * 16 + rd * 4 + lq) * XS;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = (l15 + 16 * i) * 4;
                const float4 g = *reinterpret_cast<const float4 *>(p.ln_g + col);
                const float4 b = *reinterpret_cast<const float4 *>(p.ln_b + col);
                uint2 pk;
                pk.x = pack2(fmaf(v[rd][i].x * rstd, g.x, b.x), fmaf(v[rd][i].y * rstd, g.y, b.y));
                pk.y = pack2(fmaf(v[rd][i].z * rstd, g.z, b.z), fmaf(v[rd][i].w * rstd, g.w, b.w));
                *reinterpret_cast<uint2 *>(dst + col) = pk;
            }
        }
    }
    lds_barrier();

    // ---- GEMM + GLU epilogue, two passes of 32 features.  Lane holds token nb*16 + l15, features f0 + mb*16 + j.
    const uint16_t *xfrag = xn + l15 * XS + lq * 8;
    auto read_frags = [&](int ks, bf16x8(&bf)[4]) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) bf[nb] = *reinterpret_cast<const bf16x8 *>(xfrag + nb * 16 * XS + ks * 32);
    };
    uint16_t *out = reinterpret_cast<uint16_t *>(p.out);
    // The gated outputs leave through LDS: a lane's accumulators are 4 features of 16 different tokens (8-byte pieces of 16 rows
    // per store instruction); the first pass's results wait in registers (the token tile is still being read), then both
    // passes are laid into the dead tile and stored as whole 512-byte rows.
    uint2 held[2][4];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        f32x4 acc[2][2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[h][mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 bfa[4], bfb[4];
        read_frags(0, bfa);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            bf16x8(&cur)[4] = (s & 1) ? bfb : bfa;
            bf16x8(&nxt)[4] = (s & 1) ? bfa : bfb;
            if (s + 1 < 16) read_frags((s + 1) & 7, nxt);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    acc[s >> 3][mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[s % PF][mb], cur[nb], acc[s >> 3][mb][nb], 0, 0, 0);
            if (s + PF < 16) wload(ps, s + PF, wq[s % PF]);
            else if (ps == 0) wload(1, s + PF - 16, wq[s % PF]);  // next pass's first steps
            __builtin_amdgcn_sched_barrier(0);                    // keep the refill here (see cm_ffn_fused)
        }
        const int f0 = wave * 64 + ps * 32 + lq * 4;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const float4 ba = *reinterpret_cast<const float4 *>(p.bias + f0 + mb * 16);
            const float4 bg = *reinterpret_cast<const float4 *>(p.bias + D + f0 + mb * 16);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const float o0 = (acc[0][mb][nb][0] + ba.x) * cm_sigmoid(acc[1][mb][nb][0] + bg.x);
                const float o1 = (acc[0][mb][nb][1] + ba.y) * cm_sigmoid(acc[1][mb][nb][1] + bg.y);
                const float o2 = (acc[0][mb][nb][2] + ba.z) * cm_sigmoid(acc[1][mb][nb][2] + bg.z);
                const float o3 = (acc[0][mb][nb][3] + ba.w) * cm_sigmoid(acc[1][mb][nb][3] + bg.w);
                if (ps == 0) held[mb][nb] = uint2{pack2(o0, o1), pack2(o2, o3)};
                else {
                    if (mb == 0 && nb == 0) lds_barrier();        // every wave has read its last token fragments
                    *reinterpret_cast<uint2 *>(xn + (nb * 16 + l15) * XS + f0 + mb * 16) = uint2{pack2(o0, o1), pack2(o2, o3)};
                    *reinterpret_cast<uint2 *>(xn + (nb * 16 + l15) * XS + f0 - 32 + mb * 16) = held[mb][nb];
                }
            }
        }
    }
    lds_barrier();
#pragma unroll
    for (int i = 0; i < TOK * 32 / 256; ++i) {
        const int idx = tid + 256 * i, row = idx >> 5, chunk = idx & 31;
        const uint4 v = *reinterpret_cast<const uint4 *>(xn + row * XS + chunk * 8);
        if (t0 + row < M) *reinterpret_cast<uint4 *>(out + (int64_t)(t0 + row) * D + chunk * 8) = v;
    }
}

}  // namespace

extern "C" int cm_ln_pw_glu(const cm_ln_pw_glu_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "ln_pw_glu: args is NULL");
    const cm_ln_pw_glu_args &a = *args;
    CM_REQUIRE(a.rows > 0 && a.x && a.ln_g && a.ln_b && a.w && a.bias && a.out, CM_EINVAL, "ln_pw_glu: bad sizes or NULL tensor");
    CM_REQUIRE(a.dim == D, CM_EUNSUPPORTED, "ln_pw_glu: d_model must be 256 (got %d)", a.dim);
    CM_REQUIRE(cm_aligned(a.x, 16) && cm_aligned(a.ln_g, 16) && cm_aligned(a.ln_b, 16) && cm_aligned(a.w, 16) && cm_aligned(a.bias, 16) &&
                   cm_aligned(a.out, 16) && (!a.y || cm_aligned(a.y, 8)) && (!a.x_out || cm_aligned(a.x_out, 16)),
               CM_EALIGN, "ln_pw_glu: tensors must be 16-byte aligned (y / out 8)");
    const dim3 grid((a.rows + TOK - 1) / TOK);
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    // four workgroups per CU (128 VGPRs with a 2-deep weight ring): 64 k rows = 1000 workgroups run in ONE round of 1024 slots;
    // at three per CU (4-deep ring, 154 VGPRs) the last 232 ran alone: 69 -> 58 us in the encoder (profiles/r02).
    // cm_debug_set(41) keeps the old shape for A/B runs.
#ifdef CM_ABLATE
    if (cm_debug_get() == 41) {
        hipLaunchKernelGGL((ln_pw_glu_kernel<3, 4>), grid, dim3(256), 0, st, a);
        return cm_launch_status("cm_ln_pw_glu");
    }
#endif
    hipLaunchKernelGGL((ln_pw_glu_kernel<4, 2>), grid, dim3(256), 0, st, a);
    return cm_launch_status("cm_ln_pw_glu");
}
