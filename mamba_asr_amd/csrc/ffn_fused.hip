// ffn_fused.hip — the whole position-wise feed-forward module of a ConMamba layer as ONE kernel
// (contract: cm_ffn_fused in include/conmamba_hip.h; reference modules/Conmamba.py:631-650 around
// PositionalwiseFeedForward: x + 0.5 * W2 gelu(W1 LN(x) + b1) + b2, then the layer's next LayerNorm).
//
// Through the vendor library this module was LayerNorm kernel -> GEMM (rows x 1024 hidden written to HBM) -> GEMM
// (hidden read back) -> residual + LayerNorm kernel, and the two small-K GEMMs ran at 1-2 TB/s of effective traffic.
// Here a workgroup owns 64 tokens end to end: the hidden activations never leave the CU.
//   * d_model = 256.  Four waves; LDS holds the normalised tokens (64 x 256 bf16) and one 256-wide slab of hidden
//     activations (64 x 256 bf16), rows padded to 528 B so ds_read_b128 fragments are conflict-free.
//   * v_mfma_f32_16x16x32_bf16 with the WEIGHT rows as the A operand and tokens as the B operand: a lane's accumulator
//     registers are 4 consecutive features of one token, so the hidden slab is written back with 8-byte LDS stores,
//     the output with 16-byte global stores, and the LayerNorm statistics need two cross-lane adds per token.
//   * each wave computes a 64-feature x 64-token tile (4 x 4 MFMA tiles): weight fragments come straight from
//     global memory (L2-resident, 1 MB per FFN) into a 4-deep register ring that runs across both GEMMs and across
//     slabs -- every weight byte is loaded by exactly one wave of the workgroup and never touches LDS; token fragments
//     come from LDS, 4 ds_read_b128 per 16 MFMAs (64 B/clk/CU at full MFMA rate).
//   * barriers wait on LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier), so the weight ring is not drained.
#include "cm_common.h"
#include "cm_dropout.h"
#include <type_traits>


namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int D = 256;        // d_model
constexpr int TH = 1;         // token halves per workgroup: waves (th, fq) = (token half, feature quarter); the two waves of
                              // a feature quarter stream the SAME weight fragments.  TH = 2 (128 tokens, 8 waves) was measured: 109 vs 100 us per
                              // 64k rows -- the L2 weight stream is not what bounds this kernel
constexpr int TOK = 64 * TH;  // tokens per workgroup
constexpr int NT = 256 * TH;  // threads per workgroup
constexpr int XS = 264;       // LDS row stride in bf16 elements (528 bytes)
constexpr int CH = 256;       // hidden slab
constexpr int PF = 4;         // weight-fragment ring depth (k-steps in flight)

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ uint32_t pack2(float a, float b) {         // one v_cvt_pk_bf16_f32
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}

// VAR (cm_debug_set, timing only): 1 = scalar GELU (cm_gelu_bf16 per element; 108-110 us vs 99-100 us packed at 64k rows), 2 = no GELU, 3 = no weight stream after the first fill,
// 4 = no token-fragment reads after the first, 5 = 2 + 3 + 4.  (Starting the workgroups of the second occupancy slot
// 3.4 / 10 us late, so that a CU's two workgroups are in different phases, was also tried: no effect.)
// PROJ: the Linear that consumes h (the BiMamba in_proj after the layer's first feed-forward module, reference
// bimamba.py:192-200) runs on the tile before it leaves the CU: proj_out = LN2(r) @ proj_w^T, h itself is not stored.
// TRAIN: the module's training forward (cm_ffn_args.pre_out ...): the slab goes to LDS as bf16(pre-activation), a row-wise pass stores
// it (whole 512-byte row segments) and replaces it in place by dropout(GELU(.)) -- the activation is then a function of the STORED
// value, which is what the backward differentiates and recomputes -- and the epilogue applies the second dropout.
template <bool ADD, int VAR = 0, bool PROJ = false, bool TRAIN = false>
__global__ __launch_bounds__(NT, 2) void ffn_fused_kernel(const cm_ffn_args p) {
    constexpr int PFK = TRAIN ? 2 : PF;                          // the training variant carries its row-wise pass: a 2-deep ring keeps it at 256 VGPRs
    const uint64_t seed1 = TRAIN ? cm_drop_seed(p.seed1, p.seed_epoch) : 0, seed2 = TRAIN ? cm_drop_seed(p.seed2, p.seed_epoch) : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *xn = reinterpret_cast<uint16_t *>(smem);            // [TOK][XS] normalised tokens
    uint16_t *hc = xn + TOK * XS;                                 // [TOK][XS] hidden slab
    float *red = reinterpret_cast<float *>(hc + TOK * XS);        // [4][TOK] LayerNorm partial sums
    float *b1s = red + 4 * TOK;                                   // [hidden] first bias (a global load inside the slab
                                                                  // loop would wait on vmcnt and drain the weight ring)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // 0 .. 4*TH-1: owns tokens 16 wv .. +15 in the LayerNorm phases
    const int wave = wv & 3;                                      // feature quarter (uniform: feeds the buffer loads' scalar offset)
    const int th = wv >> 2;                                       // token half in the GEMMs / epilogue
    const int l15 = lane & 15, lq = lane >> 4;
    const int t0 = blockIdx.x * TOK, M = p.rows, F = p.hidden;
    const int nch = F / CH;
    const uint16_t *W1 = reinterpret_cast<const uint16_t *>(p.w1);
    const uint16_t *W2 = reinterpret_cast<const uint16_t *>(p.w2);
    const uint16_t *addend = reinterpret_cast<const uint16_t *>(p.addend);

    // ---- weight stream.  Step s of slab c: s < 8  -> W1 rows c*CH + wave*64 + mb*16 + l15, columns s*32 + lq*8
    //                                        s >= 8 -> W2 rows wave*64 + mb*16 + l15, columns c*CH + (s-8)*32 + lq*8
    // Weights are PACKED (cm_ffn_pack_weights): 16-row x 32-column tiles, each stored as the 1 KB image of one MFMA
    // operand fragment (lane L = lq*16 + l15 owns bytes [16 L, 16 L + 16)), tiles of one 16-row band contiguous over
    // the columns.  A wave-level fragment load is then ONE contiguous kilobyte (8 full cache lines) instead of 16 half
    // lines from 16 rows -- the row-major version of this kernel was bound by L1 tag lookups (9 B/clk/CU).
    // Buffer loads: one VGPR of lane offset, everything else in SGPRs / the immediate field.
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(W1), 0, F * D * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(W2), 0, F * D * 2, 0x00020000);
    // training forward: the stored pre-activations, (rows, hidden) bf16 (element offsets fit 32 bits: rows x hidden < 2^31)
    const __amdgpu_buffer_rsrc_t rpre = __builtin_amdgcn_make_buffer_rsrc(p.pre_out, 0, TRAIN && p.pre_out ? (int)((int64_t)M * F * 2) : 0, 0x00020000);
    const int vl = lane * 16;
    const int kt2 = F / 32;                                       // tiles per 16-row band of W2
    auto wload = [&](int c, int s, bf16x8(&dst)[4], bool first = false) {
        if constexpr (VAR == 3 || VAR == 5) { if (!first) return; }
        if (s < 8) {                                              // W1 band (c*CH + wave*64)/16 + mb, tile s
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
                dst[mb] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                         r1, vl, (((c * CH + wave * 64) / 16 + mb) * (D / 32) + s) * 1024, 0));
        } else {                                                  // W2 band wave*4 + mb, tile c*CH/32 + (s-8)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
                dst[mb] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                         r2, vl, ((wave * 4 + mb) * kt2 + c * (CH / 32) + (s - 8)) * 1024, 0));
        }
    };
    bf16x8 wq[PFK][4];
#pragma unroll
    for (int s = 0; s < PFK; ++s) wload(0, s, wq[s], true);

    // first bias -> LDS: requested here with 16-byte loads, stored below once the rows have been requested too.  (As the rolled loop
    // `b1s[i] = p.b1[i]` this was hidden / 256 dependent round trips -- load, s_waitcnt vmcnt(0), ds_write -- in front of the row loads.)
    float4 b1r[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i4 = (tid + NT * k) * 4;
        b1r[k] = i4 < p.hidden ? *reinterpret_cast<const float4 *>(p.b1 + i4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // ---- phase 0: xin = x (+ add_scale * addend); xn = LayerNorm_pre(xin) in bf16.
    // Wave w owns tokens 16w .. 16w+15, four per round: a row of 16 lanes holds one token (16 floats per lane), so the
    // statistics are in-lane adds + four full-rate DPP steps (no LDS-crossbar shuffles), and all 16 loads of the wave
    // are in flight together.
    auto load_x4 = [&](int tok, int col) {
        float4 v = *reinterpret_cast<const float4 *>(p.x + (int64_t)tok * D + col);
        if constexpr (ADD) {
            const uint2 a = *reinterpret_cast<const uint2 *>(addend + (int64_t)tok * D + col);
            v.x = fmaf(p.add_scale, __uint_as_float(a.x << 16), v.x);
            v.y = fmaf(p.add_scale, __uint_as_float(a.x & 0xffff0000u), v.y);
            v.z = fmaf(p.add_scale, __uint_as_float(a.y << 16), v.z);
            v.w = fmaf(p.add_scale, __uint_as_float(a.y & 0xffff0000u), v.w);
        }
        return v;
    };
    {
        float4 v[4][4];
#pragma unroll
        for (int rd = 0; rd < 4; ++rd) {
            const int tok = min(t0 + wv * 16 + rd * 4 + lq, M - 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[rd][i] = load_x4(tok, (l15 + 16 * i) * 4);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {                                   // loads return in order: this waits for the bias, not for the rows
            const int i4 = (tid + NT * k) * 4;
            if (i4 < p.hidden) *reinterpret_cast<float4 *>(b1s + i4) = b1r[k];
        }
#pragma unroll
        for (int rd = 0; rd < 4; ++rd) {
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
            const float rstd = rsqrtf(cm_group_sum<16>(q) * (1.f / D) + p.pre_eps);
            if constexpr (TRAIN) {
                const int trow = t0 + wv * 16 + rd * 4 + lq;
                if (p.stats_out && l15 == 0 && trow < M) p.stats_out[trow] = mean, p.stats_out[(int64_t)M + trow] = rstd;
            }
            uint16_t *dst = xn + (wv * 16 + rd * 4 + lq) * XS;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = (l15 + 16 * i) * 4;
                const float4 g = *reinterpret_cast<const float4 *>(p.pre_g + col);
                const float4 b = *reinterpret_cast<const float4 *>(p.pre_b + col);
                uint2 pk;
                pk.x = pack2(fmaf(v[rd][i].x * rstd, g.x, b.x), fmaf(v[rd][i].y * rstd, g.y, b.y));
                pk.y = pack2(fmaf(v[rd][i].z * rstd, g.z, b.z), fmaf(v[rd][i].w * rstd, g.w, b.w));
                *reinterpret_cast<uint2 *>(dst + col) = pk;
            }
        }
    }
    lds_barrier();
    if constexpr (TRAIN) {
        if (p.xn_out) {                                           // the normalised tokens as the first GEMM reads them, whole rows
            uint16_t *xo = reinterpret_cast<uint16_t *>(p.xn_out);
#pragma unroll 2
            for (int i = 0; i < TOK * 32 / NT; ++i) {
                const int idx = tid + NT * i, row = idx >> 5, ch = idx & 31;
                const uint4 v = *reinterpret_cast<const uint4 *>(xn + row * XS + ch * 8);
                if (t0 + row < M) *reinterpret_cast<uint4 *>(xo + (int64_t)(t0 + row) * D + ch * 8) = v;
            }
        }
    }

    // ---- main loop over hidden slabs
    f32x4 acc2[4][4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc2[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint16_t *xfrag = xn + (th * 64 + l15) * XS + lq * 8;
    const uint16_t *hfrag = hc + (th * 64 + l15) * XS + lq * 8;
    uint16_t *hdst = hc + (th * 64 + l15) * XS + wave * 64 + lq * 4;

    auto read_frags = [&](const uint16_t *base, int ks, bf16x8(&bf)[4]) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) bf[nb] = *reinterpret_cast<const bf16x8 *>(base + nb * 16 * XS + ks * 32);
    };
    const int f0 = wave * 64 + lq * 4;
    float r[4][4][4];                                             // epilogue: [token tile][feature tile][4 features]
    // one hidden slab; the last one is peeled (LAST): it does not request the next slab's weights
    auto slab = [&](const int c, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        f32x4 acc1[4][4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc1[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        // GEMM 1: hidden slab (this wave: 64 hidden units) x 64 tokens, K = 256.  Token fragments of step s+1 are
        // read from LDS while the MFMAs of step s run.
        bf16x8 bfa[4], bfb[4];
        read_frags(xfrag, 0, bfa);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            bf16x8(&cur)[4] = (s & 1) && !(VAR == 4 || VAR == 5) ? bfb : bfa;
            bf16x8(&nxt)[4] = (s & 1) ? bfa : bfb;
            if (s + 1 < 8 && !(VAR == 4 || VAR == 5)) read_frags(xfrag, s + 1, nxt);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    acc1[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[s % PFK][mb], cur[nb], acc1[mb][nb], 0, 0, 0);
            wload(c, s + PFK, wq[s % PFK]);
            __builtin_amdgcn_sched_barrier(0);                    // keep the refill HERE: the scheduler otherwise sinks
        }                                                         // the loads to their use and the ring is gone
        // bias + GELU -> bf16 slab in LDS (token-major, hidden contiguous)
        if (c > 0) lds_barrier();                                 // every wave is done reading the previous slab
        if constexpr (TRAIN) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const float4 bv = *reinterpret_cast<const float4 *>(b1s + c * CH + wave * 64 + mb * 16 + lq * 4);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    uint2 pk;
                    pk.x = pack2(acc1[mb][nb][0] + bv.x, acc1[mb][nb][1] + bv.y);
                    pk.y = pack2(acc1[mb][nb][2] + bv.z, acc1[mb][nb][3] + bv.w);
                    *reinterpret_cast<uint2 *>(hdst + nb * 16 * XS + mb * 16) = pk;
                }
            }
            lds_barrier();
            // row-wise pass: 16-byte pieces, a wave instruction = two whole 512-byte row segments of the slab
            constexpr int NPC = TOK * (CH / 8) / NT, NPH = 4;       // pieces per thread, pieces in flight
            const bool drop = p.p1 > 0.f;
            const uint32_t th1 = cm_drop_thresh(p.p1);
            const float sc1 = drop ? cm_drop_scale(p.p1) : 1.f;
            // piece i of this thread: row (tid >> 5) + 8 i, hidden columns 8 (tid & 31) ..: one 32-bit element offset, scalar steps
            const uint32_t el0 = (uint32_t)(t0 + (tid >> 5)) * (uint32_t)F + (uint32_t)(c * CH + (tid & 31) * 8);
            uint16_t *hrow = hc + (tid >> 5) * XS + (tid & 31) * 8;
#pragma unroll
            for (int i0 = 0; i0 < NPC; i0 += NPH) {
                uint4 pv[NPH];
#pragma unroll
                for (int i = 0; i < NPH; ++i) pv[i] = *reinterpret_cast<const uint4 *>(hrow + (i0 + i) * 8 * XS);
#pragma unroll
                for (int i = 0; i < NPH; ++i) {
                    const uint32_t el = el0 + (uint32_t)((i0 + i) * 8) * (uint32_t)F;
                    if (p.pre_out)                                   // rows past the end fall outside the descriptor
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pv[i]), rpre, el * 2, 0, 0);
                    const uint32_t keep = drop ? cm_drop_keep8(seed1, (uint64_t)(el >> 3), th1) : 0xffu;
                    const uint32_t w[4] = {pv[i].x, pv[i].y, pv[i].z, pv[i].w};
                    uint32_t o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = cm_gelu_drop_bf16_pack2(__uint_as_float(w[j] << 16), __uint_as_float(w[j] & 0xffff0000u),
                                                                             (keep >> (2 * j)) & 1u, (keep >> (2 * j + 1)) & 1u, sc1);
                    *reinterpret_cast<uint4 *>(hrow + (i0 + i) * 8 * XS) = make_uint4(o[0], o[1], o[2], o[3]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const float4 bv = *reinterpret_cast<const float4 *>(b1s + c * CH + wave * 64 + mb * 16 + lq * 4);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                uint2 pk;
                if constexpr (VAR == 1) {
                    pk.x = pack2(cm_gelu_bf16(acc1[mb][nb][0] + bv.x), cm_gelu_bf16(acc1[mb][nb][1] + bv.y));
                    pk.y = pack2(cm_gelu_bf16(acc1[mb][nb][2] + bv.z), cm_gelu_bf16(acc1[mb][nb][3] + bv.w));
                } else if constexpr (VAR == 2 || VAR == 5) {
                    pk.x = pack2(acc1[mb][nb][0] + bv.x, acc1[mb][nb][1] + bv.y);
                    pk.y = pack2(acc1[mb][nb][2] + bv.z, acc1[mb][nb][3] + bv.w);
                } else {
                    pk.x = cm_gelu_bf16_pack2(acc1[mb][nb][0] + bv.x, acc1[mb][nb][1] + bv.y);
                    pk.y = cm_gelu_bf16_pack2(acc1[mb][nb][2] + bv.z, acc1[mb][nb][3] + bv.w);
                }
                *reinterpret_cast<uint2 *>(hdst + nb * 16 * XS + mb * 16) = pk;
            }
        }
        }
        lds_barrier();
        // GEMM 2: 64 output features x 64 tokens, K = this slab
        read_frags(hfrag, 0, bfa);
#pragma unroll
        for (int s = 8; s < 16; ++s) {
            bf16x8(&cur)[4] = (s & 1) && !(VAR == 4 || VAR == 5) ? bfb : bfa;
            bf16x8(&nxt)[4] = (s & 1) ? bfa : bfb;
            if (s + 1 < 16 && !(VAR == 4 || VAR == 5)) read_frags(hfrag, s + 1 - 8, nxt);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    acc2[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[s % PFK][mb], cur[nb], acc2[mb][nb], 0, 0, 0);
            if (s + PFK < 16) wload(c, s + PFK, wq[s % PFK]);
            else if constexpr (!LAST) wload(c + 1, s + PFK - 16, wq[s % PFK]);   // next slab's first steps
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int c = 0; c + 1 < nch; ++c) slab(c, std::false_type{});
    slab(nch - 1, std::true_type{});
    // Residual rows for the epilogue, requested AFTER the last GEMM.  Requested before it (to travel under its MFMAs) they sat in
    // front of its weight fragments in the in-order return queue: in-kernel stamps showed that GEMM taking 10.8 k ticks
    // against 3-4 k for the others (profiles/r02/ffn_stamps_and_variants.log); 120 / 108 / 105 -> 110 / 103 / 100 us for the
    // second FFN of a layer, 12.52 -> 12.44 ms per step.
    // They come in as whole rows (a wave instruction = one 1 KB row, + its bf16 addend row) and reach the accumulator layout
    // through LDS -- read straight into that layout they are 16-byte pieces of 16 rows per instruction, on the critical path.
    if constexpr (VAR == 25) {                                    // timing variant: the direct reads
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int tok = min(t0 + th * 64 + nb * 16 + l15, M - 1);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const float4 xv = load_x4(tok, f0 + mb * 16);
                r[nb][mb][0] = xv.x; r[nb][mb][1] = xv.y; r[nb][mb][2] = xv.z; r[nb][mb][3] = xv.w;
            }
        }
    } else {
        constexpr int SR = 260;
        float *stg = reinterpret_cast<float *>(smem);
        float4 rows_[TOK * 64 / NT];
#pragma unroll
        for (int i = 0; i < TOK * 64 / NT; ++i) {
            const int idx = tid + NT * i;
            rows_[i] = load_x4(min(t0 + (idx >> 6), M - 1), (idx & 63) * 4);
        }
        lds_barrier();                                            // every wave is done with the last GEMM's fragments
#pragma unroll
        for (int i = 0; i < TOK * 64 / NT; ++i) {
            const int idx = tid + NT * i;
            *reinterpret_cast<float4 *>(stg + (idx >> 6) * SR + (idx & 63) * 4) = rows_[i];
        }
        lds_barrier();
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const float4 xv = *reinterpret_cast<const float4 *>(stg + (th * 64 + nb * 16 + l15) * SR + f0 + mb * 16);
                r[nb][mb][0] = xv.x; r[nb][mb][1] = xv.y; r[nb][mb][2] = xv.z; r[nb][mb][3] = xv.w;
            }
    }

    // ---- epilogue: r = xin + alpha (acc2 + b2); optional LN1 -> stream; optional LN2 -> h_out.
    // Lane holds token nb*16 + l15, features wave*64 + mb*16 + lq*4 + j.
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const float4 bv = *reinterpret_cast<const float4 *>(p.b2 + f0 + mb * 16);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            if constexpr (TRAIN) {
                if (p.p2 > 0.f) {                                     // second dropout: on W2 g + b2, in front of the scaled residual add
                    const int64_t e0 = (int64_t)(t0 + th * 64 + nb * 16 + l15) * D + f0 + mb * 16;
                    const uint32_t keep = cm_drop_keep4(seed2, (uint64_t)e0 >> 3, (int)((e0 >> 2) & 1), cm_drop_thresh(p.p2));
                    const float sc2 = cm_drop_scale(p.p2);
                    const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        r[nb][mb][j] = fmaf(p.alpha, ((keep >> j) & 1u) ? (acc2[mb][nb][j] + bb[j]) * sc2 : 0.f, r[nb][mb][j]);
                    continue;
                }
            }
            r[nb][mb][0] = fmaf(p.alpha, acc2[mb][nb][0] + bv.x, r[nb][mb][0]);
            r[nb][mb][1] = fmaf(p.alpha, acc2[mb][nb][1] + bv.y, r[nb][mb][1]);
            r[nb][mb][2] = fmaf(p.alpha, acc2[mb][nb][2] + bv.z, r[nb][mb][2]);
            r[nb][mb][3] = fmaf(p.alpha, acc2[mb][nb][3] + bv.w, r[nb][mb][3]);
        }
    }
    // sum over a token's 256 features: 16 in-lane, 4 lane groups, 4 waves (through LDS)
    auto token_sums = [&](float (&v)[4]) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            v[nb] += __shfl_xor(v[nb], 16, 64);
            v[nb] += __shfl_xor(v[nb], 32, 64);
        }
        lds_barrier();                                            // previous use of red is over
        if (lq == 0) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) red[wave * TOK + th * 64 + nb * 16 + l15] = v[nb];
        }
        lds_barrier();
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int t = th * 64 + nb * 16 + l15;
            v[nb] = (red[t] + red[TOK + t]) + (red[2 * TOK + t] + red[3 * TOK + t]);
        }
    };
    auto layer_norm = [&](const float *g, const float *b, float eps) {
        float s[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            s[nb] = 0.f;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) s[nb] += (r[nb][mb][0] + r[nb][mb][1]) + (r[nb][mb][2] + r[nb][mb][3]);
        }
        token_sums(s);
        float q[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            s[nb] *= (1.f / D);
            q[nb] = 0.f;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d = r[nb][mb][j] - s[nb]; q[nb] = fmaf(d, d, q[nb]); }
        }
        token_sums(q);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const float4 gv = *reinterpret_cast<const float4 *>(g + f0 + mb * 16);
            const float4 bv = *reinterpret_cast<const float4 *>(b + f0 + mb * 16);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const float rstd = rsqrtf(q[nb] * (1.f / D) + eps);
                r[nb][mb][0] = fmaf((r[nb][mb][0] - s[nb]) * rstd, gv.x, bv.x);
                r[nb][mb][1] = fmaf((r[nb][mb][1] - s[nb]) * rstd, gv.y, bv.y);
                r[nb][mb][2] = fmaf((r[nb][mb][2] - s[nb]) * rstd, gv.z, bv.z);
                r[nb][mb][3] = fmaf((r[nb][mb][3] - s[nb]) * rstd, gv.w, bv.w);
            }
        }
    };
    if (p.n1_g) layer_norm(p.n1_g, p.n1_b, p.n1_eps);
    if (VAR == 26 && p.x_out) {                                   // timing variant: 16-byte pieces of 16 rows per store instruction
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int tok = t0 + th * 64 + nb * 16 + l15;
            if (tok < M) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
                    *reinterpret_cast<float4 *>(p.x_out + (int64_t)tok * D + f0 + mb * 16) =
                        make_float4(r[nb][mb][0], r[nb][mb][1], r[nb][mb][2], r[nb][mb][3]);
            }
        }
    } else if (p.x_out) {
        // The stream rows leave through LDS (both token tiles are dead: 64 rows x 1040 bytes over xn + hc): a lane's
        // registers are 16-byte pieces of 16 different rows per store instruction; staged, a wave instruction writes one
        // whole 1 KB row (the projection epilogue's output went from 8-byte pieces to row segments the same way:
        // 152-173 -> 136-156 us for FFN + in_proj, tools/bench_ffn_proj.py).
        constexpr int SS = 260;                                   // floats per staged row: 16-byte lane writes of 8 rows hit 32 banks
        float *stg = reinterpret_cast<float *>(smem);
        lds_barrier();                                            // every wave is done with the last GEMM's fragments
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
                *reinterpret_cast<float4 *>(stg + (th * 64 + nb * 16 + l15) * SS + f0 + mb * 16) =
                    make_float4(r[nb][mb][0], r[nb][mb][1], r[nb][mb][2], r[nb][mb][3]);
        lds_barrier();
#pragma unroll
        for (int i = 0; i < TOK * 64 / NT; ++i) {
            const int idx = tid + NT * i, row = idx >> 6, chunk = idx & 63;
            const float4 v = *reinterpret_cast<const float4 *>(stg + row * SS + chunk * 4);
            if (t0 + row < M) *reinterpret_cast<float4 *>(p.x_out + (int64_t)(t0 + row) * D + chunk * 4) = v;
        }
    }
    if constexpr (PROJ) {
        // ---- projection epilogue.  h = LN2(r) goes to LDS in bf16 (the token tile of the GEMMs above is dead), then
        // 256-wide output slabs: wave = 64 features x 64 tokens, K = 256, weights through the same ring
        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.proj_w), 0, p.proj_dim * D * 2, 0x00020000);
        auto pload = [&](int ps, int s, bf16x8(&dst)[4]) {       // band (ps*256 + wave*64)/16 + mb, k-tile s
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
                dst[mb] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                         rp, vl, (((ps * 256 + wave * 64) / 16 + mb) * (D / 32) + s) * 1024, 0));
        };
#pragma unroll
        for (int s = 0; s < PFK; ++s) pload(0, s, wq[s]);          // in flight under the LayerNorm below
        if (p.n2_g) layer_norm(p.n2_g, p.n2_b, p.n2_eps);
        lds_barrier();                                            // (layer_norm's own barriers already order the last GEMM's reads)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                uint2 pk;
                pk.x = pack2(r[nb][mb][0], r[nb][mb][1]);
                pk.y = pack2(r[nb][mb][2], r[nb][mb][3]);
                *reinterpret_cast<uint2 *>(xn + (th * 64 + nb * 16 + l15) * XS + f0 + mb * 16) = pk;
            }
        lds_barrier();
        uint16_t *po = reinterpret_cast<uint16_t *>(p.proj_out);
        const int nps = p.proj_dim / 256;
        for (int ps = 0; ps < nps; ++ps) {
            f32x4 acc[4][4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
            bf16x8 bfa[4], bfb[4];
            read_frags(xfrag, 0, bfa);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                bf16x8(&cur)[4] = (s & 1) ? bfb : bfa;
                bf16x8(&nxt)[4] = (s & 1) ? bfa : bfb;
                if (s + 1 < 8) read_frags(xfrag, s + 1, nxt);
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[s % PFK][mb], cur[nb], acc[mb][nb], 0, 0, 0);
                if (s + PFK < 8) pload(ps, s + PFK, wq[s % PFK]);
                else if (ps + 1 < nps) pload(ps + 1, s + PFK - 8, wq[s % PFK]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // bias, bf16, and out through LDS: a lane's accumulators are 4 features of 16 different tokens (8-byte pieces of
            // 16 rows per store instruction); staged in the free hidden-slab tile they leave as whole 512-byte row segments
            const int fcol = ps * 256 + f0;
            if (ps > 0) lds_barrier();                            // the previous slab's rows have been read out
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.proj_b) bv = *reinterpret_cast<const float4 *>(p.proj_b + fcol + mb * 16);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    uint2 pk;
                    pk.x = pack2(acc[mb][nb][0] + bv.x, acc[mb][nb][1] + bv.y);
                    pk.y = pack2(acc[mb][nb][2] + bv.z, acc[mb][nb][3] + bv.w);
                    *reinterpret_cast<uint2 *>(hdst + nb * 16 * XS + mb * 16) = pk;
                }
            }
            lds_barrier();
#pragma unroll
            for (int i = 0; i < TOK * 32 / NT; ++i) {
                const int idx = tid + NT * i, row = idx >> 5, chunk = idx & 31;
                const uint4 v = *reinterpret_cast<const uint4 *>(hc + row * XS + chunk * 8);
                if (t0 + row < M) *reinterpret_cast<uint4 *>(po + (int64_t)(t0 + row) * p.proj_dim + ps * 256 + chunk * 8) = v;
            }
        }
        return;
    }
    if (p.h_out) {
        if (p.n2_g) layer_norm(p.n2_g, p.n2_b, p.n2_eps);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const int tok = t0 + th * 64 + nb * 16 + l15;
            if (tok < M) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    if (p.h_dtype == CM_F32) {
                        *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p.h_out) + (int64_t)tok * D + f0 + mb * 16) =
                            make_float4(r[nb][mb][0], r[nb][mb][1], r[nb][mb][2], r[nb][mb][3]);
                    } else {
                        uint2 pk;
                        pk.x = pack2(r[nb][mb][0], r[nb][mb][1]);
                        pk.y = pack2(r[nb][mb][2], r[nb][mb][3]);
                        *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(p.h_out) + (int64_t)tok * D + f0 + mb * 16) = pk;
                    }
                }
            }
        }
    }
}

// row-major (R, K) bf16 -> fragment-tiled image (see ffn_fused_kernel).  One thread per 16-byte piece.
__global__ void ffn_pack_kernel(const uint16_t *__restrict__ w, uint16_t *__restrict__ out, int R, int K) {
    const int64_t piece = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // = tile * 64 + lane
    const int64_t npieces = (int64_t)R * K / 8;
    if (piece >= npieces) return;
    const int lane = (int)(piece & 63);
    const int64_t tile = piece >> 6;
    const int kt = K / 32;
    const int rb = (int)(tile / kt), kb = (int)(tile % kt);
    const int r = rb * 16 + (lane & 15), k = kb * 32 + (lane >> 4) * 8;
    *reinterpret_cast<uint4 *>(out + piece * 8) = *reinterpret_cast<const uint4 *>(w + (int64_t)r * K + k);
}

template <bool ADD, int VAR, bool PROJ = false, bool TRAIN = false>
int launch_var(const cm_ffn_args &a) {
    const size_t smem = (size_t)2 * TOK * XS * sizeof(uint16_t) + (size_t)4 * TOK * sizeof(float) + (size_t)a.hidden * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&ffn_fused_kernel<ADD, VAR, PROJ, TRAIN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            cm_set_error("ffn_fused: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return (int)e;
        }
        attr_done = true;
    }
    dim3 grid((a.rows + TOK - 1) / TOK);
    hipLaunchKernelGGL((ffn_fused_kernel<ADD, VAR, PROJ, TRAIN>), grid, dim3(NT), smem, reinterpret_cast<hipStream_t>(a.stream), a);
    return cm_launch_status("cm_ffn_fused");
}

template <bool ADD>
int launch(const cm_ffn_args &a) {
#ifdef CM_ABLATE
    if (a.proj_w) return cm_debug_get() == 26 ? launch_var<ADD, 26, true>(a) : (cm_debug_get() == 25 ? launch_var<ADD, 25, true>(a) : launch_var<ADD, 0, true>(a));
    switch (cm_debug_get()) {
        case 1: return launch_var<ADD, 1>(a);
        case 2: return launch_var<ADD, 2>(a);
        case 3: return launch_var<ADD, 3>(a);
        case 4: return launch_var<ADD, 4>(a);
        case 5: return launch_var<ADD, 5>(a);
        case 25: return launch_var<ADD, 25>(a);
        case 26: return launch_var<ADD, 26>(a);
        default: return launch_var<ADD, 0>(a);
    }
#else
    return a.proj_w ? launch_var<ADD, 0, true>(a) : launch_var<ADD, 0>(a);
#endif
}

}  // namespace

int cm_ffn_fused32_launch(const cm_ffn_args &a);                  // ffn_fused32.hip

extern "C" int cm_ffn_pack_weights(const void *w, int32_t rows, int32_t cols, void *out, void *stream) {
    CM_REQUIRE(w && out && rows > 0 && cols > 0, CM_EINVAL, "ffn_pack_weights: bad sizes or NULL tensor");
    CM_REQUIRE(rows % 16 == 0 && cols % 32 == 0, CM_EUNSUPPORTED, "ffn_pack_weights: needs rows %% 16 == 0 and cols %% 32 == 0");
    CM_REQUIRE(cm_aligned(w, 16) && cm_aligned(out, 16) && w != out, CM_EALIGN, "ffn_pack_weights: tensors must be distinct and 16-byte aligned");
    const int64_t npieces = (int64_t)rows * cols / 8;
    hipLaunchKernelGGL(ffn_pack_kernel, dim3((unsigned)((npieces + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const uint16_t *>(w), reinterpret_cast<uint16_t *>(out), rows, cols);
    return cm_launch_status("cm_ffn_pack_weights");
}

extern "C" int cm_ffn_fused(const cm_ffn_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "ffn_fused: args is NULL");
    const cm_ffn_args &a = *args;
    CM_REQUIRE(a.rows > 0 && a.x && a.w1 && a.b1 && a.w2 && a.b2 && a.pre_g && a.pre_b, CM_EINVAL,
               "ffn_fused: bad sizes or NULL tensor");
    CM_REQUIRE(a.dim == D, CM_EUNSUPPORTED, "ffn_fused: d_model must be 256 (got %d)", a.dim);
    CM_REQUIRE(a.hidden >= CH && a.hidden % CH == 0 && a.hidden <= 2048, CM_EUNSUPPORTED,
               "ffn_fused: hidden must be a multiple of 256, at most 2048 (got %d)", a.hidden);
    CM_REQUIRE((!a.n1_g || a.n1_b) && (!a.n2_g || a.n2_b), CM_EINVAL, "ffn_fused: LayerNorm weight without bias");
    CM_REQUIRE(a.x_out || a.h_out || a.proj_w, CM_EINVAL, "ffn_fused: neither x_out nor h_out nor a projection given");
    CM_REQUIRE(!a.proj_w || (a.proj_out && !a.h_out && a.proj_dim >= 256 && a.proj_dim % 256 == 0 && a.proj_dim <= 4096 &&
                              cm_aligned(a.proj_w, 16) && cm_aligned(a.proj_out, 16) && (!a.proj_b || cm_aligned(a.proj_b, 16))),
               CM_EINVAL, "ffn_fused: the projection needs proj_out, no h_out, proj_dim a multiple of 256 (<= 4096), aligned tensors");
    CM_REQUIRE(!a.h_out || a.h_dtype == CM_F32 || a.h_dtype == CM_BF16, CM_EINVAL, "ffn_fused: h_dtype must be f32 or bf16");
    CM_REQUIRE(cm_aligned(a.x, 16) && cm_aligned(a.w1, 16) && cm_aligned(a.w2, 16) && cm_aligned(a.b1, 16) && cm_aligned(a.b2, 16) &&
                   cm_aligned(a.pre_g, 16) && cm_aligned(a.pre_b, 16) && (!a.x_out || cm_aligned(a.x_out, 16)) &&
                   (!a.h_out || cm_aligned(a.h_out, 16)) && (!a.addend || cm_aligned(a.addend, 8)) &&
                   (!a.n1_g || (cm_aligned(a.n1_g, 16) && cm_aligned(a.n1_b, 16))) &&
                   (!a.n2_g || (cm_aligned(a.n2_g, 16) && cm_aligned(a.n2_b, 16))),
               CM_EALIGN, "ffn_fused: tensors must be 16-byte aligned");
    const bool train = a.pre_out || a.xn_out || a.stats_out || a.p1 > 0.f || a.p2 > 0.f;
    CM_REQUIRE(a.layout == 0 || a.layout == 1, CM_EINVAL, "ffn_fused: layout must be 0 or 1");
    CM_REQUIRE(a.layout == 0 || !train, CM_EUNSUPPORTED, "ffn_fused: the training forward takes layout 0 weights");
    CM_REQUIRE(a.tokens == 0 || a.tokens == 64 || (a.tokens == 32 && a.layout == 1), CM_EINVAL, "ffn_fused: tokens must be 0 / 64, or 32 with layout 1");
    if (a.layout == 1) return cm_ffn_fused32_launch(a);
    if (train) {
        CM_REQUIRE(a.x_out && !a.addend && !a.n1_g && !a.n2_g && !a.proj_w && !a.h_out, CM_EINVAL,
                   "ffn_fused: the training forward writes x_out (+ pre_out, xn_out) only: no addend, n1, n2, projection or h_out");
        CM_REQUIRE(a.p1 >= 0.f && a.p1 < 1.f && a.p2 >= 0.f && a.p2 < 1.f, CM_EINVAL, "ffn_fused: dropout probabilities must be in [0, 1)");
        CM_REQUIRE((!a.pre_out || cm_aligned(a.pre_out, 16)) && (!a.xn_out || cm_aligned(a.xn_out, 16)), CM_EALIGN,
                   "ffn_fused: pre_out / xn_out must be 16-byte aligned");
        CM_REQUIRE((int64_t)a.rows * a.hidden * 2 < ((int64_t)1 << 32), CM_EUNSUPPORTED, "ffn_fused: rows x hidden too large for the training forward's 32-bit offsets");
        return launch_var<false, 0, false, true>(a);
    }
    return a.addend ? launch<true>(a) : launch<false>(a);
}
