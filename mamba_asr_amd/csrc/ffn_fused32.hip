// ffn_fused32.hip — cm_ffn_fused on v_mfma_f32_32x32x16_bf16 (cm_ffn_args.layout = 1; weights in cm_ffn_pack_weights32's image).
// Same contract, tiling and phases as ffn_fused.hip (64 tokens per workgroup, 4 waves x 64 features, hidden slabs of 256 kept in LDS,
// weight fragments straight from their packed image in L2 through a register ring); what changes is the matrix instruction:
//   * a 32x32x16 MFMA holds the SIMD's vector issue port for 8 of its 32 cycles, the 16x16x32 form for 8 of its 16
//     (MI355X_MICROARCH.md, cycle constants): per 64-token tile and SIMD the matrix work costs 4.1 k issue cycles instead of 8.2 k of a
//     23 k-cycle issue budget that the GELU / LayerNorm / epilogue arithmetic of the partner wave shares;
//   * its token fragment (lane = token % 32, 8 consecutive k of 16 chosen by lane / 32) reads the 528-byte LDS rows without bank
//     conflicts: the 16 lanes a ds_read_b128 serves together all have the same k half, 16 different rows = 16 different 4-bank groups
//     (the 16x16x32 fragment mixes two k blocks per group: 2-way, SQ_LDS_BANK_CONFLICT 41-43 % of the kernel's LDS cycles).
// Accumulator layout (32 x 32 tile, 16 registers): lane holds token n = lane % 32 and features 8 (r / 4) + 4 (lane / 32) + r % 4:
// four groups of 4 consecutive features -> the same 8-byte LDS pieces and 16-byte staging pieces as before.
#include "cm_common.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int D = 256, NT = 256, XS = 264, CH = 256;
constexpr int PF = 8;         // weight-fragment ring depth in k16-steps (= the 4 k32-steps of the 16x16x32 kernel)

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ uint32_t pack2(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2)); }

// TOK: tokens per workgroup, 64 or 32.  32 (cm_ffn_args.tokens = 32) is for launches that leave CUs idle at 64: 16 k rows are 250
// workgroups on 512 slots (SURVEY's 16 x 40 s; per part of the two-stream encoder half of that), each streaming the same 1 MB of weights
// as a 64-token one -- twice the weight stream per token, bought back by a chip that is full.
template <bool ADD, bool PROJ, int TOK>
__global__ __launch_bounds__(NT, 2) void ffn_fused32_kernel(const cm_ffn_args p) {
    constexpr int NB = TOK / 32;                                  // 32-token tiles per workgroup
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *xn = reinterpret_cast<uint16_t *>(smem);            // [TOK][XS] normalised tokens
    uint16_t *hc = xn + TOK * XS;                                 // [TOK][XS] hidden slab
    float *red = reinterpret_cast<float *>(hc + TOK * XS);        // [4][TOK] LayerNorm partial sums
    float *b1s = red + 4 * TOK;                                   // [hidden] first bias
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;                    // phase 0 (row of 16 lanes per token)
    const int l31 = lane & 31, h = lane >> 5;                     // matrix phases: token within a 32-tile, k half / feature half
    const int t0 = blockIdx.x * TOK, M = p.rows, F = p.hidden;
    const int nch = F / CH;
    const uint16_t *addend = reinterpret_cast<const uint16_t *>(p.addend);

    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w1), 0, F * D * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w2), 0, F * D * 2, 0x00020000);
    const int vl = lane * 16;
    const int kt2 = F / 16;                                       // k16-tiles per 32-row band of W2
    // step s of slab c: s < 16 -> W1 band (c*CH + wave*64) / 32 + mb, k16-tile s;  s >= 16 -> W2 band wave*2 + mb, k16-tile c*16 + s - 16.
    // Packed image (cm_ffn_pack_weights32): 32-row x 16-column tiles of 1 KB, lane L owns row L % 32, columns 8 (L / 32) .. + 7.
    auto wload = [&](int c, int s, bf16x8(&dst)[2]) {
        if (s < 16) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
                dst[mb] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r1, vl, (((c * CH + wave * 64) / 32 + mb) * (D / 16) + s) * 1024, 0));
        } else {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
                dst[mb] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r2, vl, ((wave * 2 + mb) * kt2 + c * (CH / 16) + (s - 16)) * 1024, 0));
        }
    };
    bf16x8 wq[PF][2];
#pragma unroll
    for (int s = 0; s < PF; ++s) wload(0, s, wq[s]);

    for (int i = tid; i < p.hidden; i += NT) b1s[i] = p.b1[i];

    // ---- phase 0: xin = x (+ add_scale * addend); xn = LayerNorm_pre(xin) in bf16 (as ffn_fused.hip: a row of 16 lanes per token)
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
        constexpr int RD = TOK / 16;                              // rounds of 4 tokens per wave (wave w owns tokens (TOK / 4) w ..)
        float4 v[RD][4];
#pragma unroll
        for (int rd = 0; rd < RD; ++rd) {
            const int tok = min(t0 + wave * (TOK / 4) + rd * 4 + lq, M - 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[rd][i] = load_x4(tok, (l15 + 16 * i) * 4);
        }
#pragma unroll
        for (int rd = 0; rd < RD; ++rd) {
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
            uint16_t *dst = xn + (wave * (TOK / 4) + rd * 4 + lq) * XS;
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

    // ---- main loop over hidden slabs
    f32x16 acc2[2][NB];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[mb][nb][e] = 0.f;
    const uint16_t *xfrag = xn + l31 * XS + h * 8;                // + nb * 32 rows, + 16 columns per k16-step
    const uint16_t *hfrag = hc + l31 * XS + h * 8;
    const int f0 = wave * 64 + 4 * h;                             // + 32 mb + 8 g + i
    uint16_t *hdst = hc + l31 * XS + f0;
    auto read_frags = [&](const uint16_t *base, int ks, bf16x8(&bf)[NB]) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bf[nb] = *reinterpret_cast<const bf16x8 *>(base + nb * 32 * XS + ks * 16);
    };
    auto slab = [&](const int c, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        f32x16 acc1[2][NB];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc1[mb][nb][e] = 0.f;
        bf16x8 bfa[NB], bfb[NB];
        read_frags(xfrag, 0, bfa);
#pragma unroll
        for (int s = 0; s < 16; ++s) {                               // GEMM 1: this wave's 64 hidden units x 64 tokens, K = 256
            bf16x8(&cur)[NB] = (s & 1) ? bfb : bfa;
            bf16x8(&nxt)[NB] = (s & 1) ? bfa : bfb;
            if (s + 1 < 16) read_frags(xfrag, s + 1, nxt);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc1[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[s % PF][mb], cur[nb], acc1[mb][nb], 0, 0, 0);
            wload(c, s + PF, wq[s % PF]);
            __builtin_amdgcn_sched_barrier(0);                    // keep the refill HERE (see ffn_fused.hip)
        }
        if (c > 0) lds_barrier();                                 // every wave is done reading the previous slab
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4 *>(b1s + c * CH + f0 + mb * 32 + g * 8);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    uint2 pk;
                    pk.x = cm_gelu_bf16_pack2(acc1[mb][nb][4 * g + 0] + bv.x, acc1[mb][nb][4 * g + 1] + bv.y);
                    pk.y = cm_gelu_bf16_pack2(acc1[mb][nb][4 * g + 2] + bv.z, acc1[mb][nb][4 * g + 3] + bv.w);
                    *reinterpret_cast<uint2 *>(hdst + nb * 32 * XS + mb * 32 + g * 8) = pk;
                }
            }
        lds_barrier();
        read_frags(hfrag, 0, bfa);
#pragma unroll
        for (int s = 16; s < 32; ++s) {                              // GEMM 2: 64 output features x 64 tokens, K = this slab
            bf16x8(&cur)[NB] = (s & 1) ? bfb : bfa;
            bf16x8(&nxt)[NB] = (s & 1) ? bfa : bfb;
            if (s + 1 < 32) read_frags(hfrag, s + 1 - 16, nxt);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc2[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[s % PF][mb], cur[nb], acc2[mb][nb], 0, 0, 0);
            if (s + PF < 32) wload(c, s + PF, wq[s % PF]);
            else if constexpr (!LAST) wload(c + 1, s + PF - 32, wq[s % PF]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int c = 0; c + 1 < nch; ++c) slab(c, std::false_type{});
    slab(nch - 1, std::true_type{});

    // ---- residual rows (whole rows through LDS into the accumulator layout, requested after the last GEMM: ffn_fused.hip)
    float r[NB][2][16];                                            // [token tile][feature band][8 g + ... register order of the accumulator]
    constexpr int SR = 260;
    float *stg = reinterpret_cast<float *>(smem);
    {
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
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 xv = *reinterpret_cast<const float4 *>(stg + (nb * 32 + l31) * SR + f0 + mb * 32 + g * 8);
                    r[nb][mb][4 * g + 0] = xv.x; r[nb][mb][4 * g + 1] = xv.y; r[nb][mb][4 * g + 2] = xv.z; r[nb][mb][4 * g + 3] = xv.w;
                }
    }
    // ---- epilogue: r = xin + alpha (acc2 + b2); optional LN1 -> stream; optional LN2 -> h_out / projection
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bv = *reinterpret_cast<const float4 *>(p.b2 + f0 + mb * 32 + g * 8);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                r[nb][mb][4 * g + 0] = fmaf(p.alpha, acc2[mb][nb][4 * g + 0] + bv.x, r[nb][mb][4 * g + 0]);
                r[nb][mb][4 * g + 1] = fmaf(p.alpha, acc2[mb][nb][4 * g + 1] + bv.y, r[nb][mb][4 * g + 1]);
                r[nb][mb][4 * g + 2] = fmaf(p.alpha, acc2[mb][nb][4 * g + 2] + bv.z, r[nb][mb][4 * g + 2]);
                r[nb][mb][4 * g + 3] = fmaf(p.alpha, acc2[mb][nb][4 * g + 3] + bv.w, r[nb][mb][4 * g + 3]);
            }
        }
    // sum over a token's 256 features: 32 in-lane, the two feature halves (lane ^ 32), 4 waves (through LDS)
    auto token_sums = [&](float (&v)[NB]) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) v[nb] += __shfl_xor(v[nb], 32, 64);
        lds_barrier();                                            // previous use of red is over
        if (h == 0) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) red[wave * TOK + nb * 32 + l31] = v[nb];
        }
        lds_barrier();
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int t = nb * 32 + l31;
            v[nb] = (red[t] + red[TOK + t]) + (red[2 * TOK + t] + red[3 * TOK + t]);
        }
    };
    auto layer_norm = [&](const float *gw, const float *bw, float eps) {
        float s[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            s[nb] = 0.f;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int e = 0; e < 16; e += 4) s[nb] += (r[nb][mb][e] + r[nb][mb][e + 1]) + (r[nb][mb][e + 2] + r[nb][mb][e + 3]);
        }
        token_sums(s);
        float q[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            s[nb] *= (1.f / D);
            q[nb] = 0.f;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int e = 0; e < 16; ++e) { const float d = r[nb][mb][e] - s[nb]; q[nb] = fmaf(d, d, q[nb]); }
        }
        token_sums(q);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 gv = *reinterpret_cast<const float4 *>(gw + f0 + mb * 32 + g * 8);
                const float4 bv = *reinterpret_cast<const float4 *>(bw + f0 + mb * 32 + g * 8);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const float rstd = rsqrtf(q[nb] * (1.f / D) + eps);
                    r[nb][mb][4 * g + 0] = fmaf((r[nb][mb][4 * g + 0] - s[nb]) * rstd, gv.x, bv.x);
                    r[nb][mb][4 * g + 1] = fmaf((r[nb][mb][4 * g + 1] - s[nb]) * rstd, gv.y, bv.y);
                    r[nb][mb][4 * g + 2] = fmaf((r[nb][mb][4 * g + 2] - s[nb]) * rstd, gv.z, bv.z);
                    r[nb][mb][4 * g + 3] = fmaf((r[nb][mb][4 * g + 3] - s[nb]) * rstd, gv.w, bv.w);
                }
            }
    };
    if (p.n1_g) layer_norm(p.n1_g, p.n1_b, p.n1_eps);
    if (p.x_out) {
        constexpr int SS = 260;
        lds_barrier();                                            // (also: the staged residual rows above have been read by every wave)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4 *>(stg + (nb * 32 + l31) * SS + f0 + mb * 32 + g * 8) =
                        make_float4(r[nb][mb][4 * g + 0], r[nb][mb][4 * g + 1], r[nb][mb][4 * g + 2], r[nb][mb][4 * g + 3]);
        lds_barrier();
#pragma unroll
        for (int i = 0; i < TOK * 64 / NT; ++i) {
            const int idx = tid + NT * i, row = idx >> 6, chunk = idx & 63;
            const float4 v = *reinterpret_cast<const float4 *>(stg + row * SS + chunk * 4);
            if (t0 + row < M) *reinterpret_cast<float4 *>(p.x_out + (int64_t)(t0 + row) * D + chunk * 4) = v;
        }
    }
    if constexpr (PROJ) {
        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.proj_w), 0, p.proj_dim * D * 2, 0x00020000);
        auto pload = [&](int ps, int s, bf16x8(&dst)[2]) {       // band (ps*256 + wave*64) / 32 + mb, k16-tile s
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
                dst[mb] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rp, vl, (((ps * 256 + wave * 64) / 32 + mb) * (D / 16) + s) * 1024, 0));
        };
#pragma unroll
        for (int s = 0; s < PF; ++s) pload(0, s, wq[s]);          // in flight under the LayerNorm below
        if (p.n2_g) layer_norm(p.n2_g, p.n2_b, p.n2_eps);
        lds_barrier();                                            // the staged stream rows have left / the last GEMM's reads are over
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 pk;
                    pk.x = pack2(r[nb][mb][4 * g + 0], r[nb][mb][4 * g + 1]);
                    pk.y = pack2(r[nb][mb][4 * g + 2], r[nb][mb][4 * g + 3]);
                    *reinterpret_cast<uint2 *>(xn + (nb * 32 + l31) * XS + f0 + mb * 32 + g * 8) = pk;
                }
        lds_barrier();
        uint16_t *po = reinterpret_cast<uint16_t *>(p.proj_out);
        const int nps = p.proj_dim / 256;
        for (int ps = 0; ps < nps; ++ps) {
            f32x16 acc[2][NB];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[mb][nb][e] = 0.f;
            bf16x8 bfa[NB], bfb[NB];
            read_frags(xfrag, 0, bfa);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                bf16x8(&cur)[NB] = (s & 1) ? bfb : bfa;
                bf16x8(&nxt)[NB] = (s & 1) ? bfa : bfb;
                if (s + 1 < 16) read_frags(xfrag, s + 1, nxt);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[s % PF][mb], cur[nb], acc[mb][nb], 0, 0, 0);
                if (s + PF < 16) pload(ps, s + PF, wq[s % PF]);
                else if (ps + 1 < nps) pload(ps + 1, s + PF - 16, wq[s % PF]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ps > 0) lds_barrier();                            // the previous slab's rows have been read out
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (p.proj_b) bv = *reinterpret_cast<const float4 *>(p.proj_b + ps * 256 + f0 + mb * 32 + g * 8);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        uint2 pk;
                        pk.x = pack2(acc[mb][nb][4 * g + 0] + bv.x, acc[mb][nb][4 * g + 1] + bv.y);
                        pk.y = pack2(acc[mb][nb][4 * g + 2] + bv.z, acc[mb][nb][4 * g + 3] + bv.w);
                        *reinterpret_cast<uint2 *>(hdst + nb * 32 * XS + mb * 32 + g * 8) = pk;
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
        for (int nb = 0; nb < NB; ++nb) {
            const int tok = t0 + nb * 32 + l31;
            if (tok < M) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int f = f0 + mb * 32 + g * 8;
                        if (p.h_dtype == CM_F32) {
                            *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p.h_out) + (int64_t)tok * D + f) =
                                make_float4(r[nb][mb][4 * g + 0], r[nb][mb][4 * g + 1], r[nb][mb][4 * g + 2], r[nb][mb][4 * g + 3]);
                        } else {
                            uint2 pk;
                            pk.x = pack2(r[nb][mb][4 * g + 0], r[nb][mb][4 * g + 1]);
                            pk.y = pack2(r[nb][mb][4 * g + 2], r[nb][mb][4 * g + 3]);
                            *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(p.h_out) + (int64_t)tok * D + f) = pk;
                        }
                    }
            }
        }
    }
}

// row-major (R, K) bf16 -> 32-row x 16-column fragment tiles (lane L of a tile: row L % 32, columns 8 (L / 32) .. + 7); one thread per piece
__global__ void ffn_pack32_kernel(const uint16_t *__restrict__ w, uint16_t *__restrict__ out, int R, int K) {
    const int64_t piece = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // = tile * 64 + lane
    if (piece >= (int64_t)R * K / 8) return;
    const int lane = (int)(piece & 63);
    const int64_t tile = piece >> 6;
    const int kt = K / 16;
    const int rb = (int)(tile / kt), kb = (int)(tile % kt);
    const int r = rb * 32 + (lane & 31), k = kb * 16 + (lane >> 5) * 8;
    *reinterpret_cast<uint4 *>(out + piece * 8) = *reinterpret_cast<const uint4 *>(w + (int64_t)r * K + k);
}

template <bool ADD, bool PROJ, int TOK>
int launch32(const cm_ffn_args &a) {
    const size_t smem = (size_t)2 * TOK * XS * sizeof(uint16_t) + (size_t)4 * TOK * sizeof(float) + (size_t)a.hidden * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&ffn_fused32_kernel<ADD, PROJ, TOK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            cm_set_error("ffn_fused: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return (int)e;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL((ffn_fused32_kernel<ADD, PROJ, TOK>), dim3((a.rows + TOK - 1) / TOK), dim3(NT), smem, reinterpret_cast<hipStream_t>(a.stream), a);
    return cm_launch_status("cm_ffn_fused(32x32x16)");
}

}  // namespace

extern "C" int cm_ffn_pack_weights32(const void *w, int32_t rows, int32_t cols, void *out, void *stream) {
    CM_REQUIRE(w && out && rows > 0 && cols > 0, CM_EINVAL, "ffn_pack_weights32: bad sizes or NULL tensor");
    CM_REQUIRE(rows % 32 == 0 && cols % 16 == 0, CM_EUNSUPPORTED, "ffn_pack_weights32: needs rows %% 32 == 0 and cols %% 16 == 0");
    CM_REQUIRE(cm_aligned(w, 16) && cm_aligned(out, 16) && w != out, CM_EALIGN, "ffn_pack_weights32: tensors must be distinct and 16-byte aligned");
    const int64_t npieces = (int64_t)rows * cols / 8;
    hipLaunchKernelGGL(ffn_pack32_kernel, dim3((unsigned)((npieces + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const uint16_t *>(w), reinterpret_cast<uint16_t *>(out), rows, cols);
    return cm_launch_status("cm_ffn_pack_weights32");
}

// called by cm_ffn_fused (ffn_fused.hip) after its argument checks when args->layout == 1
int cm_ffn_fused32_launch(const cm_ffn_args &a) {
    if (a.tokens == 32) {
        if (a.proj_w) return a.addend ? launch32<true, true, 32>(a) : launch32<false, true, 32>(a);
        return a.addend ? launch32<true, false, 32>(a) : launch32<false, false, 32>(a);
    }
    if (a.proj_w) return a.addend ? launch32<true, true, 64>(a) : launch32<false, true, 64>(a);
    return a.addend ? launch32<true, false, 64>(a) : launch32<false, false, 64>(a);
}
