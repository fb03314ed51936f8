// ffn_bwd_fused.hip — the data-gradient chain of a feed-forward module's backward as ONE kernel (contract: cm_ffn_bwd_fused in
// include/conmamba_hip.h; the reference leaves it to autograd over modules/Conmamba.py:597-617: two Dropout backwards, two addmm
// backwards, a GELU backward, with the (rows, hidden) gradients round-tripping through HBM between them).
//
// It is cm_ffn_fused's shape run with the transposed weights -- 256 -> hidden -> 256 per token, the hidden tile never leaves the CU:
//     da2 = alpha * dout * keep2 / (1 - p2)                       (fp32 residual-stream gradient -> bf16; stored for dW2, db2)
//     dg  = da2 @ W2                                              (GEMM 1: "weights" = W2^T, (hidden, 256) packed)
//     da1 = dg * keep1 / (1 - p1) * GELU'(pre)                    (pre = the forward's stored bf16 pre-activation; stored for dW1, db1)
//     act = dropout1(GELU(pre))                                   (recomputed: what the forward's second GEMM saw; stored for dW2)
//     dh  = da1 @ W1                                              (GEMM 2: "weights" = W1^T, (256, hidden) packed; bf16 out)
// plus the two bias gradients as per-workgroup partial column sums (folded in a fixed order by ffn_bwd_colsum_kernel).
// Separately this was: element-wise (13 us) + GEMM (28) + element-wise (60) + 2 column-sum folds (12) + GEMM (28) per module at
// 32 k rows.  The LayerNorm backward in front of the module (dx = dout + LN'(dh)) stays cm_layernorm_bwd.
// Tiling, weight ring, fragment layouts: as ffn_fused.hip (64 tokens per workgroup, 4 waves x 64 features, v_mfma_f32_16x16x32_bf16
// with the weights as the A operand from their packed image, 2-deep ring).
#include "cm_common.h"
#include "cm_dropout.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int D = 256, TOK = 64, NT = 256, XS = 264, CH = 256, PF = 2;

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ uint32_t pack2(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2)); }

__global__ __launch_bounds__(NT, 2) void ffn_bwd_kernel(const cm_ffn_bwd_args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *xn = reinterpret_cast<uint16_t *>(smem);            // [TOK][XS] da2 tile (GEMM 1's token operand)
    uint16_t *hc = xn + TOK * XS;                                 // [TOK][XS] hidden slab: dg, then da1 in place (GEMM 2's token operand)
    float *red = reinterpret_cast<float *>(hc + TOK * XS);        // [8][256] db1 partial sums of the slab's row groups
    const uint64_t seed1 = cm_drop_seed(p.seed1, p.seed_epoch), seed2 = cm_drop_seed(p.seed2, p.seed_epoch);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int t0 = blockIdx.x * TOK, M = p.rows, F = p.hidden;
    const int nch = F / CH;

    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w2t), 0, F * D * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w1t), 0, F * D * 2, 0x00020000);
    const int rowbytes = (int)((int64_t)M * F * 2);
    const __amdgpu_buffer_rsrc_t rpre = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.pre), 0, rowbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rda1 = __builtin_amdgcn_make_buffer_rsrc(p.da1, 0, rowbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ract = __builtin_amdgcn_make_buffer_rsrc(p.act, 0, rowbytes, 0x00020000);
    const int vl = lane * 16, kt2 = F / 32;
    // step s of slab c: s < 8 -> W2^T rows c*CH + wave*64 + mb*16 (k-tile s of 256 columns);  s >= 8 -> W1^T rows wave*64 + mb*16, k-tile c*8 + s - 8
    auto wload = [&](int c, int s, bf16x8(&dst)[4]) {
        if (s < 8) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
                dst[mb] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r1, vl, (((c * CH + wave * 64) / 16 + mb) * (D / 32) + s) * 1024, 0));
        } else {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
                dst[mb] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r2, vl, ((wave * 4 + mb) * kt2 + c * (CH / 32) + (s - 8)) * 1024, 0));
        }
    };
    bf16x8 wq[PF][4];
#pragma unroll
    for (int s = 0; s < PF; ++s) wload(0, s, wq[s]);

    // ---- phase 0: da2 = alpha * dout * keep2 / (1 - p2) -> bf16 tile.  Wave w owns tokens 16 w .. 16 w + 15, four per round; a row of
    // 16 lanes holds one token, lane l15 columns (l15 + 16 i) * 4 .. + 3
    {
        const bool drop = p.p2 > 0.f;
        const uint32_t th2 = cm_drop_thresh(p.p2);
        const float sc = p.alpha * (drop ? cm_drop_scale(p.p2) : 1.f);
#pragma unroll
        for (int rd = 0; rd < 4; ++rd) {
            const int trow = t0 + wave * 16 + rd * 4 + lq, tok = min(trow, M - 1);
            float4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float4 *>(p.dout + (int64_t)tok * D + (l15 + 16 * i) * 4);
            uint16_t *dst = xn + (wave * 16 + rd * 4 + lq) * XS;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = (l15 + 16 * i) * 4;
                const int64_t e0 = (int64_t)trow * D + col;
                const uint32_t keep = drop ? cm_drop_keep4(seed2, (uint64_t)e0 >> 3, (int)((e0 >> 2) & 1), th2) : 0xfu;
                const bool live = trow < M;                          // rows past the end: zero gradient (they would enter db2 / dg otherwise)
                uint2 pk;
                pk.x = pack2((live && (keep & 1u)) ? v[i].x * sc : 0.f, (live && (keep & 2u)) ? v[i].y * sc : 0.f);
                pk.y = pack2((live && (keep & 4u)) ? v[i].z * sc : 0.f, (live && (keep & 8u)) ? v[i].w * sc : 0.f);
                *reinterpret_cast<uint2 *>(dst + col) = pk;
            }
        }
    }
    lds_barrier();
    {
        // da2 out (whole rows) and its column sums over the tile's 64 tokens (what the weight-gradient GEMM sees: the rounded values)
        uint16_t *o = reinterpret_cast<uint16_t *>(p.da2);
#pragma unroll 2
        for (int i = 0; i < TOK * 32 / NT; ++i) {
            const int idx = tid + NT * i, row = idx >> 5, ch = idx & 31;
            const uint4 v = *reinterpret_cast<const uint4 *>(xn + row * XS + ch * 8);
            if (t0 + row < M) *reinterpret_cast<uint4 *>(o + (int64_t)(t0 + row) * D + ch * 8) = v;
        }
        float s = 0.f;
#pragma unroll 8
        for (int r = 0; r < TOK; ++r) s += __uint_as_float((uint32_t)xn[r * XS + tid] << 16);
        p.db2_part[(int64_t)blockIdx.x * (F + D) + tid] = s;      // a workgroup's partial row: [db1 (hidden) | db2 (256)]
    }

    f32x4 acc2[4][4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc2[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint16_t *xfrag = xn + l15 * XS + lq * 8;
    const uint16_t *hfrag = hc + l15 * XS + lq * 8;
    uint16_t *hdst = hc + l15 * XS + wave * 64 + lq * 4;
    auto read_frags = [&](const uint16_t *base, int ks, bf16x8(&bf)[4]) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) bf[nb] = *reinterpret_cast<const bf16x8 *>(base + nb * 16 * XS + ks * 32);
    };
    const bool drop1 = p.p1 > 0.f;
    const uint32_t th1 = cm_drop_thresh(p.p1);
    const float sc1 = drop1 ? cm_drop_scale(p.p1) : 1.f;

    for (int c = 0; c < nch; ++c) {
        f32x4 acc1[4][4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc1[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 bfa[4], bfb[4];
        read_frags(xfrag, 0, bfa);
#pragma unroll
        for (int s = 0; s < 8; ++s) {                                // GEMM 1: dg slab (this wave: 64 hidden units) x 64 tokens, K = 256
            bf16x8(&cur)[4] = (s & 1) ? bfb : bfa;
            bf16x8(&nxt)[4] = (s & 1) ? bfa : bfb;
            if (s + 1 < 8) read_frags(xfrag, s + 1, nxt);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc1[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[s % PF][mb], cur[nb], acc1[mb][nb], 0, 0, 0);
            wload(c, s + PF, wq[s % PF]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c > 0) lds_barrier();                                    // every wave is done reading the previous slab (and its red sums)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                uint2 pk;
                pk.x = pack2(acc1[mb][nb][0], acc1[mb][nb][1]);
                pk.y = pack2(acc1[mb][nb][2], acc1[mb][nb][3]);
                *reinterpret_cast<uint2 *>(hdst + nb * 16 * XS + mb * 16) = pk;          // dg, rounded as the library GEMM's bf16 output was
            }
        lds_barrier();
        // row-wise pass: piece i of this thread = row (tid >> 5) + 8 i, hidden columns 8 (tid & 31) .. + 7 of the slab
        {
            constexpr int NPC = TOK * (CH / 8) / NT, NPH = 4;
            const uint32_t el0 = (uint32_t)(t0 + (tid >> 5)) * (uint32_t)F + (uint32_t)(c * CH + (tid & 31) * 8);
            uint16_t *hrow = hc + (tid >> 5) * XS + (tid & 31) * 8;
            float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i0 = 0; i0 < NPC; i0 += NPH) {
                u32x4 pv[NPH];
                uint4 gv[NPH];
#pragma unroll
                for (int i = 0; i < NPH; ++i) {
                    pv[i] = __builtin_amdgcn_raw_buffer_load_b128(rpre, (el0 + (uint32_t)((i0 + i) * 8) * (uint32_t)F) * 2, 0, 0);   // rows past the end read 0
                    gv[i] = *reinterpret_cast<const uint4 *>(hrow + (i0 + i) * 8 * XS);
                }
#pragma unroll
                for (int i = 0; i < NPH; ++i) {
                    const uint32_t el = el0 + (uint32_t)((i0 + i) * 8) * (uint32_t)F;
                    const uint32_t keep = drop1 ? cm_drop_keep8(seed1, (uint64_t)(el >> 3), th1) : 0xffu;
                    const uint32_t pw[4] = {pv[i][0], pv[i][1], pv[i][2], pv[i][3]}, gw[4] = {gv[i].x, gv[i].y, gv[i].z, gv[i].w};
                    uint32_t ao[4], dv_[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t ka = (keep >> (2 * j)) & 1u, kb = (keep >> (2 * j + 1)) & 1u;
                        const cm_v2f gg = cm_gelu_grad_and_act_bf16_2(cm_v2f{__uint_as_float(pw[j] << 16), __uint_as_float(pw[j] & 0xffff0000u)}, ka, kb, sc1, ao[j]);
                        const float d0 = ka ? __uint_as_float(gw[j] << 16) * sc1 * gg.x : 0.f, d1 = kb ? __uint_as_float(gw[j] & 0xffff0000u) * sc1 * gg.y : 0.f;
                        dv_[j] = pack2(d0, d1);
                        cs[2 * j] += __uint_as_float(dv_[j] << 16), cs[2 * j + 1] += __uint_as_float(dv_[j] & 0xffff0000u);   // the rounded values, as the GEMMs see them
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{ao[0], ao[1], ao[2], ao[3]}, ract, el * 2, 0, 0);        // rows past the end are dropped
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{dv_[0], dv_[1], dv_[2], dv_[3]}, rda1, el * 2, 0, 0);
                    *reinterpret_cast<uint4 *>(hrow + (i0 + i) * 8 * XS) = make_uint4(dv_[0], dv_[1], dv_[2], dv_[3]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            float *rr = red + (tid >> 5) * CH + (tid & 31) * 8;
            *reinterpret_cast<float4 *>(rr) = make_float4(cs[0], cs[1], cs[2], cs[3]);
            *reinterpret_cast<float4 *>(rr + 4) = make_float4(cs[4], cs[5], cs[6], cs[7]);
        }
        lds_barrier();
        {
            float s = 0.f;                                           // fixed order over the 8 row groups
#pragma unroll
            for (int g8 = 0; g8 < 8; ++g8) s += red[g8 * CH + tid];
            p.db1_part[(int64_t)blockIdx.x * (F + D) + c * CH + tid] = s;
        }
        read_frags(hfrag, 0, bfa);
#pragma unroll
        for (int s = 8; s < 16; ++s) {                               // GEMM 2: 64 output features x 64 tokens, K = this slab
            bf16x8(&cur)[4] = (s & 1) ? bfb : bfa;
            bf16x8(&nxt)[4] = (s & 1) ? bfa : bfb;
            if (s + 1 < 16) read_frags(hfrag, s + 1 - 8, nxt);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc2[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[s % PF][mb], cur[nb], acc2[mb][nb], 0, 0, 0);
            if (s + PF < 16) wload(c, s + PF, wq[s % PF]);
            else if (c + 1 < nch) wload(c + 1, s + PF - 16, wq[s % PF]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- dh out: bf16, through the dead da2 tile as whole 512-byte rows (a lane's accumulators are 8-byte pieces of 16 rows)
    lds_barrier();                                                   // every wave is done with the last GEMM's fragments (hc) -- xn is long dead
    {
        uint16_t *xd = xn + l15 * XS + wave * 64 + lq * 4;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                uint2 pk;
                pk.x = pack2(acc2[mb][nb][0], acc2[mb][nb][1]);
                pk.y = pack2(acc2[mb][nb][2], acc2[mb][nb][3]);
                *reinterpret_cast<uint2 *>(xd + nb * 16 * XS + mb * 16) = pk;
            }
    }
    lds_barrier();
    {
        uint16_t *o = reinterpret_cast<uint16_t *>(p.dh);
#pragma unroll 2
        for (int i = 0; i < TOK * 32 / NT; ++i) {
            const int idx = tid + NT * i, row = idx >> 5, ch = idx & 31;
            const uint4 v = *reinterpret_cast<const uint4 *>(xn + row * XS + ch * 8);
            if (t0 + row < M) *reinterpret_cast<uint4 *>(o + (int64_t)(t0 + row) * D + ch * 8) = v;
        }
    }
}

// out[col] (= or +=) sum over the workgroups' partial rows, fixed order: 32 columns x 32 row groups per workgroup
__global__ __launch_bounds__(1024) void ffn_bwd_colsum_kernel(const float *__restrict__ part, const int nrow, const int dim, float *__restrict__ out) {
    __shared__ float red[32][32];           // 32 row groups: 8 left each of the few workgroups walking 250 dependent loads at 32 k rows (14 us)
    const int col = blockIdx.x * 32 + (threadIdx.x & 31), grp = threadIdx.x >> 5;
    float s = 0.f;
    if (col < dim) {
#pragma unroll 8
        for (int b = grp; b < nrow; b += 32) s += part[(int64_t)b * dim + col];
    }
    red[grp][threadIdx.x & 31] = s;
    __syncthreads();
    if (grp == 0 && col < dim) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += red[k][threadIdx.x & 31];
        out[col] = t;
    }
}

}  // namespace

extern "C" int64_t cm_ffn_bwd_workspace_floats(int32_t rows, int32_t hidden) {
    if (rows <= 0 || hidden <= 0) return 0;
    return (int64_t)((rows + TOK - 1) / TOK) * (hidden + D);
}

extern "C" int cm_ffn_bwd_fused(const cm_ffn_bwd_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "ffn_bwd_fused: args is NULL");
    cm_ffn_bwd_args a = *args;
    CM_REQUIRE(a.rows > 0 && a.dout && a.w2t && a.w1t && a.pre && a.da2 && a.da1 && a.act && a.dh && a.db1 && a.db2 && a.workspace, CM_EINVAL,
               "ffn_bwd_fused: bad sizes or NULL tensor");
    CM_REQUIRE(a.dim == D, CM_EUNSUPPORTED, "ffn_bwd_fused: d_model must be 256 (got %d)", a.dim);
    CM_REQUIRE(a.hidden >= CH && a.hidden % CH == 0 && a.hidden <= 2048, CM_EUNSUPPORTED, "ffn_bwd_fused: hidden must be a multiple of 256, at most 2048 (got %d)", a.hidden);
    CM_REQUIRE(a.p1 >= 0.f && a.p1 < 1.f && a.p2 >= 0.f && a.p2 < 1.f, CM_EINVAL, "ffn_bwd_fused: dropout probabilities must be in [0, 1)");
    CM_REQUIRE((int64_t)a.rows * a.hidden * 2 < ((int64_t)1 << 31), CM_EUNSUPPORTED, "ffn_bwd_fused: rows x hidden too large for 32-bit offsets");
    CM_REQUIRE(cm_aligned(a.dout, 16) && cm_aligned(a.w2t, 16) && cm_aligned(a.w1t, 16) && cm_aligned(a.pre, 16) && cm_aligned(a.da2, 16) && cm_aligned(a.da1, 16) &&
                   cm_aligned(a.act, 16) && cm_aligned(a.dh, 16) && cm_aligned(a.workspace, 16),
               CM_EALIGN, "ffn_bwd_fused: tensors must be 16-byte aligned");
    const int nwg = (a.rows + TOK - 1) / TOK;
    CM_REQUIRE(a.workspace_floats >= cm_ffn_bwd_workspace_floats(a.rows, a.hidden), CM_EINVAL, "ffn_bwd_fused: workspace smaller than cm_ffn_bwd_workspace_floats()");
    CM_REQUIRE(a.db2 == a.db1 + a.hidden, CM_EINVAL, "ffn_bwd_fused: db2 must follow db1 (one (hidden + 256) fp32 buffer: one fold launch for both)");
    a.db1_part = a.workspace;                                        // rows of (hidden + 256): [db1 | db2] per workgroup
    a.db2_part = a.workspace + a.hidden;
    const size_t smem = (size_t)2 * TOK * XS * sizeof(uint16_t) + (size_t)8 * CH * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&ffn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            cm_set_error("ffn_bwd_fused: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return (int)e;
        }
        attr_done = true;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    hipLaunchKernelGGL(ffn_bwd_kernel, dim3(nwg), dim3(NT), smem, st, a);
    if (int rc = cm_launch_status("cm_ffn_bwd_fused")) return rc;
    hipLaunchKernelGGL(ffn_bwd_colsum_kernel, dim3((a.hidden + D + 31) / 32), dim3(1024), 0, st, a.workspace, nwg, a.hidden + D, a.db1);
    return cm_launch_status("cm_ffn_bwd_fused(bias sums)");
}
