// gemm_bf16.hip — bf16 MFMA GEMM with fused epilogues (contract: cm_gemm_bf16 in include/conmamba_hip.h).
//
// Shapes on this path: M = batch*time rows (16k..64k), N in {256, 512, 1024}, K in {256, 512, 640, 1024}.  The layer
// is a chain of such GEMMs separated by bias/GELU or residual-add + LayerNorm; through the vendor library each seam
// was an extra pass over (M, N) and the GEMMs ran at 11-18 % of MFMA peak.  Here one workgroup computes a
// (64 rows x 256 columns) tile and applies the seam in registers.
//   * orientation: v_mfma_f32_16x16x32_bf16 with the WEIGHT tile as the A operand and the activation rows as the
//     B operand, so the accumulator has the token on the lane (col = lane & 15) and 4 consecutive output features in
//     its 4 registers: a token's 256 outputs live in 4 lanes x 16 tiles -> LayerNorm statistics are 64 adds and two
//     cross-lane exchanges, stores are 8/16-byte pieces of a 64-byte row segment;
//   * wave w owns tokens [16w, 16w+16) x all 256 columns (16 accumulator tiles = 64 VGPRs);
//   * the 256 x 64 weight tile is staged once per K-step in LDS (rows padded to 160 B: conflict-free
//     ds_read_b128 fragments) and double-buffered; activation fragments (16 B per lane per k-iteration) are read
//     straight from global memory one K-step ahead.
#include "cm_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int BM = 64, BN = 256, BK = 64;
constexpr int LDSROW = 80;                       // elements per LDS weight row (64 + 16 pad) = 160 bytes
constexpr int TILE_ELEMS = BN * LDSROW;          // one buffered weight tile

__device__ __forceinline__ float gelu_erf_f(float x) { return cm_gelu(x); }

template <int EPI>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const cm_gemm_args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *wt = reinterpret_cast<uint16_t *>(smem);             // [2][BN][LDSROW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int M = p.M, K = p.K;
    const uint16_t *A = reinterpret_cast<const uint16_t *>(p.A);
    const uint16_t *W = reinterpret_cast<const uint16_t *>(p.W);
    const int nk = K / BK;

    // activation fragment source: token row of this lane, clamped (rows >= M are computed but never stored)
    const int mrow = m0 + wave * 16 + (lane & 15);
    const uint16_t *arow = A + (int64_t)(mrow < M ? mrow : M - 1) * p.lda + 8 * (lane >> 4);
    // weight tile vectors of this thread: v = tid + 256 i -> row (tid >> 3) + 32 i, k-chunk (tid & 7) * 8
    const uint16_t *wsrc0 = W + (int64_t)(n0 + (tid >> 3)) * p.ldw + (tid & 7) * 8;
    const int64_t wsrc_step = 32 * p.ldw;
    const int wdst0 = (tid >> 3) * LDSROW + (tid & 7) * 8;

    f32x4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 wreg[8];
    bf16x8 bcur[2], bnxt[2];
    auto load_w = [&](int ks) {
#pragma unroll
        for (int i = 0; i < 8; ++i) wreg[i] = *reinterpret_cast<const uint4 *>(wsrc0 + i * wsrc_step + ks * BK);
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4 *>(wt + buf * TILE_ELEMS + wdst0 + i * 32 * LDSROW) = wreg[i];
    };
    auto load_a = [&](int ks, bf16x8 (&b)[2]) {
#pragma unroll
        for (int it = 0; it < 2; ++it) b[it] = *reinterpret_cast<const bf16x8 *>(arow + ks * BK + 32 * it);
    };

    load_w(0);
    load_a(0, bcur);
    store_w(0);
    __syncthreads();
    const int frag_off = (lane & 15) * LDSROW + 8 * (lane >> 4);   // this lane's fragment inside a 16-row tile
    for (int ks = 0; ks < nk; ++ks) {
        if (ks + 1 < nk) {
            load_w(ks + 1);
            load_a(ks + 1, bnxt);
        }
        const uint16_t *wb = wt + (ks & 1) * TILE_ELEMS + frag_off;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const bf16x8 a = *reinterpret_cast<const bf16x8 *>(wb + t * 16 * LDSROW + 32 * it);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bcur[it], acc[t], 0, 0, 0);
            }
        }
        if (ks + 1 < nk) {
            store_w((ks + 1) & 1);
            bcur[0] = bnxt[0];
            bcur[1] = bnxt[1];
        }
        __syncthreads();
    }

    // ---------------- epilogue: lane holds token mrow, features n0 + 16 t + 4 (lane >> 4) + j
    const bool row_ok = mrow < M;
    const int nq = 4 * (lane >> 4);
    if constexpr (EPI == 0 || EPI == 1) {
        uint16_t *orow = reinterpret_cast<uint16_t *>(p.out) + (int64_t)mrow * p.ldo + n0 + nq;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            float v[4] = {acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
            if (p.bias) {
                const float4 bv = *reinterpret_cast<const float4 *>(p.bias + n0 + 16 * t + nq);
                v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
            }
            if constexpr (EPI == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = gelu_erf_f(v[j]);
            }
            uint2 pk;
            pk.x = (uint32_t)cm_elem<cm_bf16>::to_bits(v[0]) | ((uint32_t)cm_elem<cm_bf16>::to_bits(v[1]) << 16);
            pk.y = (uint32_t)cm_elem<cm_bf16>::to_bits(v[2]) | ((uint32_t)cm_elem<cm_bf16>::to_bits(v[3]) << 16);
            if (row_ok) *reinterpret_cast<uint2 *>(orow + 16 * t) = pk;
        }
    } else {
        // r = x + alpha (acc + bias); optional LN1 -> x; optional LN2 -> out.  N == 256: the row is complete here.
        float *xrow = p.x + (int64_t)(row_ok ? mrow : 0) * BN + nq;
        float r[16][4];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            float4 xv = *reinterpret_cast<const float4 *>(xrow + 16 * t);
            float4 bv = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) bv = *reinterpret_cast<const float4 *>(p.bias + 16 * t + nq);
            r[t][0] = fmaf(p.alpha, acc[t][0] + bv.x, xv.x);
            r[t][1] = fmaf(p.alpha, acc[t][1] + bv.y, xv.y);
            r[t][2] = fmaf(p.alpha, acc[t][2] + bv.z, xv.z);
            r[t][3] = fmaf(p.alpha, acc[t][3] + bv.w, xv.w);
        }
        auto row_sum = [&](float v) {                              // over the 4 lanes that share a token
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            return v;
        };
        auto layer_norm = [&](const float *g, const float *b, float eps) {
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t) s += (r[t][0] + r[t][1]) + (r[t][2] + r[t][3]);
            const float mean = row_sum(s) * (1.f / BN);
            float sq = 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d = r[t][j] - mean; sq = fmaf(d, d, sq); }
            const float rstd = rsqrtf(row_sum(sq) * (1.f / BN) + eps);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float4 gv = *reinterpret_cast<const float4 *>(g + 16 * t + nq);
                const float4 bv = *reinterpret_cast<const float4 *>(b + 16 * t + nq);
                r[t][0] = (r[t][0] - mean) * rstd * gv.x + bv.x;
                r[t][1] = (r[t][1] - mean) * rstd * gv.y + bv.y;
                r[t][2] = (r[t][2] - mean) * rstd * gv.z + bv.z;
                r[t][3] = (r[t][3] - mean) * rstd * gv.w + bv.w;
            }
        };
        if (p.g1) layer_norm(p.g1, p.b1, p.eps1);
        if (row_ok) {
#pragma unroll
            for (int t = 0; t < 16; ++t) *reinterpret_cast<float4 *>(xrow + 16 * t) = make_float4(r[t][0], r[t][1], r[t][2], r[t][3]);
        }
        if (p.out) {
            if (p.g2) layer_norm(p.g2, p.b2, p.eps2);
            uint16_t *orow = reinterpret_cast<uint16_t *>(p.out) + (int64_t)mrow * p.ldo + nq;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                uint2 pk;
                pk.x = (uint32_t)cm_elem<cm_bf16>::to_bits(r[t][0]) | ((uint32_t)cm_elem<cm_bf16>::to_bits(r[t][1]) << 16);
                pk.y = (uint32_t)cm_elem<cm_bf16>::to_bits(r[t][2]) | ((uint32_t)cm_elem<cm_bf16>::to_bits(r[t][3]) << 16);
                if (row_ok) *reinterpret_cast<uint2 *>(orow + 16 * t) = pk;
            }
        }
    }
}

template <int EPI>
int launch(const cm_gemm_args &a) {
    const size_t smem = (size_t)2 * TILE_ELEMS * sizeof(uint16_t);        // 81,920 bytes
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_tn_kernel<EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) {
            cm_set_error("gemm_bf16: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return (int)e;
        }
        attr_done = true;
    }
    dim3 grid((a.M + BM - 1) / BM, a.N / BN);
    hipLaunchKernelGGL((gemm_tn_kernel<EPI>), grid, dim3(256), smem, reinterpret_cast<hipStream_t>(a.stream), a);
    return cm_launch_status("cm_gemm_bf16");
}

}  // namespace

extern "C" int cm_gemm_bf16(const cm_gemm_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "gemm_bf16: args is NULL");
    const cm_gemm_args &a = *args;
    CM_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.A && a.W, CM_EINVAL, "gemm_bf16: bad sizes or NULL operand");
    CM_REQUIRE(a.K % BK == 0 && a.N % BN == 0, CM_EUNSUPPORTED, "gemm_bf16: needs K %% 64 == 0 and N %% 256 == 0 (K=%d N=%d)", a.K, a.N);
    CM_REQUIRE(cm_aligned(a.A, 16) && cm_aligned(a.W, 16) && a.lda % 8 == 0 && a.ldw % 8 == 0, CM_EALIGN,
               "gemm_bf16: A/W must be 16-byte aligned with leading dimensions multiple of 8");
    CM_REQUIRE((int64_t)((a.M + BM - 1) / BM) <= 2147483647, CM_EINVAL, "gemm_bf16: M too large");
    switch (a.epilogue) {
        case 0:
        case 1:
            CM_REQUIRE(a.out && cm_aligned(a.out, 8) && a.ldo % 4 == 0, CM_EINVAL, "gemm_bf16: out missing or misaligned");
            return a.epilogue == 0 ? launch<0>(a) : launch<1>(a);
        case 2:
            CM_REQUIRE(a.N == BN && a.x, CM_EUNSUPPORTED, "gemm_bf16: epilogue 2 needs N == 256 and the residual x");
            CM_REQUIRE((!a.g1 || a.b1) && (!a.g2 || a.b2), CM_EINVAL, "gemm_bf16: LayerNorm weight without bias");
            CM_REQUIRE(!a.out || (cm_aligned(a.out, 8) && a.ldo % 4 == 0), CM_EALIGN, "gemm_bf16: out misaligned");
            return launch<2>(a);
        default:
            cm_set_error("gemm_bf16: unknown epilogue %d", a.epilogue);
            return CM_EINVAL;
    }
}
