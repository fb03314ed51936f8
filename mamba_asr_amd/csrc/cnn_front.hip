// cnn_front.hip — both blocks of the CNN front end in ONE kernel (contract: cm_cnn_front in include/conmamba_hip.h;
// reference hparams/CTC/conmamba_large.yaml:187-194 -> speechbrain ConvolutionFrontEnd, 2 x [Conv2d 3x3 stride 2 ->
// LayerNorm(freq, channel) -> LeakyReLU], channels (64, 32), on 80 mel bins).
//
// cm_cnn_block1 wrote its (batch, T/2 + 2, 42, 64) bf16 output to HBM (688 MB at 64 x 40 s) and cm_cnn_block2 read it
// back: 1.19 ms per step for ~50 GFLOP.  Block 1 is cheap to compute (one input channel, 9 taps), so here its rows are
// produced straight into the LDS tile block 2's implicit GEMM reads, and the intermediate never exists in memory:
//   * a persistent workgroup (8 waves) walks CONSECUTIVE tiles of 4 output steps of one utterance.  A tile needs 9
//     block-1 rows (stride 2, 3 taps); the last row of a tile is the first of the next and stays in LDS, so each tile
//     computes exactly 8 new rows = one per wave;
//   * a wave computes its block-1 row alone: 3 reflect-padded feature rows in a private LDS patch (prefetched into
//     registers one tile ahead), lane = (channel pair, frequency parity), 40 conv outputs per lane in registers,
//     LayerNorm statistics by wave reduction (no workgroup barrier), LeakyReLU, bf16 pair stores into the padded
//     [row][42][64 + 4] tile including the reflected frequency border;
//   * block 2 is cm_cnn_block2's scheme: v_mfma_f32_16x16x32_bf16 with the 32 x 576 weights as A
//     operands (fragment images staged once in LDS), positions as B operands read from the tile, conv outputs to LDS in fp32, then one wave per output
//     step does LayerNorm over 20 x 32 values + LeakyReLU and writes the bf16 row.
// HBM traffic: the features once (each input row is read by the 1-2 waves that need it) and the output once.
#include "cm_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int F0 = 80;                    // mel bins
constexpr int C1 = 64, F1 = 40, F1P = 42; // block 1: channels, output bins, padded bins
constexpr int C2 = 32, F2 = 20;           // block 2
constexpr int TT = 4;                     // block-2 output steps per tile
constexpr int NR = 2 * TT + 1;            // block-1 rows per tile
constexpr int CS = 68;                    // tile column stride in bf16 elements (136 bytes: conflict-free fragments)
constexpr int OTS = 36;                   // block-2 output-tile row stride in floats
constexpr int FW = 84;                    // floats per staged feature row (82 used)
constexpr int NB = (TT * F2 + 15) / 16;   // waves with MFMA work (5)

__device__ __forceinline__ int reflect_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
__device__ __forceinline__ float wave_sum64(float v) {
    v = cm_group_sum<16>(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ uint32_t pack2(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}

struct front_lds {
    uint16_t rows[NR][F1P][CS];           // block-1 output tile (bf16), slot = padded row index mod NR
    float ot[NB * 16][OTS];               // block-2 conv outputs of the tile
    float fin[8][3][2][FW];               // per-wave reflect-padded feature rows, twice: copy 1 is copy 0 shifted by two bins,
                                          // so that a lane of either frequency parity reads its 3-tap window 16-byte aligned
    float ln1[2][F1 * C1];                // block-1 LayerNorm weight / bias
    bf16x8 wfrag[2][18][64];              // block-2 weights as MFMA A-operand fragment images (36 KB; in registers they
};                                        // cost 144 VGPRs and pushed the block-1 row computation into scratch)

__global__ __launch_bounds__(512) void cnn_front_kernel(const cm_cnn_front_args p, int T1, int T2, int tiles_per_utt, int chunk, int chunks_per_utt,
                                                        int nchunks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    front_lds &L = *reinterpret_cast<front_lds *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int T = p.T, T1P = T1 + 2;

    // ---- constants held in registers for the whole kernel
    const uint16_t *W2 = reinterpret_cast<const uint16_t *>(p.w2);                    // (32, 3, 3, 64) bf16
    for (int i = tid; i < 2 * 18 * 64; i += 512) {
        const int cb = i / (18 * 64), ks = (i / 64) % 18, ln = i % 64;
        L.wfrag[cb][ks][ln] = *reinterpret_cast<const bf16x8 *>(W2 + ((int64_t)(cb * 16 + (ln & 15)) * 9 + (ks >> 1)) * C1 + (ks & 1) * 32 + (ln >> 4) * 8);
    }
    float4 bias2[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
        bias2[cb] = p.b2 ? *reinterpret_cast<const float4 *>(p.b2 + cb * 16 + lq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int cp = lane & 31, fh = lane >> 5;                     // block 1: channel pair, frequency parity
    f32x2 w1[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w1[k] = f32x2{p.w1[(2 * cp) * 9 + k], p.w1[(2 * cp + 1) * 9 + k]};
    const float b1a = p.b1 ? p.b1[2 * cp] : 0.f, b1b = p.b1 ? p.b1[2 * cp + 1] : 0.f;
    for (int i = tid; i < F1 * C1; i += 512) { L.ln1[0][i] = p.ln1_g[i]; L.ln1[1][i] = p.ln1_b[i]; }

    // this lane's block-2 output position inside a tile (clamped duplicates past the tile are not used)
    const int pidx = min(wave * 16 + l15, TT * F2 - 1);
    const int pr = pidx / F2, pf = pidx % F2;
    uint16_t *out = reinterpret_cast<uint16_t *>(p.out);

    // feature rows of padded block-1 row tp of utterance b: 3 rows x 82 reflect-padded bins, 4 values per lane
    auto fetch_rows = [&](const float *feats, int tp, float (&r)[4]) {
        const int t1 = reflect_idx(min(tp, T1P - 1) - 1, T1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = lane + 64 * k;
            r[k] = 0.f;
            if (idx < 3 * 82) {
                const int rr = idx / 82, fc = idx - rr * 82;
                r[k] = feats[(int64_t)reflect_idx(2 * t1 + rr - 1, T) * F0 + reflect_idx(fc - 1, F0)];
            }
        }
    };
    auto stage_rows = [&](const float (&r)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = lane + 64 * k;
            if (idx < 3 * 82) {
                const int rr = idx / 82, fc = idx % 82;
                L.fin[wave][rr][0][fc] = r[k];
                if (fc >= 2) L.fin[wave][rr][1][fc - 2] = r[k];
            }
        }
    };
    // one block-1 row (from this wave's staged feature rows) -> tile slot
    auto block1_row = [&](int slot) {
        float v[2 * (F1 / 2)];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < F1 / 2; ++i) {
            // output bin f1 = 2 i + fh reads bins 2 f1 .. 2 f1 + 2 of the padded rows = floats 4 i .. 4 i + 2 of copy fh: one
            // 16-byte read per input row (the lanes of one parity share the address: broadcast), both channels of the
            // lane's pair per v_pk_fma_f32
            f32x2 a2 = {b1a, b1b};
#pragma unroll
            for (int dt = 0; dt < 3; ++dt) {
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(&L.fin[wave][dt][fh][4 * i]);
#pragma unroll
                for (int df = 0; df < 3; ++df) a2 = __builtin_elementwise_fma(w1[dt * 3 + df], f32x2{xv[df], xv[df]}, a2);
            }
            v[2 * i] = a2.x; v[2 * i + 1] = a2.y;
            s += a2.x + a2.y;
            if (i & 1) __builtin_amdgcn_sched_barrier(0);        // keeps the compiler from hoisting all 60 LDS reads (spills)
        }
        const float mean = wave_sum64(s) * (1.f / (F1 * C1));
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < F1; ++i) { v[i] -= mean; sq = fmaf(v[i], v[i], sq); }
        const float rstd = rsqrtf(wave_sum64(sq) * (1.f / (F1 * C1)) + p.eps1);
#pragma unroll
        for (int i = 0; i < F1 / 2; ++i) {
            const int f1 = 2 * i + fh, o = f1 * C1 + 2 * cp;
            const float2 g = *reinterpret_cast<const float2 *>(&L.ln1[0][o]);
            const float2 bt = *reinterpret_cast<const float2 *>(&L.ln1[1][o]);
            float y0 = fmaf(v[2 * i] * rstd, g.x, bt.x), y1 = fmaf(v[2 * i + 1] * rstd, g.y, bt.y);
            y0 = y0 > 0.f ? y0 : p.slope * y0;
            y1 = y1 > 0.f ? y1 : p.slope * y1;
            const uint32_t pk = pack2(y0, y1);
            *reinterpret_cast<uint32_t *>(&L.rows[slot][f1 + 1][2 * cp]) = pk;
            if (f1 == 1) *reinterpret_cast<uint32_t *>(&L.rows[slot][0][2 * cp]) = pk;              // reflected borders
            if (f1 == F1 - 2) *reinterpret_cast<uint32_t *>(&L.rows[slot][F1P - 1][2 * cp]) = pk;
        }
    };

    for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        const int b = ch / chunks_per_utt, tile0 = (ch % chunks_per_utt) * chunk;
        const int tile1 = min(tile0 + chunk, tiles_per_utt);
        const float *feats = p.feats + (int64_t)b * T * F0;
        float pre[4];
        __syncthreads();                                          // previous chunk's tile is dead (and ln1 is staged)
        // chunk prologue: padded row 8*tile0 (the row every later tile inherits) by wave 0
        if (wave == 0) {
            fetch_rows(feats, 2 * TT * tile0, pre);
            stage_rows(pre);
            block1_row((2 * TT * tile0) % NR);
        }
        fetch_rows(feats, 2 * TT * tile0 + 1 + wave, pre);
        for (int tile = tile0; tile < tile1; ++tile) {
            const int tp0 = 2 * TT * tile, r0 = TT * tile;
            // ---- block 1: wave w -> padded row tp0 + 1 + w
            stage_rows(pre);
            if (tile + 1 < tile1) fetch_rows(feats, tp0 + 2 * TT + 1 + wave, pre);      // next tile's rows, one tile ahead
            block1_row((tp0 + 1 + wave) % NR);
            cm_lds_barrier();
            // ---- block 2: implicit GEMM, 16 positions x 32 channels per wave
            if (wave < NB) {
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                const uint16_t *fr[3];
#pragma unroll
                for (int dt = 0; dt < 3; ++dt) fr[dt] = &L.rows[(tp0 + 2 * pr + dt) % NR][2 * pf][lq * 8];
#pragma unroll
                for (int ks = 0; ks < 18; ++ks) {
                    const int tap = ks >> 1, dt = tap / 3, df = tap % 3;
                    const bf16x8 bfr = *reinterpret_cast<const bf16x8 *>(fr[dt] + df * CS + (ks & 1) * 32);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(L.wfrag[0][ks][lane], bfr, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(L.wfrag[1][ks][lane], bfr, acc1, 0, 0, 0);
                }
                float *o = &L.ot[wave * 16 + l15][lq * 4];
                *reinterpret_cast<float4 *>(o) = make_float4(acc0[0] + bias2[0].x, acc0[1] + bias2[0].y, acc0[2] + bias2[0].z, acc0[3] + bias2[0].w);
                *reinterpret_cast<float4 *>(o + 16) = make_float4(acc1[0] + bias2[1].x, acc1[1] + bias2[1].y, acc1[2] + bias2[1].z, acc1[3] + bias2[1].w);
            }
            cm_lds_barrier();
            // ---- LayerNorm over (freq, channel) + LeakyReLU, one wave per output step
            if (wave < TT && r0 + wave < T2) {
                constexpr int nfeat = F2 * C2;
                const float *orow = &L.ot[wave * F2][0];
                float2 vv[nfeat / 128];
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < nfeat / 128; ++k) {
                    const int i = lane + 64 * k;
                    vv[k] = *reinterpret_cast<const float2 *>(orow + (i >> 4) * OTS + 2 * (i & 15));
                    s += vv[k].x + vv[k].y;
                }
                const float mean = wave_sum64(s) * (1.f / nfeat);
                float sq = 0.f;
#pragma unroll
                for (int k = 0; k < nfeat / 128; ++k) {
                    vv[k].x -= mean; vv[k].y -= mean;
                    sq = fmaf(vv[k].x, vv[k].x, fmaf(vv[k].y, vv[k].y, sq));
                }
                const float rstd = rsqrtf(wave_sum64(sq) * (1.f / nfeat) + p.eps2);
                uint16_t *dst = out + ((int64_t)b * T2 + r0 + wave) * nfeat;
#pragma unroll
                for (int k = 0; k < nfeat / 128; ++k) {
                    const int i = lane + 64 * k;
                    const float2 g = *reinterpret_cast<const float2 *>(p.ln2_g + 2 * i);
                    const float2 bt = *reinterpret_cast<const float2 *>(p.ln2_b + 2 * i);
                    float y0 = fmaf(vv[k].x * rstd, g.x, bt.x), y1 = fmaf(vv[k].y * rstd, g.y, bt.y);
                    y0 = y0 > 0.f ? y0 : y0 * p.slope;
                    y1 = y1 > 0.f ? y1 : y1 * p.slope;
                    *reinterpret_cast<uint32_t *>(dst + 2 * i) = pack2(y0, y1);
                }
            }
            // the next tile overwrites 8 of the 9 row slots and ot: everyone must be done reading them
            cm_lds_barrier();
        }
    }
}

}  // namespace

extern "C" int cm_cnn_front(const cm_cnn_front_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "cnn_front: args is NULL");
    const cm_cnn_front_args &a = *args;
    CM_REQUIRE(a.batch > 0 && a.T >= 5 && a.feats && a.w1 && a.ln1_g && a.ln1_b && a.w2 && a.ln2_g && a.ln2_b && a.out, CM_EINVAL,
               "cnn_front: bad sizes or NULL tensor");
    CM_REQUIRE(a.F == F0 && a.C1 == C1 && a.C2 == C2, CM_EUNSUPPORTED, "cnn_front: only 80 bins, channels (64, 32) (got F %d, C %d, %d)", a.F,
               a.C1, a.C2);
    CM_REQUIRE(cm_aligned(a.w2, 16) && cm_aligned(a.out, 4) && cm_aligned(a.ln1_g, 8) && cm_aligned(a.ln1_b, 8) && cm_aligned(a.ln2_g, 8) &&
                   cm_aligned(a.ln2_b, 8) && (!a.b2 || cm_aligned(a.b2, 16)),
               CM_EALIGN, "cnn_front: w2 / b2 must be 16-byte aligned, LayerNorm parameters 8-byte aligned");
    const int T1 = (a.T + 1) / 2, T2 = (T1 + 2 - 3) / 2 + 1;
    const int tiles_per_utt = (T2 + TT - 1) / TT;
    const int chunk = 16;
    const int chunks_per_utt = (tiles_per_utt + chunk - 1) / chunk;
    const int64_t nchunks = (int64_t)a.batch * chunks_per_utt;
    CM_REQUIRE(nchunks <= 2147483647, CM_EINVAL, "cnn_front: too many chunks");
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(cnn_front_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) {
            cm_set_error("cnn_front: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return (int)e;
        }
        attr_done = true;
    }
    const int64_t grid = nchunks < 256 ? nchunks : 256;
    hipLaunchKernelGGL(cnn_front_kernel, dim3((unsigned)grid), dim3(512), sizeof(front_lds), reinterpret_cast<hipStream_t>(a.stream), a, T1, T2,
                       tiles_per_utt, chunk, chunks_per_utt, (int)nchunks);
    return cm_launch_status("cm_cnn_front");
}
