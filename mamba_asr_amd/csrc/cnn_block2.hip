// cnn_block2.hip — second block of the CNN front end as one kernel (contract: cm_cnn_block2 in include/conmamba_hip.h;
// reference hparams/CTC/conmamba_large.yaml:187-194 -> speechbrain ConvolutionFrontEnd block 2):
// Conv2d(64 -> 32, 3x3, stride 2 in time and frequency) on the reflect-padded channels-last output of cm_cnn_block1,
// LayerNorm over (freq, channel), LeakyReLU.
//
// Through the vendor library this was a conv whose algorithm is picked WITHOUT a search when the launch sequence is
// being captured into a hipGraph (1.0-1.4 ms for the headline batch instead of 0.16 ms), plus layout transposes, dtype
// casts and a separate LayerNorm kernel.  Here it is an implicit GEMM on v_mfma_f32_16x16x32_bf16:
//   * persistent workgroups loop over tiles of TT output time steps of one utterance (TT * F2 output positions, 16 per
//     wave); the 2*TT+1 input rows a tile needs are staged in LDS once (columns padded from 128 to 136 bytes: the
//     stride-2 position fragments are then conflict-free ds_read_b128);
//   * K = 9 taps x 64 input channels = 18 MFMA k-steps; the WEIGHTS are the A operand and live in registers for the
//     whole kernel (2 x 18 fragments = 144 VGPRs per lane, loaded once per workgroup from the OHWI weight tensor);
//     positions are the B operand, so a lane's accumulator is 4 consecutive output channels of one position;
//   * the 3x3 window is not materialised: a fragment is 8 consecutive input channels of one (row, column) of the
//     staged rows;
//   * conv outputs of the tile go to LDS (fp32), then one wave per output time step does LayerNorm over its
//     F2 x 32 values + LeakyReLU and writes the bf16 row.
#include "cm_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int CIN = 64, COUT = 32;
constexpr int CS = 68;          // LDS column stride in bf16 elements (136 bytes)
constexpr int OTS = 36;         // LDS output-tile row stride in floats

__device__ __forceinline__ float wave_sum64(float v) {
    v = cm_group_sum<16>(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

__global__ __launch_bounds__(512) void cnn_block2_kernel(const cm_cnn_block2_args p, int TT, int NB, int tiles_per_utt, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int T1p = p.T_in, F1p = p.F_in, T2 = (T1p - 3) / 2 + 1, F2 = (F1p - 3) / 2 + 1;
    const int nrows = 2 * TT + 1, npos = TT * F2, nfeat = F2 * COUT;
    uint16_t *rows = reinterpret_cast<uint16_t *>(smem);                              // [nrows][F1p][CS]
    float *ot = reinterpret_cast<float *>(rows + (size_t)nrows * F1p * CS);          // [NB*16][OTS]
    const int tid = threadIdx.x, lane = tid & 63, nthreads = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const uint16_t *W = reinterpret_cast<const uint16_t *>(p.weight);                 // (COUT, 3, 3, CIN)
    const uint16_t *in = reinterpret_cast<const uint16_t *>(p.in);
    uint16_t *out = reinterpret_cast<uint16_t *>(p.out);

    // weights: A fragments for both 16-channel halves and all 18 k-steps (tap-major, 32 input channels per step)
    bf16x8 wf[2][18];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int ks = 0; ks < 18; ++ks)
            wf[cb][ks] = *reinterpret_cast<const bf16x8 *>(W + ((int64_t)(cb * 16 + l15) * 9 + (ks >> 1)) * CIN + (ks & 1) * 32 + lq * 8);
    float4 bias4[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
        bias4[cb] = p.bias ? *reinterpret_cast<const float4 *>(p.bias + cb * 16 + lq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);

    // this lane's output position inside a tile (clamped: lanes past the tile compute a duplicate that is not used)
    const int pidx = min(wave * 16 + l15, npos - 1);
    const int pr = pidx / F2, pf = pidx % F2;
    const uint16_t *frag0 = rows + ((size_t)(2 * pr) * F1p + 2 * pf) * CS + lq * 8;

    // The input rows of a tile travel global -> registers -> LDS; the registers of tile i+1 are loaded while tile i is
    // computed (a first version loaded and stored piece by piece and spent 17 us per tile on exposed HBM latency).
    constexpr int MAXP = 12;                                      // 16-byte pieces per thread per tile (host-checked)
    const int pieces_per_row = F1p * 8, npieces = nrows * pieces_per_row;
    uint4 pre[MAXP];
    auto fetch = [&](int tile) {
        const int b = tile / tiles_per_utt, r0 = (tile % tiles_per_utt) * TT;
        const uint16_t *src = in + (int64_t)b * T1p * F1p * CIN;
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            const int i = tid + k * nthreads;
            if (i < npieces) {
                const int row = i / pieces_per_row, rem = i - row * pieces_per_row;
                const int tr = min(2 * r0 + row, T1p - 1);
                pre[k] = *reinterpret_cast<const uint4 *>(src + ((int64_t)tr * F1p + (rem >> 3)) * CIN + (rem & 7) * 8);
            }
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_utt, r0 = (tile % tiles_per_utt) * TT;
        // ---- input rows 2*r0 .. 2*r0 + 2*TT (clamped at the end of the utterance) -> LDS
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            const int i = tid + k * nthreads;
            if (i < npieces) {
                const int row = i / pieces_per_row, rem = i - row * pieces_per_row;
                *reinterpret_cast<uint4 *>(rows + ((size_t)row * F1p + (rem >> 3)) * CS + (rem & 7) * 8) = pre[k];
            }
        }
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
        __syncthreads();
        // ---- implicit GEMM: 16 positions x 32 channels per wave
        if (wave < NB) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 18; ++ks) {
                const int tap = ks >> 1, dt = tap / 3, df = tap % 3;
                const bf16x8 bf = *reinterpret_cast<const bf16x8 *>(frag0 + ((size_t)dt * F1p + df) * CS + (ks & 1) * 32);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][ks], bf, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][ks], bf, acc1, 0, 0, 0);
            }
            float *o = ot + (wave * 16 + l15) * OTS + lq * 4;
            *reinterpret_cast<float4 *>(o) = make_float4(acc0[0] + bias4[0].x, acc0[1] + bias4[0].y, acc0[2] + bias4[0].z, acc0[3] + bias4[0].w);
            *reinterpret_cast<float4 *>(o + 16) = make_float4(acc1[0] + bias4[1].x, acc1[1] + bias4[1].y, acc1[2] + bias4[1].z, acc1[3] + bias4[1].w);
        }
        __syncthreads();
        // ---- LayerNorm over (freq, channel) + LeakyReLU, one wave per output time step
        const int nwaves = nthreads >> 6;
        for (int r = wave; r < TT; r += nwaves) {
            const int t2 = r0 + r;
            if (t2 >= T2) break;
            const float *orow = ot + (size_t)r * F2 * OTS;
            float s = 0.f;
            for (int i = lane; i < nfeat / 2; i += 64) {
                const float2 v = *reinterpret_cast<const float2 *>(orow + (i >> 4) * OTS + 2 * (i & 15));
                s += v.x + v.y;
            }
            const float mean = wave_sum64(s) / nfeat;
            float sq = 0.f;
            for (int i = lane; i < nfeat / 2; i += 64) {
                const float2 v = *reinterpret_cast<const float2 *>(orow + (i >> 4) * OTS + 2 * (i & 15));
                sq = fmaf(v.x - mean, v.x - mean, fmaf(v.y - mean, v.y - mean, sq));
            }
            const float rstd = rsqrtf(wave_sum64(sq) / nfeat + p.eps);
            uint16_t *dst = out + ((int64_t)b * T2 + t2) * nfeat;
            for (int i = lane; i < nfeat / 2; i += 64) {
                const float2 v = *reinterpret_cast<const float2 *>(orow + (i >> 4) * OTS + 2 * (i & 15));
                const float2 g = *reinterpret_cast<const float2 *>(p.ln_g + 2 * i);
                const float2 bt = *reinterpret_cast<const float2 *>(p.ln_b + 2 * i);
                float y0 = fmaf((v.x - mean) * rstd, g.x, bt.x), y1 = fmaf((v.y - mean) * rstd, g.y, bt.y);
                y0 = y0 > 0.f ? y0 : y0 * p.slope;
                y1 = y1 > 0.f ? y1 : y1 * p.slope;
                *reinterpret_cast<uint32_t *>(dst + 2 * i) =
                    (uint32_t)cm_elem<cm_bf16>::to_bits(y0) | ((uint32_t)cm_elem<cm_bf16>::to_bits(y1) << 16);
            }
        }
        __syncthreads();                                          // rows / ot are rewritten by the next tile
    }
}

}  // namespace

extern "C" int cm_cnn_block2(const cm_cnn_block2_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "cnn_block2: args is NULL");
    const cm_cnn_block2_args &a = *args;
    CM_REQUIRE(a.batch > 0 && a.T_in >= 3 && a.F_in >= 3 && a.in && a.weight && a.ln_g && a.ln_b && a.out, CM_EINVAL,
               "cnn_block2: bad sizes or NULL tensor");
    CM_REQUIRE(a.C_in == CIN && a.C_out == COUT, CM_EUNSUPPORTED, "cnn_block2: only 64 -> 32 channels (got %d -> %d)", a.C_in, a.C_out);
    CM_REQUIRE(cm_aligned(a.in, 16) && cm_aligned(a.weight, 16) && cm_aligned(a.out, 4) && cm_aligned(a.ln_g, 8) && cm_aligned(a.ln_b, 8) &&
                   (!a.bias || cm_aligned(a.bias, 16)),
               CM_EALIGN, "cnn_block2: tensors must be 16-byte aligned");
    const int T2 = (a.T_in - 3) / 2 + 1, F2 = (a.F_in - 3) / 2 + 1;
    CM_REQUIRE(F2 >= 1 && F2 <= 128, CM_EUNSUPPORTED, "cnn_block2: output frequency bins %d out of range (1..128)", F2);
    int TT = 128 / F2;                                            // positions per tile <= 128 (8 waves)
    if (TT > 4) TT = 4;
    if (TT < 1) TT = 1;
    const int NB = (TT * F2 + 15) / 16;
    int nwaves = NB < TT ? TT : NB;
    if (nwaves > 8) nwaves = 8;
    const size_t smem = (size_t)(2 * TT + 1) * a.F_in * CS * 2 + (size_t)NB * 16 * OTS * 4;
    CM_REQUIRE(smem <= 150 * 1024, CM_EUNSUPPORTED, "cnn_block2: F_in %d needs %zu bytes of LDS", a.F_in, smem);
    CM_REQUIRE((2 * TT + 1) * a.F_in * 8 <= 12 * 64 * nwaves, CM_EUNSUPPORTED, "cnn_block2: F_in %d too wide for the row prefetch", a.F_in);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(cnn_block2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) {
            cm_set_error("cnn_block2: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return (int)e;
        }
        attr_done = true;
    }
    const int tiles_per_utt = (T2 + TT - 1) / TT;
    const int64_t ntiles = (int64_t)a.batch * tiles_per_utt;
    CM_REQUIRE(ntiles <= 2147483647, CM_EINVAL, "cnn_block2: too many tiles");
    const int per_cu = smem <= 75 * 1024 ? 2 : 1;
    const int64_t grid = ntiles < 256 * per_cu ? ntiles : 256 * per_cu;
    hipLaunchKernelGGL(cnn_block2_kernel, dim3((unsigned)grid), dim3(64 * nwaves), smem, reinterpret_cast<hipStream_t>(a.stream), a, TT, NB,
                       tiles_per_utt, (int)ntiles);
    return cm_launch_status("cm_cnn_block2");
}
