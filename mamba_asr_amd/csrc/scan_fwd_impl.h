// scan_fwd_impl.h — selective-scan forward kernel template for gfx950 (MI355X); instantiated per
// dtype pair by scan_fwd_<dtypes>.hip, entry point in selective_scan_fwd.hip.
//
// Contract: cm_selective_scan_fwd in include/conmamba_hip.h (replaces selective_scan_cuda.fwd,
// reference modules/mamba/selective_scan_interface.py:42/218; math = selective_scan_ref :91-157).
//
// Mapping (wave64, no time-parallel scan):
//   * the recurrence is sequential in time inside a lane; parallelism comes from
//     (batch, channel, state-group).  A channel is owned by S adjacent lanes, each carrying
//     NS = dstate/S states in registers; y_t = sum_n C h is a log2(S)-step DPP reduction.
//     S is picked at launch so that the grid fills the chip (small batch -> large S).
//   * a wave owns 64/S channels and walks time in tiles of TT = 16*S steps, so one tile of
//     one tensor is always 1024 elements = whole 16-byte vectors along the contiguous time axis
//     (coalesced row segments), staged through LDS to turn "time along lanes" (global) into
//     "time along the loop" (compute).
//   * per-(channel,t) work (softplus, delta*u, gate) is split over Q = min(S,4) lanes of the
//     channel and re-distributed with quad-perm DPP broadcasts.
//   * B_t / C_t (shared by all channels of a batch row) are staged once per workgroup as fp32.
//   * next tile's global loads are issued before the current tile's compute (register prefetch).
#pragma once
#include "cm_common.h"

namespace {

constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;

template <typename IO, bool VECOK>
__device__ __forceinline__ uint4 load_vec_guarded(const IO *src, int nvalid) {
    constexpr int VEC = cm_elem<IO>::kVec;
    uint4 r = {0u, 0u, 0u, 0u};
    if constexpr (VECOK) {   // rows are 16-B aligned and seqlen % VEC == 0: a vector is all-valid or all-padding
        if (nvalid > 0) r = *reinterpret_cast<const uint4 *>(src);
        return r;
    }
    uint32_t *w = reinterpret_cast<uint32_t *>(&r);
    if constexpr (sizeof(IO) == 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < nvalid) w[j] = reinterpret_cast<const uint32_t *>(src)[j];
    } else {
        const uint16_t *s = reinterpret_cast<const uint16_t *>(src);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < nvalid) w[j >> 1] |= (uint32_t)s[j] << ((j & 1) * 16);
    }
    return r;
}

template <typename IO, bool VECOK>
__device__ __forceinline__ void store_vec_guarded(IO *dst, uint4 v, int nvalid) {
    if constexpr (VECOK) {
        if (nvalid > 0) *reinterpret_cast<uint4 *>(dst) = v;
        return;
    }
    const uint32_t *w = reinterpret_cast<const uint32_t *>(&v);
    if constexpr (sizeof(IO) == 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < nvalid) reinterpret_cast<uint32_t *>(dst)[j] = w[j];
    } else {
        uint16_t *d = reinterpret_cast<uint16_t *>(dst);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < nvalid) d[j] = (uint16_t)(w[j >> 1] >> ((j & 1) * 16));
    }
}

// 4 consecutive B/C elements -> float4 (zero padded past nvalid)
template <typename BC, bool VECOK>
__device__ __forceinline__ float4 load_bc4(const BC *src, int nvalid) {
    float4 r = {0.f, 0.f, 0.f, 0.f};
    float *f = reinterpret_cast<float *>(&r);
    if constexpr (VECOK) {
        if (nvalid <= 0) return r;
        if constexpr (sizeof(BC) == 4) {
            r = *reinterpret_cast<const float4 *>(src);
        } else {
            const uint2 raw = *reinterpret_cast<const uint2 *>(src);
            f[0] = cm_elem<BC>::from_bits((uint16_t)(raw.x & 0xffffu));
            f[1] = cm_elem<BC>::from_bits((uint16_t)(raw.x >> 16));
            f[2] = cm_elem<BC>::from_bits((uint16_t)(raw.y & 0xffffu));
            f[3] = cm_elem<BC>::from_bits((uint16_t)(raw.y >> 16));
        }
        return r;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < nvalid) f[j] = cm_elem<BC>::load(src + j);
        return r;
    }
}

template <typename IO, typename BC, int S, int NS, bool REV, bool VECOK>
__global__ __launch_bounds__(kThreads) void scan_fwd_kernel(const cm_scan_fwd_args p) {
    constexpr int N = S * NS;
    constexpr int Q = S < 4 ? S : 4;
    constexpr int OWN = 4 / Q;
    constexpr int CPW = 64 / S;
    constexpr int TT = 16 * S;
    constexpr int VEC = cm_elem<IO>::kVec;
    constexpr int VPR = TT / VEC;
    constexpr int NV = CPW * VPR / 64;
    constexpr int ROWB = TT * (int)sizeof(IO) + 16;
    constexpr int BCROW = TT + 4;
    constexpr int BCV = N * TT / 4;                       // float4 groups per B (or C) tile
    constexpr int BCI = (BCV + kThreads - 1) / kThreads;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int b = blockIdx.y;
    const int L = p.seqlen;
    const int dim = p.dim;
    const int ew0 = (blockIdx.x * kWaves + wave) * CPW;   // first channel of this wave
    const int c_local = lane / S;
    const int sg = lane % S;
    const int q = lane % Q;
    const int e = ew0 + c_local;
    const bool e_ok = e < dim;
    const int e_c = e_ok ? e : dim - 1;
    const bool has_z = p.z != nullptr;
    const bool want_out = p.out != nullptr;

    unsigned char *utile = smem + wave * (3 * CPW * ROWB);
    unsigned char *dtile = utile + CPW * ROWB;
    unsigned char *ztile = dtile + CPW * ROWB;
    float *Bt = reinterpret_cast<float *>(smem + kWaves * 3 * CPW * ROWB);
    float *Ct = Bt + N * BCROW;

    const IO *ug = reinterpret_cast<const IO *>(p.u) + (int64_t)b * p.u_bs;
    const IO *dg = reinterpret_cast<const IO *>(p.delta) + (int64_t)b * p.delta_bs;
    const IO *zg = has_z ? reinterpret_cast<const IO *>(p.z) + (int64_t)b * p.z_bs : nullptr;
    const BC *Bg = reinterpret_cast<const BC *>(p.B) + (int64_t)b * p.B_bs;
    const BC *Cg = reinterpret_cast<const BC *>(p.C) + (int64_t)b * p.C_bs;

    float Ap[NS], h[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        Ap[i] = p.A[(int64_t)e_c * N + sg * NS + i] * CM_LOG2E;
        h[i] = p.h0 ? p.h0[((int64_t)b * dim + e_c) * N + sg * NS + i] : 0.f;      // carry of a time-split scan, else zero
    }
    const float Dv = p.D ? p.D[e_c] : 0.f;
    const float bias = p.delta_bias ? p.delta_bias[e_c] : 0.f;
    const bool softplus = p.delta_softplus != 0;
    const int nchunks = (L + CM_SCAN_CHUNK - 1) / CM_SCAN_CHUNK;
    float dsum = 0.f;

    const int nT = (L + TT - 1) / TT;
    uint4 ru[NV], rd[NV], rz[NV];
    float4 rB[BCI], rC[BCI];

    auto fetch = [&](int t0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + 64 * i;
            const int row = v / VPR, col = (v % VPR) * VEC;
            const int er = ew0 + row, t = t0 + col;
            int nvalid = (er < dim) ? (L - t) : 0;
            nvalid = nvalid < 0 ? 0 : (nvalid > VEC ? VEC : nvalid);
            const int64_t off_u = (int64_t)er * p.u_ds + t;
            const int64_t off_d = (int64_t)er * p.delta_ds + t;
            ru[i] = load_vec_guarded<IO, VECOK>(ug + off_u, nvalid);
            rd[i] = load_vec_guarded<IO, VECOK>(dg + off_d, nvalid);
            if (has_z) rz[i] = load_vec_guarded<IO, VECOK>(zg + (int64_t)er * p.z_ds + t, nvalid);
        }
#pragma unroll
        for (int i = 0; i < BCI; ++i) {
            const int g = tid + kThreads * i;
            const int n = g / (TT / 4), tq = (g % (TT / 4)) * 4;
            int nvalid = (g < BCV) ? (L - (t0 + tq)) : 0;
            nvalid = nvalid < 0 ? 0 : (nvalid > 4 ? 4 : nvalid);
            rB[i] = load_bc4<BC, VECOK>(Bg + (int64_t)n * p.B_ns + t0 + tq, nvalid);
            rC[i] = load_bc4<BC, VECOK>(Cg + (int64_t)n * p.C_ns + t0 + tq, nvalid);
        }
    };

    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + 64 * i;
            const int row = v / VPR, col = (v % VPR) * VEC;
            const int o = row * ROWB + col * (int)sizeof(IO);
            *reinterpret_cast<uint4 *>(utile + o) = ru[i];
            *reinterpret_cast<uint4 *>(dtile + o) = rd[i];
            if (has_z) *reinterpret_cast<uint4 *>(ztile + o) = rz[i];
        }
#pragma unroll
        for (int i = 0; i < BCI; ++i) {
            const int g = tid + kThreads * i;
            if (g < BCV) {
                const int n = g / (TT / 4), tq = (g % (TT / 4)) * 4;
                *reinterpret_cast<float4 *>(Bt + n * BCROW + tq) = rB[i];
                *reinterpret_cast<float4 *>(Ct + n * BCROW + tq) = rC[i];
            }
        }
    };

    fetch((REV ? nT - 1 : 0) * TT);

    for (int it = 0; it < nT; ++it) {
        const int kt = REV ? nT - 1 - it : it;
        const int t0 = kt * TT;
        stage();
        __syncthreads();
        if (it + 1 < nT) fetch((REV ? kt - 1 : kt + 1) * TT);

        const IO *urow = reinterpret_cast<const IO *>(utile + c_local * ROWB);
        const IO *drow = reinterpret_cast<const IO *>(dtile + c_local * ROWB);
        const IO *zrow = reinterpret_cast<const IO *>(ztile + c_local * ROWB);

        for (int blk = 0; blk < TT / 4; ++blk) {
            const int tb = (REV ? TT / 4 - 1 - blk : blk) * 4;
            if (t0 + tb >= L) continue;                   // whole block is padding (wave-uniform)
            // --- per-(channel,t) quantities, OWN timesteps per lane
            float dt_o[OWN], w_o[OWN], u_o[OWN], z_o[OWN], y_o[OWN];
#pragma unroll
            for (int j = 0; j < OWN; ++j) {
                const int tl = tb + q * OWN + j;
                const bool valid = t0 + tl < L;
                float dv = cm_elem<IO>::load(drow + tl) + bias;
                if (softplus) dv = cm_softplus(dv);
                float uv = cm_elem<IO>::load(urow + tl);
                dt_o[j] = valid ? dv : 0.f;
                u_o[j] = valid ? uv : 0.f;
                w_o[j] = dt_o[j] * u_o[j];
                z_o[j] = has_z ? cm_elem<IO>::load(zrow + tl) : 0.f;
                y_o[j] = 0.f;
            }
            // --- broadcast the four steps' (delta, delta*u) to every lane of the channel
            float dtk[4], wk[4], ypk[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                switch (k) {   // owner of tile-local step k: group lane k / OWN, slot k % OWN
                    case 0: dtk[k] = cm_group_bcast<Q, 0 / OWN>(dt_o[0 % OWN]); wk[k] = cm_group_bcast<Q, 0 / OWN>(w_o[0 % OWN]); break;
                    case 1: dtk[k] = cm_group_bcast<Q, 1 / OWN>(dt_o[1 % OWN]); wk[k] = cm_group_bcast<Q, 1 / OWN>(w_o[1 % OWN]); break;
                    case 2: dtk[k] = cm_group_bcast<Q, 2 / OWN>(dt_o[2 % OWN]); wk[k] = cm_group_bcast<Q, 2 / OWN>(w_o[2 % OWN]); break;
                    default: dtk[k] = cm_group_bcast<Q, 3 / OWN>(dt_o[3 % OWN]); wk[k] = cm_group_bcast<Q, 3 / OWN>(w_o[3 % OWN]); break;
                }
                ypk[k] = 0.f;
            }
            dsum += (dtk[0] + dtk[1]) + (dtk[2] + dtk[3]);
            // --- 4 recurrence steps, states in chunks of <= 4 to bound live registers
            constexpr int SC = NS < 4 ? NS : 4;
#pragma unroll
            for (int i0 = 0; i0 < NS; i0 += SC) {
                float4 Bq[SC], Cq[SC];
#pragma unroll
                for (int i = 0; i < SC; ++i) {
                    Bq[i] = *reinterpret_cast<const float4 *>(Bt + (sg * NS + i0 + i) * BCROW + tb);
                    Cq[i] = *reinterpret_cast<const float4 *>(Ct + (sg * NS + i0 + i) * BCROW + tb);
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int k = REV ? 3 - kk : kk;
#pragma unroll
                    for (int i = 0; i < SC; ++i) {
                        const float bv = reinterpret_cast<const float *>(&Bq[i])[k];
                        const float cv = reinterpret_cast<const float *>(&Cq[i])[k];
                        const float a = cm_exp2(dtk[k] * Ap[i0 + i]);
                        h[i0 + i] = fmaf(a, h[i0 + i], wk[k] * bv);
                        ypk[k] = fmaf(cv, h[i0 + i], ypk[k]);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float y = cm_group_sum<S>(ypk[k]);
                if (q == k / OWN) y_o[k % OWN] = y;
            }
            // --- finalize: skip connection, gate; results overwrite the consumed LDS slots
            if (sg < Q) {
#pragma unroll
                for (int j = 0; j < OWN; ++j) {
                    const int tl = tb + q * OWN + j;
                    float o = fmaf(Dv, u_o[j], y_o[j]);
                    if (has_z) {
                        if (want_out) cm_elem<IO>::store(const_cast<IO *>(urow) + tl, o);
                        o *= z_o[j] * cm_sigmoid(z_o[j]);
                        cm_elem<IO>::store(const_cast<IO *>(zrow) + tl, o);
                    } else {
                        cm_elem<IO>::store(const_cast<IO *>(urow) + tl, o);
                    }
                }
            }
            // --- checkpoint at chunk ends (forward: last step of a chunk or of the sequence;
            //     reverse: first time index of a chunk)
            const int tg = t0 + tb;
            const bool ck = REV ? (tg % CM_SCAN_CHUNK == 0)
                                : (((tg + 4) % CM_SCAN_CHUNK == 0) || (tg + 4 >= L));
            if (ck) {
                if (p.x != nullptr && e_ok) {
                    float *xp = p.x + (((int64_t)b * dim + e) * nchunks + tg / CM_SCAN_CHUNK) * (2 * N) + 2 * sg * NS;
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        xp[2 * i] = cm_exp2(dsum * Ap[i]);
                        xp[2 * i + 1] = h[i];
                    }
                }
                dsum = 0.f;
            }
        }
        __syncthreads();
        // --- write the tile's outputs (wave-private LDS region -> global, coalesced vectors)
        IO *og = reinterpret_cast<IO *>(has_z ? p.out_z : p.out) + (int64_t)b * p.out_bs;
        IO *pg = (has_z && want_out) ? reinterpret_cast<IO *>(p.out) + (int64_t)b * p.out_bs : nullptr;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + 64 * i;
            const int row = v / VPR, col = (v % VPR) * VEC;
            const int er = ew0 + row, t = t0 + col;
            int nvalid = (er < dim) ? (L - t) : 0;
            nvalid = nvalid < 0 ? 0 : (nvalid > VEC ? VEC : nvalid);
            if (nvalid > 0) {
                const int o = row * ROWB + col * (int)sizeof(IO);
                const int64_t off = (int64_t)er * p.out_ds + t;
                store_vec_guarded<IO, VECOK>(og + off, *reinterpret_cast<const uint4 *>((has_z ? ztile : utile) + o), nvalid);
                if (pg) store_vec_guarded<IO, VECOK>(pg + off, *reinterpret_cast<const uint4 *>(utile + o), nvalid);
            }
        }
    }
}

template <typename IO, int S>
constexpr size_t scan_fwd_smem(int N) {
    return (size_t)kWaves * 3 * (64 / S) * (16 * S * sizeof(IO) + 16) + (size_t)2 * N * (16 * S + 4) * sizeof(float);
}

template <typename IO, typename BC, int S, int NS>
int launch_scan_fwd(const cm_scan_fwd_args &a, bool vecok) {
    constexpr int CPW = 64 / S;
    const size_t smem = scan_fwd_smem<IO, S>(S * NS);
    dim3 grid((a.dim + kWaves * CPW - 1) / (kWaves * CPW), a.batch);
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    auto go = [&](auto kern) -> int {
        if (smem > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) {
                cm_set_error("scan_fwd: hipFuncSetAttribute(%zu B LDS) failed: %s", smem, hipGetErrorString(e));
                return (int)e;
            }
        }
        hipLaunchKernelGGL(kern, grid, dim3(kThreads), smem, st, a);
        return cm_launch_status("cm_selective_scan_fwd");
    };
    if (vecok) {
        if (a.reverse_time) return go(scan_fwd_kernel<IO, BC, S, NS, true, true>);
        return go(scan_fwd_kernel<IO, BC, S, NS, false, true>);
    }
    if constexpr (S == 4) {   // unaligned tensors: element-wise staging, one lane split only
        if (a.reverse_time) return go(scan_fwd_kernel<IO, BC, S, NS, true, false>);
        return go(scan_fwd_kernel<IO, BC, S, NS, false, false>);
    }
    cm_set_error("scan_fwd: internal: unaligned path requested with S=%d", S);
    return CM_EUNSUPPORTED;
}


// per-dtype-pair dispatcher on (dstate, lane split S); explicit instantiations live in scan_fwd_*.hip
template <typename IO, typename BC>
int cm_scan_fwd_dispatch(const cm_scan_fwd_args &a, int S, bool vecok) {
#define CM_CASE(N_, S_) \
    if (a.dstate == (N_) && S == (S_)) return launch_scan_fwd<IO, BC, (S_), (N_) / (S_)>(a, vecok);
    CM_CASE(16, 1) CM_CASE(16, 2) CM_CASE(16, 4) CM_CASE(16, 8) CM_CASE(16, 16)
    CM_CASE(8, 1) CM_CASE(8, 2) CM_CASE(8, 4) CM_CASE(8, 8)
#undef CM_CASE
    cm_set_error("scan_fwd: no kernel for dstate=%d with lane split S=%d (supported dstate: 8, 16)", a.dstate, S);
    return CM_EUNSUPPORTED;
}

}  // namespace
