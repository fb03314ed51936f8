// scan_bwd_impl.h — selective-scan backward kernel template for gfx950 (MI355X).
//
// Contract: cm_selective_scan_bwd in include/conmamba_hip.h (replaces selective_scan_cuda.bwd,
// reference modules/mamba/selective_scan_interface.py:67/252).  Gradient formulas: SURVEY.md
// Appendix A; pinned by tests/golden/g2_scan_bwd.npz (autograd through selective_scan_ref).
//
// Same lane mapping as the forward (S lanes per channel, NS states per lane, time sequential in
// the lane).  The adjoint recurrence lambda_t = C_t g_t + a_{t+1} lambda_{t+1} runs against the
// forward's time direction and needs h_{t-1}; the forward stored h only at CM_SCAN_CHUNK
// boundaries, so each 64-step chunk is processed as
//   sweep 1: forward recompute through the chunk, keeping h at the start of every 8-step
//            sub-block (register stack),
//   per sub-block, last to first:
//     sweep 2: forward recompute of the 8 steps keeping a_t and h_t in registers,
//     sweep 3: the adjoint steps in reverse, producing du, ddelta, dz (in place in the LDS
//              tiles), per-lane dA/dD/ddelta_bias partial sums (registers, one atomic per lane at
//              the end) and dB/dC contributions.
// dB[n,t], dC[n,t] sum over channels: DPP reduction over the channel lanes of a 16-lane row,
// 16 partials per workgroup through LDS (double-buffered), then per (workgroup, n, t) either one
// fp32 atomic, or -- with a workspace -- a plain store of the workgroup's partial that
// scan_bwd_reduce_kernel sums over the workgroups in a fixed order (deterministic gradients); the
// per-channel dA / dD / ddelta_bias partials of a batch element are handled the same way.
#pragma once
#include "cm_common.h"

namespace {

constexpr int kBwdWaves = 4;
constexpr int kBwdThreads = kBwdWaves * 64;

template <typename IO>
__device__ __forceinline__ uint4 bwd_load_vec(const IO *src, int nvalid, bool vecok) {
    constexpr int VEC = cm_elem<IO>::kVec;
    uint4 r = {0u, 0u, 0u, 0u};
    if (nvalid <= 0) return r;
    if (vecok) return *reinterpret_cast<const uint4 *>(src);
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
    (void)VEC;
    return r;
}

template <typename IO>
__device__ __forceinline__ void bwd_store_vec(IO *dst, uint4 v, int nvalid, bool vecok) {
    if (nvalid <= 0) return;
    if (vecok) {
        *reinterpret_cast<uint4 *>(dst) = v;
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

// sum over the channel lanes of a 16-lane row that share one state group (lane stride S);
// the result is valid in lanes 16-S .. 15 of each row.
template <int S> __device__ __forceinline__ float row_channel_sum(float v) {
    if constexpr (S <= 8) v += cm_dpp<0x118>(v);                 // row_shr:8
    if constexpr (S <= 4) v += cm_dpp<0x114>(v);                 // row_shr:4  (applied to the shr:8 sums)
    return v;
}

template <typename IO, typename BC, int S, int NS, bool REV>
// Two waves per SIMD (<= 256 VGPRs) where that costs few spills: NS = 2 spills 25-39 registers and runs 526 us instead
// of 714 us per direction at 32 x 512 channels x 1000 steps; NS = 4 would spill 190 (940 us vs 572 us) and keeps one wave.
__global__ __launch_bounds__(kBwdThreads, NS <= 2 ? 2 : 1) void scan_bwd_kernel(const cm_scan_bwd_args p, int vecok) {
    static_assert(S == 4 || S == 8 || S == 16, "backward supports lane splits 4, 8, 16");
    constexpr int N = S * NS;
    constexpr int CPW = 64 / S;
    constexpr int CK = CM_SCAN_CHUNK;           // 64 timesteps per tile = one checkpoint chunk
    constexpr int SBK = 8, NSB = CK / SBK;
    constexpr int VEC = cm_elem<IO>::kVec;
    constexpr int VPR = CK / VEC;
    constexpr int ROWB = CK * (int)sizeof(IO) + 16;
    constexpr int BCROW = CK + 4;
    constexpr int NPART = kBwdWaves * 4;        // dB/dC partials per workgroup (one per 16-lane row)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const cm_scan_fwd_args &f = p.fwd;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = blockIdx.y, L = f.seqlen, dim = f.dim;
    const int ew0 = (blockIdx.x * kBwdWaves + wave) * CPW;
    const int c_local = lane / S, sg = lane % S, q = lane & 3;
    const int e = ew0 + c_local;
    const bool e_ok = e < dim;
    const int e_c = e_ok ? e : dim - 1;
    const bool has_z = f.z != nullptr;
    const bool softplus = f.delta_softplus != 0;
    const bool want_oz = has_z && f.out_z != nullptr;

    unsigned char *utile = smem + wave * (4 * CPW * ROWB);
    unsigned char *dtile = utile + CPW * ROWB;
    unsigned char *ztile = dtile + CPW * ROWB;
    unsigned char *gtile = ztile + CPW * ROWB;                   // dout, later out_z
    float *Bt = reinterpret_cast<float *>(smem + kBwdWaves * 4 * CPW * ROWB);
    float *Ct = Bt + N * BCROW;
    float *pbuf = Ct + N * BCROW;                                // [2 buffers][2 tensors][NPART][N*SBK]

    const IO *ug = reinterpret_cast<const IO *>(f.u) + (int64_t)b * f.u_bs;
    const IO *dg = reinterpret_cast<const IO *>(f.delta) + (int64_t)b * f.delta_bs;
    const IO *zg = has_z ? reinterpret_cast<const IO *>(f.z) + (int64_t)b * f.z_bs : nullptr;
    const IO *gg = reinterpret_cast<const IO *>(p.dout) + (int64_t)b * p.dout_bs;
    const BC *Bg = reinterpret_cast<const BC *>(f.B) + (int64_t)b * f.B_bs;
    const BC *Cg = reinterpret_cast<const BC *>(f.C) + (int64_t)b * f.C_bs;
    IO *dug = reinterpret_cast<IO *>(p.du) + (int64_t)b * p.du_bs;
    IO *ddg = reinterpret_cast<IO *>(p.ddelta) + (int64_t)b * p.ddelta_bs;
    IO *dzg = has_z ? reinterpret_cast<IO *>(p.dz) + (int64_t)b * p.dz_bs : nullptr;
    IO *ozg = want_oz ? reinterpret_cast<IO *>(f.out_z) + (int64_t)b * f.out_bs : nullptr;

    float Ap[NS], Av[NS], lamc[NS], dAacc[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        Av[i] = f.A[(int64_t)e_c * N + sg * NS + i];
        Ap[i] = Av[i] * CM_LOG2E;
        lamc[i] = 0.f;
        dAacc[i] = 0.f;
    }
    const float Dv = f.D ? f.D[e_c] : 0.f;
    const float bias = f.delta_bias ? f.delta_bias[e_c] : 0.f;
    float dDacc = 0.f, dbacc = 0.f;
    const int nchunks = (L + CK - 1) / CK;
    int pb_sel = 0;
    // deterministic path: per-workgroup partials (layout: scan_bwd_ws)
    float *wsBC = reinterpret_cast<float *>(p.workspace);        // [2][gridDim.x][batch][N][L]
    const int64_t ws_bc_stride = (int64_t)gridDim.x * gridDim.y * N * L;

    for (int ic = 0; ic < nchunks; ++ic) {
        const int c = REV ? ic : nchunks - 1 - ic;               // chunks in descending processing order
        const int t0 = c * CK;
        __syncthreads();                                         // previous chunk's tile reads are done
        // ---------------- stage the chunk's tiles
        for (int v = lane; v < CPW * VPR; v += 64) {
            const int row = v / VPR, col = (v % VPR) * VEC;
            const int er = ew0 + row, t = t0 + col;
            int nvalid = (er < dim) ? (L - t) : 0;
            nvalid = nvalid < 0 ? 0 : (nvalid > VEC ? VEC : nvalid);
            const int o = row * ROWB + col * (int)sizeof(IO);
            *reinterpret_cast<uint4 *>(utile + o) = bwd_load_vec<IO>(ug + (int64_t)er * f.u_ds + t, nvalid, vecok);
            *reinterpret_cast<uint4 *>(dtile + o) = bwd_load_vec<IO>(dg + (int64_t)er * f.delta_ds + t, nvalid, vecok);
            *reinterpret_cast<uint4 *>(gtile + o) = bwd_load_vec<IO>(gg + (int64_t)er * p.dout_ds + t, nvalid, vecok);
            if (has_z)
                *reinterpret_cast<uint4 *>(ztile + o) = bwd_load_vec<IO>(zg + (int64_t)er * f.z_ds + t, nvalid, vecok);
        }
        for (int g4 = tid; g4 < N * CK / 4; g4 += kBwdThreads) {
            const int n = g4 / (CK / 4), tq = (g4 % (CK / 4)) * 4;
            float4 vb = {0.f, 0.f, 0.f, 0.f}, vc = {0.f, 0.f, 0.f, 0.f};
            float *fb = reinterpret_cast<float *>(&vb), *fc = reinterpret_cast<float *>(&vc);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (t0 + tq + j < L) {
                    fb[j] = cm_elem<BC>::load(Bg + (int64_t)n * f.B_ns + t0 + tq + j);
                    fc[j] = cm_elem<BC>::load(Cg + (int64_t)n * f.C_ns + t0 + tq + j);
                }
            }
            *reinterpret_cast<float4 *>(Bt + n * BCROW + tq) = vb;
            *reinterpret_cast<float4 *>(Ct + n * BCROW + tq) = vc;
        }
        // state entering the chunk (in the scan's own direction)
        float h[NS];
        {
            const int cprev = REV ? c + 1 : c - 1;
            const bool have = e_ok && cprev >= 0 && cprev < nchunks;
            const float *xp = f.x + (((int64_t)b * dim + e_c) * nchunks + (have ? cprev : 0)) * (2 * N) + 2 * sg * NS;
#pragma unroll
            for (int i = 0; i < NS; ++i) h[i] = have ? xp[2 * i + 1] : 0.f;
        }
        __syncthreads();

        const IO *urow = reinterpret_cast<const IO *>(utile + c_local * ROWB);
        const IO *drow = reinterpret_cast<const IO *>(dtile + c_local * ROWB);
        const IO *zrow = reinterpret_cast<const IO *>(ztile + c_local * ROWB);
        const IO *grow = reinterpret_cast<const IO *>(gtile + c_local * ROWB);

        // per-(channel, t) inputs of one 4-step block, for the timestep this lane owns (tile-local tb+q)
        auto own_dt_w = [&](int tb, float &dt, float &w) {
            const int tl = tb + q;
            const bool valid = t0 + tl < L;
            float dv = cm_elem<IO>::load(drow + tl) + bias;
            if (softplus) dv = cm_softplus(dv);
            dt = valid ? dv : 0.f;
            w = valid ? dt * cm_elem<IO>::load(urow + tl) : 0.f;
        };

        // ---------------- sweep 1: states at the start of every sub-block (register stack)
        float hsb[NSB][NS];
        for (int sb = 0; sb < NSB; ++sb) {
#pragma unroll
            for (int k = 0; k < NSB - 1; ++k)
#pragma unroll
                for (int i = 0; i < NS; ++i) hsb[k][i] = hsb[k + 1][i];
#pragma unroll
            for (int i = 0; i < NS; ++i) hsb[NSB - 1][i] = h[i];
            if (sb == NSB - 1) break;                            // the last sub-block's end state is not needed
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) {
                const int pb = sb * 2 + b2;
                const int tb = (REV ? 15 - pb : pb) * 4;
                float dt_o, w_o;
                own_dt_w(tb, dt_o, w_o);
                float4 Bq[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) Bq[i] = *reinterpret_cast<const float4 *>(Bt + (sg * NS + i) * BCROW + tb);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int k = REV ? 3 - kk : kk;
                    float dt, w;
                    switch (k) {
                        case 0: dt = cm_group_bcast<4, 0>(dt_o); w = cm_group_bcast<4, 0>(w_o); break;
                        case 1: dt = cm_group_bcast<4, 1>(dt_o); w = cm_group_bcast<4, 1>(w_o); break;
                        case 2: dt = cm_group_bcast<4, 2>(dt_o); w = cm_group_bcast<4, 2>(w_o); break;
                        default: dt = cm_group_bcast<4, 3>(dt_o); w = cm_group_bcast<4, 3>(w_o); break;
                    }
#pragma unroll
                    for (int i = 0; i < NS; ++i)
                        h[i] = fmaf(cm_exp2(dt * Ap[i]), h[i], w * reinterpret_cast<const float *>(&Bq[i])[k]);
                }
            }
        }

        // ---------------- sub-blocks, last processed first
        for (int sbi = NSB - 1; sbi >= 0; --sbi) {
            float hstart[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) hstart[i] = hsb[NSB - 1][i];
#pragma unroll
            for (int k = NSB - 1; k > 0; --k)                     // pop the stack
#pragma unroll
                for (int i = 0; i < NS; ++i) hsb[k][i] = hsb[k - 1][i];

            // owner inputs for the two blocks of this sub-block
            float dt_o[2], w_o[2], u_o[2], g_o[2], z_o[2], go_o[2], pre_o[2], y_o[2];
            int tbv[2];
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) {
                const int pb = sbi * 2 + b2;
                const int tb = (REV ? 15 - pb : pb) * 4;
                tbv[b2] = tb;
                const int tl = tb + q;
                const bool valid = t0 + tl < L;
                float pre = cm_elem<IO>::load(drow + tl) + bias;
                float dv = softplus ? cm_softplus(pre) : pre;
                float uv = cm_elem<IO>::load(urow + tl);
                float gv = cm_elem<IO>::load(grow + tl);
                float zv = has_z ? cm_elem<IO>::load(zrow + tl) : 0.f;
                pre_o[b2] = pre;
                dt_o[b2] = valid ? dv : 0.f;
                u_o[b2] = valid ? uv : 0.f;
                w_o[b2] = dt_o[b2] * u_o[b2];
                go_o[b2] = valid ? gv : 0.f;
                z_o[b2] = zv;
                g_o[b2] = has_z ? go_o[b2] * zv * cm_sigmoid(zv) : go_o[b2];
                y_o[b2] = 0.f;
            }

            // ---- sweep 2: forward through the 8 steps, keep a_t, h_t
            float ast[SBK][NS], hst[SBK][NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) h[i] = hstart[i];
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) {
                const int tb = tbv[b2];
                float4 Bq[NS], Cq[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    Bq[i] = *reinterpret_cast<const float4 *>(Bt + (sg * NS + i) * BCROW + tb);
                    Cq[i] = *reinterpret_cast<const float4 *>(Ct + (sg * NS + i) * BCROW + tb);
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int k = REV ? 3 - kk : kk;
                    const int s = b2 * 4 + kk;
                    float dt, w;
                    switch (k) {
                        case 0: dt = cm_group_bcast<4, 0>(dt_o[b2]); w = cm_group_bcast<4, 0>(w_o[b2]); break;
                        case 1: dt = cm_group_bcast<4, 1>(dt_o[b2]); w = cm_group_bcast<4, 1>(w_o[b2]); break;
                        case 2: dt = cm_group_bcast<4, 2>(dt_o[b2]); w = cm_group_bcast<4, 2>(w_o[b2]); break;
                        default: dt = cm_group_bcast<4, 3>(dt_o[b2]); w = cm_group_bcast<4, 3>(w_o[b2]); break;
                    }
                    float yp = 0.f;
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        const float a = cm_exp2(dt * Ap[i]);
                        h[i] = fmaf(a, h[i], w * reinterpret_cast<const float *>(&Bq[i])[k]);
                        ast[s][i] = a;
                        hst[s][i] = h[i];
                        yp = fmaf(reinterpret_cast<const float *>(&Cq[i])[k], h[i], yp);
                    }
                    if (has_z) {
                        const float y = cm_group_sum<S>(yp);
                        if (q == k) y_o[b2] = y;
                    }
                }
            }

            // ---- sweep 3: adjoint steps, reverse processing order
            float *pB = pbuf + (pb_sel * 2 + 0) * NPART * (N * SBK);
            float *pC = pbuf + (pb_sel * 2 + 1) * NPART * (N * SBK);
            const int part = wave * 4 + (lane >> 4);
            const int tbs = REV ? 56 - 8 * sbi : 8 * sbi;         // lowest tile-local time of the sub-block
#pragma unroll
            for (int b2 = 1; b2 >= 0; --b2) {
                const int tb = tbv[b2];
                float4 Bq[NS], Cq[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    Bq[i] = *reinterpret_cast<const float4 *>(Bt + (sg * NS + i) * BCROW + tb);
                    Cq[i] = *reinterpret_cast<const float4 *>(Ct + (sg * NS + i) * BCROW + tb);
                }
                float4 accB[NS], accC[NS];                      // dB / dC contributions for tile-local tb..tb+3
                float sb_o = 0.f, sa_o = 0.f;                    // owner's sums over states
#pragma unroll
                for (int kk = 3; kk >= 0; --kk) {
                    const int k = REV ? 3 - kk : kk;
                    const int s = b2 * 4 + kk;
                    float dt, w, g;
                    switch (k) {
                        case 0: dt = cm_group_bcast<4, 0>(dt_o[b2]); w = cm_group_bcast<4, 0>(w_o[b2]); g = cm_group_bcast<4, 0>(g_o[b2]); break;
                        case 1: dt = cm_group_bcast<4, 1>(dt_o[b2]); w = cm_group_bcast<4, 1>(w_o[b2]); g = cm_group_bcast<4, 1>(g_o[b2]); break;
                        case 2: dt = cm_group_bcast<4, 2>(dt_o[b2]); w = cm_group_bcast<4, 2>(w_o[b2]); g = cm_group_bcast<4, 2>(g_o[b2]); break;
                        default: dt = cm_group_bcast<4, 3>(dt_o[b2]); w = cm_group_bcast<4, 3>(w_o[b2]); g = cm_group_bcast<4, 3>(g_o[b2]); break;
                    }
                    float sbl = 0.f, sal = 0.f;
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        const float Bv = reinterpret_cast<const float *>(&Bq[i])[k];
                        const float Cv = reinterpret_cast<const float *>(&Cq[i])[k];
                        const float lam = fmaf(Cv, g, lamc[i]);
                        const float hprev = s > 0 ? hst[s > 0 ? s - 1 : 0][i] : hstart[i];
                        reinterpret_cast<float *>(&accC[i])[k] = g * hst[s][i];
                        reinterpret_cast<float *>(&accB[i])[k] = lam * w;
                        sbl = fmaf(lam, Bv, sbl);
                        const float r = lam * hprev * ast[s][i];
                        dAacc[i] = fmaf(r, dt, dAacc[i]);
                        sal = fmaf(r, Av[i], sal);
                        lamc[i] = lam * ast[s][i];
                    }
                    const float sbs = cm_group_sum<S>(sbl);
                    const float sas = cm_group_sum<S>(sal);
                    if (q == k) { sb_o = sbs; sa_o = sas; }
                }
                // channel reduction of the block's dB/dC contributions -> LDS partials
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    float4 rb, rc;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        reinterpret_cast<float *>(&rb)[k] = row_channel_sum<S>(reinterpret_cast<const float *>(&accB[i])[k]);
                        reinterpret_cast<float *>(&rc)[k] = row_channel_sum<S>(reinterpret_cast<const float *>(&accC[i])[k]);
                    }
                    if ((lane & 15) >= 16 - S) {
                        const int o = part * (N * SBK) + (sg * NS + i) * SBK + (tb - tbs);
                        *reinterpret_cast<float4 *>(pB + o) = rb;
                        *reinterpret_cast<float4 *>(pC + o) = rc;
                    }
                }
                // owner finalises its timestep of this block: du, ddelta, dz (+ recomputed out_z)
                {
                    const int tl = tb + q;
                    const bool valid = t0 + tl < L;
                    const float g = g_o[b2], u = u_o[b2], dt = dt_o[b2];
                    const float du = fmaf(Dv, g, sb_o * dt);
                    const float ddt = fmaf(sb_o, u, sa_o);
                    float dd = ddt;
                    if (softplus) dd = pre_o[b2] > 20.f ? ddt : ddt * cm_sigmoid(pre_o[b2]);
                    dd = valid ? dd : 0.f;
                    if (sg < 4) {
                        dDacc = fmaf(g, u, dDacc);
                        dbacc += dd;
                        cm_elem<IO>::store(const_cast<IO *>(urow) + tl, du);
                        cm_elem<IO>::store(const_cast<IO *>(drow) + tl, dd);
                        if (has_z) {
                            const float zv = z_o[b2];
                            const float sgm = cm_sigmoid(zv);
                            const float y = fmaf(Dv, u, y_o[b2]);
                            cm_elem<IO>::store(const_cast<IO *>(zrow) + tl, go_o[b2] * y * sgm * (1.f + zv * (1.f - sgm)));
                            if (want_oz) cm_elem<IO>::store(const_cast<IO *>(grow) + tl, y * zv * sgm);
                        }
                    }
                }
            }
            // ---- flush this sub-block's dB/dC partials: one atomic per (workgroup, n, t)
            __syncthreads();
            if (tid < 2 * N * SBK) {
                const int which = tid / (N * SBK);
                const int r = tid % (N * SBK);
                const int n = r / SBK, j = r % SBK;
                const float *src = pbuf + (pb_sel * 2 + which) * NPART * (N * SBK) + r;
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < NPART; ++k) acc += src[k * (N * SBK)];
                const int t = t0 + tbs + j;
                if (t < L) {
                    if (wsBC) wsBC[which * ws_bc_stride + (((int64_t)blockIdx.x * gridDim.y + b) * N + n) * L + t] = acc;
                    else atomicAdd((which ? p.dC : p.dB) + ((int64_t)b * N + n) * L + t, acc);
                }
            }
            pb_sel ^= 1;
        }
        __syncthreads();
        // ---------------- write du / ddelta / dz (/ out_z) tiles
        for (int v = lane; v < CPW * VPR; v += 64) {
            const int row = v / VPR, col = (v % VPR) * VEC;
            const int er = ew0 + row, t = t0 + col;
            int nvalid = (er < dim) ? (L - t) : 0;
            nvalid = nvalid < 0 ? 0 : (nvalid > VEC ? VEC : nvalid);
            const int o = row * ROWB + col * (int)sizeof(IO);
            bwd_store_vec<IO>(dug + (int64_t)er * p.du_ds + t, *reinterpret_cast<const uint4 *>(utile + o), nvalid, vecok);
            bwd_store_vec<IO>(ddg + (int64_t)er * p.ddelta_ds + t, *reinterpret_cast<const uint4 *>(dtile + o), nvalid, vecok);
            if (has_z) {
                bwd_store_vec<IO>(dzg + (int64_t)er * p.dz_ds + t, *reinterpret_cast<const uint4 *>(ztile + o), nvalid, vecok);
                if (want_oz)
                    bwd_store_vec<IO>(ozg + (int64_t)er * f.out_ds + t, *reinterpret_cast<const uint4 *>(gtile + o), nvalid, vecok);
            }
        }
    }
    // ---------------- per-channel parameter gradients
    if (wsBC) {
        float *wsA = wsBC + 2 * ws_bc_stride;                    // [batch][dim][N]
        float *wsD = wsA + (int64_t)gridDim.y * dim * N;         // [batch][dim]
        float *wsb = wsD + (int64_t)gridDim.y * dim;             // [batch][dim]
        const float dDq = cm_group_sum<4>(dDacc), dbq = cm_group_sum<4>(dbacc);   // the 4 owner lanes (sg 0..3) of the channel
        if (e_ok) {
#pragma unroll
            for (int i = 0; i < NS; ++i) wsA[((int64_t)b * dim + e) * N + sg * NS + i] = dAacc[i];
            if (sg == 0) {
                wsD[(int64_t)b * dim + e] = dDq;
                wsb[(int64_t)b * dim + e] = dbq;
            }
        }
    } else if (e_ok) {
#pragma unroll
        for (int i = 0; i < NS; ++i) atomicAdd(p.dA + (int64_t)e * N + sg * NS + i, dAacc[i]);
        if (sg < 4) {
            if (p.dD) atomicAdd(p.dD + e, dDacc);
            if (p.ddelta_bias) atomicAdd(p.ddelta_bias + e, dbacc);
        }
    }
}

// second pass of the deterministic path: fixed-order sums of the per-workgroup partials, ACCUMULATED into the outputs
__global__ __launch_bounds__(256) void scan_bwd_reduce_kernel(const cm_scan_bwd_args p, const int gx) {
    const cm_scan_fwd_args &f = p.fwd;
    const int64_t nbc = (int64_t)f.batch * f.dstate * f.seqlen, na = (int64_t)f.dim * f.dstate;
    const float *wsBC = reinterpret_cast<const float *>(p.workspace);
    const int64_t ws_bc_stride = (int64_t)gx * nbc;
    const float *wsA = wsBC + 2 * ws_bc_stride, *wsD = wsA + (int64_t)f.batch * na, *wsb = wsD + (int64_t)f.batch * f.dim;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * nbc + na + 2 * f.dim; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < 2 * nbc) {                                       // dB, dC: sum over the channel-tile workgroups
            const int which = i >= nbc;
            const int64_t j = i - which * nbc;
            float acc = 0.f;
            for (int g = 0; g < gx; ++g) acc += wsBC[which * ws_bc_stride + (int64_t)g * nbc + j];
            (which ? p.dC : p.dB)[j] += acc;
        } else if (i < 2 * nbc + na) {                           // dA: sum over the batch
            const int64_t j = i - 2 * nbc;
            float acc = 0.f;
            for (int b = 0; b < f.batch; ++b) acc += wsA[(int64_t)b * na + j];
            p.dA[j] += acc;
        } else {                                                 // dD, ddelta_bias: sum over the batch
            const int64_t j = i - 2 * nbc - na;
            const int which = j >= f.dim;
            const int e = (int)(j - which * f.dim);
            float *dst = which ? p.ddelta_bias : p.dD;
            if (!dst) continue;
            const float *src = which ? wsb : wsD;
            float acc = 0.f;
            for (int b = 0; b < f.batch; ++b) acc += src[(int64_t)b * f.dim + e];
            dst[e] += acc;
        }
    }
}

// workspace bytes of the deterministic path for a launch with `gx` channel-tile workgroups per batch element
inline int64_t scan_bwd_ws_bytes(const cm_scan_fwd_args &f, int gx) {
    return 4 * (2 * (int64_t)gx * f.batch * f.dstate * f.seqlen + (int64_t)f.batch * f.dim * f.dstate + 2 * (int64_t)f.batch * f.dim);
}

template <typename IO, int S>
constexpr size_t scan_bwd_smem(int N) {
    return (size_t)kBwdWaves * 4 * (64 / S) * (CM_SCAN_CHUNK * sizeof(IO) + 16) +
           (size_t)2 * N * (CM_SCAN_CHUNK + 4) * sizeof(float) + (size_t)2 * 2 * (kBwdWaves * 4) * N * 8 * sizeof(float);
}

template <typename IO, typename BC, int S, int NS>
int launch_scan_bwd(const cm_scan_bwd_args &a, bool vecok) {
    constexpr int CPW = 64 / S;
    const size_t smem = scan_bwd_smem<IO, S>(S * NS);
    dim3 grid((a.fwd.dim + kBwdWaves * CPW - 1) / (kBwdWaves * CPW), a.fwd.batch);
    hipStream_t st = reinterpret_cast<hipStream_t>(a.fwd.stream);
    if (a.workspace && a.workspace_bytes < scan_bwd_ws_bytes(a.fwd, (int)grid.x)) {
        cm_set_error("scan_bwd: workspace of %lld bytes is smaller than the %lld this launch needs", (long long)a.workspace_bytes,
                     (long long)scan_bwd_ws_bytes(a.fwd, (int)grid.x));
        return CM_EINVAL;
    }
    auto go = [&](auto kern) -> int {
        if (smem > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) {
                cm_set_error("scan_bwd: hipFuncSetAttribute(%zu B LDS) failed: %s", smem, hipGetErrorString(e));
                return (int)e;
            }
        }
        hipLaunchKernelGGL(kern, grid, dim3(kBwdThreads), smem, st, a, (int)vecok);
        if (a.workspace) {
            const int64_t n = 2 * (int64_t)a.fwd.batch * a.fwd.dstate * a.fwd.seqlen + (int64_t)a.fwd.dim * a.fwd.dstate + 2 * a.fwd.dim;
            const int64_t blocks = (n + 255) / 256;
            hipLaunchKernelGGL(scan_bwd_reduce_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, st, a, (int)grid.x);
        }
        return cm_launch_status("cm_selective_scan_bwd");
    };
    if (a.fwd.reverse_time) return go(scan_bwd_kernel<IO, BC, S, NS, true>);
    return go(scan_bwd_kernel<IO, BC, S, NS, false>);
}

template <typename IO, typename BC>
int cm_scan_bwd_dispatch(const cm_scan_bwd_args &a, int S, bool vecok) {
#define CM_CASE(N_, S_) \
    if (a.fwd.dstate == (N_) && S == (S_)) return launch_scan_bwd<IO, BC, (S_), (N_) / (S_)>(a, vecok);
    CM_CASE(16, 4) CM_CASE(16, 8) CM_CASE(16, 16)
    CM_CASE(8, 4) CM_CASE(8, 8)
#undef CM_CASE
    cm_set_error("scan_bwd: no kernel for dstate=%d with lane split S=%d (supported dstate: 8, 16)", a.fwd.dstate, S);
    return CM_EUNSUPPORTED;
}

}  // namespace
