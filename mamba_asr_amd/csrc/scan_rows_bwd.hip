// scan_rows_bwd.hip — channels-last selective scan BACKWARD on the forward's row-group design (contract: cm_scan_cl_bwd;
// reference semantics: selective_scan_cuda.bwd as MambaInnerFnNoOutProj.backward calls it, selective_scan_interface.py:252-256,
// plus the dt_proj gradients of :258-279; gradient formulas SURVEY.md Appendix A, pinned by tests/golden/g2_scan_bwd.npz through
// the oracle).
//
// One workgroup = 4 waves = 64 channels of one (sequence, direction); both BiMamba directions in one launch.  The forward saved
// the state entering every 8-step half block (cm_scan_cl_dir.ckpt), so a block costs ONE recompute sweep and ONE adjoint sweep,
// run as two halves of 8 steps so that the (a_t, h_t) of a sweep fit in registers (16 steps: 128 registers, 88-179 spilled):
//   owner phase  (lane = step s of the block, 4 channels 4g..4g+3 of the wave's 16): delta = W_dt . dt on the matrix pipe
//       (operands swapped w.r.t. the forward, so that the result lands one (step, 4 channels) per lane and u / z / dout / ypre
//       are ONE 8-byte global load per lane and tensor -- no LDS staging of the big tensors), softplus, gate; (delta',
//       delta' u, dout silu(z)) go to the recurrence lanes through a per-wave LDS patch;
//   recurrence   (lane = (state quad q, channel cl of row g): 4 states of one channel, packed fp32): sweep 1 recomputes h_t
//       from the checkpoint keeping a_t and h_t of the 8 steps in registers; sweep 2 runs the adjoint lambda_t = C_t g_t +
//       a_{t+1} lambda_{t+1} backwards.  a_t h_{t-1} is h_t - w_t B_t, so no second copy of the state is kept.
//       Sums over the 16 states of a channel (-> du, ddelta) are two quad DPP adds; sums over channels (dB, dC) are two
//       row DPP adds over the 4 channels of a 16-lane row, then the 4 rows and the 4 waves through LDS, then one fp32 row of
//       48 partial sums per (workgroup, step) to the workspace;
//   epilogue     (owner layout again): du, dz, ddelta -> d dt (contracted with W_dt on the matrix pipe, into the same
//       workspace row) and ddt_weight (accumulated in MFMA registers over the whole sequence), dD / ddelta_bias / dA in
//       registers; 8-byte stores of du / dz.
// scan_rows_bwd_reduce_kernel then sums the workspace in a fixed order: dxdbl rows over the dim / 64 channel groups, the
// parameter gradients over the batch.
#include "scan_rows_common.h"

namespace {

constexpr int PIN = 17 * 4;   // floats per channel in the (delta', delta' u, g) patch: 16 steps x float4 + 16 bytes (bank spread)
constexpr int POUT = 17 * 2;  // floats per channel in the (sum lam B, sum r A) patch
constexpr int TRS = 17;       // floats per channel in the ddelta transposition patch

template <int DTR> struct bwd_lds {
    static constexpr int RW = DTR + 32;
    static constexpr int kXT = TB * XS * 4;                                    // staged x_dbl tile (raw dt columns + fp32 B, C)
    static constexpr int XSB = XS;                                             // floats per staged row
    static constexpr int kX = 0;
    static constexpr int kWave = 2 * kXT;                                      // two x tiles, then the per-wave region
    static constexpr int kPin = 0, kPout = 16 * PIN * 4, kRed = kPout + 16 * POUT * 4, kTr = kRed + 4 * 4 * 32 * 4;
    static constexpr int kWaveBytes = kTr + 16 * TRS * 4;
    static constexpr int kXw = kWave + 4 * kWaveBytes;                         // cross-wave partial rows: [wave][16 steps][RW]
    static constexpr int kBytes = kXw + 4 * TB * RW * 4;
};

// Time chunks (launches of fewer than 512 workgroups leave CUs idle: config 5 trains 4 x 160 s per GPU = 128 workgroups at
// E = 1024): the adjoint lambda_t = C_t g_t + a_{t+1} lambda_{t+1} is linear in what enters a chunk from later steps,
// L_out = P L_in + E with P = prod a_t over the chunk and E the carry the chunk produces from L_in = 0.  Pass 1 (SUM) runs only
// the adjoint of every chunk from zero (no state recompute, no outputs: a_t, C_t g_t), rows_bwd_carry_kernel folds (P, E) against
// the scan, pass 2 is the full kernel per chunk started from its L_in; the forward's checkpoints give every chunk its states.
struct bwd_plan {
    int nx;                    // 64-channel groups
    int chunks, chunk_len;     // chunk_len: multiple of TB
    int pass;                  // 0 unchunked, 1 summaries, 2 full from carried-in adjoints
    float *wsx;                // (ndir, nx, batch, seqlen, RW) fp32 partial dxdbl rows
    float *wsp;                // (ndir, batch, chunks, dim, 16 + P + 4) fp32 per-(sequence, chunk) parameter gradients
    float *lin, *le, *lp;      // (ndir, batch, chunks, dim, 16) fp32: adjoint entering a chunk, its E, its P
};

__device__ __forceinline__ u32x2 ld8(const __amdgpu_buffer_rsrc_t r, int off) {
    return __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
}

// 4 consecutive channels of one row: bf16 (8 bytes) or fp32 (16 bytes)
template <typename IO> struct quad_io;
template <> struct quad_io<cm_bf16> {
    u32x2 v;
    __device__ __forceinline__ void load(const __amdgpu_buffer_rsrc_t r, int off) { v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0); }
    __device__ __forceinline__ float get(int i) const { return (i & 1) ? cm_bf16_hi(v[i >> 1]) : cm_bf16_lo(v[i >> 1]); }
    static __device__ __forceinline__ void store(const __amdgpu_buffer_rsrc_t r, int off, const float *x) {
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{cm_pack_bf16(x[0], x[1]), cm_pack_bf16(x[2], x[3])}, r, off, 0, 0);
    }
};
template <> struct quad_io<float> {
    u32x4 v;
    __device__ __forceinline__ void load(const __amdgpu_buffer_rsrc_t r, int off) { v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); }
    __device__ __forceinline__ float get(int i) const { return __uint_as_float(v[i]); }
    static __device__ __forceinline__ void store(const __amdgpu_buffer_rsrc_t r, int off, const float *x) {
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])}, r, off, 0, 0);
    }
};

// sum over the 4 channel lanes of a 16-lane row that share one state quad (lane stride 4); valid in lanes 12..15 of the row
__device__ __forceinline__ float row_quad_channel_sum(float v) {
    v += cm_dpp<0x118>(v);                 // row_shr:8
    v += cm_dpp<0x114>(v);                 // row_shr:4
    return v;
}

// Runs steps [t_lo, t_lo + T) of sequence b.  lam_in: (dim, 16) adjoint carry entering the range from later scan steps, or NULL.
// SUM: summary pass -- only the adjoint recurrence; writes e_out (its carry from zero) and p_out (decay product), (dim, 16) each.
template <typename IO, bool REV, int DTR, bool SUM>
__device__ __forceinline__ void scan_rows_bwd(const cm_scan_cl_bwd_args &p, const cm_scan_cl_bwd_dir &d, unsigned char *lds,
                                              const int cx, const int b, const int t_lo, const int T, float *wsx, float *wsp,
                                              const float *lam_in, float *e_out, float *p_out) {
    using L = bwd_lds<DTR>;
    constexpr int S = (int)sizeof(IO);
    constexpr int VEC = cm_elem<IO>::kVec;
    constexpr int RW = DTR + 32;
    constexpr int XCPR = RW / VEC;               // 16-byte chunks per x_dbl row
    constexpr int DTC = DTR / 8;                 // bf16 I/O: chunks of raw dt columns per row
    constexpr int XSB = L::XSB;
    constexpr int NP = 16 + DTR + 4;             // per-channel parameter gradients: dA (16) | ddt_weight (DTR) | dD | ddelta_bias | 2 pad (16-byte rows)
    static_assert(DTR == 16 || (DTR == 32 && sizeof(IO) == 2), "dt_rank 32 needs bf16 I/O");
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s16 = lane & 15, g = lane >> 4, q = lane & 3, cl = (lane >> 2) & 3;
    const int E = p.dim, c0 = cx * 64;
    const int nblk = (T + TB - 1) / TB, nblk_seq = (p.seqlen + TB - 1) / TB;
    constexpr int DIR = REV ? -1 : 1;
    // owner layout: step s16 of the block, channels co .. co + 3;  recurrence layout: channel cr, states 4q .. 4q + 3
    const int co = c0 + 16 * w + 4 * g, cr = co + cl;
    const bool ok = co < E;                                          // dim % 4 == 0: a channel quad is in range or not
    const int coc = ok ? co : E - 4, crc = ok ? cr : E - 1;
    const int u_ts = (int)d.u_ts, z_ts = (int)p.z_ts, g_ts = (int)d.dout_ts, y_ts = (int)d.ypre_ts, x_ts = (int)d.xdbl_ts;
    const int du_ts = (int)d.du_ts, dz_ts = (int)d.dz_ts;
    auto rs = [&](const void *base, int64_t bs, int ts, int width) {
        return make_rsrc(reinterpret_cast<const IO *>(base) + (int64_t)b * bs + (int64_t)t_lo * ts, ((int64_t)(T - 1) * ts + width) * S);
    };
    const __amdgpu_buffer_rsrc_t ur = rs(d.u, d.u_bs, u_ts, E), zr = rs(p.z, p.z_bs, z_ts, E), gr = rs(d.dout, d.dout_bs, g_ts, E),
                                 yr = rs(d.ypre, d.ypre_bs, y_ts, E), xr = rs(d.xdbl, d.xdbl_bs, x_ts, RW),
                                 dur = rs(d.du, d.du_bs, du_ts, E), dzr = rs(d.dz, d.dz_bs, dz_ts, E);
    // blocks are visited against the scan: the forward's last block first
    const int tbb0 = (REV ? 0 : nblk - 1) * TB;
    constexpr int BDIR = -DIR;                                        // base step moves by BDIR * TB per iteration

    float *const xw = reinterpret_cast<float *>(lds + L::kXw);
    unsigned char *const wl = lds + L::kWave + w * L::kWaveBytes;
    float *const pin = reinterpret_cast<float *>(wl + L::kPin);
    float *const pout = reinterpret_cast<float *>(wl + L::kPout);
    float *const red = reinterpret_cast<float *>(wl + L::kRed);
    float *const tr = reinterpret_cast<float *>(wl + L::kTr);

    // ---- x_dbl tile staging (as the forward: raw bf16 dt columns in front, B / C widened to fp32)
    const bool x_thread = tid < TB * XCPR;
    int x_off = x_thread ? ((tbb0 + tid / XCPR) * x_ts + (tid % XCPR) * VEC) * S : 0x7fffffff;
    const int x_step = x_thread ? BDIR * TB * x_ts * S : 0;
    const bool x_raw = S == 2 && tid % XCPR < DTC;
    // staged row (floats): bf16: [raw dt: DTR/2 floats][B 16][C 16]; fp32: [dt 16][B 16][C 16]
    constexpr int BOFF = (S == 2) ? DTR / 2 : DTR;                   // float offset of B in a staged row
    const int x_lds = (tid / XCPR) * XSB * 4 + (x_raw ? (tid % XCPR) * 16 : (S == 2 ? (BOFF + (tid % XCPR - DTC) * VEC) * 4 : (tid % XCPR) * VEC * 4));
    u32x4 rx;
    // ---- owner-layout row loads (one per tensor per block) and the block's entry state
    int r_u = ok ? ((tbb0 + s16) * u_ts + co) * S : 0x7fffffff, r_z = ok ? ((tbb0 + s16) * z_ts + co) * S : 0x7fffffff;
    int r_g = ok ? ((tbb0 + s16) * g_ts + co) * S : 0x7fffffff, r_y = ok ? ((tbb0 + s16) * y_ts + co) * S : 0x7fffffff;
    int w_u = ok ? ((tbb0 + s16) * du_ts + co) * S : 0x7fffffff, w_z = ok ? ((tbb0 + s16) * dz_ts + co) * S : 0x7fffffff;
    const int s_u = ok ? BDIR * TB * u_ts * S : 0, s_z = ok ? BDIR * TB * z_ts * S : 0, s_g = ok ? BDIR * TB * g_ts * S : 0,
              s_y = ok ? BDIR * TB * y_ts * S : 0, sw_u = ok ? BDIR * TB * du_ts * S : 0, sw_z = ok ? BDIR * TB * dz_ts * S : 0;
    // (batch, 2 nblk, dim, 16): entry state of every half block of 8 steps [8 m, 8 m + 8), in scan order
    const float *ckp = d.ckpt + (((int64_t)b * 2 * nblk_seq + 2 * ((t_lo + tbb0) / TB)) * E + crc) * 16 + 4 * q;
    const int64_t ck_step = (int64_t)BDIR * 2 * E * 16;
    quad_io<IO> nu, nz, ng, ny;
    float4 nh[2];                                                    // [time half of the block]
    auto issue = [&]() {
        rx = __builtin_amdgcn_raw_buffer_load_b128(xr, x_off, 0, 0);
        x_off += x_step;
        nz.load(zr, r_z), ng.load(gr, r_g);
        r_z += s_z, r_g += s_g;
        if constexpr (!SUM) {
            nu.load(ur, r_u), ny.load(yr, r_y);
            r_u += s_u, r_y += s_y;
            nh[0] = *reinterpret_cast<const float4 *>(ckp);
            nh[1] = *reinterpret_cast<const float4 *>(ckp + (int64_t)E * 16);
            ckp += ck_step;
        }
    };
    auto commit = [&](const int buf_x) {
        if (x_thread) {
            if (x_raw) *reinterpret_cast<u32x4 *>(lds + L::kX + x_lds + buf_x) = rx;
            else unpack_store(reinterpret_cast<float *>(lds + L::kX + x_lds + buf_x), uint4{rx[0], rx[1], rx[2], rx[3]}, IO{});
        }
    };

    // ---- per-lane constants
    f32x2 Ap01, Ap23;                                                // recurrence lane: A log2(e) of its 4 states
    {
        const float4 a4 = *reinterpret_cast<const float4 *>(d.A + (int64_t)crc * 16 + 4 * q);
        Ap01 = f32x2{a4.x * CM_LOG2E, a4.y * CM_LOG2E};
        Ap23 = f32x2{a4.z * CM_LOG2E, a4.w * CM_LOG2E};
    }
    float bias[4], Dv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bias[i] = d.delta_bias ? d.delta_bias[coc + i] : 0.f;
        Dv[i] = d.D ? d.D[coc + i] : 0.f;
    }
    // delta MFMA, A operand: lane (m = channel lane%16 of the wave, k block = lane/16) holds W_dt[channel][k ..]
    const int cm = min(c0 + 16 * w + s16, E - 1);
    bf16x8 WdtA8;
    float WdtA[4];
    // d dt MFMA, B operand: lane (n = feature lane%16, k block g) holds W_dt[co + i][feature]   (second tile: feature + 16)
    bf16x4 WdtB4[DTR / 16];
    float WdtB[4];
    {
        const float4 w4 = *reinterpret_cast<const float4 *>(d.dt_weight + (int64_t)cm * DTR + 4 * g);
        WdtA[0] = w4.x, WdtA[1] = w4.y, WdtA[2] = w4.z, WdtA[3] = w4.w;
        if constexpr (S == 2) {
            const float *wr = d.dt_weight + (int64_t)cm * DTR + 8 * (DTR == 32 ? g : (g & 1));
            const float4 lo = *reinterpret_cast<const float4 *>(wr), hi = *reinterpret_cast<const float4 *>(wr + 4);
            const float sc = (DTR == 32 || g < 2) ? 1.f : 0.f;
            typedef float f32x8 __attribute__((ext_vector_type(8)));
            WdtA8 = __builtin_convertvector(f32x8{lo.x * sc, lo.y * sc, lo.z * sc, lo.w * sc, hi.x * sc, hi.y * sc, hi.z * sc, hi.w * sc}, bf16x8);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) WdtB[i] = d.dt_weight[(int64_t)(coc + i) * DTR + s16];
        if constexpr (S == 2) {
#pragma unroll
            for (int t2 = 0; t2 < DTR / 16; ++t2)
                WdtB4[t2] = __builtin_convertvector(f32x4{d.dt_weight[(int64_t)(coc + 0) * DTR + s16 + 16 * t2], d.dt_weight[(int64_t)(coc + 1) * DTR + s16 + 16 * t2],
                                                          d.dt_weight[(int64_t)(coc + 2) * DTR + s16 + 16 * t2], d.dt_weight[(int64_t)(coc + 3) * DTR + s16 + 16 * t2]}, bf16x4);
        }
    }
    if (!ok) {                                                       // a channel quad past dim contributes nothing
#pragma unroll
        for (int i = 0; i < 4; ++i) WdtB[i] = 0.f;
        if constexpr (S == 2) {
#pragma unroll
            for (int t2 = 0; t2 < DTR / 16; ++t2) WdtB4[t2] = bf16x4{0, 0, 0, 0};
        }
    }

    // ---- state carried over the whole sequence
    f32x2 lam01 = {0.f, 0.f}, lam23 = {0.f, 0.f};                    // adjoint state a_{t+1} lambda_{t+1}
    if (lam_in && ok) {
        const float4 l4 = *reinterpret_cast<const float4 *>(lam_in + (int64_t)cr * 16 + 4 * q);
        lam01 = f32x2{l4.x, l4.y}, lam23 = f32x2{l4.z, l4.w};
    }
    float dsum[4] = {0.f, 0.f, 0.f, 0.f};                            // SUM: delta' of the owned (step, channels) over the chunk
    f32x2 dA01 = {0.f, 0.f}, dA23 = {0.f, 0.f};
    float dDacc[4] = {0.f, 0.f, 0.f, 0.f}, dbacc[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 dWacc[DTR / 16];                                           // ddt_weight tile: lane (feature lane%16, gq) reg i <-> channel 4 gq + i of the wave
#pragma unroll
    for (int t2 = 0; t2 < DTR / 16; ++t2) dWacc[t2] = f32x4{0.f, 0.f, 0.f, 0.f};

    int x_cur = 0, x_nxt = L::kXT;
    issue();
    commit(x_cur);
    quad_io<IO> cu = nu, cz = nz, cg = ng, cy = ny;
    float4 ch[2] = {nh[0], nh[1]};
    (void)cu, (void)cy, (void)ch;
    __syncthreads();
    int tb = tbb0;
    for (int k = 0; k < nblk; ++k) {
        const bool more = k + 1 < nblk;
        if (more) issue();
        const float *xt = reinterpret_cast<const float *>(lds + L::kX + x_cur);
        // ================= owner phase: delta', delta' u, g = dout silu(z) for (step s16, channels co + i)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if constexpr (S == 2) {
            // B operand: lane (n = step lane%16, k block) holds dt[step][8 kb .. 8 kb + 7] (raw bf16 columns of the staged row)
            const bf16x8 dt8 = *reinterpret_cast<const bf16x8 *>(xt + s16 * XSB + 4 * (DTR == 32 ? g : (g & 1)));
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WdtA8, dt8, acc, 0, 0, 0);
        } else {
            const f32x4 dtf = *reinterpret_cast<const f32x4 *>(xt + s16 * XSB + 4 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(WdtA[i], dtf[i], acc, 0, 0, 0);
        }
        // acc[i] = delta_raw[channel 4 g + i of the wave][step s16]
        const bool valid = tb + s16 < T;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float pre = acc[i] + bias[i];
            float dv = softplus_rows(pre);
            dv = (valid && ok) ? dv : 0.f;                           // padded steps / channels: a = 1, b = 0, nothing flows
            const float uv = SUM ? 0.f : cu.get(i), zv = cz.get(i), gv = cg.get(i);
            const float gzv = (valid && ok) ? gv * zv * cm_sigmoid(zv) : 0.f;
            if constexpr (SUM) dsum[i] += dv;
            // the 4th slot keeps the pre-activation for the epilogue's softplus' (re-read there instead of living in registers)
            *reinterpret_cast<f32x4 *>(pin + (4 * g + i) * PIN + 4 * s16) = f32x4{dv, dv * uv, gzv, pre};
        }
        if constexpr (SUM) {
            // summary pass: the adjoint alone.  lambda_t = C_t g_t + carry;  carry = a_t lambda_t
            const float *pl = pin + (4 * g + cl) * PIN;
            const float *xb = xt + BOFF + 4 * q;
#pragma unroll
            for (int sp = TB - 1; sp >= 0; --sp) {
                const int j = REV ? TB - 1 - sp : sp;
                const f32x4 dwg = *reinterpret_cast<const f32x4 *>(pl + 4 * j);
                const f32x4 Cv = *reinterpret_cast<const f32x4 *>(xb + j * XSB + 16);
                const f32x2 d2 = {dwg[0], dwg[0]}, g2 = {dwg[2], dwg[2]};
                const f32x2 x01 = d2 * Ap01, x23 = d2 * Ap23;
                const f32x2 a01 = f32x2{cm_exp2(x01.x), cm_exp2(x01.y)}, a23 = f32x2{cm_exp2(x23.x), cm_exp2(x23.y)};
                lam01 = __builtin_elementwise_fma(f32x2{Cv[0], Cv[1]}, g2, lam01) * a01;
                lam23 = __builtin_elementwise_fma(f32x2{Cv[2], Cv[3]}, g2, lam23) * a23;
            }
            tb += BDIR * TB;
            if (more) commit(x_nxt);
            cm_lds_barrier();
            cz = nz, cg = ng;
            const int x_old = x_cur;
            x_cur = x_nxt, x_nxt = x_old;
            continue;
        }

        // ================= recurrence phase
        auto slot = [](int sp) { return REV ? TB - 1 - sp : sp; };
        constexpr int HB = TB / 2;
        const float *pl = pin + (4 * g + cl) * PIN;
        const float *xb = xt + BOFF + 4 * q;
        float *const redw = red + g * (4 * 32) + 4 * q;              // [row g][step & 3][dB 16 | dC 16]
#pragma unroll
        for (int hf = 1; hf >= 0; --hf) {                            // the half the scan ran last comes first
            const float4 hc = ch[REV ? 1 - hf : hf];                 // scan-order half hf covers the time half hf (forward) / 1 - hf (reverse)
            f32x2 h01 = {hc.x, hc.y}, h23 = {hc.z, hc.w};
            if (!ok) h01 = h23 = f32x2{0.f, 0.f};
            f32x2 a01s[HB], a23s[HB], h01s[HB], h23s[HB];
#pragma unroll
            for (int si = 0; si < HB; ++si) {                        // sweep 1: states of the half block
                const int j = slot(HB * hf + si);
                const float2 dw = *reinterpret_cast<const float2 *>(pl + 4 * j);
                const f32x4 Bv = *reinterpret_cast<const f32x4 *>(xb + j * XSB);
                const f32x2 d2 = {dw.x, dw.x}, w2 = {dw.y, dw.y};
                const f32x2 x01 = d2 * Ap01, x23 = d2 * Ap23;
#if defined(CM_BWD_ABL) && CM_BWD_ABL == 3
                const f32x2 a01 = x01 * 0.5f + 1.0f, a23 = x23 * 0.5f + 1.0f;           // timing ablation: no exponentials
#else
                const f32x2 a01 = f32x2{cm_exp2(x01.x), cm_exp2(x01.y)}, a23 = f32x2{cm_exp2(x23.x), cm_exp2(x23.y)};
#endif
                h01 = __builtin_elementwise_fma(a01, h01, w2 * f32x2{Bv[0], Bv[1]});
                h23 = __builtin_elementwise_fma(a23, h23, w2 * f32x2{Bv[2], Bv[3]});
                a01s[si] = a01, a23s[si] = a23, h01s[si] = h01, h23s[si] = h23;
            }
#pragma unroll
            for (int si = HB - 1; si >= 0; --si) {                   // sweep 2: adjoint
                const int sp = HB * hf + si, j = slot(sp);
                const f32x4 dwg = *reinterpret_cast<const f32x4 *>(pl + 4 * j);          // delta', delta' u, g
                const f32x4 Bv = *reinterpret_cast<const f32x4 *>(xb + j * XSB);
                const f32x4 Cv = *reinterpret_cast<const f32x4 *>(xb + j * XSB + 16);
                const f32x2 g2 = {dwg[2], dwg[2]}, w2 = {dwg[1], dwg[1]}, d2 = {dwg[0], dwg[0]};
                const f32x2 B01 = {Bv[0], Bv[1]}, B23 = {Bv[2], Bv[3]};
                const f32x2 l01 = __builtin_elementwise_fma(f32x2{Cv[0], Cv[1]}, g2, lam01);   // lambda_t
                const f32x2 l23 = __builtin_elementwise_fma(f32x2{Cv[2], Cv[3]}, g2, lam23);
                const f32x2 aC01 = g2 * h01s[si], aC23 = g2 * h23s[si];                        // dC terms
                const f32x2 aB01 = l01 * w2, aB23 = l23 * w2;                                  // dB terms
                f32x2 sb = l01 * B01;
                sb = __builtin_elementwise_fma(l23, B23, sb);
                const f32x2 hb01 = __builtin_elementwise_fma(-w2, B01, h01s[si]);              // a_t h_{t-1}
                const f32x2 hb23 = __builtin_elementwise_fma(-w2, B23, h23s[si]);
                const f32x2 r01 = l01 * hb01, r23 = l23 * hb23;
                dA01 = __builtin_elementwise_fma(r01, d2, dA01);
                dA23 = __builtin_elementwise_fma(r23, d2, dA23);
                f32x2 sa = r01 * Ap01;
                sa = __builtin_elementwise_fma(r23, Ap23, sa);
                lam01 = l01 * a01s[si];
                lam23 = l23 * a23s[si];
                // sums over the 16 states of the channel (in-lane, then the quad: 2 DPP adds each) and over the 4 channels of the row
                // (row_shr:8, row_shr:4: valid in lanes 12..15), as ONE block of v_add_f32 with the DPP operand: through the
                // builtin the compiler emitted v_mov_b32_dpp + packed adds + hazard nops (704 moves, 175 nops in the kernel), 6-8
                // issue cycles per sum instead of 4.  One s_nop covers the VALU-write -> DPP-read hazard of every input; later
                // reads are >= 2 instructions behind their writes.
                float sbs = sb.x + sb.y, sas = sa.x + sa.y;
                float b0 = aB01.x, b1 = aB01.y, b2 = aB23.x, b3 = aB23.y, c0_ = aC01.x, c1 = aC01.y, c2 = aC23.x, c3 = aC23.y;
#if !defined(CM_BWD_ABL) || CM_BWD_ABL != 1
#define CM_DPP_ADD(r, ctl) "v_add_f32_dpp " r ", " r ", " r " " ctl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                asm volatile("s_nop 1\n\t"
                             CM_DPP_ADD("%0", "quad_perm:[1,0,3,2]") CM_DPP_ADD("%1", "quad_perm:[1,0,3,2]")
                             CM_DPP_ADD("%2", "row_shr:8") CM_DPP_ADD("%3", "row_shr:8") CM_DPP_ADD("%4", "row_shr:8") CM_DPP_ADD("%5", "row_shr:8")
                             CM_DPP_ADD("%6", "row_shr:8") CM_DPP_ADD("%7", "row_shr:8") CM_DPP_ADD("%8", "row_shr:8") CM_DPP_ADD("%9", "row_shr:8")
                             CM_DPP_ADD("%0", "quad_perm:[2,3,0,1]") CM_DPP_ADD("%1", "quad_perm:[2,3,0,1]")
                             CM_DPP_ADD("%2", "row_shr:4") CM_DPP_ADD("%3", "row_shr:4") CM_DPP_ADD("%4", "row_shr:4") CM_DPP_ADD("%5", "row_shr:4")
                             CM_DPP_ADD("%6", "row_shr:4") CM_DPP_ADD("%7", "row_shr:4") CM_DPP_ADD("%8", "row_shr:4") CM_DPP_ADD("%9", "row_shr:4")
                             : "+v"(sbs), "+v"(sas), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(c0_), "+v"(c1), "+v"(c2), "+v"(c3));
#undef CM_DPP_ADD
#else
                sbs = cm_group_sum<4>(sbs), sas = cm_group_sum<4>(sas);
#endif
                if (q == 0) *reinterpret_cast<float2 *>(pout + (4 * g + cl) * POUT + 2 * j) = make_float2(sbs, sas);
#if !defined(CM_BWD_ABL) || CM_BWD_ABL != 1
                const f32x4 rB = {b0, b1, b2, b3}, rC = {c0_, c1, c2, c3};
                if (cl == 3) {
                    *reinterpret_cast<f32x4 *>(redw + (j & 3) * 32) = rB;
                    *reinterpret_cast<f32x4 *>(redw + (j & 3) * 32 + 16) = rC;
                }
                if ((sp & 3) == 0) {
                    // flush 4 steps: sum the wave's 4 rows (LDS operations of one wave execute in order: no barrier) -> cross-wave tile
                    const int o = 2 * lane, jj = o >> 5, col = o & 31;
                    f32x2 v = *reinterpret_cast<const f32x2 *>(red + jj * 32 + col);
#pragma unroll
                    for (int gg = 1; gg < 4; ++gg) v += *reinterpret_cast<const f32x2 *>(red + gg * 128 + jj * 32 + col);
                    *reinterpret_cast<f32x2 *>(xw + (w * TB + (j & ~3) + jj) * RW + DTR + col) = v;
                }
#else
                asm volatile("" ::"v"(aB01), "v"(aB23), "v"(aC01), "v"(aC23));          // timing ablation: no dB / dC reduction
#endif
                asm volatile("" : "+v"(lam01), "+v"(lam23), "+v"(dA01), "+v"(dA23));
            }
        }

        // ================= epilogue (owner layout): du, dz, ddelta_raw for (step s16, channels co + i)
        float duv[4], dzv[4], ddr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float2 sv = *reinterpret_cast<const float2 *>(pout + (4 * g + i) * POUT + 2 * s16);
            const f32x4 own = *reinterpret_cast<const f32x4 *>(pin + (4 * g + i) * PIN + 4 * s16);   // delta', delta' u, g, pre-activation
            const float uv = cu.get(i), zv = cz.get(i), gv = cg.get(i), yv = cy.get(i);
            const float sbl = sv.x, sal = sv.y * CM_LN2;             // the recurrence lanes contracted with A log2(e)
            duv[i] = fmaf(Dv[i], own[2], sbl * own[0]);
            const float ddt = fmaf(sbl, uv, sal);
            float dd = ddt * cm_sigmoid(own[3]);                     // softplus'
            dd = (valid && ok) ? dd : 0.f;
            ddr[i] = dd;
            dbacc[i] += dd;
            dDacc[i] = fmaf(own[2], uv, dDacc[i]);
            const float sg = cm_sigmoid(zv);
            dzv[i] = gv * yv * sg * fmaf(zv, 1.f - sg, 1.f);
            tr[(4 * g + i) * TRS + s16] = dd;
        }
        quad_io<IO>::store(dur, w_u, duv);
        quad_io<IO>::store(dzr, w_z, dzv);
        w_u += sw_u, w_z += sw_z;
#if !defined(CM_BWD_ABL) || CM_BWD_ABL != 2
        // d dt[t][r] = sum over the wave's channels of ddelta_raw[c][t] W_dt[c][r]   (A: lane (m = t, k = channel), B: lane (n = r, k = channel))
        // ddt_weight[c][r] += sum_t ddelta_raw[c][t] dt[t][r]                        (A: lane (m = c, k = t) from the transposition patch)
        if constexpr (S == 2) {
            const bf16x4 a4 = __builtin_convertvector(f32x4{ddr[0], ddr[1], ddr[2], ddr[3]}, bf16x4);
            const bf16x4 at4 = __builtin_convertvector(f32x4{tr[s16 * TRS + 4 * g], tr[s16 * TRS + 4 * g + 1], tr[s16 * TRS + 4 * g + 2], tr[s16 * TRS + 4 * g + 3]}, bf16x4);
            const uint16_t *xraw = reinterpret_cast<const uint16_t *>(xt);
#pragma unroll
            for (int t2 = 0; t2 < DTR / 16; ++t2) {
                const f32x4 dd = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, WdtB4[t2], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) xw[(w * TB + 4 * g + i) * RW + 16 * t2 + s16] = dd[i];
                typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));
                const u16x4 dtc = {xraw[(4 * g + 0) * XSB * 2 + 16 * t2 + s16], xraw[(4 * g + 1) * XSB * 2 + 16 * t2 + s16],
                                   xraw[(4 * g + 2) * XSB * 2 + 16 * t2 + s16], xraw[(4 * g + 3) * XSB * 2 + 16 * t2 + s16]};
                dWacc[t2] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(at4, __builtin_bit_cast(bf16x4, dtc), dWacc[t2], 0, 0, 0);
            }
        } else {
            f32x4 dd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) dd = __builtin_amdgcn_mfma_f32_16x16x4f32(ddr[i], WdtB[i], dd, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) xw[(w * TB + 4 * g + i) * RW + s16] = dd[i];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                dWacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(tr[s16 * TRS + 4 * g + i], xt[(4 * g + i) * XSB + s16], dWacc[0], 0, 0, 0);
        }
#endif

        tb += BDIR * TB;
        if (more) commit(x_nxt);
        cm_lds_barrier();
#if !defined(CM_BWD_ABL) || CM_BWD_ABL != 4
        // ================= the workgroup's partial dxdbl rows of this block: sum of the 4 waves -> workspace
        {
            const int tbk = tb - BDIR * TB;                          // this block's base step
#pragma unroll
            for (int r = 0; r < (TB * RW + 255) / 256; ++r) {
                const int o = tid + 256 * r;
                if (o < TB * RW) {
                    const float v = (xw[o] + xw[TB * RW + o]) + (xw[2 * TB * RW + o] + xw[3 * TB * RW + o]);
                    if (tbk + o / RW < T) wsx[(int64_t)(t_lo + tbk) * RW + o] = v;
                }
            }
        }
        cm_lds_barrier();                                            // the cross-wave tile is rewritten during the next block
#endif
        cu = nu, cz = nz, cg = ng, cy = ny, ch[0] = nh[0], ch[1] = nh[1];
        const int x_old = x_cur;
        x_cur = x_nxt, x_nxt = x_old;
    }

    if constexpr (SUM) {
        // E = the carry this chunk hands on from a zero entry; P = exp(A sum of delta') per (channel, state)
        float *sl = tr;                                              // 16 floats per wave: sum of delta' per channel of the wave
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float sd = cm_group_sum<16>(dsum[i]);
            if (s16 == 0) sl[4 * g + i] = sd;
        }
        const float sc = sl[4 * g + cl];                             // in-order LDS within the wave
        if (ok) {
            *reinterpret_cast<float4 *>(e_out + (int64_t)cr * 16 + 4 * q) = make_float4(lam01.x, lam01.y, lam23.x, lam23.y);
            *reinterpret_cast<float4 *>(p_out + (int64_t)cr * 16 + 4 * q) =
                make_float4(cm_exp2(Ap01.x * sc), cm_exp2(Ap01.y * sc), cm_exp2(Ap23.x * sc), cm_exp2(Ap23.y * sc));
        }
        return;
    }
    // ---- per-sequence parameter gradients -> workspace (summed over the batch by the reduce kernel)
    float *pp = wsp + (int64_t)crc * NP;
    if (ok) *reinterpret_cast<float4 *>(pp + 4 * q) = make_float4(dA01.x, dA01.y, dA23.x, dA23.y);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float sD = cm_group_sum<16>(dDacc[i]), sb = cm_group_sum<16>(dbacc[i]);
        if (ok && s16 == 0) {
            wsp[(int64_t)(co + i) * NP + 16 + DTR] = sD;
            wsp[(int64_t)(co + i) * NP + 16 + DTR + 1] = sb;
        }
    }
#pragma unroll
    for (int t2 = 0; t2 < DTR / 16; ++t2)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (ok) wsp[(int64_t)(co + i) * NP + 16 + 16 * t2 + s16] = dWacc[t2][i];
}

// bf16: two workgroups per CU (256 VGPRs); the fp32 instantiation (parity tests) carries twice the row registers: one per CU
template <typename IO, int DTR, bool SUM>
__global__ __launch_bounds__(256, sizeof(IO) == 2 ? 2 : 1) void scan_rows_bwd_kernel(const cm_scan_cl_bwd_args p, const bwd_plan pl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int RW = DTR + 32, NP = 16 + DTR + 4;
    const int total = gridDim.x, nx = pl.nx, nbk = p.batch * pl.chunks;
    int id = blockIdx.x;
    if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);   // workgroups of one (sequence, direction) on one XCD
    const int cx = id % nx, bk = (id / nx) % nbk, z = id / (nx * nbk);
    const int b = bk / pl.chunks, k = bk % pl.chunks;
    const cm_scan_cl_bwd_dir &d = p.dir[z];
    const int t_lo = k * pl.chunk_len, T = min(p.seqlen - t_lo, pl.chunk_len);
    float *wsx = pl.wsx + (((int64_t)z * nx + cx) * p.batch + b) * p.seqlen * RW;
    const int64_t slot = ((int64_t)z * p.batch + b) * pl.chunks + k;
    float *wsp = pl.wsp + slot * p.dim * NP;
    const float *lam_in = pl.pass == 2 ? pl.lin + slot * p.dim * 16 : nullptr;
    float *e_out = SUM ? pl.le + slot * p.dim * 16 : nullptr, *p_out = SUM ? pl.lp + slot * p.dim * 16 : nullptr;
    if (d.reverse_time) scan_rows_bwd<IO, true, DTR, SUM>(p, d, lds, cx, b, t_lo, T, wsx, wsp, lam_in, e_out, p_out);
    else scan_rows_bwd<IO, false, DTR, SUM>(p, d, lds, cx, b, t_lo, T, wsx, wsp, lam_in, e_out, p_out);
}

// adjoint entering every chunk, folded against the scan: one thread per (direction, sequence, channel, state quad)
__global__ __launch_bounds__(256) void rows_bwd_carry_kernel(const cm_scan_cl_bwd_args p, const bwd_plan pl) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int E = p.dim, C = pl.chunks;
    if (idx >= (int64_t)p.ndir * p.batch * E * 4) return;
    const int q = (int)(idx & 3), c = (int)((idx >> 2) % E), b = (int)((idx >> 2) / E % p.batch), z = (int)((idx >> 2) / E / p.batch);
    const bool rev = p.dir[z].reverse_time != 0;
    float4 Lc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < C; ++i) {
        const int k = rev ? i : C - 1 - i;                           // the chunk the scan ran last comes first
        const int64_t off = ((((int64_t)z * p.batch + b) * C + k) * E + c) * 16 + 4 * q;
        *reinterpret_cast<float4 *>(pl.lin + off) = Lc;
        const float4 a = *reinterpret_cast<const float4 *>(pl.lp + off), e = *reinterpret_cast<const float4 *>(pl.le + off);
        Lc = make_float4(fmaf(a.x, Lc.x, e.x), fmaf(a.y, Lc.y, e.y), fmaf(a.z, Lc.z, e.z), fmaf(a.w, Lc.w, e.w));
    }
}

// fixed-order second pass: dxdbl rows = sum over the channel-group workgroups (stored in the I/O dtype); parameter gradients =
// sum over the batch, ACCUMULATED into the caller's fp32 tensors
template <typename IO, int DTR>
__global__ __launch_bounds__(256) void scan_rows_bwd_reduce_kernel(const cm_scan_cl_bwd_args p, const bwd_plan pl, const int row_blocks) {
    constexpr int RW = DTR + 32, NP = 16 + DTR + 4, G = RW / 4;
    const int64_t rows = (int64_t)p.batch * p.seqlen;
    if ((int)blockIdx.x < row_blocks) {
        // dxdbl rows: thread = 4 consecutive columns of one (direction = blockIdx.y, sequence, step): 16-byte loads of the nx partial rows,
        // one 8- / 16-byte store, 32-bit index arithmetic (the first version summed one float per thread behind four 64-bit divisions: 42 us
        // for 98 MB at 32 x 1000 x 2)
        const int z = blockIdx.y;
        const uint32_t j = blockIdx.x * 256u + threadIdx.x;          // (row, column group) of direction z; rows * G < 2^31 (checked by the host)
        if (j >= (uint32_t)(rows * G)) return;
        const uint32_t bt = j / G, cg = j % G;
        const float *src = pl.wsx + ((int64_t)z * pl.nx * rows + bt) * RW + 4 * cg;
        f32x4 acc = *reinterpret_cast<const f32x4 *>(src);
        for (int x = 1; x < pl.nx; ++x) acc += *reinterpret_cast<const f32x4 *>(src + (int64_t)x * rows * RW);
        const uint32_t b = bt / (uint32_t)p.seqlen, t = bt - b * (uint32_t)p.seqlen;
        const cm_scan_cl_bwd_dir &d = p.dir[z];
        IO *dst = reinterpret_cast<IO *>(d.dxdbl) + (int64_t)b * d.dxdbl_bs + (int64_t)t * d.dxdbl_ts + 4 * cg;
        if constexpr (sizeof(IO) == 2) *reinterpret_cast<u32x2 *>(dst) = u32x2{cm_pack_bf16(acc[0], acc[1]), cm_pack_bf16(acc[2], acc[3])};
        else *reinterpret_cast<f32x4 *>(dst) = acc;
        return;
    }
    if (blockIdx.y != 0) return;
    // parameter gradients: sum over the (sequence, chunk) slabs
    const int64_t npar = (int64_t)p.ndir * p.dim * NP;
    for (int64_t j = (int64_t)(blockIdx.x - row_blocks) * 256 + threadIdx.x; j < npar; j += (int64_t)(gridDim.x - row_blocks) * 256) {
        const int z = (int)(j / ((int64_t)p.dim * NP));
        const int64_t cj = j - (int64_t)z * p.dim * NP;              // (channel, slot)
        const int c = (int)(cj / NP), slot = (int)(cj % NP);
        const int64_t nslab = (int64_t)p.batch * pl.chunks;
        const float *src = pl.wsp + (int64_t)z * nslab * p.dim * NP + cj;
        float acc = 0.f;
        for (int64_t b = 0; b < nslab; ++b) acc += src[b * p.dim * NP];
        const cm_scan_cl_bwd_dir &d = p.dir[z];
        auto put = [&](float *dst) { *dst = p.overwrite ? acc : *dst + acc; };
        if (slot < 16) {
            if (p.da_log) acc *= d.A[(int64_t)c * 16 + slot];
            put(d.dA + (int64_t)c * 16 + slot);
        }
        else if (slot < 16 + DTR) put(d.ddt_weight + (int64_t)c * DTR + slot - 16);
        else if (slot == 16 + DTR) { if (d.dD) put(d.dD + c); }
        else if (d.ddelta_bias) put(d.ddelta_bias + c);
    }
}

inline int bwd_chunk_len(int seqlen, int chunks) { return ((seqlen + chunks - 1) / chunks + TB - 1) / TB * TB; }

// chunk count for a launch: below 512 workgroups (two per CU) the sequences are cut so that about 1024 run, chunks >= 128 steps
inline int bwd_auto_chunks(int batch, int seqlen, int dim, int ndir, int want) {
    if (want >= 1) {
        if (want == 1 || seqlen < 2 * TB) return 1;
        const int len = bwd_chunk_len(seqlen, want);
        return (seqlen + len - 1) / len;
    }
    const long wgs = (long)((dim + 63) / 64) * batch * ndir;
    if (wgs >= 512) return 1;
    long c = (1024 + wgs - 1) / wgs;
    if (c > seqlen / 128) c = seqlen / 128;
    if (c < 2) return 1;
    const int len = bwd_chunk_len(seqlen, (int)c);
    return (seqlen + len - 1) / len;
}

template <typename IO, int DTR>
int launch_bwd(const cm_scan_cl_bwd_args &a) {
    constexpr int RW = DTR + 32, NP = 16 + DTR + 4;
    bwd_plan pl{};
    pl.nx = (a.dim + 63) / 64;
    pl.chunks = bwd_auto_chunks(a.batch, a.seqlen, a.dim, a.ndir, a.time_chunks);
    pl.chunk_len = pl.chunks > 1 ? bwd_chunk_len(a.seqlen, pl.chunks) : (a.seqlen + TB - 1) / TB * TB;
    pl.wsx = reinterpret_cast<float *>(a.workspace);
    pl.wsp = pl.wsx + (int64_t)a.ndir * pl.nx * a.batch * a.seqlen * RW;
    pl.lin = pl.wsp + (int64_t)a.ndir * a.batch * pl.chunks * a.dim * NP;
    pl.le = pl.lin + (int64_t)a.ndir * a.batch * pl.chunks * a.dim * 16;
    pl.lp = pl.le + (int64_t)a.ndir * a.batch * pl.chunks * a.dim * 16;
    const size_t smem = bwd_lds<DTR>::kBytes;
    static bool attr_done = false;
    if (!attr_done && smem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&scan_rows_bwd_kernel<IO, DTR, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&scan_rows_bwd_kernel<IO, DTR, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) {
            cm_set_error("scan_cl_bwd: hipFuncSetAttribute(%zu B LDS) failed: %s", smem, hipGetErrorString(e));
            return (int)e;
        }
        attr_done = true;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    const long total = (long)pl.nx * a.batch * pl.chunks * a.ndir;
    if (pl.chunks > 1) {
        pl.pass = 1;
        hipLaunchKernelGGL((scan_rows_bwd_kernel<IO, DTR, true>), dim3((unsigned)total), dim3(256), smem, st, a, pl);
        if (int rc = cm_launch_status("cm_scan_cl_bwd(chunk summaries)")) return rc;
        const long nthr = (long)a.ndir * a.batch * a.dim * 4;
        hipLaunchKernelGGL(rows_bwd_carry_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, a, pl);
        if (int rc = cm_launch_status("cm_scan_cl_bwd(carry)")) return rc;
        pl.pass = 2;
    }
    hipLaunchKernelGGL((scan_rows_bwd_kernel<IO, DTR, false>), dim3((unsigned)total), dim3(256), smem, st, a, pl);
    if (int rc = cm_launch_status("cm_scan_cl_bwd")) return rc;
    const int64_t nrow4 = (int64_t)a.batch * a.seqlen * (RW / 4);            // per direction
    const int row_blocks = (int)((nrow4 + 255) / 256);
    const int64_t par_blocks = ((int64_t)a.ndir * a.dim * NP + 255) / 256;
    hipLaunchKernelGGL((scan_rows_bwd_reduce_kernel<IO, DTR>), dim3((unsigned)(row_blocks + (par_blocks > 1024 ? 1024 : par_blocks)), (unsigned)a.ndir), dim3(256), 0,
                       st, a, pl, row_blocks);
    return cm_launch_status("cm_scan_cl_bwd(reduce)");
}

inline int bwd_dtr(const cm_scan_cl_bwd_args &a) { return a.dir[0].dt_rank > 16 ? 32 : 16; }

}  // namespace

extern "C" int cm_scan_cl_bwd_auto_chunks(int batch, int seqlen, int dim, int ndir) { return bwd_auto_chunks(batch, seqlen, dim, ndir, 0); }

extern "C" int64_t cm_scan_cl_bwd_workspace_bytes(const cm_scan_cl_bwd_args *a) {
    if (!a || a->batch <= 0 || a->seqlen <= 0 || a->dim <= 0 || a->ndir <= 0) return 0;
    const int64_t P = bwd_dtr(*a), RW = P + 32, NP = 16 + P + 4, nx = (a->dim + 63) / 64;
    const int64_t chunks = bwd_auto_chunks(a->batch, a->seqlen, a->dim, a->ndir, a->time_chunks);
    return 4 * ((int64_t)a->ndir * nx * a->batch * a->seqlen * RW + (int64_t)a->ndir * a->batch * chunks * a->dim * (NP + 3 * 16));
}

extern "C" int cm_scan_cl_bwd(const cm_scan_cl_bwd_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "scan_cl_bwd: args is NULL");
    const cm_scan_cl_bwd_args &a = *args;
    CM_REQUIRE(a.batch > 0 && a.dim > 0 && a.seqlen > 0, CM_EINVAL, "scan_cl_bwd: bad sizes batch=%d dim=%d seqlen=%d", a.batch, a.dim, a.seqlen);
    CM_REQUIRE(a.dstate == 16, CM_EUNSUPPORTED, "scan_cl_bwd: dstate %d unsupported (16 only)", a.dstate);
    CM_REQUIRE(a.ndir == 1 || a.ndir == 2, CM_EINVAL, "scan_cl_bwd: ndir must be 1 or 2");
    CM_REQUIRE(a.io_dtype == CM_BF16 || a.io_dtype == CM_F32, CM_EUNSUPPORTED, "scan_cl_bwd: io dtype %d unsupported", a.io_dtype);
    const int vec = a.io_dtype == CM_BF16 ? 8 : 4;
    CM_REQUIRE(a.dim % vec == 0, CM_EUNSUPPORTED, "scan_cl_bwd: dim %d must be a multiple of %d", a.dim, vec);
    CM_REQUIRE(a.z && cm_aligned(a.z, 16) && a.z_bs % 4 == 0 && a.z_ts % 4 == 0, CM_EALIGN, "scan_cl_bwd: z is required, 16-byte aligned, strides multiples of 4");
    CM_REQUIRE((long)((a.dim + 63) / 64) * a.batch * a.ndir < (1L << 31), CM_EINVAL, "scan_cl_bwd: grid too large");
    const int64_t need = cm_scan_cl_bwd_workspace_bytes(&a);
    CM_REQUIRE(a.workspace && cm_aligned(a.workspace, 16) && a.workspace_bytes >= need, CM_EINVAL,
               "scan_cl_bwd: needs a 16-byte aligned workspace of %lld bytes (cm_scan_cl_bwd_workspace_bytes), got %lld", (long long)need,
               (long long)a.workspace_bytes);
    const int P = bwd_dtr(a);
    CM_REQUIRE((int64_t)a.batch * a.seqlen * ((P + 32) / 4) < ((int64_t)1 << 31), CM_EUNSUPPORTED, "scan_cl_bwd: batch x seqlen too large for the reduce pass's 32-bit row index");
    for (int i = 0; i < a.ndir; ++i) {
        const cm_scan_cl_bwd_dir &d = a.dir[i];
        CM_REQUIRE(d.u && d.xdbl && d.A && d.dt_weight && d.ckpt && d.ypre && d.dout && d.du && d.dz && d.dxdbl && d.dA && d.ddt_weight, CM_EINVAL,
                   "scan_cl_bwd: dir %d has a NULL tensor", i);
        CM_REQUIRE((d.dt_rank == 16 || d.dt_rank == 32) && d.dt_rank == P, CM_EUNSUPPORTED, "scan_cl_bwd: dt_rank (padded) %d: 16, or 32 for every direction", d.dt_rank);
        CM_REQUIRE(P == 16 || a.io_dtype == CM_BF16, CM_EUNSUPPORTED, "scan_cl_bwd: 64-wide x_dbl rows (dt_rank > 16) are built for bf16 I/O only");
        const int64_t st[] = {d.u_bs, d.u_ts, d.ypre_bs, d.ypre_ts, d.dout_bs, d.dout_ts, d.du_bs, d.du_ts, d.dz_bs, d.dz_ts, d.dxdbl_bs, d.dxdbl_ts};
        bool al = cm_aligned(d.dxdbl, 16) && cm_aligned(d.u, 16) && cm_aligned(d.ypre, 16) && cm_aligned(d.dout, 16) && cm_aligned(d.du, 16) && cm_aligned(d.dz, 16) &&
                  cm_aligned(d.xdbl, 16) && d.xdbl_bs % vec == 0 && d.xdbl_ts % vec == 0 && cm_aligned(d.A, 16) && cm_aligned(d.dt_weight, 16) &&
                  cm_aligned(d.ckpt, 16);
        for (int64_t s : st) al = al && s % 4 == 0;
        CM_REQUIRE(al, CM_EALIGN, "scan_cl_bwd: dir %d: tensors must be 16-byte aligned; row strides multiples of 4 (x_dbl: %d) elements", i, vec);
    }
    if (a.io_dtype == CM_BF16) return P == 32 ? launch_bwd<cm_bf16, 32>(a) : launch_bwd<cm_bf16, 16>(a);
    return launch_bwd<float, 16>(a);
}
