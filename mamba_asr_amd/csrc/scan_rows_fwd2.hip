// scan_rows_fwd2.hip — the row-group scan forward with TWO states per lane, for launches that leave SIMDs idle.
//
// scan_rows_fwd.hip puts the 16 states of a channel in 4 lanes (4 states each): a wave covers 16 channels, a workgroup 64, and
// a launch of fewer than 512 workgroups (16 utterances x 40 s at E = 512: 256) gives one wave per SIMD, which issues at about
// half the rate of three.  Cutting the sequences in time (the chunked launch) doubles the recurrence work.
// Here a channel takes 8 lanes (2 states each): a wave covers 8 channels, a workgroup 32, the same launch has twice the
// waves, and the per-lane recurrence is half as long (one packed instruction per quantity instead of two).  Per channel the
// owner work (softplus, gate) is unchanged -- the delta MFMA's tile has the wave's 8 channels twice, and the two copies
// split the block's 16 steps -- so the total vector work is ~1.15x the 4-state kernel's for twice the resident waves.
// Same contract as cm_scan_cl_fwd's xdbl mode with z and softplus, dt_rank <= 16, unchunked.
// MEASURED AND NOT SELECTED BY SIZE (profiles/r03/scan_small_batches.log): 16 x 1000 x 512 133.7 us against 110.6 us for the
// 4-state kernel at one wave per SIMD, 32 x 1000 x 512 191.6 vs 145.4 us.  A SIMD holding one 4-state wave is short of
// independent recurrence chains (two packed chains per lane-step), and two 2-state waves carry exactly as many, at 1.3x the
// owner / staging work.  Kept as an explicit tuning choice (cm_scan_cl_args.lanes_per_channel = 8) and as a second,
// independently written implementation the parity tests run beside the first.
#include "scan_rows_common.h"

namespace {

constexpr int NB2 = 3;        // staged input tiles
constexpr int PW2 = 36;       // floats per channel in the per-wave (delta', delta' u) patch (16 steps x 2 + pad)
constexpr int CW = 32;        // channels per workgroup

template <typename IO> struct rows2_lds {
    static constexpr int kTile = TB * CW * (int)sizeof(IO);
    static constexpr int kU = 0, kZ = NB2 * kTile, kX = 2 * NB2 * kTile;
    static constexpr int kPw = kX + NB2 * TB * XS * 4;
    static constexpr int kPatch = 8 * 8 * 16 * 4;         // per-wave patch: [state pair group 8][channel 8][16 steps] fp32
    static constexpr int kBytes = kPw + 4 * kPatch;
};

template <typename IO, bool REV>
__device__ __forceinline__ void scan_rows2(const cm_scan_cl_args &p, const cm_scan_cl_dir &d, unsigned char *lds, const int cx, const int b) {
    using L = rows2_lds<IO>;
    constexpr int S = (int)sizeof(IO);
    constexpr int VEC = cm_elem<IO>::kVec;
    constexpr int CPR = CW / VEC;                // 16-byte chunks per 32-channel row
    constexpr int NCH = TB * CPR;                // chunks per (u or z) tile
    constexpr int RW = 48, XCPR = RW / VEC, NXC = TB * XCPR;
    constexpr int NTOT = 2 * NCH + NXC, NV = (NTOT + 255) / 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c8 = lane & 7, gg = lane >> 3;     // channel of the wave, state pair (2 gg, 2 gg + 1) == owned steps (2 gg, 2 gg + 1)
    const int half = gg & 1;                     // which two of the MFMA tile's four accumulator rows this lane owns
    const int E = p.dim, T = p.seqlen, c0 = cx * CW;
    const int c = c0 + 8 * w + c8;
    const bool c_ok = c < E;
    const int cc = c_ok ? c : E - 1;
    const int nblk = (T + TB - 1) / TB;
    const int u_ts = (int)d.u_ts, z_ts = (int)p.z_ts, x_ts = (int)d.xdbl_ts, o_ts = (int)d.out_ts, p_ts = (int)d.ypre_ts;
    const bool has_pre = d.ypre != nullptr;
    auto rs = [&](const void *base, int64_t bs, int ts, int width) {
        return make_rsrc(base ? reinterpret_cast<const IO *>(base) + (int64_t)b * bs : nullptr, base ? ((int64_t)(T - 1) * ts + width) * S : 0);
    };
    const __amdgpu_buffer_rsrc_t ur = rs(d.u, d.u_bs, u_ts, E), zr = rs(p.z, p.z_bs, z_ts, E), xr = rs(d.xdbl, d.xdbl_bs, x_ts, RW),
                                 orr = rs(d.out, d.out_bs, o_ts, E), prr = rs(d.ypre, d.ypre_bs, p_ts, E);
    const int tb0 = (REV ? nblk - 1 : 0) * TB;
    constexpr int DIR = REV ? -1 : 1;

    // ---- staging: chunk idx of the block = [u tile | z tile | x_dbl rows]; out-of-range channel chunks read as zero
    int g_off[NV], g_step[NV], l_off[NV], kind[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = tid + 256 * i;
        if (idx < 2 * NCH) {
            const int tens = idx / NCH, within = idx % NCH, ts = tens ? z_ts : u_ts;
            const int ch = c0 + (within % CPR) * VEC;
            g_off[i] = ch < E ? ((tb0 + within / CPR) * ts + ch) * S : 0x7fffffff;
            g_step[i] = ch < E ? DIR * TB * ts * S : 0;
            l_off[i] = (tens ? L::kZ : L::kU) + within * 16;
            kind[i] = tens;
        } else if (idx < NTOT) {
            const int xi = idx - 2 * NCH, row = xi / XCPR, col = xi % XCPR;
            g_off[i] = ((tb0 + row) * x_ts + col * VEC) * S;
            g_step[i] = DIR * TB * x_ts * S;
            // bf16: chunks 0, 1 (dt) stay raw in the first 32 bytes of the staged row; B / C widened to fp32 at floats 16..47
            const bool raw = S == 2 && col < 2;
            l_off[i] = L::kX + row * XS * 4 + (raw ? col * 16 : (S == 2 ? (16 + (col - 2) * VEC) * 4 : col * VEC * 4));
            kind[i] = raw ? 2 : 3;
        } else {
            g_off[i] = 0x7fffffff, g_step[i] = 0, l_off[i] = 0, kind[i] = -1;
        }
    }
    u32x4 rg[NV];
    auto issue = [&]() {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (kind[i] >= 0) rg[i] = __builtin_amdgcn_raw_buffer_load_b128(kind[i] == 0 ? ur : (kind[i] == 1 ? zr : xr), g_off[i], 0, 0);
            g_off[i] += g_step[i];
        }
    };
    auto commit = [&](const int buf_tile, const int buf_x) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (kind[i] == 0 || kind[i] == 1 || kind[i] == 2) *reinterpret_cast<u32x4 *>(lds + l_off[i] + (kind[i] == 2 ? buf_x : buf_tile)) = rg[i];
            else if (kind[i] == 3) unpack_store(reinterpret_cast<float *>(lds + l_off[i] + buf_x), uint4{rg[i][0], rg[i][1], rg[i][2], rg[i][3]}, IO{});
        }
    };

    // ---- per-lane constants
    f32x2 Ap, h = {0.f, 0.f};
    {
        const float2 a2 = *reinterpret_cast<const float2 *>(d.A + (int64_t)cc * 16 + 2 * gg);
        Ap = f32x2{a2.x * CM_LOG2E, a2.y * CM_LOG2E};
    }
    // delta MFMA: lane (n = lane % 16, k block = lane / 16); both copies n and n + 8 carry channel (n & 7) of the wave
    const int cn = min(c0 + 8 * w + (lane & 7), E - 1), kb = lane >> 4;
    float Wdt[4];
    bf16x8 Wdt8;
    {
        const float4 w4 = *reinterpret_cast<const float4 *>(d.dt_weight + (int64_t)cn * 16 + 4 * kb);
        Wdt[0] = w4.x, Wdt[1] = w4.y, Wdt[2] = w4.z, Wdt[3] = w4.w;
        if constexpr (S == 2) {
            const float *wr = d.dt_weight + (int64_t)cn * 16 + 8 * (kb & 1);
            const float4 lo = *reinterpret_cast<const float4 *>(wr), hi = *reinterpret_cast<const float4 *>(wr + 4);
            const float sc = kb < 2 ? 1.f : 0.f;
            typedef float f32x8 __attribute__((ext_vector_type(8)));
            Wdt8 = __builtin_convertvector(f32x8{lo.x * sc, lo.y * sc, lo.z * sc, lo.w * sc, hi.x * sc, hi.y * sc, hi.z * sc, hi.w * sc}, bf16x8);
        }
    }
    if (d.h0) {
        const float2 h2 = *reinterpret_cast<const float2 *>(d.h0 + ((int64_t)b * E + cc) * 16 + 2 * gg);
        h = f32x2{h2.x, h2.y};
    }
    float dsum = 0.f;
    const float bias = d.delta_bias ? d.delta_bias[cc] : 0.f, Dv = d.D ? d.D[cc] : 0.f;
    float uq[2], zq[2];
    float *patch = reinterpret_cast<float *>(lds + L::kPw + w * L::kPatch);
    float *pww = patch + c8 * PW2;
    // partial-output exchange [state pair group][channel][16 steps], 4-step quads XOR-swizzled by channel / 4 (conflict-free 16-byte writes)
    float *red_w = patch + (gg * 8 + c8) * 16;
    const float *red_r = patch + c8 * 16 + 4 * ((gg >> 1) ^ (c8 >> 2)) + 2 * (gg & 1);
    int o_off = c_ok ? ((tb0 + 2 * gg) * o_ts + c) * S : 0x7fffffff, p_off = (c_ok && has_pre) ? ((tb0 + 2 * gg) * p_ts + c) * S : 0x7fffffff;
    const int o_step = c_ok ? DIR * TB * o_ts * S : 0, p_step = (c_ok && has_pre) ? DIR * TB * p_ts * S : 0;
    float *ck = (d.ckpt && c_ok) ? d.ckpt + (((int64_t)b * 2 * nblk + 2 * (tb0 / TB)) * E + c) * 16 + 2 * gg : nullptr;
    const int64_t ck_step = (int64_t)DIR * 2 * E * 16, ck_half = (int64_t)E * 16;

    auto produce = [&](const int buf_tile, const int buf_x, const int tb) {
        const float *xt = reinterpret_cast<const float *>(lds + L::kX + buf_x);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int s16 = lane & 15;
        if constexpr (S == 2) {
            const bf16x8 dt8 = *reinterpret_cast<const bf16x8 *>(xt + s16 * XS + 4 * (kb & 1));
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dt8, Wdt8, acc, 0, 0, 0);
        } else {
            const f32x4 dtf = *reinterpret_cast<const f32x4 *>(xt + s16 * XS + 4 * kb);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dtf[i], Wdt[i], acc, 0, 0, 0);
        }
        // acc[i] = delta_raw[step 4 kb + i][channel lane & 7]; this lane owns i = 2 half, 2 half + 1, i.e. steps 2 gg, 2 gg + 1
        const IO *ut = reinterpret_cast<const IO *>(lds + L::kU + buf_tile) + 8 * w + c8 + 2 * gg * CW;
        const IO *zt = reinterpret_cast<const IO *>(lds + L::kZ + buf_tile) + 8 * w + c8 + 2 * gg * CW;
        float o[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float dv = softplus_rows((half ? acc[2 + i] : acc[i]) + bias);
            dv = tb + 2 * gg + i < T ? dv : 0.f;                     // padded steps: a = 1, b = 0
            dsum += dv;
            uq[i] = ld_io(ut + i * CW);
            zq[i] = ld_io(zt + i * CW);
            o[2 * i] = dv, o[2 * i + 1] = dv * uq[i];
        }
        *reinterpret_cast<f32x4 *>(pww + 4 * gg) = f32x4{o[0], o[1], o[2], o[3]};
    };

    auto recur = [&](const int buf_x) -> f32x2 {
        const float *xt = reinterpret_cast<const float *>(lds + L::kX + buf_x) + 2 * gg;
        auto slot = [](int sp) { return REV ? TB - 1 - sp : sp; };
        float part[TB];
#pragma unroll
        for (int sp = 0; sp < TB; ++sp) {
            const int j = slot(sp);
            const float2 dw = *reinterpret_cast<const float2 *>(pww + 2 * j);
            const f32x2 Bv = *reinterpret_cast<const f32x2 *>(xt + j * XS + 16), Cv = *reinterpret_cast<const f32x2 *>(xt + j * XS + 32);
            const f32x2 x2 = f32x2{dw.x, dw.x} * Ap;
            const f32x2 a2 = {cm_exp2(x2.x), cm_exp2(x2.y)};
            h = __builtin_elementwise_fma(a2, h, f32x2{dw.y, dw.y} * Bv);
            const f32x2 p2 = Cv * h;
            part[j] = p2.x + p2.y;
            if (sp == TB / 2 - 1 && ck) {                            // entry state of the block's second half (scan order)
                *reinterpret_cast<float2 *>(ck + (REV ? 0 : ck_half)) = make_float2(h.x, h.y);
                ck += ck_step;
            }
            if ((sp & 1) == 1) {
                asm volatile("" : "+v"(h), "+v"(part[j]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // sum over the 8 state-pair groups through the per-wave patch (in-order LDS within a wave: no barrier)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4 *>(red_w + 4 * (q ^ (c8 >> 2))) = f32x4{part[4 * q], part[4 * q + 1], part[4 * q + 2], part[4 * q + 3]};
        f32x2 y = *reinterpret_cast<const f32x2 *>(red_r);
#pragma unroll
        for (int g2 = 1; g2 < 8; ++g2) y += *reinterpret_cast<const f32x2 *>(red_r + g2 * 128);
        return y;
    };

    auto gate = [&](const f32x2 &y) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float ov = fmaf(Dv, uq[i], y[i]);
            if (has_pre) st_io(prr, p_off, i * p_ts * S, ov, IO{});
            ov *= zq[i] * cm_sigmoid(zq[i]);
            st_io(orr, o_off, i * o_ts * S, ov, IO{});
        }
        o_off += o_step, p_off += p_step;
    };

    constexpr int XT = TB * XS * 4;
    int t_cur = 0, t_nxt = L::kTile, t_fill = 2 * L::kTile;
    int x_cur = 0, x_nxt = XT, x_fill = 2 * XT;
    issue();
    commit(t_cur, x_cur);
    if (nblk > 1) {
        issue();
        commit(t_nxt, x_nxt);
    }
    __syncthreads();
    produce(t_cur, x_cur, tb0);
    int tb = tb0;
    for (int k = 0; k < nblk; ++k) {
        const bool more = k + 2 < nblk;
        if (more) issue();
        if (ck) *reinterpret_cast<float2 *>(ck + (REV ? ck_half : 0)) = make_float2(h.x, h.y);
        const f32x2 y = recur(x_cur);
        gate(y);
        tb += DIR * TB;
        if (k + 1 < nblk) produce(t_nxt, x_nxt, tb);
        if (more) commit(t_fill, x_fill);
        cm_lds_barrier();
        const int t_old = t_cur, x_old = x_cur;
        t_cur = t_nxt, t_nxt = t_fill, t_fill = t_old;
        x_cur = x_nxt, x_nxt = x_fill, x_fill = x_old;
    }
    if (d.h_last && c_ok) *reinterpret_cast<float2 *>(d.h_last + ((int64_t)b * E + c) * 16 + 2 * gg) = make_float2(h.x, h.y);
    if (d.decay) {
        patch[gg * 8 + c8] = dsum;                                   // owned steps of the 8 lanes of a channel cover the sequence
        float Ssum = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < 8; ++g2) Ssum += patch[g2 * 8 + c8];
        if (c_ok) *reinterpret_cast<float2 *>(d.decay + ((int64_t)b * E + c) * 16 + 2 * gg) = make_float2(cm_exp2(Ap.x * Ssum), cm_exp2(Ap.y * Ssum));
    }
}

template <typename IO>
__global__ __launch_bounds__(256, 4) void scan_rows_fwd2_kernel(const cm_scan_cl_args p, const int nx) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[rows2_lds<IO>::kBytes];
    const int total = gridDim.x;
    int id = blockIdx.x;
    if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);   // workgroups of one (sequence, direction) on one XCD
    const int cx = id % nx, b = (id / nx) % p.batch, z = id / (nx * p.batch);
    const cm_scan_cl_dir &d = p.dir[z];
    if (d.reverse_time) scan_rows2<IO, true>(p, d, lds, cx, b);
    else scan_rows2<IO, false>(p, d, lds, cx, b);
}

}  // namespace

// called by launch_rows (scan_rows_fwd.hip) for unchunked z + softplus launches with dt_rank <= 16 that would otherwise
// leave SIMDs idle (fewer than 512 workgroups of 64 channels)
int cm_scan_rows_fwd2(const cm_scan_cl_args &a) {
    const int nx = (a.dim + CW - 1) / CW;
    const long total = (long)nx * a.batch * a.ndir;
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    if (a.io_dtype == CM_BF16) hipLaunchKernelGGL((scan_rows_fwd2_kernel<cm_bf16>), dim3((unsigned)total), dim3(256), 0, st, a, nx);
    else hipLaunchKernelGGL((scan_rows_fwd2_kernel<float>), dim3((unsigned)total), dim3(256), 0, st, a, nx);
    return cm_launch_status("cm_scan_cl_fwd(rows, 2 states per lane)");
}
