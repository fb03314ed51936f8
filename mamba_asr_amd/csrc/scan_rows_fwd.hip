// scan_rows_fwd.hip — channels-last selective scan forward, "row-group" variant (contract: cm_scan_cl_fwd with
// cm_scan_cl_dir.xdbl set; reference semantics selective_scan_interface.py:91-157 + the dt_proj GEMM of :187).
//
// Why a second shape.  scan_cl_fwd.hip splits the 16 states of a channel over 8 WAVES: every step each wave reads
// (delta', delta'*u) from LDS, writes a partial output to LDS, an "owner" wave sums 8 partials, and the workgroup
// barriers every 8 steps — ~30 LDS instructions per wave per block and a barrier domain per CU set the pace.
// Here the 16 states of a channel sit in 4 LANES of ONE wave (lane = channel%16 + 16*group, 4 states per lane), and the
// otherwise idle matrix pipe does the cross-lane work in fp32:
//   * y[t][c] = sum over the 4 lane groups of the per-lane partial sum_i C[t][4g+i] h_i is ONE v_mfma_f32_16x16x4_f32
//     per step: A = one-hot row selector (row t), B = the partials (k = lane group, n = channel), accumulated over the 16
//     steps of a block, so that afterwards lane (c, g) holds y for steps 4g..4g+3 of its channel;
//   * delta[t][c] = W_dt[c,:] . dt[t,:] is four more MFMAs per 16-step block and lands in the SAME lane layout, so
//     softplus, delta*u, the D skip and the SiLU gate are done exactly once per (channel, step), spread over all 64
//     lanes, with no cross-wave exchange at all;
//   * (delta', delta'*u) go from that layout to the recurrence layout through a 2.3 KB per-wave LDS patch (in-order LDS
//     within a wave: no barrier); B_t/C_t come from the staged x_dbl tile with 16-byte broadcast reads;
//   * the four waves of a workgroup (64 channels = one 128-byte row segment) only share the staged input tiles
//     (u, z, x_dbl rows: 16 steps per tile, 3 tiles in flight, every thread one 16-byte global load per tensor), which
//     costs ONE barrier per 16 steps, a block behind the loads.
// x_dbl is read as the x_proj GEMM wrote it: rows (batch*time, [dt(16, zero padded) | B(16) | C(16)]) in the I/O dtype —
// no transposed copy, no fp32 staging buffer.
#include "scan_rows_common.h"

namespace {

constexpr int NBUF = 3;       // staged input tiles
constexpr int PWS = 36;       // floats per channel in the per-wave (delta', delta'*u) patch (32 + pad)

template <typename IO> struct rows_lds {
    static constexpr int kTile = TB * 64 * (int)sizeof(IO);
    static constexpr int kU = 0, kZ = NBUF * kTile, kX = 2 * NBUF * kTile;
    static constexpr int kPw = kX + NBUF * TB * XS * 4;
    static constexpr int kPatch = 4 * 16 * 16 * 4;        // per-wave patch: 4 KB partial-output exchange, overlaid by the
    static constexpr int kBytes = kPw + 4 * kPatch;       // 2.3 KB (delta', delta'*u) patch while that one is live
};

// How a launch covers the time axis.  chunks == 1: one workgroup runs a whole sequence.  chunks > 1 (small batches: too few
// (sequence, 64-channel group) pairs to fill 1024 SIMDs): sequences are cut into chunks of chunk_len steps and the launch
// becomes three -- pass 1 runs the recurrence of every chunk from a zero state and keeps only its summary (decay product,
// end state), rows_carry_kernel folds the summaries in scan order into each chunk's entry state
// (H_k = decay_{k-1} H_{k-1} + end_{k-1}), pass 2 runs every chunk again from its entry state and writes the outputs.
// It is the algebra of the time-split scan across GPUs (seqpar.py) applied inside one.
struct rows_plan {
    int nx;                  // 64-channel groups
    int chunks, chunk_len;   // chunk_len is a multiple of TB
    int pass;                // 0: single pass, 1: summaries, 2: outputs from carried-in states
    int need_last;           // pass 1 also runs the last chunk in scan order (the caller asked for h_last / decay)
    float *h0, *hl, *dc;     // workspace, each (ndir, batch, chunks, dim, 16) fp32
};

// FULL: z gate and softplus present (the BiMamba layer's call), resolved at compile time
// SUM: summary pass of a chunked launch -- no output contraction, gate, z or stores
// ABL (timing-only ablations, cm_debug_set): 1 = exp replaced by a multiply-add, 2 = B/C not read from LDS,
// 3 = no per-(channel,step) owner work (softplus / gate), 4 = plain add instead of the output MFMA, 5 = no staging
// DTR: zero-padded dt_rank of the x_dbl rows, 16 or 32 (32: bf16 only; rows are then [dt32 | B16 | C16] and the K = 32 bf16
// MFMA that forms delta contracts real columns in all four lane groups -- the S2S-large encoder, d_model 512)
// Runs steps [t_lo, t_lo + T) of sequence b; h0 / h_last / decay point at this run's (dim, 16) carry slabs or are NULL.
template <typename IO, bool REV, bool FULL, int ABL = 0, int DTR = 16, bool SUM = false>
__device__ __forceinline__ void scan_rows(const cm_scan_cl_args &p, const cm_scan_cl_dir &d, unsigned char *lds,
                                          const int cx, const int b, const int t_lo, const int T,
                                          const float *h0, float *h_last, float *decay) {
    using L = rows_lds<IO>;
    constexpr int S = (int)sizeof(IO);
    constexpr int VEC = cm_elem<IO>::kVec;       // elements per 16-byte vector
    constexpr int CPR = 64 / VEC;                // 16-byte chunks per 64-channel row
    constexpr int NCH = TB * CPR;                // chunks per (u or z) tile
    constexpr int NV = 2 * NCH / 256;            // chunks per thread per block (u and z together)
    constexpr int RW = DTR + 32;                 // x_dbl row width
    constexpr int XCPR = RW / VEC;               // chunks per x_dbl row
    constexpr int DTC = DTR / 8;                 // bf16 I/O: 16-byte chunks of raw dt columns per row
    static_assert(DTR == 16 || (DTR == 32 && sizeof(IO) == 2), "dt_rank 32 needs bf16 I/O");
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    const int E = p.dim, c0 = cx * 64;
    const int c = c0 + 16 * w + c16;
    const bool c_ok = c < E;
    const int cc = c_ok ? c : E - 1;
    const bool has_z = !SUM && (FULL || p.z != nullptr);
    const bool softplus = FULL || p.delta_softplus != 0;
    const int nblk = (T + TB - 1) / TB;
    const int u_ts = (int)d.u_ts, z_ts = (int)p.z_ts, x_ts = (int)d.xdbl_ts, o_ts = (int)d.out_ts;
    const __amdgpu_buffer_rsrc_t ur = make_rsrc(reinterpret_cast<const IO *>(d.u) + (int64_t)b * d.u_bs + (int64_t)t_lo * u_ts,
                                                ((int64_t)(T - 1) * u_ts + E) * S);
    const __amdgpu_buffer_rsrc_t zr = make_rsrc(has_z ? reinterpret_cast<const IO *>(p.z) + (int64_t)b * p.z_bs + (int64_t)t_lo * z_ts : nullptr,
                                                has_z ? ((int64_t)(T - 1) * z_ts + E) * S : 0);
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(reinterpret_cast<const IO *>(d.xdbl) + (int64_t)b * d.xdbl_bs + (int64_t)t_lo * x_ts,
                                                ((int64_t)(T - 1) * x_ts + RW) * S);
    const __amdgpu_buffer_rsrc_t orr = make_rsrc(SUM ? nullptr : reinterpret_cast<IO *>(d.out) + (int64_t)b * d.out_bs + (int64_t)t_lo * o_ts,
                                                 SUM ? 0 : ((int64_t)(T - 1) * o_ts + E) * S);
    const bool has_pre = !SUM && d.ypre != nullptr;          // training forward: pre-gate output for cm_scan_cl_bwd
    const int p_ts = (int)d.ypre_ts;
    const __amdgpu_buffer_rsrc_t prr = make_rsrc(has_pre ? reinterpret_cast<IO *>(d.ypre) + (int64_t)b * d.ypre_bs + (int64_t)t_lo * p_ts : nullptr,
                                                 has_pre ? ((int64_t)(T - 1) * p_ts + E) * S : 0);
    const int tb0 = (REV ? nblk - 1 : 0) * TB;               // first block's base step; blocks advance by +-TB steps
    constexpr int DIR = REV ? -1 : 1;

    // ---- staging: global -> registers (issue) -> LDS (commit), one block of 16 steps at a time.
    // Per-thread byte offsets into the batch slice advance by one block per issue (one add per tensor).
    int uz_off[NV], uz_step[NV], uz_lds[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = tid + 256 * i, tens = idx / NCH, within = idx % NCH;
        const int ts = tens ? z_ts : u_ts;
        uz_off[i] = ((tb0 + within / CPR) * ts + c0 + (within % CPR) * VEC) * S;
        uz_step[i] = DIR * TB * ts * S;
        uz_lds[i] = (tens ? L::kZ : L::kU) + within * 16;
    }
    const bool x_thread = tid < TB * XCPR;
    int x_off = x_thread ? ((tb0 + tid / XCPR) * x_ts + (tid % XCPR) * VEC) * S : 0x7fffffff;
    const int x_step = x_thread ? DIR * TB * x_ts * S : 0;
    // bf16 I/O: the dt columns (chunks 0, 1 of a row) stay RAW bf16 in the first 32 bytes of the staged row -- they are
    // the A operand of one bf16 MFMA; B and C (chunks 2..5) are widened to fp32 at floats 16..47 for the recurrence
    const bool x_raw = S == 2 && tid % XCPR < DTC;
    const int x_lds = L::kX + (tid / XCPR) * XS * 4 +
                      (x_raw ? (tid % XCPR) * 16 : (S == 2 ? (16 + (tid % XCPR - DTC) * VEC) * 4 : (tid % XCPR) * VEC * 4));
    u32x4 ruz[NV], rx;
    auto issue = [&]() {                                          // next block in processing order
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int tens = (w * 64 + 256 * i) / NCH;                // wave-uniform (NCH is a multiple of 64)
            ruz[i] = (tens == 0 || has_z) ? __builtin_amdgcn_raw_buffer_load_b128(tens ? zr : ur, uz_off[i], 0, 0) : u32x4{0u, 0u, 0u, 0u};
            uz_off[i] += uz_step[i];
        }
        rx = __builtin_amdgcn_raw_buffer_load_b128(xr, x_off, 0, 0);
        x_off += x_step;
    };
    auto commit = [&](const int buf_tile, const int buf_x) {      // byte offsets of the destination tiles
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int tens = (w * 64 + 256 * i) / NCH;
            if (tens == 0 || has_z) *reinterpret_cast<u32x4 *>(lds + uz_lds[i] + buf_tile) = ruz[i];
        }
        if (x_thread) {
            if (x_raw) *reinterpret_cast<u32x4 *>(lds + x_lds + buf_x) = rx;
            else unpack_store(reinterpret_cast<float *>(lds + x_lds + buf_x), uint4{rx[0], rx[1], rx[2], rx[3]}, IO{});
        }
    };

    // ---- per-lane constants: 4 states of one channel
    f32x2 Ap01, Ap23, h01 = {0.f, 0.f}, h23 = {0.f, 0.f};
    float Wdt[4];
    bf16x8 Wdt8;                                                  // bf16 I/O: W_dt[c][8g .. 8g+7] for g < 2, zero above (K = 32 padded)
    {
        const float4 a4 = *reinterpret_cast<const float4 *>(d.A + (int64_t)cc * 16 + 4 * g);
        Ap01 = f32x2{a4.x * CM_LOG2E, a4.y * CM_LOG2E};
        Ap23 = f32x2{a4.z * CM_LOG2E, a4.w * CM_LOG2E};
        // B operand of the delta MFMAs: lane (n = channel, k = lane group) holds W_dt[c][4k + q], q = 0..3
        const float4 w4 = *reinterpret_cast<const float4 *>(d.dt_weight + (int64_t)cc * 16 + 4 * g);
        Wdt[0] = w4.x; Wdt[1] = w4.y; Wdt[2] = w4.z; Wdt[3] = w4.w;
        if constexpr (S == 2) {
            // The reference runs this product as a bf16 GEMM under autocast (selective_scan_interface.py:187: weight and
            // x_dbl both cast to bf16, fp32 accumulate); here too, except that delta stays fp32 afterwards.
            const float *wr = d.dt_weight + (int64_t)cc * DTR + 8 * (DTR == 32 ? g : (g & 1));
            const float4 lo = *reinterpret_cast<const float4 *>(wr), hi = *reinterpret_cast<const float4 *>(wr + 4);
            const float sc = (DTR == 32 || g < 2) ? 1.f : 0.f;
            typedef float f32x8 __attribute__((ext_vector_type(8)));
            Wdt8 = __builtin_convertvector(f32x8{lo.x * sc, lo.y * sc, lo.z * sc, lo.w * sc, hi.x * sc, hi.y * sc, hi.z * sc, hi.w * sc}, bf16x8);
        }
    }
    if (h0) {                                                     // carry of a time-split scan (seqpar.py / chunked launch), else zero
        const float4 h4 = *reinterpret_cast<const float4 *>(h0 + (int64_t)cc * 16 + 4 * g);
        h01 = f32x2{h4.x, h4.y};
        h23 = f32x2{h4.z, h4.w};
    }
    float dsum = 0.f;                                             // sum of this lane's owned delta' (for the decay output)
    const float bias = d.delta_bias ? d.delta_bias[cc] : 0.f;
    const float Dv = d.D ? d.D[cc] : 0.f;
    float uq[4], zq[4];                                           // (u, z) of the lane's 4 owned steps of the staged block
    float *patch = reinterpret_cast<float *>(lds + L::kPw + w * L::kPatch);
    float *pww = patch + c16 * PWS;
    // partial-output exchange: [group][channel][16 steps], the 4-step quads of a row XOR-swizzled by channel/4 so that
    // both the 16-byte writes (lane = channel) and the 16-byte reads are bank-conflict free without padding
    float *red_w = patch + (g * 16 + c16) * 16;
    const float *red_r = patch + c16 * 16 + 4 * (g ^ (c16 >> 2));
    int o_off = c_ok ? ((tb0 + 4 * g) * o_ts + c) * S : 0x7fffffff;   // out-of-range offset: stores dropped
    const int o_step = c_ok ? DIR * TB * o_ts * S : 0;
    int p_off = (c_ok && has_pre) ? ((tb0 + 4 * g) * p_ts + c) * S : 0x7fffffff;
    const int p_step = (c_ok && has_pre) ? DIR * TB * p_ts * S : 0;
    // training forward: the state each half block of 8 steps [8 m, 8 m + 8) is entered with (scan order),
    // (batch, 2 ceil(seqlen / 16), dim, 16) fp32; ck points at the block's first time half
    float *ck = (!SUM && d.ckpt && c_ok) ? d.ckpt + (((int64_t)b * 2 * ((p.seqlen + TB - 1) / TB) + 2 * ((t_lo + tb0) / TB)) * E + c) * 16 + 4 * g : nullptr;
    const int64_t ck_step = (int64_t)DIR * 2 * E * 16, ck_half = (int64_t)E * 16;

    // delta' = softplus(W_dt . dt + bias), delta'*u for the block staged at (buf_tile, buf_x) -> per-wave patch;
    // owned (u, z) -> registers.  tb = the block's base step.
    auto produce_impl = [&](const int buf_tile, const int buf_x, const int tb, auto ragged_tag) {
        constexpr bool ragged = decltype(ragged_tag)::value;      // only the sequence's last block has padded steps
        const float *xt = reinterpret_cast<const float *>(lds + L::kX + buf_x);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if constexpr (S == 2) {
            // ONE v_mfma_f32_16x16x32_bf16 (4 passes) instead of four fp32 MFMAs (8 passes each).  A operand: lane
            // (m = step = lane%16, k block = lane/16) holds dt[step][8k .. 8k+7]; the blocks k = 2, 3 re-read blocks
            // 0, 1 (finite data) against zero weights.
            const bf16x8 dt8 = *reinterpret_cast<const bf16x8 *>(xt + c16 * XS + 4 * (DTR == 32 ? g : (g & 1)));
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dt8, Wdt8, acc, 0, 0, 0);
        } else {
            // A operand: lane (m = step = lane%16, k = lane/16) holds dt[step][4k + q]
            const f32x4 dtf = *reinterpret_cast<const f32x4 *>(xt + c16 * XS + 4 * g);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dtf[0], Wdt[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dtf[1], Wdt[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dtf[2], Wdt[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dtf[3], Wdt[3], acc, 0, 0, 0);
        }
        // acc[i] = delta_raw[step 4g+i][channel c16]
        const IO *ut = reinterpret_cast<const IO *>(lds + L::kU + buf_tile) + 16 * w + c16 + 4 * g * 64;
        const IO *zt = reinterpret_cast<const IO *>(lds + L::kZ + buf_tile) + 16 * w + c16 + 4 * g * 64;
        float o[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float dv = acc[i] + bias;
            if (softplus && ABL != 3) dv = softplus_rows(dv);
            if (ragged) dv = tb + 4 * g + i < T ? dv : 0.f;       // padded steps: a = 1, b = 0 (state passes through)
            dsum += dv;
            const float uv = ld_io(ut + i * 64);
            uq[i] = uv;
            zq[i] = has_z ? ld_io(zt + i * 64) : 0.f;
            o[2 * i] = dv;
            o[2 * i + 1] = dv * uv;
        }
        *reinterpret_cast<f32x4 *>(pww + 8 * g) = f32x4{o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f32x4 *>(pww + 8 * g + 4) = f32x4{o[4], o[5], o[6], o[7]};
    };

    auto produce = [&](const int buf_tile, const int buf_x, const int tb) {
        if (tb + TB > T) produce_impl(buf_tile, buf_x, tb, std::true_type{});      // uniform branch: the selects that zero
        else produce_impl(buf_tile, buf_x, tb, std::false_type{});                // padded steps stay out of the common path
    };

    // 16 recurrence steps on the block staged at buf_x; y[step][channel] accumulated by the matrix pipe.
    // Operands of steps 2s+2, 2s+3 are read from LDS while steps 2s, 2s+1 compute; the scheduling barriers keep the
    // compiler from hoisting all 16 steps' reads to the top (it did: 204 VGPRs, half the occupancy).
    struct StepOps { float2 dw; f32x4 B, C; };
    auto recur = [&](const int buf_x) -> f32x4 {
        const float *xt = reinterpret_cast<const float *>(lds + L::kX + buf_x) + 4 * g;
        auto slot = [](int sp) { return REV ? TB - 1 - sp : sp; };
        auto fetch = [&](int sp, StepOps &o) {
            const int j = slot(sp);
            o.dw = *reinterpret_cast<const float2 *>(pww + 2 * j);
            if constexpr (ABL == 2) { o.B = f32x4{o.dw.x, o.dw.y, o.dw.x, o.dw.y}; o.C = o.B; }
            else {
                o.B = *reinterpret_cast<const f32x4 *>(xt + j * XS + 16);
                if constexpr (!SUM) o.C = *reinterpret_cast<const f32x4 *>(xt + j * XS + 32);
            }
        };
        float part[TB];                                           // this lane's 4-state partial outputs, by step
        // packed fp32 (v_pk_mul_f32 / v_pk_fma_f32 are full rate: two states per instruction)
        auto step = [&](int sp, const StepOps &o) {
            const f32x2 d2 = {o.dw.x, o.dw.x}, du2 = {o.dw.y, o.dw.y};
            const f32x2 x01 = d2 * Ap01, x23 = d2 * Ap23;
            f32x2 a01, a23;
            if constexpr (ABL == 1) { a01 = x01 * 0.5f + 1.0f; a23 = x23 * 0.5f + 1.0f; }
            else { a01 = f32x2{cm_exp2(x01.x), cm_exp2(x01.y)}; a23 = f32x2{cm_exp2(x23.x), cm_exp2(x23.y)}; }
            const f32x2 b01 = du2 * f32x2{o.B[0], o.B[1]}, b23 = du2 * f32x2{o.B[2], o.B[3]};
            h01 = __builtin_elementwise_fma(a01, h01, b01);
            h23 = __builtin_elementwise_fma(a23, h23, b23);
            if constexpr (!SUM) {
                f32x2 p2 = f32x2{o.C[0], o.C[1]} * h01;
                p2 = __builtin_elementwise_fma(f32x2{o.C[2], o.C[3]}, h23, p2);
                part[slot(sp)] = p2.x + p2.y;
            }
        };
        StepOps q[2][2];
        fetch(0, q[0][0]);
        fetch(1, q[0][1]);
#pragma unroll
        for (int sg = 0; sg < TB / 2; ++sg) {
            if (sg + 1 < TB / 2) {
                fetch(2 * sg + 2, q[(sg + 1) & 1][0]);
                fetch(2 * sg + 3, q[(sg + 1) & 1][1]);
            }
            step(2 * sg, q[sg & 1][0]);
            step(2 * sg + 1, q[sg & 1][1]);
            if constexpr (!SUM) {
                if (sg == TB / 4 - 1 && ck) {                         // 8 steps done: entry state of the block's second half (scan order)
                    *reinterpret_cast<float4 *>(ck + (REV ? 0 : ck_half)) = make_float4(h01.x, h01.y, h23.x, h23.y);
                    ck += ck_step;
                }
            }
            // pin the state here: machine-sink otherwise moves the whole h chain below the last scheduling barrier
            // (its results are only consumed at the end of the block) and keeps 64 exp results alive instead
            if constexpr (SUM) asm volatile("" : "+v"(h01), "+v"(h23));
            else asm volatile("" : "+v"(h01), "+v"(h23), "+v"(part[slot(2 * sg)]), "+v"(part[slot(2 * sg + 1)]));
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (SUM) return f32x4{0.f, 0.f, 0.f, 0.f};
        // sum over the 4 lane groups through the per-wave patch (the (delta', delta'*u) patch is dead by now; LDS
        // operations of one wave execute in order, so no barrier): afterwards lane (c, g) holds y of steps 4g..4g+3.
        // (An MFMA with a one-hot A operand per step did this too, but v_mfma_f32_16x16x4_f32 holds the SIMD for 8
        // passes: 16 of them cost 22 % of the kernel.)
        if constexpr (ABL == 4) return f32x4{part[0], part[1], part[2], part[3]};
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4 *>(red_w + 4 * (q ^ (c16 >> 2))) = f32x4{part[4 * q], part[4 * q + 1], part[4 * q + 2], part[4 * q + 3]};
        f32x4 y = *reinterpret_cast<const f32x4 *>(red_r);
#pragma unroll
        for (int gg = 1; gg < 4; ++gg) y += *reinterpret_cast<const f32x4 *>(red_r + gg * 256);
        return y;
    };

    auto gate = [&](const f32x4 &y) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float ov = fmaf(Dv, uq[i], y[i]);
            if (has_pre) st_io(prr, p_off, i * p_ts * S, ov, IO{});
            if (has_z && ABL != 3) ov *= zq[i] * cm_sigmoid(zq[i]);
            st_io(orr, o_off, i * o_ts * S, ov, IO{});
        }
        o_off += o_step;
        p_off += p_step;
    };

    // tile ring: byte offsets of the (u/z, x_dbl) tiles holding block k, k+1 and the one block k+2 is committed to
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
        const bool more = ABL != 5 && k + 2 < nblk;
        if (more) issue();
        if constexpr (!SUM) {
            if (ck) *reinterpret_cast<float4 *>(ck + (REV ? ck_half : 0)) = make_float4(h01.x, h01.y, h23.x, h23.y);
        }
        const f32x4 y = recur(x_cur);
        if constexpr (!SUM) gate(y);
        tb += DIR * TB;
        if (k + 1 < nblk) produce(t_nxt, x_nxt, tb);
        if (more) commit(t_fill, x_fill);
        if (ABL != 5) cm_lds_barrier();                          // LDS-only: output stores and prefetch loads stay in flight
        const int t_old = t_cur, x_old = x_cur;
        t_cur = t_nxt; t_nxt = t_fill; t_fill = t_old;
        x_cur = x_nxt; x_nxt = x_fill; x_fill = x_old;
    }
    // ---- carry outputs of a time shard: last state, and the factor a state entering the shard is multiplied by
    if (h_last && c_ok)
        *reinterpret_cast<float4 *>(h_last + (int64_t)c * 16 + 4 * g) = make_float4(h01.x, h01.y, h23.x, h23.y);
    if (decay) {
        // sum of delta' over the sequence for channel c16: this lane's owned steps + the other three lane groups', through the
        // per-wave patch (LDS operations of one wave execute in order: no barrier)
        patch[g * 16 + c16] = dsum;
        const float S = (patch[c16] + patch[16 + c16]) + (patch[32 + c16] + patch[48 + c16]);
        if (c_ok)
            *reinterpret_cast<float4 *>(decay + (int64_t)c * 16 + 4 * g) =
                make_float4(cm_exp2(Ap01.x * S), cm_exp2(Ap01.y * S), cm_exp2(Ap23.x * S), cm_exp2(Ap23.y * S));
    }
}

template <typename IO, int ABL, int DTR = 16, bool SUM = false>
__global__ __launch_bounds__(256, sizeof(IO) == 2 ? 4 : 3) void scan_rows_fwd_kernel(const cm_scan_cl_args p, const rows_plan pl) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[rows_lds<IO>::kBytes];
    // workgroups of one (batch, direction) share x_dbl rows and neighbouring row segments: keep them on one XCD
    // (consecutive workgroup ids are dealt round-robin to the 8 XCDs)
    const int total = gridDim.x, nx = pl.nx, nbk = p.batch * pl.chunks;
    int id = blockIdx.x;
    if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
    const int cx = id % nx, bk = (id / nx) % nbk, z = id / (nx * nbk);
    const int b = bk / pl.chunks, k = bk % pl.chunks;
    const cm_scan_cl_dir &d = p.dir[z];
    const int t_lo = k * pl.chunk_len, T = min(p.seqlen - t_lo, pl.chunk_len);
    const float *h0;
    float *h_last, *decay;
    if (pl.pass == 0) {
        const int64_t slab = (int64_t)b * p.dim * 16;
        h0 = d.h0 ? d.h0 + slab : nullptr;
        h_last = d.h_last ? d.h_last + slab : nullptr;
        decay = d.decay ? d.decay + slab : nullptr;
    } else {
        const int64_t slab = (((int64_t)z * p.batch + b) * pl.chunks + k) * p.dim * 16;
        if (SUM) {
            // the last chunk in scan order hands nothing on
            if (!pl.need_last && k == (d.reverse_time ? 0 : pl.chunks - 1)) return;
            h0 = nullptr, h_last = pl.hl + slab, decay = pl.dc + slab;
        } else {
            h0 = pl.h0 + slab, h_last = nullptr, decay = nullptr;
        }
    }
    const bool full = p.z != nullptr && p.delta_softplus != 0;
    if (full) {
        if (d.reverse_time) scan_rows<IO, true, true, ABL, DTR, SUM>(p, d, lds, cx, b, t_lo, T, h0, h_last, decay);
        else scan_rows<IO, false, true, ABL, DTR, SUM>(p, d, lds, cx, b, t_lo, T, h0, h_last, decay);
    } else if constexpr (ABL == 0) {
        if (d.reverse_time) scan_rows<IO, true, false, 0, DTR, SUM>(p, d, lds, cx, b, t_lo, T, h0, h_last, decay);
        else scan_rows<IO, false, false, 0, DTR, SUM>(p, d, lds, cx, b, t_lo, T, h0, h_last, decay);
    }
}

// Entry state of every chunk from the chunk summaries, in scan order; the caller's h0 enters the first chunk, the caller's
// h_last / decay (whole-sequence values) leave the last.  One thread per (direction, sequence, channel, 4 states).
__global__ __launch_bounds__(256) void rows_carry_kernel(const cm_scan_cl_args p, const rows_plan pl) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int E = p.dim, C = pl.chunks;
    if (idx >= (int64_t)p.ndir * p.batch * E * 4) return;
    const int q = (int)(idx & 3), c = (int)((idx >> 2) % E), b = (int)((idx >> 2) / E % p.batch), z = (int)((idx >> 2) / E / p.batch);
    const cm_scan_cl_dir &d = p.dir[z];
    const int64_t user = ((int64_t)b * E + c) * 16 + 4 * q;
    float4 H = d.h0 ? *reinterpret_cast<const float4 *>(d.h0 + user) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 P = make_float4(1.f, 1.f, 1.f, 1.f);
    for (int i = 0; i < C; ++i) {
        const int k = d.reverse_time ? C - 1 - i : i;
        const int64_t off = ((((int64_t)z * p.batch + b) * C + k) * E + c) * 16 + 4 * q;
        *reinterpret_cast<float4 *>(pl.h0 + off) = H;
        if (i + 1 < C || pl.need_last) {
            const float4 a = *reinterpret_cast<const float4 *>(pl.dc + off), e = *reinterpret_cast<const float4 *>(pl.hl + off);
            H = make_float4(fmaf(a.x, H.x, e.x), fmaf(a.y, H.y, e.y), fmaf(a.z, H.z, e.z), fmaf(a.w, H.w, e.w));
            P = make_float4(P.x * a.x, P.y * a.y, P.z * a.z, P.w * a.w);
        }
    }
    if (d.h_last) *reinterpret_cast<float4 *>(d.h_last + user) = H;
    if (d.decay) *reinterpret_cast<float4 *>(d.decay + user) = P;
}

// chunk length for a requested chunk count: whole 16-step blocks, so only a sequence's last chunk is ragged
inline int rows_chunk_len(int seqlen, int chunks) { return ((seqlen + chunks - 1) / chunks + TB - 1) / TB * TB; }

template <typename IO, int DTR>
int launch_rows_chunked(const cm_scan_cl_args &a, rows_plan pl) {
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    const dim3 grid((unsigned)((long)pl.nx * a.batch * pl.chunks * a.ndir)), block(256);
    pl.pass = 1;
    hipLaunchKernelGGL((scan_rows_fwd_kernel<IO, 0, DTR, true>), grid, block, 0, st, a, pl);
    if (int rc = cm_launch_status("cm_scan_cl_fwd(rows, chunk summaries)")) return rc;
    const long nthr = (long)a.ndir * a.batch * a.dim * 4;
    hipLaunchKernelGGL(rows_carry_kernel, dim3((unsigned)((nthr + 255) / 256)), block, 0, st, a, pl);
    if (int rc = cm_launch_status("cm_scan_cl_fwd(rows, carry)")) return rc;
    pl.pass = 2;
    hipLaunchKernelGGL((scan_rows_fwd_kernel<IO, 0, DTR, false>), grid, block, 0, st, a, pl);
    return cm_launch_status("cm_scan_cl_fwd(rows, chunked)");
}

}  // namespace
int cm_scan_rows_fwd2(const cm_scan_cl_args &a);        // scan_rows_fwd2.hip: 2 states per lane
namespace {

template <typename IO>
int launch_rows(const cm_scan_cl_args &a) {
    rows_plan pl{};
    pl.nx = (a.dim + 63) / 64;
    pl.chunks = 1;
    pl.chunk_len = (a.seqlen + TB - 1) / TB * TB;
    if (a.time_chunks > 1) {
        pl.chunk_len = rows_chunk_len(a.seqlen, a.time_chunks);
        pl.chunks = (a.seqlen + pl.chunk_len - 1) / pl.chunk_len;
    }
    if (pl.chunks > 1) {
        const int64_t slab = (int64_t)a.ndir * a.batch * pl.chunks * a.dim * 16;
        pl.h0 = reinterpret_cast<float *>(a.workspace);
        pl.hl = pl.h0 + slab;
        pl.dc = pl.hl + slab;
        for (int i = 0; i < a.ndir; ++i) pl.need_last |= a.dir[i].h_last != nullptr || a.dir[i].decay != nullptr;
        if constexpr (sizeof(IO) == 2) {
            if (a.dir[0].dt_rank > 16) return launch_rows_chunked<IO, 32>(a, pl);
        }
        return launch_rows_chunked<IO, 16>(a, pl);
    }
    const long total = (long)pl.nx * a.batch * a.ndir;
    // lanes_per_channel == 8: the 2-states-per-lane kernel (scan_rows_fwd2.hip: twice the waves for the same launch).  Measured
    // (tools/bench_scan_small.py, profiles/r03/scan_small_batches.log) it is SLOWER at every size -- 16 x 1000 x 512: 133.7 vs
    // 110.6 us, 32 x: 191.6 vs 145.4 -- because what a SIMD with one 4-state wave lacks is independent recurrence chains, and
    // two 2-state waves carry exactly as many (plus 1.3x the owner / staging work); it is therefore never chosen by size.
    if (a.lanes_per_channel == 8 && a.z && a.delta_softplus && a.dir[0].dt_rank <= 16 && cm_debug_get() == 0) return cm_scan_rows_fwd2(a);
    const dim3 grid((unsigned)total), block(256);
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    if constexpr (sizeof(IO) == 2) {
        if (a.dir[0].dt_rank > 16) {                    // 64-wide rows, dt features padded to 32
            hipLaunchKernelGGL((scan_rows_fwd_kernel<IO, 0, 32>), grid, block, 0, st, a, pl);
            return cm_launch_status("cm_scan_cl_fwd(rows, dt_rank 32)");
        }
#ifdef CM_ABLATE
        switch (cm_debug_get()) {                       // ablation builds exist for the bf16 kernel only
            case 1: hipLaunchKernelGGL((scan_rows_fwd_kernel<IO, 1>), grid, block, 0, st, a, pl); return cm_launch_status("rows abl1");
            case 2: hipLaunchKernelGGL((scan_rows_fwd_kernel<IO, 2>), grid, block, 0, st, a, pl); return cm_launch_status("rows abl2");
            case 3: hipLaunchKernelGGL((scan_rows_fwd_kernel<IO, 3>), grid, block, 0, st, a, pl); return cm_launch_status("rows abl3");
            case 4: hipLaunchKernelGGL((scan_rows_fwd_kernel<IO, 4>), grid, block, 0, st, a, pl); return cm_launch_status("rows abl4");
            case 5: hipLaunchKernelGGL((scan_rows_fwd_kernel<IO, 5>), grid, block, 0, st, a, pl); return cm_launch_status("rows abl5");
            default: break;
        }
#endif
    }
    hipLaunchKernelGGL((scan_rows_fwd_kernel<IO, 0>), grid, block, 0, st, a, pl);
    return cm_launch_status("cm_scan_cl_fwd(rows)");
}

}  // namespace

extern "C" int64_t cm_scan_cl_fwd_workspace_bytes(const cm_scan_cl_args *a) {
    if (!a || a->time_chunks <= 1 || a->seqlen <= 0) return 0;
    const int len = rows_chunk_len(a->seqlen, a->time_chunks);
    const int64_t chunks = (a->seqlen + len - 1) / len;
    return chunks > 1 ? 3 * (int64_t)a->ndir * a->batch * chunks * a->dim * 16 * (int64_t)sizeof(float) : 0;
}

// Chunk count that fills the chip.  Chunking runs the recurrence twice (the summary pass keeps 56 of the 68 issue cycles of a
// step: the four exponentials stay), so it pays only while compute units sit idle: measured on MI355X (tools/
// bench_scan_chunks.py, profiles/r02/scan_time_chunks.log) 8 x 1000 x 512 97 -> 68 us, 4 x 4000 x 1024 378 -> 215 us,
// 1 x 4000 x 512 376 -> 61 us, but 16 x 1000 x 512 (one workgroup per CU already) 106 -> 112-116 us.  Below 256 workgroups the
// sequences are cut so that (64-channel groups) x batch x directions x chunks reaches 1024, chunks no shorter than 128 steps.
extern "C" int32_t cm_scan_cl_fwd_auto_chunks(int32_t batch, int32_t seqlen, int32_t dim, int32_t ndir) {
    if (batch <= 0 || seqlen <= 0 || dim <= 0 || ndir <= 0) return 1;
    const long wgs = (long)((dim + 63) / 64) * batch * ndir;
    if (wgs >= 256) return 1;
    long c = 1024 / wgs;
    if (c > seqlen / 128) c = seqlen / 128;
    return c < 2 ? 1 : (int32_t)c;
}

// called by cm_scan_cl_fwd (scan_cl_fwd.hip) when every direction carries xdbl
int cm_scan_rows_fwd(const cm_scan_cl_args &a) {
    const int vec = a.io_dtype == CM_BF16 ? 8 : 4;
    CM_REQUIRE(a.io_dtype == CM_BF16 || a.io_dtype == CM_F32, CM_EUNSUPPORTED, "scan_cl_fwd(xdbl): io dtype %d unsupported", a.io_dtype);
    CM_REQUIRE(a.dim % vec == 0, CM_EUNSUPPORTED, "scan_cl_fwd(xdbl): dim %d must be a multiple of %d", a.dim, vec);
    CM_REQUIRE(!a.z || (cm_aligned(a.z, 16) && a.z_bs % vec == 0 && a.z_ts % vec == 0), CM_EALIGN,
               "scan_cl_fwd(xdbl): z must be 16-byte aligned with strides that are multiples of %d", vec);
    CM_REQUIRE(a.time_chunks >= 0 && a.time_chunks <= 4096, CM_EINVAL, "scan_cl_fwd(xdbl): time_chunks %d out of range", a.time_chunks);
    const int chunks = a.time_chunks > 1 ? (a.seqlen + rows_chunk_len(a.seqlen, a.time_chunks) - 1) / rows_chunk_len(a.seqlen, a.time_chunks) : 1;
    CM_REQUIRE((long)((a.dim + 63) / 64) * a.batch * a.ndir * chunks < (1L << 31), CM_EINVAL, "scan_cl_fwd(xdbl): grid too large");
    if (chunks > 1) {
        const int64_t need = cm_scan_cl_fwd_workspace_bytes(&a);
        CM_REQUIRE(a.workspace && cm_aligned(a.workspace, 16) && a.workspace_bytes >= need, CM_EINVAL,
                   "scan_cl_fwd(xdbl): time_chunks %d needs a 16-byte aligned workspace of %lld bytes (cm_scan_cl_fwd_workspace_bytes), got %lld",
                   a.time_chunks, (long long)need, (long long)a.workspace_bytes);
    }
    for (int i = 0; i < a.ndir; ++i) {
        const cm_scan_cl_dir &d = a.dir[i];
        CM_REQUIRE(d.u && d.xdbl && d.A && d.dt_weight && d.out, CM_EINVAL, "scan_cl_fwd(xdbl): dir %d has a NULL tensor", i);
        CM_REQUIRE(d.dt_rank >= 0 && d.dt_rank <= 32 && (d.dt_rank > 16) == (a.dir[0].dt_rank > 16), CM_EUNSUPPORTED,
                   "scan_cl_fwd(xdbl): dt_rank %d: at most 32, and every direction on the same side of 16", d.dt_rank);
        CM_REQUIRE(d.dt_rank <= 16 || a.io_dtype == CM_BF16, CM_EUNSUPPORTED,
                   "scan_cl_fwd(xdbl): dt_rank %d > 16 (64-wide rows) is built for bf16 I/O only", d.dt_rank);
        CM_REQUIRE(cm_aligned(d.u, 16) && d.u_bs % vec == 0 && d.u_ts % vec == 0 && cm_aligned(d.xdbl, 16) &&
                       d.xdbl_bs % vec == 0 && d.xdbl_ts % vec == 0 && cm_aligned(d.A, 16) && cm_aligned(d.dt_weight, 16),
                   CM_EALIGN, "scan_cl_fwd(xdbl): dir %d: u / xdbl / A / dt_weight must be 16-byte aligned, strides multiples of %d", i, vec);
    }
    return a.io_dtype == CM_BF16 ? launch_rows<cm_bf16>(a) : launch_rows<float>(a);
}
