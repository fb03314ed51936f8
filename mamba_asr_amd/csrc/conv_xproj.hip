// conv_xproj.hip — both BiMamba directions' depthwise conv + SiLU AND their x_proj GEMMs in one pass over x
// (contract: cm_conv_xproj in include/conmamba_hip.h; reference bimamba.py:223-248 for the two convolutions,
// selective_scan_interface.py:182-186 for conv -> x_dbl = conv_out @ x_proj.weight^T).
//
// Separately this was conv_cl_kernel (read x, write u_fwd | u_bwd) followed by a library GEMM that read the 2E-wide u
// rows back to produce 96 numbers per row.  Here a workgroup owns 16 steps x all channels of one utterance:
//   phase 1: every thread convolves 4 channels x 8 steps from a 14-row register window (8-byte loads, 512 B per
//            wave-instruction), applies SiLU, stores u_fwd / u_bwd to HBM (the scan reads them) and keeps a bf16 copy
//            in LDS, token-major with a 32-byte row pad (conflict-free 16-byte fragment reads);
//   phase 2: wave w < 2 = direction multiplies the 16-token tile by that direction's x_proj weight
//            (48 x E: [dt rows zero-padded to 16 | B | C]) with v_mfma_f32_16x16x32_bf16, weights as the A operand
//            straight from their packed image in L2 (cm_ffn_pack_weights layout, 1 KB per fragment), so a lane ends
//            up with 4 consecutive features of one token = one 8-byte store into the x_dbl row.
// HBM traffic: x once (+ a 6-row halo per 16, served by the XCD's L2), u twice E per step written, 192 B of x_dbl per step.
#include "cm_common.h"


namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// steps per workgroup: TT (template), steps per thread TT / 2.  TT = 32 streams each x_proj weight fragment from L2 once per
// TWO token tiles (at TT = 16 the 96 KB of weights per workgroup are more L2 traffic than the kernel's HBM bytes) and reads
// 22 rows per 16 outputs instead of 14 per 8; its LDS tiles (67 KB) leave two workgroups per CU.
constexpr int W = 4;            // conv width
constexpr int PF = 8;           // weight-fragment ring depth

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ uint32_t pack2(float a, float b) {         // one v_cvt_pk_bf16_f32
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ float silu(float a) { return a * cm_sigmoid(a); }

// cm_debug_set(18): per-phase s_memtime stamps of one mid-grid workgroup (waves 0 and 3), read back with cm_debug_read_stamps_cx
__device__ unsigned long long g_cx_stamps[16];
__device__ __forceinline__ bool cm_ksplit_ok(int stamp) { return stamp != 2; }     // cm_debug_set(19): one wave per direction (A / B)
__device__ __forceinline__ unsigned long long cx_now() {
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// NB: 16-column bands of x_dbl per direction: 3 = [dt16 | B | C], 4 = [dt32 | B | C] (dt_rank 17..32: the S2S-large encoder)
template <int TT, int NB>
__global__ __launch_bounds__(256, (TT == 16 && NB == 3) ? 4 : 2) void conv_xproj_kernel(const cm_conv_xproj_args p, const int ntile, const int stamp_arg = 0) {
#ifdef CM_ABLATE
    const int stamp = stamp_arg;
#else
    constexpr int stamp = 0;                                          // product build: no stamps, no A / B variant
#endif
    const bool st_wg = stamp == 1 && blockIdx.x == gridDim.x / 2 + 3;
    unsigned long long ts_[6] = {0, 0, 0, 0, 0, 0};
    if (st_wg) ts_[0] = cx_now();
    constexpr int TH = TT / 2;
    constexpr int NP = NB * 16;                                   // x_dbl columns per direction
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int E = p.dim, T = p.seqlen;
    const int XS = E + 16;                                        // LDS row stride in bf16 elements
    uint16_t *ut[2] = {reinterpret_cast<uint16_t *>(smem), reinterpret_cast<uint16_t *>(smem) + TT * XS};
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // consecutive step tiles of an utterance share halo rows: keep them on one XCD (workgroup ids are dealt round-robin
    // to the 8 XCDs)
    const int total = gridDim.x;
    int id = blockIdx.x;
    if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
    const int b = id / ntile, t0 = (id % ntile) * TT;
    // buffer descriptors over this utterance's slices: rows outside [0, T) fall outside the buffer (reads return 0 = the
    // conv's zero padding, stores are dropped) and per-row addresses are an SGPR offset, so no per-access address math
    const int x_ts = (int)p.x_ts * 2, yf_ts = (int)p.yf_ts * 2, yb_ts = (int)p.yb_ts * 2;      // row strides in bytes
    auto rsrc = [&](const void *base, int64_t bs, int row_bytes) {
        const int64_t bytes = (int64_t)(T - 1) * row_bytes + (int64_t)E * 2;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(reinterpret_cast<const uint16_t *>(base) + (int64_t)b * bs), 0,
                                                 bytes > 0x7fffffff ? 0x7fffffff : (int)bytes, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t xr = rsrc(p.x, p.x_bs, x_ts), fr = rsrc(p.y_fwd, p.yf_bs, yf_ts), br = rsrc(p.y_bwd, p.yb_bs, yb_ts);

    // ---- phase 2's operands, set up first: wave = (direction, K half); the 32-step build requests its weights now
    constexpr bool EARLY = TT == 32 || NB == 4;              // builds with a 256-VGPR budget
    constexpr int NTL = TT / 16;                                  // token tiles per workgroup
    const int l15 = lane & 15, lq = lane >> 4;
    const int nks = E / 32;
    const int dir = wave & 1, kh = wave >> 1;
    const bool ksplit = (nks & 1) == 0 && cm_ksplit_ok(stamp);
    const int ks_lo = ksplit ? kh * (nks / 2) : 0, ks_hi = ksplit ? ks_lo + nks / 2 : (kh == 0 ? nks : 0);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void *>(dir ? p.wx_b : p.wx_f), 0, NP * E * 2, 0x00020000);
    const int vl = lane * 16;
    bf16x8 wq[PF][NB];
    auto wload = [&](int ks, bf16x8(&dst)[NB]) {                   // fragment (band mt, k-tile ks) = 1 KB at (mt*nks + ks)*1024
#pragma unroll
        for (int mt = 0; mt < NB; ++mt)
            dst[mt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, vl, (mt * nks + ks) * 1024, 0));
    };
    bool wq_requested = !EARLY;

    // ---- phase 1: conv + SiLU, both directions.  thread = (4-channel group, half of the tile's steps)
    const int half = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int ts = t0 + TH * half;                                // first step of this thread (wave-uniform)
    for (int cg = tid & 127; cg < E / 4; cg += 128) {
        const int c0 = cg * 4;
        u32x2 raw[TH + 2 * (W - 1)];                              // rows ts-3 .. ts+TH+2, all loads issued first
#pragma unroll
        for (int r = 0; r < TH + 2 * (W - 1); ++r)
            raw[r] = __builtin_amdgcn_raw_buffer_load_b64(xr, c0 * 2, (ts - (W - 1) + r) * x_ts, 0);
        if (!wq_requested) {                                      // behind the rows (HBM) in the in-order return queue: phase 1 waits
            wq_requested = true;                                  // for the rows only, the L2-resident weights land during it
#pragma unroll
            for (int s = 0; s < PF; ++s)
                if (ks_lo + s < ks_hi) wload(ks_lo + s, wq[s]);
        }
        if (st_wg) ts_[1] = cx_now();                             // rows have arrived
        // two channel PAIRS per thread: a staged dword is one pair, so taps, bias and SiLU run as v_pk_*_f32 on (lo, hi),
        // and every staged row is widened to fp32 once, when it enters the 7-row window (the scalar form widened each
        // element at each of its 8 uses: as many shifts as multiply-adds)
        f32x2 wf[2][W], wb[2][W], bf[2], bb[2];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const float4 f0 = *reinterpret_cast<const float4 *>(p.weight_f + (c0 + 2 * pr) * W);
            const float4 f1 = *reinterpret_cast<const float4 *>(p.weight_f + (c0 + 2 * pr + 1) * W);
            const float4 g0 = *reinterpret_cast<const float4 *>(p.weight_b + (c0 + 2 * pr) * W);
            const float4 g1 = *reinterpret_cast<const float4 *>(p.weight_b + (c0 + 2 * pr + 1) * W);
            wf[pr][0] = f32x2{f0.x, f1.x}; wf[pr][1] = f32x2{f0.y, f1.y}; wf[pr][2] = f32x2{f0.z, f1.z}; wf[pr][3] = f32x2{f0.w, f1.w};
            wb[pr][0] = f32x2{g0.x, g1.x}; wb[pr][1] = f32x2{g0.y, g1.y}; wb[pr][2] = f32x2{g0.z, g1.z}; wb[pr][3] = f32x2{g0.w, g1.w};
            bf[pr] = p.bias_f ? f32x2{p.bias_f[c0 + 2 * pr], p.bias_f[c0 + 2 * pr + 1]} : f32x2{0.f, 0.f};
            bb[pr] = p.bias_b ? f32x2{p.bias_b[c0 + 2 * pr], p.bias_b[c0 + 2 * pr + 1]} : f32x2{0.f, 0.f};
        }
        auto widen = [&](int r, int pr) -> f32x2 { return f32x2{cm_bf16_lo(raw[r][pr]), cm_bf16_hi(raw[r][pr])}; };
        auto silu2 = [](f32x2 a) -> f32x2 {                       // a / (1 + 2^(-a log2 e))
            const f32x2 e = a * f32x2{-CM_LOG2E, -CM_LOG2E};
            const f32x2 d = f32x2{cm_exp2(e.x), cm_exp2(e.y)} + f32x2{1.0f, 1.0f};
            return a * f32x2{cm_rcp(d.x), cm_rcp(d.y)};
        };
        f32x2 xw[2 * (W - 1) + 1][2];                             // rows i .. i+6 of the window, [row][pair]
#pragma unroll
        for (int r = 0; r < 2 * (W - 1); ++r) { xw[r][0] = widen(r, 0); xw[r][1] = widen(r, 1); }
#pragma unroll
        for (int i = 0; i < TH; ++i) {
            xw[2 * (W - 1)][0] = widen(i + 2 * (W - 1), 0);
            xw[2 * (W - 1)][1] = widen(i + 2 * (W - 1), 1);
            u32x2 pf, pb;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                f32x2 af = bf[pr], ab = bb[pr];
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    af = __builtin_elementwise_fma(wf[pr][k], xw[k][pr], af);                     // x[t-(W-1)+k]
                    ab = __builtin_elementwise_fma(wb[pr][k], xw[2 * (W - 1) - k][pr], ab);       // x[t+(W-1)-k]
                }
                const f32x2 of = silu2(af), ob = silu2(ab);
                pf[pr] = pack2(of.x, of.y);
                pb[pr] = pack2(ob.x, ob.y);
            }
            const int tl = TH * half + i;
            *reinterpret_cast<u32x2 *>(ut[0] + tl * XS + c0) = pf;
            *reinterpret_cast<u32x2 *>(ut[1] + tl * XS + c0) = pb;
            __builtin_amdgcn_raw_buffer_store_b64(pf, fr, c0 * 2, (ts + i) * yf_ts, 0);     // steps >= T: out of range, dropped
            __builtin_amdgcn_raw_buffer_store_b64(pb, br, c0 * 2, (ts + i) * yb_ts, 0);
#pragma unroll
            for (int r = 0; r < 2 * (W - 1); ++r) { xw[r][0] = xw[r + 1][0]; xw[r][1] = xw[r + 1][1]; }   // slide (register renaming)
        }
    }
    if (!wq_requested) {                                          // lanes without a channel group (dim < 512) skipped the loop
#pragma unroll
        for (int s = 0; s < PF; ++s)
            if (ks_lo + s < ks_hi) wload(ks_lo + s, wq[s]);
    }
    if (st_wg) {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");      // phase 1 issued (stores still in flight)
        ts_[2] = t;
    }
    cm_lds_barrier();                                             // the u stores to HBM stay in flight under phase 2
    if (st_wg) {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        ts_[3] = t;
        if (wave == 3 && lane == 0) { for (int i = 0; i < 4; ++i) g_cx_stamps[8 + i] = ts_[i]; }
    }

    // ---- phase 2: x_dbl[token][dir*48 + f] = sum_c W_dir[f][c] u_dir[token][c].  All four waves: wave = (direction, K half);
    // the halves meet through LDS (the token tiles are dead by then).  In-kernel stamps of the version with one wave per
    // direction: phase 1 9.8 k ticks, phase 2 10.7 k (96 MFMAs = 1.5 k of them: the rest was the wave waiting on its own
    // 48 KB weight stream from L2) -- so the weights of the 32-step build are requested BEFORE phase 1 (96 VGPRs held across
    // it, 232 in all) and phase 2 is MFMAs and fragment reads only.
    // an OFFSET into the shared array, not ut[dir]: indexing the pointer array with a run-time value loses the LDS address space, the
    // fragment reads became flat_load + s_waitcnt vmcnt(0) lgkmcnt(0) and every k-step drained the weight ring it was meant to overlap
    const uint16_t *frag = reinterpret_cast<const uint16_t *>(smem) + dir * (TT * XS) + l15 * XS + lq * 8;
    f32x4 acc[NTL][NB];
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
        for (int mt = 0; mt < NB; ++mt) acc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!EARLY) {
#pragma unroll
        for (int s = 0; s < PF; ++s) wload(ks_lo + s, wq[s]);     // past-the-end fragments read as zeros (buffer bounds)
    }
    for (int ks0 = ks_lo; ks0 < ks_hi; ks0 += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            const int ks = ks0 + s;
            if (ks < ks_hi) {
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) {
                    const bf16x8 tok = *reinterpret_cast<const bf16x8 *>(frag + nt * 16 * XS + ks * 32);
#pragma unroll
                    for (int mt = 0; mt < NB; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[s][mt], tok, acc[nt][mt], 0, 0, 0);
                }
                if (ks + PF < ks_hi) wload(ks + PF, wq[s]);
            }
        }
    }
    if (st_wg) {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");      // MFMA loop issued
        ts_[4] = t;
    }
    if (ksplit) {
        cm_lds_barrier();                                         // every wave is done with the token tiles
        float *xch = reinterpret_cast<float *>(smem) + (dir * 64 + lane) * (NTL * NB * 4);
        if (kh == 1) {
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
                for (int mt = 0; mt < NB; ++mt) *reinterpret_cast<f32x4 *>(xch + (nt * NB + mt) * 4) = acc[nt][mt];
        }
        cm_lds_barrier();
        if (kh == 1) return;
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
            for (int mt = 0; mt < NB; ++mt) acc[nt][mt] += *reinterpret_cast<const f32x4 *>(xch + (nt * NB + mt) * 4);
    } else if (kh == 1) {
        return;
    }
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) {
        const int t = t0 + 16 * nt + l15;
        if (t < T) {
            uint16_t *xo = reinterpret_cast<uint16_t *>(p.xdbl) + (int64_t)b * p.xdbl_bs + (int64_t)t * p.xdbl_ts + dir * NP + lq * 4;
#pragma unroll
            for (int mt = 0; mt < NB; ++mt)
                *reinterpret_cast<uint2 *>(xo + mt * 16) = uint2{pack2(acc[nt][mt][0], acc[nt][mt][1]), pack2(acc[nt][mt][2], acc[nt][mt][3])};
        }
    }
    if (st_wg && wave == 0) {
        ts_[5] = cx_now();                                        // everything of this wave has landed
        if (lane == 0) { for (int i = 0; i < 6; ++i) g_cx_stamps[i] = ts_[i]; }
    }
}

template <int TT, int NB>
int launch_cx(const cm_conv_xproj_args &a) {
    const size_t smem = (size_t)2 * TT * (a.dim + 16) * sizeof(uint16_t);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_xproj_kernel<TT, NB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) {
            cm_set_error("conv_xproj: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return (int)e;
        }
        attr_done = true;
    }
    const int ntile = (a.seqlen + TT - 1) / TT;
    CM_REQUIRE((long)ntile * a.batch < (1L << 31), CM_EINVAL, "conv_xproj: grid too large");
    hipLaunchKernelGGL((conv_xproj_kernel<TT, NB>), dim3((unsigned)(ntile * a.batch)), dim3(256), smem, reinterpret_cast<hipStream_t>(a.stream), a, ntile, cm_debug_get() == 18 ? 1 : (cm_debug_get() == 19 ? 2 : 0));
    return cm_launch_status("cm_conv_xproj");
}

}  // namespace

#ifdef CM_ABLATE
extern "C" int cm_debug_read_stamps_cx(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cx_stamps), sizeof(unsigned long long) * 16);
}
#endif

extern "C" int cm_conv_xproj(const cm_conv_xproj_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "conv_xproj: args is NULL");
    const cm_conv_xproj_args &a = *args;
    CM_REQUIRE(a.batch > 0 && a.seqlen > 0 && a.dim > 0 && a.x && a.weight_f && a.weight_b && a.wx_f && a.wx_b && a.y_fwd &&
                   a.y_bwd && a.xdbl, CM_EINVAL, "conv_xproj: bad sizes or NULL tensor");
    CM_REQUIRE(a.width == W, CM_EUNSUPPORTED, "conv_xproj: conv width %d unsupported (4 only)", a.width);
    CM_REQUIRE(a.dim % 32 == 0 && a.dim <= 2048, CM_EUNSUPPORTED, "conv_xproj: dim %d must be a multiple of 32, at most 2048", a.dim);
    CM_REQUIRE(cm_aligned(a.x, 8) && cm_aligned(a.y_fwd, 8) && cm_aligned(a.y_bwd, 8) && cm_aligned(a.xdbl, 8) &&
                   cm_aligned(a.weight_f, 16) && cm_aligned(a.weight_b, 16) && cm_aligned(a.wx_f, 16) && cm_aligned(a.wx_b, 16) &&
                   a.x_bs % 4 == 0 && a.x_ts % 4 == 0 && a.yf_bs % 4 == 0 && a.yf_ts % 4 == 0 && a.yb_bs % 4 == 0 &&
                   a.yb_ts % 4 == 0 && a.xdbl_bs % 4 == 0 && a.xdbl_ts % 4 == 0,
               CM_EALIGN, "conv_xproj: tensors must be 8-byte aligned (weights 16) with strides that are multiples of 4 elements");
    // 32-step tiles when the LDS tiles fit twice per CU and the sequence is long enough to fill the chip with them
    const bool wide = a.variant != 1 && (size_t)2 * 32 * (a.dim + 16) * 2 <= 72 * 1024 && (long)a.batch * ((a.seqlen + 31) / 32) >= 512;
    CM_REQUIRE(a.dt_pad == 0 || a.dt_pad == 16 || a.dt_pad == 32, CM_EUNSUPPORTED, "conv_xproj: dt_pad %d (16 or 32)", a.dt_pad);
    if (a.dt_pad == 32) {
        CM_REQUIRE((size_t)2 * 16 * (a.dim + 16) * 2 <= 80 * 1024, CM_EUNSUPPORTED, "conv_xproj: dim %d too wide for 64-column rows", a.dim);
        return wide ? launch_cx<32, 4>(a) : launch_cx<16, 4>(a);
    }
    return wide ? launch_cx<32, 3>(a) : launch_cx<16, 3>(a);
}
