// wgrad_bf16.hip — weight gradients of the Linear layers of a training step: C (M, N) fp32 = sum over rows k of A[k, m] B[k, n] with
// A (rows, M), B (rows, N) bf16 row-major -- the "dW = dY^T X" products (contract: cm_wgrad_bf16 in include/conmamba_hip.h; the
// reference leaves them to autograd's addmm backward: one GEMM with K = batch x time per Linear, reference modules/Conmamba.py:597-650,
// modules/mamba/selective_scan_interface.py:262-284).
//
// Shape: the contraction index is the ROW index of both operands (K-major), the output is small (256 x 256 ... 1024 x 1024) and
// K = 32 k-64 k rows.  The vendor library has no good kernel for it (one GEMM: 160 us at 1024 x 256 x 32000; as 32 per-utterance
// batched GEMMs + a fold: 35 + 5 us = 480 TFLOP/s).  Here:
//   * split-K: a workgroup owns a 128 x 128 output tile and one chunk of rows; partial tiles go to a workspace
//     (ksplit, M, N) fp32 and cm_sum_leading folds them in a fixed order (deterministic, as the batched form was);
//   * both operands travel as they lie in memory -- 64 rows x 128 columns per step, whole 256-byte row segments -- into an LDS image
//     with the XOR swizzle of cdna_hip_programming.md T10 (b), and reach the MFMA operand layout (8 consecutive k per lane) through
//     ds_read_b64_tr_b16 (two per operand fragment): no transposing pass over HBM;
//   * 4 waves = 2 x 2 quadrants of 64 x 64 (4 x 4 v_mfma_f32_16x16x32_bf16 tiles, 64 accumulators); the next step's rows are
//     requested before the current step's MFMAs and land in the other LDS buffer behind one barrier per 64 rows.
#include "cm_common.h"

extern "C" int cm_sum_leading(const void *in, void *out, int32_t nbatch, int64_t n, int32_t in_dtype, int32_t out_dtype, void *stream);

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int TM = 128, TN = 128, KB = 64;        // output tile, rows per step
constexpr int TILE_BYTES = KB * 256;               // one operand's (64 rows x 128 columns) bf16 image

struct wgrad_plan { int ksplit, chunk; };           // chunk: rows per workgroup, a multiple of KB

// byte offset of 16-byte chunk ch (0..15) of row `row` in a [rows][128 x bf16] image (T10 (b)): ds_read_b128-free, and the
// transposed reads of two 4-row blocks 8 rows apart are conflict-free
__device__ __forceinline__ int img_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

__device__ __forceinline__ bf16x4 tr_read(const unsigned char *lds_base, int off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
        reinterpret_cast<bf16x4 __attribute__((address_space(3))) *>(reinterpret_cast<uintptr_t>(lds_base + off)));
}

// DMA: the tiles go global -> LDS directly (buffer_load_dwordx4 ... lds: a wave instruction fills four whole rows of the image; the
// XOR of the image is applied to the SOURCE column chunk, the destination is lane-linear) instead of through registers and
// ds_write_b128 (~80 B/clk/CU: 32 KB per step and workgroup = 415 LDS cycles against 512 MFMA cycles per wave).
template <bool DMA>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const cm_wgrad_args p, const wgrad_plan pl) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][TILE_BYTES];      // [buffer][A | B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mw = wave >> 1, nw = wave & 1;
    const int tiles_n = p.n / TN;
    // one row chunk's tiles share its rows of A and B: keep them on ONE XCD (workgroups are dealt to the 8 XCDs round-robin), whose
    // L2 then holds the chunk once -- dealt as launched, every XCD fetched every chunk
    const int ntile = (p.m / TM) * tiles_n, total = gridDim.x;
    int id = blockIdx.x;
    if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
    const int tile = id % ntile, ks = id / ntile;
    const int m0 = (tile / tiles_n) * TM, n0 = (tile % tiles_n) * TN;
    const int k_lo = ks * pl.chunk, k_hi = min(p.rows, k_lo + pl.chunk);
    // rows at or past k_hi fall outside the descriptors: they read as zero
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.a), 0, (int)((int64_t)k_hi * p.lda * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.b), 0, (int)((int64_t)k_hi * p.ldb * 2), 0x00020000);
    // staging: 64 rows x 16 chunks per operand = 1024 chunks, 4 per thread.  Register path: chunk c = tid + 256 i -> row c / 16, column
    // chunk c % 16, written at img_off.  DMA path: piece pi = 4 wave + i = rows 4 pi .. 4 pi + 3; lane -> row 4 pi + lane / 16, IMAGE
    // chunk lane % 16, which holds the logical chunk (lane % 16) ^ swizzle(row) -- the same involution the reads apply.
    int go_a[4], go_b[4], lo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row, ch;
        if constexpr (DMA) {
            const int pi = 4 * wave + i;
            row = 4 * pi + (lane >> 4);
            ch = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
            lo[i] = pi * 1024;                                       // wave-uniform piece base
        } else {
            const int c = tid + 256 * i;
            row = c >> 4, ch = c & 15;
            lo[i] = img_off(row, ch);
        }
        go_a[i] = (int)(((int64_t)(k_lo + row) * p.lda + m0 + ch * 8) * 2);
        go_b[i] = (int)(((int64_t)(k_lo + row) * p.ldb + n0 + ch * 8) * 2);
    }
    const int step_a = (int)(p.lda * KB * 2), step_b = (int)(p.ldb * KB * 2);
    u32x4 va[4], vb[4];
    auto request = [&](int t, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (DMA) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void *)(reinterpret_cast<uintptr_t>(&lds[buf][0][0]) + __builtin_amdgcn_readfirstlane(lo[i])),
                                                         16, go_a[i], t * step_a, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void *)(reinterpret_cast<uintptr_t>(&lds[buf][1][0]) + __builtin_amdgcn_readfirstlane(lo[i])),
                                                         16, go_b[i], t * step_b, 0, 0);
            } else {
                va[i] = __builtin_amdgcn_raw_buffer_load_b128(ra, go_a[i], t * step_a, 0);
                vb[i] = __builtin_amdgcn_raw_buffer_load_b128(rb, go_b[i], t * step_b, 0);
            }
        }
    };
    auto commit = [&](int buf) {
        if constexpr (!DMA) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<u32x4 *>(&lds[buf][0][lo[i]]) = va[i];
                *reinterpret_cast<u32x4 *>(&lds[buf][1][lo[i]]) = vb[i];
            }
        }
    };
    // transposed reads: group g = lane / 16 takes k rows 8 g .. 8 g + 7 of a 32-row k-step (two blocks of 4 rows); lane 4 q + p of the
    // group addresses row q, columns 4 p .. 4 p + 3 of the block's 16 columns (T10)
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    int tr_a[4][2], tr_b[4][2];                                       // [tile of 16 columns][first / second 4 rows]; + 32 rows per k-step
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = 8 * g + 4 * h + q;
            tr_a[t][h] = img_off(row, 2 * (mw * 4 + t) + (pp >> 1)) + 8 * (pp & 1);
            tr_b[t][h] = img_off(row, 2 * (nw * 4 + t) + (pp >> 1)) + 8 * (pp & 1);
        }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nstep = (k_hi - k_lo + KB - 1) / KB;
    if (nstep > 0) {
        request(0, 0);
        commit(0);
        __syncthreads();                                             // (drains the DMA: hipcc waits vmcnt(0) in front of a barrier)
    }
    for (int t = 0; t < nstep; ++t) {
        const int buf = t & 1;
        if (t + 1 < nstep) request(t + 1, buf ^ 1);                  // the other buffer: its last readers passed the previous barrier
#pragma unroll
        for (int kk = 0; kk < KB / 32; ++kk) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // img_off is linear in row blocks of 32 (only row & 15 enters the swizzle): + 32 rows = + 8192 bytes
                const bf16x4 a0 = tr_read(lds[buf][0], tr_a[i][0] + kk * 32 * 256), a1 = tr_read(lds[buf][0], tr_a[i][1] + kk * 32 * 256);
                const bf16x4 b0 = tr_read(lds[buf][1], tr_b[i][0] + kk * 32 * 256), b1 = tr_read(lds[buf][1], tr_b[i][1] + kk * 32 * 256);
                fa[i] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                fb[i] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < nstep) {
            commit(buf ^ 1);                                         // the other buffer: its last readers passed the previous barrier
            __syncthreads();
        }
    }
    // partial tile: lane holds column n = lane % 16 of tile j, rows 4 (lane / 16) + r of tile i
    float *out = p.workspace + (int64_t)ks * p.m * p.n;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(int64_t)(m0 + mw * 64 + i * 16 + 4 * g + r) * p.n + n0 + nw * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
}

// Deeper pipeline for the LDS-DMA form (PMC of the training step: the two-buffer kernel ran at 17 % MFMA busy with 23 % of wave
// cycles waiting -- a step's rows are requested ONE 0.25 us compute step ahead of their use against ~1-2 us of memory latency, and
// two workgroups x one outstanding step = 64 KB in flight per CU).  Here: 32-row steps, a ring of NST = 4 LDS stages (64 KB), three
// steps (48 KB per workgroup) in flight; per step a counted s_waitcnt vmcnt (the wave's own pieces of the oldest step), ONE raw
// s_barrier (everyone's pieces have landed, and everyone is done with the stage the next request overwrites), the request for
// step + 3, then the step's 16 MFMAs per wave.  Requests past the chunk's end address rows outside the descriptors: they write zeros and
// keep the counts uniform.
constexpr int KB2 = 32, NST = 4, STAGE_BYTES = KB2 * 256;           // one operand's (32 rows x 128 columns) image

// One LDS-DMA instruction, issued where it is written and invisible to the compiler's wait-count pass: through the builtin, hipcc
// orders the LDS write against the following ds_reads with s_waitcnt vmcnt(0), i.e. it drains the whole ring every step.  The
// descriptor is built by hand (raw buffer, num_records bytes from base); M0 = the wave-uniform LDS destination, restored behind.
__device__ __forceinline__ u32x4 raw_rsrc(const void *base, uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    return u32x4{(uint32_t)a, (uint32_t)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}
__device__ __forceinline__ void dma16(const u32x4 rsrc, uint32_t lds_addr, int voff, int soff) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}

__global__ __launch_bounds__(256, 2) void wgrad_pipe_kernel(const cm_wgrad_args p, const wgrad_plan pl) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[NST][2][STAGE_BYTES];     // [stage][A | B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mw = wave >> 1, nw = wave & 1;
    const int tiles_n = p.n / TN;
    const int ntile = (p.m / TM) * tiles_n, total = gridDim.x;
    int id = blockIdx.x;
    if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);  // a chunk's tiles on one XCD
    const int tile = id % ntile, ks = id / ntile;
    const int m0 = (tile / tiles_n) * TM, n0 = (tile % tiles_n) * TN;
    const int k_lo = ks * pl.chunk, k_hi = min(p.rows, k_lo + pl.chunk);
    const u32x4 ra = raw_rsrc(p.a, (uint32_t)((int64_t)k_hi * p.lda * 2)), rb = raw_rsrc(p.b, (uint32_t)((int64_t)k_hi * p.ldb * 2));
    const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>(&lds[0][0][0]);      // the LDS offset is the low half of the flat address
    // a step's 8 pieces per operand (4 rows each): this wave fills pieces 2 wave, 2 wave + 1; lane -> row 4 piece + lane / 16, IMAGE chunk
    // lane % 16 = logical chunk ^ swizzle(row)
    int go_a[2], go_b[2], lo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pi = 2 * wave + i, row = 4 * pi + (lane >> 4);
        const int ch = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
        lo[i] = pi * 1024;
        go_a[i] = (int)(((int64_t)(k_lo + row) * p.lda + m0 + ch * 8) * 2);
        go_b[i] = (int)(((int64_t)(k_lo + row) * p.ldb + n0 + ch * 8) * 2);
    }
    const int step_a = (int)(p.lda * KB2 * 2), step_b = (int)(p.ldb * KB2 * 2);
    const int nstep = (k_hi - k_lo + KB2 - 1) / KB2;
    auto request = [&](int t) {                                      // 4 LDS-DMA instructions per wave; steps >= nstep read zeros
        const int st = t % NST;
        const bool in = t < nstep;                                   // (past the end: offsets that are out of range whatever the stride)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t dst = lds0 + (uint32_t)(st * 2 * STAGE_BYTES) + (uint32_t)__builtin_amdgcn_readfirstlane(lo[i]);
            dma16(ra, dst, in ? go_a[i] : 0x7ffffff0, in ? t * step_a : 0);
            dma16(rb, dst + STAGE_BYTES, in ? go_b[i] : 0x7ffffff0, in ? t * step_b : 0);
        }
    };
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    int tr_a[4][2], tr_b[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = 8 * g + 4 * h + q;
            tr_a[t][h] = img_off(row, 2 * (mw * 4 + t) + (pp >> 1)) + 8 * (pp & 1);
            tr_b[t][h] = img_off(row, 2 * (nw * 4 + t) + (pp >> 1)) + 8 * (pp & 1);
        }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NST - 1; ++t) request(t);
    for (int t = 0; t < nstep; ++t) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");             // this wave's 4 pieces of step t (8 of steps t + 1, t + 2 may be in flight)
        __builtin_amdgcn_s_barrier();                                // everyone's pieces of step t are in LDS; everyone has read stage (t - 1) % NST
        request(t + NST - 1);                                        // -> stage (t - 1) % NST
        const int st = t % NST;
        bf16x8 fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16x4 a0 = tr_read(lds[st][0], tr_a[i][0]), a1 = tr_read(lds[st][0], tr_a[i][1]);
            const bf16x4 b0 = tr_read(lds[st][1], tr_b[i][0]), b1 = tr_read(lds[st][1], tr_b[i][1]);
            fa[i] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
            fb[i] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the zero-filling requests past the end have landed before the LDS is released
    float *out = p.workspace + (int64_t)ks * p.m * p.n;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(int64_t)(m0 + mw * 64 + i * 16 + 4 * g + r) * p.n + n0 + nw * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
}

inline wgrad_plan plan_for(int rows, int m, int n) {
    const int tiles = (m / TM) * (n / TN);
    int ks = (512 + tiles - 1) / tiles;
    ks = ks < 4 ? 4 : (ks > 32 ? 32 : ks);
    int chunk = ((rows + ks - 1) / ks + KB - 1) / KB * KB;
    ks = (rows + chunk - 1) / chunk;
    return wgrad_plan{ks, chunk};
}

}  // namespace

extern "C" int cm_wgrad_supported(int32_t rows, int32_t m, int32_t n) {
    return rows > 0 && m > 0 && n > 0 && m % TM == 0 && n % TN == 0 && m <= 4096 && n <= 4096;
}

extern "C" int64_t cm_wgrad_workspace_floats(int32_t rows, int32_t m, int32_t n) {
    if (!cm_wgrad_supported(rows, m, n)) return 0;
    return (int64_t)plan_for(rows, m, n).ksplit * m * n;
}

extern "C" int cm_wgrad_bf16(const cm_wgrad_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "wgrad_bf16: args is NULL");
    const cm_wgrad_args &a = *args;
    CM_REQUIRE(a.a && a.b && a.out && a.workspace, CM_EINVAL, "wgrad_bf16: a / b / out / workspace must be non-NULL");
    CM_REQUIRE(cm_wgrad_supported(a.rows, a.m, a.n), CM_EUNSUPPORTED, "wgrad_bf16: needs m, n multiples of 128 (<= 4096), got rows %d, m %d, n %d", a.rows, a.m, a.n);
    CM_REQUIRE(a.lda >= a.m && a.ldb >= a.n && a.lda % 8 == 0 && a.ldb % 8 == 0 && cm_aligned(a.a, 16) && cm_aligned(a.b, 16) && cm_aligned(a.out, 16) &&
                   cm_aligned(a.workspace, 16),
               CM_EALIGN, "wgrad_bf16: row strides must be multiples of 8 elements covering the columns, tensors 16-byte aligned");
    CM_REQUIRE((int64_t)a.rows * a.lda * 2 < 2147483647LL && (int64_t)a.rows * a.ldb * 2 < 2147483647LL, CM_EUNSUPPORTED, "wgrad_bf16: operand larger than 2 GiB");
    const wgrad_plan pl = plan_for(a.rows, a.m, a.n);
    CM_REQUIRE(a.workspace_floats >= (int64_t)pl.ksplit * a.m * a.n, CM_EINVAL, "wgrad_bf16: workspace smaller than cm_wgrad_workspace_floats()");
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    // measured at 32 k rows, fold included (us; 1024x256 / 256x256 / 512x256 / 1024x1024): 4-stage LDS-DMA ring 30.2 / 19.5 / 22.4 / 76.1,
    // 2-buffer LDS-DMA 31.6 / 23.4 / 26.8 / 86.9, through registers 32.9 / 19.7 / 23.0 / 91.4 -> the ring unless asked otherwise
    const bool regs = a.variant == 1;
    if (regs) hipLaunchKernelGGL(wgrad_kernel<false>, dim3((a.m / TM) * (a.n / TN) * pl.ksplit), dim3(256), 0, st, a, pl);
    else if (a.variant == 3) hipLaunchKernelGGL(wgrad_kernel<true>, dim3((a.m / TM) * (a.n / TN) * pl.ksplit), dim3(256), 0, st, a, pl);
    else hipLaunchKernelGGL(wgrad_pipe_kernel, dim3((a.m / TM) * (a.n / TN) * pl.ksplit), dim3(256), 0, st, a, pl);
    if (int rc = cm_launch_status("cm_wgrad_bf16")) return rc;
    return cm_sum_leading(a.workspace, a.out, pl.ksplit, (int64_t)a.m * a.n, CM_F32, CM_F32, a.stream);
}
