// scan_cl_fwd.hip — channels-last selective scan forward for gfx950 (contract: cm_scan_cl_fwd).
//
// Why this shape (measured on MI355X, profiles/r01/ubench_valu.log): the scan is VALU-bound, not
// HBM-bound (exp ~3.3 issue slots, DPP adds ~2.7), and a single wave per SIMD issues at < half rate.
// So: no cross-lane arithmetic in the hot loop, as many waves as the problem allows, and everything
// that is uniform across channels kept out of the vector pipe.
//   * lane = channel (64 consecutive channels = one 128-byte row segment per load);
//   * wave w of a workgroup owns states [w*NS, (w+1)*NS) of those 64 channels, so B_t[n], C_t[n] are
//     wave-uniform: staged once per block into LDS and fetched with broadcast reads (a first version
//     used scalar loads; their L2 round trip per block could not be hidden at 1-2 waves per SIMD);
//   * per-(channel,t) work (softplus(delta+bias), delta*u, D skip, SiLU gate) is done once: wave j
//     "owns" time slot j of every TB-step block, publishes (delta', delta'*u) through LDS, and later
//     sums the W partial outputs of that slot from LDS and writes the gated result;
//   * one workgroup barrier per TB steps; three blocks in flight (produce k+1, compute k, finalize k-1);
//   * both BiMamba directions are grid.z of one launch (independent parameter sets, shared z).
#include "cm_common.h"
#include <atomic>

namespace {

template <typename IO> __device__ __forceinline__ float ld_io(const IO *p) { return cm_elem<IO>::load(p); }

// Wave-uniform read-only data through the scalar cache: loads from the constant address space with a
// uniform address select s_load_dword* (SGPR destination), keeping B_t / C_t out of the vector registers.
typedef const float __attribute__((address_space(4))) *cm_kfloat;
__device__ __forceinline__ cm_kfloat as_k(const float *p) { return (cm_kfloat)(uintptr_t)p; }

// LDS layout per workgroup (floats):
//   pw   [2][TB][64] float2   (delta', delta'*u) per (time slot, channel), double-buffered by block parity
//   ybuf [2][W][TB][64]       per-wave partial outputs sum_{n in wave} C_n h_n
//   raw  [2][3][TB][64] IO    u / delta / z tiles exactly as they sit in HBM (row = time step)
template <typename IO, int NS, int TB> struct scan_cl_lds {
    static constexpr int W = 16 / NS;
    static constexpr int kPw = 2 * TB * 64 * 2, kY = 2 * W * TB * 64;
    static constexpr int kRawBytes = 2 * 3 * TB * 64 * (int)sizeof(IO);
    static constexpr int kDtFloats = 2 * TB * 16;                 // [2][TB][16] low-rank time-step features
    static constexpr int kBytes = (kPw + kY) * 4 + kRawBytes + kDtFloats * 4;
};

// operands of one block's recurrence steps
template <int NS, int TB> struct scan_cl_operands {
    float2 dw[TB];          // per lane: (delta', delta'*u) of each time slot          (VGPR)
    float Bs[NS][TB];       // wave-uniform                                            (SGPR)
    float Cs[NS][TB];
};

constexpr int kLoaderDepth = 3;     // input tiles (of TB steps) the loader wave keeps in flight

// ---------------------------------------------------------------------------------------------------
// loader wave: streams the (TB x 64 channel) tiles of u, delta, z from HBM into the LDS `raw` ring with
// 16-byte-per-lane loads, kLoaderDepth tiles ahead (memory-level parallelism the compute waves' own
// just-in-time loads could not provide: ~3 KB in flight per CU measured 2x slower).  It executes exactly
// the barriers the compute waves execute.
// ---------------------------------------------------------------------------------------------------
template <typename IO, int TB, bool REV>
__device__ __forceinline__ void scan_cl_loader(const cm_scan_cl_args &p, const cm_scan_cl_dir &d, unsigned char *raw,
                                               float *dtl, const bool vec_ok) {
    constexpr int VEC = cm_elem<IO>::kVec;                       // elements per 16-byte vector
    constexpr int CPR = 64 / VEC;                                // 16-byte chunks per 64-channel row
    constexpr int NV = TB * CPR / 64;                            // vectors per lane per tensor per tile (1 or 2)
    constexpr int TILE = TB * 64 * (int)sizeof(IO);              // bytes of one tensor's tile
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.y;
    const int c0 = blockIdx.x * 64;
    const int T = p.seqlen, E = p.dim;
    const bool has_z = p.z != nullptr;
    const int nblk = (T + TB - 1) / TB;
    const IO *ug = reinterpret_cast<const IO *>(d.u) + (int64_t)b * d.u_bs;
    const IO *dg = reinterpret_cast<const IO *>(d.delta) + (int64_t)b * d.delta_bs;
    const IO *zg = has_z ? reinterpret_cast<const IO *>(p.z) + (int64_t)b * p.z_bs : nullptr;

    const bool has_dt = d.dt_low != nullptr;
    constexpr int NDT = TB * 16 / 64;                            // dt_low values per lane per tile
    const float *dtg = has_dt ? d.dt_low + (int64_t)b * d.bc_bs : nullptr;
    const float *Bg = d.B + (int64_t)b * d.bc_bs;
    const float *Cg = d.C + (int64_t)b * d.bc_bs;
    struct Tile { uint4 v[3][NV]; float dt[NDT]; float touch; };
    auto load_row_vec = [&](const IO *base, int64_t ts, int t, int ch) -> uint4 {
        uint4 r = {0u, 0u, 0u, 0u};
        if (t >= T || ch >= E) return r;
        const IO *src = base + (int64_t)t * ts + ch;
        if (vec_ok && ch + VEC <= E) return *reinterpret_cast<const uint4 *>(src);
        uint32_t *w = reinterpret_cast<uint32_t *>(&r);
        const unsigned char *sb = reinterpret_cast<const unsigned char *>(src);
        const int nbytes = (E - ch < VEC ? E - ch : VEC) * (int)sizeof(IO);
#pragma unroll 1
        for (int j = 0; j < nbytes; ++j) w[j >> 2] |= (uint32_t)sb[j] << ((j & 3) * 8);      // rare path: byte loads
        return r;
    };
    auto issue = [&](int k, Tile &tile) {                        // global loads of block k's tiles
        if (k >= nblk) return;
        const int tb = (REV ? nblk - 1 - k : k) * TB;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + 64 * i;
            const int t = tb + v / CPR, ch = c0 + (v % CPR) * VEC;
            tile.v[0][i] = load_row_vec(ug, d.u_ts, t, ch);
            if (!has_dt) tile.v[1][i] = load_row_vec(dg, d.delta_ts, t, ch);
            if (has_z) tile.v[2][i] = load_row_vec(zg, p.z_ts, t, ch);
        }
        if (has_dt) {
#pragma unroll
            for (int i = 0; i < NDT; ++i) {
                const int g = lane + 64 * i;                     // g = r * TB + j
                const int r = g / TB, j = g % TB;
                tile.dt[i] = (r < d.dt_rank && tb + j < T) ? dtg[(int64_t)r * d.bc_ns + tb + j] : 0.f;
            }
        }
        // L2 warm-up of this block's B_t / C_t (32 rows x TB floats), three blocks before the compute waves' scalar
        // loads ask for them: measured 880 -> 340 cycles of exposed scalar-load latency per block
        {
            const int which = lane >> 5, n = (lane >> 1) & 15, j = (lane & 1) * (TB / 2);
            tile.touch = (tb + j < T) ? (which ? Cg : Bg)[(int64_t)n * d.bc_ns + tb + j] : 0.f;
        }
    };
    auto commit = [&](int k, const Tile &tile) {                 // registers -> raw[k & 1]
        if (k >= nblk) return;
        unsigned char *dst = raw + (k & 1) * 3 * TILE;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = lane + 64 * i;
            *reinterpret_cast<uint4 *>(dst + 0 * TILE + v * 16) = tile.v[0][i];
            if (!has_dt) *reinterpret_cast<uint4 *>(dst + 1 * TILE + v * 16) = tile.v[1][i];
            if (has_z) *reinterpret_cast<uint4 *>(dst + 2 * TILE + v * 16) = tile.v[2][i];
        }
        if (has_dt) {
#pragma unroll
            for (int i = 0; i < NDT; ++i) {
                const int g = lane + 64 * i;
                const int r = g / TB, j = g % TB;
                dtl[((k & 1) * TB + j) * 16 + r] = tile.dt[i];    // [slot][rank]: one slot's features contiguous
            }
        }
        asm volatile("" ::"v"(tile.touch));                       // keep the warm-up load alive
    };

    Tile ring[kLoaderDepth];
    // ring[j % depth] holds block j.  Prologue: blocks 0..depth-1 in flight; commit 0, 1, 2 (each before a barrier).
#pragma unroll
    for (int j = 0; j < kLoaderDepth; ++j) issue(j, ring[j]);
    commit(0, ring[0]); issue(kLoaderDepth + 0, ring[0]);
    __syncthreads();
    commit(1, ring[1]); issue(kLoaderDepth + 1, ring[1]);
    __syncthreads();
    commit(2, ring[2]); issue(kLoaderDepth + 2, ring[2]);
    __syncthreads();
    // iteration i (matching the compute waves' iteration i): commit block i+3, refill its ring entry
    for (int i0 = 0; i0 <= nblk; i0 += kLoaderDepth) {
#pragma unroll
        for (int r = 0; r < kLoaderDepth; ++r) {
            const int i = i0 + r;
            if (i <= nblk) {
                Tile &tile = ring[(r + 3) % kLoaderDepth];
                commit(i + 3, tile);
                issue(i + 3 + kLoaderDepth, tile);
                __syncthreads();
            }
        }
    }
}

__device__ unsigned long long g_stamps[8];
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// ABL (timing-only ablations, selected by cm_debug_set): 1 = no barrier, 2 = exp replaced by a multiply,
// 4 = no LDS exchange of partial outputs, 5 = per-phase cycle stamps of workgroup 0 / wave 0 into g_stamps
template <typename IO, int NS, int TB, bool REV, int ABL>
__device__ __forceinline__ void scan_cl_compute(const cm_scan_cl_args &p, const cm_scan_cl_dir &d, float *lds,
                                                const unsigned char *raw, const float *dtl) {
    constexpr int W = 16 / NS;                          // compute waves per workgroup (dstate == 16)
    constexpr int OWN = TB >= W ? TB / W : 1;           // time slots of a block owned by one wave
    constexpr int TILE = TB * 64 * (int)sizeof(IO);
    using Operands = scan_cl_operands<NS, TB>;
    float2 *pw = reinterpret_cast<float2 *>(lds);
    float *ybuf = lds + scan_cl_lds<IO, NS, TB>::kPw;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int c = blockIdx.x * 64 + lane;
    const int T = p.seqlen, E = p.dim;
    const bool c_ok = c < E;
    const int cc = c_ok ? c : E - 1;
    const bool has_z = p.z != nullptr;
    const bool softplus = p.delta_softplus != 0;
    const bool owner = wave * OWN < TB;                 // this wave produces / finalises slots wave*OWN .. +OWN

    IO *og = reinterpret_cast<IO *>(d.out) + (int64_t)b * d.out_bs + cc;
    const cm_kfloat Bg = as_k(d.B + (int64_t)b * d.bc_bs + (int64_t)(wave * NS) * d.bc_ns);
    const cm_kfloat Cg = as_k(d.C + (int64_t)b * d.bc_bs + (int64_t)(wave * NS) * d.bc_ns);

    float Ap[NS], h[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        Ap[i] = d.A[(int64_t)cc * 16 + wave * NS + i] * CM_LOG2E;
        h[i] = 0.f;
    }
    const float bias = d.delta_bias ? d.delta_bias[cc] : 0.f;
    const float Dv = d.D ? d.D[cc] : 0.f;
    const int nblk = (T + TB - 1) / TB;
    const bool has_dt = d.dt_low != nullptr;
    float Wdt[16];                                       // this channel's dt_proj row (zero padded to 16)
#pragma unroll
    for (int r = 0; r < 16; ++r) Wdt[r] = (has_dt && owner && r < d.dt_rank) ? d.dt_weight[(int64_t)cc * d.dt_rank + r] : 0.f;

    // owner state: (u, z) of the blocks whose gate is still to be applied, oldest first
    float uq[3][OWN], zq[3][OWN];
#pragma unroll
    for (int o = 0; o < OWN; ++o) uq[0][o] = uq[1][o] = uq[2][o] = zq[0][o] = zq[1][o] = zq[2][o] = 0.f;

    auto chunk_base = [&](int k) { return (REV ? nblk - 1 - k : k) * TB; };   // time index of slot 0 of block k
    auto produce = [&](int k) {                         // publish block k's (delta', delta'*u); queue (u, z)
        if (!owner) return;
#pragma unroll
        for (int o = 0; o < OWN; ++o) { uq[0][o] = uq[1][o]; zq[0][o] = zq[1][o]; uq[1][o] = uq[2][o]; zq[1][o] = zq[2][o]; }
        if (k >= nblk) return;
        const int tb = chunk_base(k);
        const unsigned char *rk = raw + (k & 1) * 3 * TILE;
#pragma unroll
        for (int o = 0; o < OWN; ++o) {
            const int slot = wave * OWN + o;
            const int ro = (slot * 64 + lane) * (int)sizeof(IO);
            const float uv = ld_io(reinterpret_cast<const IO *>(rk + 0 * TILE + ro));
            const float zv = has_z ? ld_io(reinterpret_cast<const IO *>(rk + 2 * TILE + ro)) : 0.f;
            float dv;
            if (has_dt) {                                // delta = dt_proj.weight[c, :] . dt_low[t, :]   (uniform LDS reads)
                const float4 *f4 = reinterpret_cast<const float4 *>(dtl + ((k & 1) * TB + slot) * 16);
                float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4 f = f4[q4];
                    acc0 = fmaf(Wdt[4 * q4 + 0], f.x, acc0);
                    acc1 = fmaf(Wdt[4 * q4 + 1], f.y, acc1);
                    acc0 = fmaf(Wdt[4 * q4 + 2], f.z, acc0);
                    acc1 = fmaf(Wdt[4 * q4 + 3], f.w, acc1);
                }
                dv = acc0 + acc1;
            } else {
                dv = ld_io(reinterpret_cast<const IO *>(rk + 1 * TILE + ro));
            }
            float dt = dv + bias;
            if (softplus) dt = cm_softplus(dt);
            dt = tb + slot < T ? dt : 0.f;
            pw[((k & 1) * TB + slot) * 64 + lane] = make_float2(dt, dt * uv);
            uq[2][o] = uv;
            zq[2][o] = zv;
        }
    };
    auto load_operands = [&](int k, Operands &op) {     // LDS + scalar loads of block k (issued one block ahead)
        if (k >= nblk) return;
        const int tb = chunk_base(k);
        const float2 *pwk = pw + ((k & 1) * TB) * 64 + lane;
#pragma unroll
        for (int j = 0; j < TB; ++j) op.dw[j] = pwk[j * 64];
        // B/C rows are readable (and finite) up to the next multiple of 16 steps: include/conmamba_hip.h
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                op.Bs[i][j] = Bg[(int64_t)i * d.bc_ns + tb + j];
                op.Cs[i][j] = Cg[(int64_t)i * d.bc_ns + tb + j];
            }
    };

    // one pipeline iteration: loads for block k+1, recurrence of block k, gate of block k-1, publish block k+2
    unsigned long long acc_t[5] = {0, 0, 0, 0, 0};
    const bool stamping = ABL == 5 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && wave == 0;
    auto iteration = [&](int k) {
        Operands cur;
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
        if (ABL == 5) { __builtin_amdgcn_sched_barrier(0); t0 = stamp(); __builtin_amdgcn_sched_barrier(0); }
        float ypart[OWN][W];
        if (owner && k >= 1) {                          // partial sums of block k-1 (written before the last barrier)
#pragma unroll
            for (int o = 0; o < OWN; ++o) {
                const float *yb = ybuf + (((k - 1) & 1) * W * TB + wave * OWN + o) * 64 + lane;
#pragma unroll
                for (int w = 0; w < W; ++w) ypart[o][w] = ABL == 4 ? (float)w : yb[w * TB * 64];
            }
        }
        load_operands(k, cur);
        if (ABL == 5) { __builtin_amdgcn_sched_barrier(0); t1 = stamp(); __builtin_amdgcn_sched_barrier(0); }
        if (k < nblk) {
            float yp[TB];
#pragma unroll
            for (int sp = 0; sp < TB; ++sp) {
                const int j = REV ? TB - 1 - sp : sp;   // time slot processed at step sp
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const float a = ABL == 2 ? cur.dw[j].x * Ap[i] * 0.5f : cm_exp2(cur.dw[j].x * Ap[i]);
                    h[i] = fmaf(a, h[i], cur.dw[j].y * cur.Bs[i][j]);
                    acc = fmaf(cur.Cs[i][j], h[i], acc);
                }
                yp[j] = acc;
            }
            float *yb = ybuf + ((k & 1) * W * TB + wave * TB) * 64 + lane;
            if (ABL == 4) { float sacc = 0.f; for (int j = 0; j < TB; ++j) sacc += yp[j]; if (sacc == 123.f) yb[0] = sacc; }
            else {
#pragma unroll
                for (int j = 0; j < TB; ++j) yb[j * 64] = yp[j];
            }
        }
        if (ABL == 5) { __builtin_amdgcn_sched_barrier(0); t2 = stamp(); __builtin_amdgcn_sched_barrier(0); }
        if (owner && k >= 1) {                          // F(k-1): skip term, gate, store
            const int tb = chunk_base(k - 1);
#pragma unroll
            for (int o = 0; o < OWN; ++o) {
                const int slot = wave * OWN + o;
                float y = 0.f;
#pragma unroll
                for (int w = 0; w < W; ++w) y += ypart[o][w];
                float ov = fmaf(Dv, uq[0][o], y);
                if (has_z) ov *= zq[0][o] * cm_sigmoid(zq[0][o]);
                if (c_ok && tb + slot < T) cm_elem<IO>::store(og + (int64_t)(tb + slot) * d.out_ts, ov);
            }
        }
        produce(k + 2);                                 // P(k+2) (also advances the (u, z) queue)
        if (ABL == 5) { __builtin_amdgcn_sched_barrier(0); t3 = stamp(); __builtin_amdgcn_sched_barrier(0); }
        if (ABL != 1) __syncthreads();
        if (ABL == 5) {
            __builtin_amdgcn_sched_barrier(0); t4 = stamp(); __builtin_amdgcn_sched_barrier(0);
            acc_t[0] += t1 - t0; acc_t[1] += t2 - t1; acc_t[2] += t3 - t2; acc_t[3] += t4 - t3; acc_t[4] += 1;
        }
    };

    // prologue (three barriers, mirrored by the loader): raw tiles 0, 1, 2 arrive one barrier apart
    __syncthreads();
    produce(0);
    __syncthreads();
    produce(1);
    __syncthreads();
    // (u, z) queue: produce() shifts it, so that at iteration k's gate stage uq[0] holds block k-1
    for (int k = 0; k <= nblk; ++k) iteration(k);
    if (ABL == 5 && stamping && lane == 0) {
#pragma unroll
        for (int i = 0; i < 5; ++i) g_stamps[i] = acc_t[i];
    }
}

template <typename IO, int NS, int TB, int ABL>
__global__ __launch_bounds__(64 * (16 / NS + 1)) void scan_cl_fwd_kernel(const cm_scan_cl_args p, int vec_ok) {
    constexpr int W = 16 / NS;
    __shared__ __attribute__((aligned(16))) unsigned char lds[scan_cl_lds<IO, NS, TB>::kBytes];
    float *fl = reinterpret_cast<float *>(lds);
    unsigned char *raw = lds + (scan_cl_lds<IO, NS, TB>::kPw + scan_cl_lds<IO, NS, TB>::kY) * 4;
    float *dtl = reinterpret_cast<float *>(raw + scan_cl_lds<IO, NS, TB>::kRawBytes);
    const cm_scan_cl_dir &d = p.dir[blockIdx.z];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == W) {                                    // last wave = loader
        if (d.reverse_time) scan_cl_loader<IO, TB, true>(p, d, raw, dtl, vec_ok != 0);
        else scan_cl_loader<IO, TB, false>(p, d, raw, dtl, vec_ok != 0);
    } else {
        if (d.reverse_time) scan_cl_compute<IO, NS, TB, true, ABL>(p, d, fl, raw, dtl);
        else scan_cl_compute<IO, NS, TB, false, ABL>(p, d, fl, raw, dtl);
    }
}

#ifdef CM_ABLATE
std::atomic<int> g_debug{0};
#endif

template <typename IO, int NS, int TB>
int launch(const cm_scan_cl_args &a) {
    dim3 grid((a.dim + 63) / 64, a.batch, a.ndir);
    dim3 block(64 * (16 / NS + 1));
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    // tile rows can be fetched as 16-byte vectors when every row start keeps 16-byte alignment
    constexpr int VEC = cm_elem<IO>::kVec;
    int vec_ok = a.dim % VEC == 0 && (!a.z || (cm_aligned(a.z, 16) && a.z_bs % VEC == 0 && a.z_ts % VEC == 0));
    for (int i = 0; i < a.ndir; ++i) {
        const cm_scan_cl_dir &d = a.dir[i];
        vec_ok = vec_ok && cm_aligned(d.u, 16) && d.u_bs % VEC == 0 && d.u_ts % VEC == 0;
        if (!d.dt_low) vec_ok = vec_ok && cm_aligned(d.delta, 16) && d.delta_bs % VEC == 0 && d.delta_ts % VEC == 0;
    }
#ifdef CM_ABLATE
    if constexpr (sizeof(IO) == 2 && NS == 2) {       // ablation builds exist for the bf16 NS=2 kernel only
        switch (g_debug.load()) {
            case 1: hipLaunchKernelGGL((scan_cl_fwd_kernel<IO, NS, TB, 1>), grid, block, 0, st, a, vec_ok); return cm_launch_status("abl1");
            case 2: hipLaunchKernelGGL((scan_cl_fwd_kernel<IO, NS, TB, 2>), grid, block, 0, st, a, vec_ok); return cm_launch_status("abl2");
            case 4: hipLaunchKernelGGL((scan_cl_fwd_kernel<IO, NS, TB, 4>), grid, block, 0, st, a, vec_ok); return cm_launch_status("abl4");
            case 5: hipLaunchKernelGGL((scan_cl_fwd_kernel<IO, NS, TB, 5>), grid, block, 0, st, a, vec_ok); return cm_launch_status("abl5");
            default: break;
        }
    }
#endif
    hipLaunchKernelGGL((scan_cl_fwd_kernel<IO, NS, TB, 0>), grid, block, 0, st, a, vec_ok);
    return cm_launch_status("cm_scan_cl_fwd");
}

template <typename IO>
int by_split(const cm_scan_cl_args &a, int ns) {
    switch (ns) {
        case 4: return launch<IO, 4, 8>(a);
        default: return launch<IO, 2, 8>(a);
    }
}

}  // namespace

int cm_scan_rows_fwd(const cm_scan_cl_args &a);      // scan_rows_fwd.hip

#ifdef CM_ABLATE
extern "C" int cm_debug_set(int v) { return g_debug.exchange(v); }
extern "C" int cm_debug_get() { return g_debug.load(); }
extern "C" int cm_debug_read_stamps(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 8);
}
#endif

extern "C" int cm_scan_cl_fwd(const cm_scan_cl_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "scan_cl_fwd: args is NULL");
    const cm_scan_cl_args &a = *args;
    CM_REQUIRE(a.batch > 0 && a.dim > 0 && a.seqlen > 0, CM_EINVAL, "scan_cl_fwd: bad sizes batch=%d dim=%d seqlen=%d",
               a.batch, a.dim, a.seqlen);
    CM_REQUIRE(a.dstate == 16, CM_EUNSUPPORTED, "scan_cl_fwd: dstate %d unsupported (16 only)", a.dstate);
    CM_REQUIRE(a.ndir == 1 || a.ndir == 2, CM_EINVAL, "scan_cl_fwd: ndir must be 1 or 2");
    {
        int with_rows = 0;
        for (int i = 0; i < a.ndir; ++i) with_rows += a.dir[i].xdbl != nullptr;
        CM_REQUIRE(with_rows == 0 || with_rows == a.ndir, CM_EINVAL, "scan_cl_fwd: xdbl must be set for every direction or none");
        if (with_rows) return cm_scan_rows_fwd(a);
    }
    CM_REQUIRE(a.time_chunks <= 1, CM_EUNSUPPORTED, "scan_cl_fwd: time_chunks is built for the xdbl mode only");
    CM_REQUIRE(a.batch <= 65535, CM_EINVAL, "scan_cl_fwd: batch %d exceeds the grid limit", a.batch);
    for (int i = 0; i < a.ndir; ++i) {
        const cm_scan_cl_dir &d = a.dir[i];
        CM_REQUIRE(d.u && (d.delta || d.dt_low) && d.A && d.B && d.C && d.out, CM_EINVAL, "scan_cl_fwd: dir %d has a NULL tensor", i);
        CM_REQUIRE(!d.dt_low || (d.dt_weight && d.dt_rank >= 1 && d.dt_rank <= 16), CM_EUNSUPPORTED,
                   "scan_cl_fwd: in-kernel dt_proj needs dt_weight and 1 <= dt_rank <= 16 (got %d)", d.dt_rank);
    }
    // states per lane: fewest waves that still give >= 2 waves per SIMD (2048 waves), else the finest split
    const long wg = (long)((a.dim + 63) / 64) * a.batch * a.ndir;
    const int lanes_per_channel = a.lanes_per_channel;     // tuning field: 4, 8, 16; 0 = automatic
    int ns;
    if (lanes_per_channel == 4 || lanes_per_channel == 8 || lanes_per_channel == 16) ns = 16 / lanes_per_channel;
    else ns = wg * 4 >= 2048 ? 4 : 2;
    switch (a.io_dtype) {
        case CM_BF16: return by_split<cm_bf16>(a, ns);
        case CM_F32: return by_split<float>(a, ns);
        default:
            cm_set_error("scan_cl_fwd: unsupported io dtype %d", a.io_dtype);
            return CM_EUNSUPPORTED;
    }
}
