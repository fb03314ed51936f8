// ctc.hip — CTC loss and its gradient in one call (contract: cm_ctc_loss; replaces torch.nn.functional.ctc_loss behind
// speechbrain.nnet.losses.ctc_loss, call site reference train_CTC.py:405 with blank_index 0, reduction 'batchmean', zero_infinity).
// Graves' alpha / beta recursions in log space on the extended label sequence l' (blank, y1, blank, ..., yS, blank):
//   alpha_t(s) = lp_t(l'_s) + logsumexp(alpha_{t-1}(s), alpha_{t-1}(s-1), alpha_{t-1}(s-2) if l'_s != blank and l'_s != l'_{s-2})
//   nll = -logsumexp(alpha_{T-1}(S'-1), alpha_{T-1}(S'-2));  grad_t(v) = exp(lp_t(v)) - sum_{s: l'_s = v} exp(alpha_t(s) + beta_t(s) + nll - lp_t(v))
// (the gradient torch returns for log-softmax inputs, eq. 16 of the paper).  Kernel 1: one workgroup per (utterance, direction):
// alpha and beta run CONCURRENTLY (torch: two launches in sequence, 1.3 + 1.2 ms at 32 x 1000 x 31 with 500 labels), one thread
// per extended label, the previous row in LDS, rows written to a workspace.  Kernel 2: one wave per (utterance, step): posteriors
// summed per class in a FIXED order (torch's collect kernel accumulates with atomics: its gradient is not bit-reproducible).
#include "cm_common.h"

namespace {

constexpr float NEG_INF = -__builtin_huge_valf();

__device__ __forceinline__ float lse3(float a, float b, float c) {
    const float m = fmaxf(a, fmaxf(b, c));
    if (m == NEG_INF) return NEG_INF;
    return m + CM_LN2 * cm_log2(cm_exp2(CM_LOG2E * (a - m)) + cm_exp2(CM_LOG2E * (b - m)) + cm_exp2(CM_LOG2E * (c - m)));
}

// label of extended position s for utterance b (blank at even s)
__device__ __forceinline__ int ext_label(const int64_t *tg, int s, int blank) { return (s & 1) ? (int)tg[s >> 1] : blank; }

// grid (batch, 2): y = 0 alpha, y = 1 beta.  block = 1024 threads (S' <= 1024), two LDS rows.
__global__ __launch_bounds__(1024) void ctc_alpha_beta_kernel(const cm_ctc_args p) {
    __shared__ float row[2][1024 + 2];
    const int b = blockIdx.x, beta = blockIdx.y, s = threadIdx.x;
    const int T = min(p.input_lengths[b], p.T), S = min(p.target_lengths[b], p.S), Sx = 2 * S + 1;
    const int64_t *tg = p.targets + (int64_t)b * p.S;
    const float *lp = p.log_probs + (int64_t)b * p.T * p.V;
    float *tab = (beta ? p.beta : p.alpha) + (int64_t)b * p.T * p.Sx_max;
    if (T <= 0) return;
    const bool live = s < Sx;
    const int lab = live ? ext_label(tg, s, p.blank) : p.blank;
    // the neighbour two positions away is reachable when the label differs from it and is not blank
    const int s2 = beta ? s + 2 : s - 2;
    const bool skip_ok = live && (s & 1) && s2 >= 0 && s2 < Sx && ext_label(tg, s2, p.blank) != lab;
    // first row
    const int t_first = beta ? T - 1 : 0;
    float cur = NEG_INF;
    if (live) {
        const bool init = beta ? (s >= Sx - 2) : (s <= 1);
        if (init) cur = lp[(int64_t)t_first * p.V + lab];
        tab[(int64_t)t_first * p.Sx_max + s] = cur;
    }
    float *r0 = row[0] + 1, *r1 = row[1] + 1;                        // one guard cell either side... (index -1 / Sx handled below)
    r0[s] = cur;
    if (s == 0) { row[0][0] = NEG_INF; row[1][0] = NEG_INF; }
    __syncthreads();
    // the step's own log-probability lp_t(l'_s) is a gather from a fresh (V floats) row every step: requested PF steps ahead (as a
    // load inside the step it put a global-memory latency on every one of the T dependent steps: 0.8 us per step at V = 5000).
    // The loop body is BRANCH-FREE -- dead lanes and the steps past T of the last group of PF address outside the buffer
    // descriptors (loads return 0, stores are dropped) -- so that the compiler counts the outstanding loads / stores exactly: with
    // per-lane branches around them it waited vmcnt(0) twice per step, i.e. for the prefetch it had just issued.
    constexpr int PF = 8;
    const int OOB = 0x7fffffff;
    const __amdgpu_buffer_rsrc_t rlp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(lp), 0, (int)min((int64_t)T * p.V * 4, (int64_t)0x7ffffff0), 0x00020000);
    const __amdgpu_buffer_rsrc_t rtab = __builtin_amdgcn_make_buffer_rsrc(tab, 0, (int)min((int64_t)T * p.Sx_max * 4, (int64_t)0x7ffffff0), 0x00020000);
    const int vo_lp = live ? lab * 4 : OOB, vo_tab = live ? s * 4 : OOB;
    auto step_t = [&](int k) { return beta ? T - 1 - k : k; };       // steps k >= T: t outside [0, T) -> outside the descriptors
    auto lp_at = [&](int k) -> float {
        const int t = step_t(k);
        const bool in = t >= 0 && t < T;                             // uniform; (the scalar offset is not range-checked)
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rlp, in ? vo_lp : OOB, in ? t * p.V * 4 : 0, 0));
    };
    const int s1 = beta ? s + 1 : s - 1;
    const bool has1 = live && s1 >= 0 && s1 < Sx;
    const int i1 = has1 ? s1 : s, i2 = skip_ok ? s2 : s;             // in-bounds LDS indices for every lane
    float q[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) q[i] = lp_at(1 + i);
    for (int k0 = 1; k0 < T; k0 += PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int k = k0 + i, t = step_t(k);
            const float *prev = (k & 1) ? r0 : r1;
            float *next = (k & 1) ? r1 : r0;
            const float lpv = q[i];
            q[i] = lp_at(k + PF);
            const float a0 = prev[s], a1 = has1 ? prev[i1] : NEG_INF, a2 = skip_ok ? prev[i2] : NEG_INF;
            const float v = live ? lse3(a0, a1, a2) + lpv : NEG_INF;
            const bool in = t >= 0 && t < T;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rtab, in ? vo_tab : OOB, in ? t * p.Sx_max * 4 : 0, 0);
            next[s] = v;
            cm_lds_barrier();                                        // LDS only: __syncthreads() also waits for the step's global store (vmcnt)
        }
    }
}

// Small vocabularies (the 31-character CTC recipe): one wave per (utterance, step), a wave-wide fixed-order reduction per CLASS --
// V x 14 issue slots against the list walk's S x ceil(S / 64) x 3 below (V = 31, S = 500: 0.17 vs 1.3 ms at 32 x 1000)
__global__ __launch_bounds__(256) void ctc_grad_classes_kernel(const cm_ctc_args p) {
    const int wave = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (wave >= p.batch * p.T) return;
    const int b = wave / p.T, t = wave % p.T;
    const int T = min(p.input_lengths[b], p.T), S = min(p.target_lengths[b], p.S), Sx = 2 * S + 1;
    float *g = p.grad + ((int64_t)b * p.T + t) * p.V;
    const float *lp = p.log_probs + ((int64_t)b * p.T + t) * p.V;
    const float *al = p.alpha + (int64_t)b * p.T * p.Sx_max, *be = p.beta + (int64_t)b * p.T * p.Sx_max;
    float nll = __builtin_huge_valf();
    if (T > 0) {
        const float aN = al[(int64_t)(T - 1) * p.Sx_max + Sx - 1], aM = Sx > 1 ? al[(int64_t)(T - 1) * p.Sx_max + Sx - 2] : NEG_INF;
        nll = -lse3(aN, aM, NEG_INF);
    }
    const bool inf = !(nll < __builtin_huge_valf());                 // infeasible alignment (or NaN): zero_infinity
    if (t == 0 && lane == 0) p.nll[b] = inf ? 0.f : nll;
    if (t >= T || inf) {
        for (int v = lane; v < p.V; v += 64) g[v] = 0.f;
        return;
    }
    const int64_t *tg = p.targets + (int64_t)b * p.S;
    const float *ar = al + (int64_t)t * p.Sx_max, *br = be + (int64_t)t * p.Sx_max;
    // blank: even positions; this lane's share, then a fixed-order wave sum
    float blank_sum = 0.f;
    const float lpb = lp[p.blank];
    for (int s = 2 * lane; s < Sx; s += 128) blank_sum += cm_exp2(CM_LOG2E * (ar[s] + br[s] + nll - lpb));
    for (int off = 32; off > 0; off >>= 1) blank_sum += __shfl_xor(blank_sum, off, 64);
    // labels: odd positions; every lane keeps (label, posterior mass) of its positions, classes are visited in order
    constexpr int PER = 8;                                           // 64 lanes x 8 = 512 labels
    int lab[PER];
    float mass[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int j = lane + 64 * i, s = 2 * j + 1;
        lab[i] = -1, mass[i] = 0.f;
        if (j < S) {
            lab[i] = (int)tg[j];
            mass[i] = cm_exp2(CM_LOG2E * (ar[s] + br[s] + nll - lp[lab[i]]));
        }
    }
    for (int v = 0; v < p.V; ++v) {
        float sum = 0.f;
        if (v != p.blank) {
#pragma unroll
            for (int i = 0; i < PER; ++i) sum += lab[i] == v ? mass[i] : 0.f;
            for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        } else sum = blank_sum;
        if (lane == 0) g[v] = cm_exp2(CM_LOG2E * lp[v]) - sum;
    }
}


// one wave per (utterance, step): nll, the gradient row exp(lp) in one coalesced pass, then the posterior mass of every class that
// occurs in the target subtracted: the FIRST position of a class sums the masses of all its positions in increasing position order
// (fixed order: deterministic), found by walking the wave's (label, mass) list in LDS -- S compares per position instead of a
// wave-wide reduction per CLASS (the first version: 7 ms at V = 5000, S = 400, T = 4000; 170 us at V = 31)
__global__ __launch_bounds__(256) void ctc_grad_kernel(const cm_ctc_args p) {
    __shared__ int s_lab[4][512];
    __shared__ float s_mass[4][512];
    const int wv = threadIdx.x >> 6;
    const int wave = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (wave >= p.batch * p.T) return;
    const int b = wave / p.T, t = wave % p.T;
    const int T = min(p.input_lengths[b], p.T), S = min(p.target_lengths[b], p.S), Sx = 2 * S + 1;
    float *g = p.grad + ((int64_t)b * p.T + t) * p.V;
    const float *lp = p.log_probs + ((int64_t)b * p.T + t) * p.V;
    const float *al = p.alpha + (int64_t)b * p.T * p.Sx_max, *be = p.beta + (int64_t)b * p.T * p.Sx_max;
    float nll = __builtin_huge_valf();
    if (T > 0) {
        const float aN = al[(int64_t)(T - 1) * p.Sx_max + Sx - 1], aM = Sx > 1 ? al[(int64_t)(T - 1) * p.Sx_max + Sx - 2] : NEG_INF;
        nll = -lse3(aN, aM, NEG_INF);
    }
    const bool inf = !(nll < __builtin_huge_valf());                 // infeasible alignment (or NaN): zero_infinity
    if (t == 0 && lane == 0) p.nll[b] = inf ? 0.f : nll;
    if (t >= T || inf) {
        for (int v = lane; v < p.V; v += 64) g[v] = 0.f;
        return;
    }
    const int64_t *tg = p.targets + (int64_t)b * p.S;
    const float *ar = al + (int64_t)t * p.Sx_max, *br = be + (int64_t)t * p.Sx_max;
    // blank: even positions; this lane's share, then a fixed-order wave sum
    float blank_sum = 0.f;
    const float lpb = lp[p.blank];
    for (int s = 2 * lane; s < Sx; s += 128) blank_sum += cm_exp2(CM_LOG2E * (ar[s] + br[s] + nll - lpb));
    for (int off = 32; off > 0; off >>= 1) blank_sum += __shfl_xor(blank_sum, off, 64);
    // labels: odd positions; lane keeps positions lane + 64 i
    constexpr int PER = 8;                                           // 64 lanes x 8 = 512 labels
    int lab[PER];
    float lpl[PER], sum[PER];
    bool first[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int j = lane + 64 * i, s = 2 * j + 1;
        lab[i] = -1, lpl[i] = 0.f, sum[i] = 0.f, first[i] = true;
        float m = 0.f;
        if (j < S) {
            lab[i] = (int)tg[j];
            lpl[i] = lp[lab[i]];
            m = cm_exp2(CM_LOG2E * (ar[s] + br[s] + nll - lpl[i]));
        }
        s_lab[wv][j] = lab[i];
        s_mass[wv][j] = m;
    }
    // the gradient row without the posterior terms (classes absent from the target keep it)
    for (int v = lane; v < p.V; v += 64) g[v] = cm_exp2(CM_LOG2E * lp[v]);
    // (the wave's own LDS writes above are visible to its later reads: one wave's LDS operations execute in order)
    for (int k = 0; k < S; ++k) {
        const int lk = s_lab[wv][k];                                 // broadcast reads
        const float mk = s_mass[wv][k];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const bool same = lab[i] == lk;
            first[i] = first[i] && !(same && k < lane + 64 * i);
            sum[i] += same ? mk : 0.f;                               // k increasing: a fixed order (positions before the first add 0 to a non-owner)
        }
    }
    __builtin_amdgcn_s_waitcnt(0);                                    // the row's plain stores are done before the scattered ones to the same row
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < PER; ++i)
        if (lab[i] >= 0 && first[i] && lab[i] != p.blank) g[lab[i]] = cm_exp2(CM_LOG2E * lpl[i]) - sum[i];
    if (lane == 0) g[p.blank] = cm_exp2(CM_LOG2E * lpb) - blank_sum;
}

}  // namespace

extern "C" int64_t cm_ctc_workspace_floats(int32_t batch, int32_t T, int32_t S) {
    if (batch <= 0 || T <= 0 || S < 0) return 0;
    return 2 * (int64_t)batch * T * (2 * S + 1);
}

extern "C" int cm_ctc_loss(const cm_ctc_args *args) {
    CM_REQUIRE(args != nullptr, CM_EINVAL, "ctc_loss: args is NULL");
    cm_ctc_args a = *args;
    CM_REQUIRE(a.batch > 0 && a.T > 0 && a.V > 1 && a.S >= 0, CM_EINVAL, "ctc_loss: bad sizes batch=%d T=%d V=%d S=%d", a.batch, a.T, a.V, a.S);
    CM_REQUIRE(a.S <= 511, CM_EUNSUPPORTED, "ctc_loss: at most 511 labels per utterance (got %d)", a.S);
    CM_REQUIRE(a.blank >= 0 && a.blank < a.V, CM_EINVAL, "ctc_loss: blank %d out of range", a.blank);
    CM_REQUIRE(a.log_probs && a.targets && a.input_lengths && a.target_lengths && a.nll && a.grad && a.workspace, CM_EINVAL, "ctc_loss: NULL tensor");
    CM_REQUIRE(a.workspace_floats >= cm_ctc_workspace_floats(a.batch, a.T, a.S), CM_EINVAL, "ctc_loss: workspace smaller than cm_ctc_workspace_floats()");
    a.Sx_max = 2 * a.S + 1;
    a.alpha = a.workspace;
    a.beta = a.workspace + (int64_t)a.batch * a.T * a.Sx_max;
    hipStream_t st = reinterpret_cast<hipStream_t>(a.stream);
    hipLaunchKernelGGL(ctc_alpha_beta_kernel, dim3(a.batch, 2), dim3(1024), 0, st, a);
    if (int rc = cm_launch_status("cm_ctc_loss(alpha, beta)")) return rc;
    const int64_t waves = (int64_t)a.batch * a.T;
    // per-class reductions cost ~ V x 14 issue slots per (utterance, step), the list walk ~ S x ceil(S / 64) x 3
    const bool by_class = (int64_t)a.V * 14 <= (int64_t)a.S * ((a.S + 63) / 64) * 3;
    if (by_class) hipLaunchKernelGGL(ctc_grad_classes_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(ctc_grad_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, a);
    return cm_launch_status("cm_ctc_loss(gradient)");
}
