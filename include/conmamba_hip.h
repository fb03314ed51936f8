/* conmamba_hip.h — C ABI of libconmamba_hip.so (MI355X / gfx950 native kernels for the
 * ConMamba ASR encoder hot path).
 *
 * This is the drop-in boundary: the entry points below are what the reference's Python
 * binds where it today imports the CUDA-only extension modules
 *     selective_scan_cuda   (reference modules/mamba/selective_scan_interface.py:16)
 *     causal_conv1d_cuda    (reference modules/mamba/selective_scan_interface.py:15)
 * Plain pointers and sizes only — no torch types.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (outputs and scratch included);
 *     the library never allocates, frees or retains memory;
 *   - calls are asynchronous on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream) and use the caller's current device; no device synchronisation inside;
 *   - return value: CM_OK (0) on success, a negative CM_E* code for rejected arguments, a
 *     positive value = hipError_t from the launch; cm_last_error() returns a thread-local
 *     message; nothing aborts or throws across this boundary;
 *   - re-entrant, callable from any thread (autograd backward threads included);
 *   - "time-contiguous" tensors are laid out (batch, dim, seqlen) with element stride 1 along
 *     seqlen, as the reference makes them before calling its kernels
 *     (selective_scan_interface.py:24-35, 177-178); batch/dim strides are passed in elements so
 *     that x / z (and dx / dz) may be the two halves of one xz tensor (:180, :249-256);
 *   - fp32 reduction outputs (dA, dD, ddelta_bias, dB, dC, dweight, dbias) are ACCUMULATED
 *     into: the caller zero-initialises them (or passes running sums on purpose).
 */
#ifndef CONMAMBA_HIP_H
#define CONMAMBA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CM_ABI_VERSION 10

/* error codes */
#define CM_OK            0
#define CM_EINVAL       (-1)   /* inconsistent sizes / null required pointer           */
#define CM_EUNSUPPORTED (-2)   /* valid request this build has no kernel for           */
#define CM_EALIGN       (-3)   /* pointer/stride violates a documented alignment rule  */

/* element types of activation tensors */
typedef enum { CM_F32 = 0, CM_BF16 = 1, CM_F16 = 2 } cm_dtype;

/* timesteps per scan checkpoint: the `x` tensor of the scan holds the recurrent state at the
 * end of every chunk of this many steps (the role of selective_scan_cuda.fwd's second return
 * value, selective_scan_interface.py:42-45: last_state = x[:, :, -1, 1::2]). */
#define CM_SCAN_CHUNK 64

int         cm_abi_version(void);
const char *cm_last_error(void);
/* number of checkpoint chunks for a sequence length: ceil(seqlen / CM_SCAN_CHUNK) */
int         cm_scan_num_chunks(int seqlen);
#ifdef CM_ABLATE
/* Ablation build only (make -C mamba_asr_amd/csrc ablate -> lib/libconmamba_hip_ablate.so, loaded by the tools through
 * CM_LIB_PATH): a process-wide switch selecting timing-only kernel variants with one stage removed (results are then WRONG by
 * construction; tools/bench_scan.py, bench_ffn.py produce DESIGN.md's ablation tables with it).  The product library is
 * compiled without CM_ABLATE: it has no such switch, no ablation kernels and no process-global mutable state. */
int         cm_debug_set(int ablation);
int         cm_debug_get(void);
#endif

/* ---------------------------------------------------------------------------------------
 * Selective scan forward — replaces selective_scan_cuda.fwd
 *   (call sites selective_scan_interface.py:42, 218, 359, 504, 508; spec = selective_scan_ref
 *    :91-157).  Real A, input-dependent B/C with one group: B, C are (batch, 1, dstate, seqlen).
 *
 *   delta' = delta + delta_bias[d]            (if delta_bias)
 *   delta' = softplus(delta')                 (if delta_softplus; threshold 20)
 *   h_t = exp(delta'_t A[d,n]) h_{t-1} + delta'_t B[n,t] u_t ,  h_{-1} = 0
 *   out_t   = sum_n C[n,t] h_t[n] + D[d] u_t  (D optional)
 *   out_z_t = out_t * silu(z_t)               (z optional)
 * reverse_time != 0 runs the recurrence from t = seqlen-1 down to 0 on the tensors as stored —
 * numerically what the reference obtains by .flip(-1) on every input and on the output
 * (modules/mamba/bimamba.py:237, 253) without materialising the flips.
 * ------------------------------------------------------------------------------------- */
typedef struct cm_scan_fwd_args {
    int32_t batch, dim, seqlen, dstate;
    int32_t io_dtype;        /* cm_dtype of u, delta, z, out, out_z                       */
    int32_t bc_dtype;        /* cm_dtype of B, C                                          */
    int32_t delta_softplus;  /* bool                                                      */
    int32_t reverse_time;    /* bool                                                      */
    const void  *u;          /* (batch, dim, seqlen)                                      */
    const void  *delta;      /* (batch, dim, seqlen)                                      */
    const float *A;          /* (dim, dstate) fp32, contiguous                            */
    const void  *B;          /* (batch, 1, dstate, seqlen)                                */
    const void  *C;          /* (batch, 1, dstate, seqlen)                                */
    const float *D;          /* (dim) fp32 or NULL                                        */
    const void  *z;          /* (batch, dim, seqlen) or NULL                              */
    const float *delta_bias; /* (dim) fp32 or NULL                                        */
    void        *out;        /* (batch, dim, seqlen) pre-gate output; may be NULL if z    */
    void        *out_z;      /* gated output; required iff z != NULL                      */
    float       *x;          /* (batch, dim, nchunks, 2*dstate) fp32 checkpoints or NULL:
                                [2n] = product of exp(delta' A) over the chunk, [2n+1] = h at
                                the chunk's last processed step                            */
    int64_t u_bs, u_ds;          /* batch / dim strides in elements (time stride is 1)    */
    int64_t delta_bs, delta_ds;
    int64_t z_bs, z_ds;
    int64_t out_bs, out_ds;      /* shared by out and out_z                               */
    int64_t B_bs, B_ns;          /* batch / state strides of B                            */
    int64_t C_bs, C_ns;
    void   *stream;
    const float *h0;         /* (batch, dim, dstate) fp32, contiguous, or NULL: the state the
                                recurrence starts from instead of zero (in scan order: before
                                step 0, or behind step seqlen-1 with reverse_time).  Forward only
                                (cm_selective_scan_bwd rejects it).  With it, a sequence cut into
                                time shards gives the unsplit result: shard k starts from shard
                                k-1's last state x[:, :, last chunk, 1::2] -- the carry the
                                time-split scan across GPUs exchanges (no
                                reference counterpart: the reference has no sequence parallelism,
                                SURVEY.md §5)                                               */
    int32_t lanes_per_channel; /* tuning: lanes one channel's states are split over (1, 2, 4, 8, 16; backward: 4, 8,
                                16); 0 = the library's choice from the problem size.  Results do not depend on it
                                beyond fp32 summation order                                  */
    int32_t pad4_;
} cm_scan_fwd_args;

int cm_selective_scan_fwd(const cm_scan_fwd_args *args);

/* ---------------------------------------------------------------------------------------
 * Selective scan backward — replaces selective_scan_cuda.bwd
 *   (call sites selective_scan_interface.py:67, 252, 394, 546, 553).
 * Inputs as forward plus dout (gradient of out_z if z else of out) and the forward's x.
 * du, ddelta, dz have io_dtype; dz may alias a slice of a larger tensor (the reference writes
 * it into dxz, :249-256).  dA (dim,dstate), dD (dim), ddelta_bias (dim), dB, dC
 * (batch,1,dstate,seqlen) are fp32 and ACCUMULATED into.  If out_z != NULL (and z != NULL) the
 * gated forward output is recomputed into it (the reference's recompute_out_z flag).
 * ------------------------------------------------------------------------------------- */
typedef struct cm_scan_bwd_args {
    cm_scan_fwd_args fwd;    /* same tensors as forward; fwd.out / fwd.out_z: see above;
                                fwd.x = checkpoints written by the forward (required)      */
    const void *dout;        /* (batch, dim, seqlen), io_dtype                            */
    int64_t dout_bs, dout_ds;
    void  *du;               /* (batch, dim, seqlen), io_dtype                            */
    void  *ddelta;
    void  *dz;               /* or NULL when z == NULL                                    */
    int64_t du_bs, du_ds, ddelta_bs, ddelta_ds, dz_bs, dz_ds;
    float *dA;               /* (dim, dstate)                                             */
    float *dB;               /* (batch, 1, dstate, seqlen) contiguous fp32                */
    float *dC;
    float *dD;               /* (dim) or NULL                                             */
    float *ddelta_bias;      /* (dim) or NULL                                             */
    void  *workspace;        /* optional scratch of cm_selective_scan_bwd_workspace_bytes()
                                bytes (16-byte aligned), or NULL.  With it every reduction
                                across workgroups (dB / dC over channel tiles; dA, dD,
                                ddelta_bias over the batch) is a per-workgroup partial + a
                                second kernel that sums in a FIXED order: gradients are
                                bit-identical from run to run.  Without it they are fp32
                                atomics, as in the CUDA kernels the reference binds          */
    int64_t workspace_bytes; /* size of workspace; 0 with NULL                            */
} cm_scan_bwd_args;

int cm_selective_scan_bwd(const cm_scan_bwd_args *args);
/* bytes of workspace the deterministic path needs for these sizes (uses batch, dim, seqlen, dstate only) */
int64_t cm_selective_scan_bwd_workspace_bytes(const cm_scan_bwd_args *args);

/* ---------------------------------------------------------------------------------------
 * Causal depthwise conv1d (+ optional SiLU) — replaces causal_conv1d_cuda.causal_conv1d_fwd /
 * causal_conv1d_bwd (call sites selective_scan_interface.py:182, 244, 286, 323, 385, 430; the
 * reference's own definition of the op: modules/mamba/bimamba.py:83-91, 278-279).
 *   y[b,d,t] = act( bias[d] + sum_{k<width} weight[d,k] * x[b,d,t-(width-1)+k] ),  x[<0] = 0
 * reverse_time != 0 mirrors the time axis (anti-causal conv), i.e. conv(flip(x)) flipped back.
 * ------------------------------------------------------------------------------------- */
typedef struct cm_conv_args {
    int32_t batch, dim, seqlen, width;     /* width 2..4                                   */
    int32_t io_dtype;                      /* cm_dtype of x, y, dy, dx                      */
    int32_t silu;                          /* bool                                          */
    int32_t reverse_time;                  /* bool                                          */
    const void  *x;                        /* (batch, dim, seqlen)                          */
    const float *weight;                   /* (dim, width) fp32 contiguous                  */
    const float *bias;                     /* (dim) fp32 or NULL                            */
    void        *y;                        /* forward output (fwd only)                     */
    int64_t x_bs, x_ds, y_bs, y_ds;
    /* backward only */
    const void  *dy;                       /* (batch, dim, seqlen)                          */
    void        *dx;                       /* may alias a slice of a larger tensor          */
    float       *dweight;                  /* (dim, width) fp32, accumulated                */
    float       *dbias;                    /* (dim) fp32 or NULL, accumulated               */
    int64_t dy_bs, dy_ds, dx_bs, dx_ds;
    void   *stream;
    float  *workspace;                     /* backward, optional: batch * dim * (width + 1) floats, or NULL.  With it the
                                              per-(batch, channel) partial sums of dweight / dbias are stored and summed over
                                              the batch in a fixed order by a second kernel (bit-identical from run to run);
                                              without it they are fp32 atomics                                          */
} cm_conv_args;

int cm_causal_conv1d_fwd(const cm_conv_args *args);
int cm_causal_conv1d_bwd(const cm_conv_args *args);


/* ---------------------------------------------------------------------------------------
 * Channels-last selective scan forward — the MI355X-native layout of the fused BiMamba path.
 * Same math as cm_selective_scan_fwd, but activations are (batch, seqlen, dim) with the channel
 * axis contiguous (what the in_proj GEMM writes: reference bimamba.py:192-196 produces the same
 * numbers transposed), B/C are fp32 (dstate, batch, seqlen) with time contiguous, and up to two
 * independent directions (the two halves of BiMamba v2, reference bimamba.py:223-248) run in ONE
 * launch: direction d uses its own u/delta/A/B/C/D/delta_bias/out and walks time forward
 * (reverse[d] == 0) or backward.  z is shared.  Row (time) and batch strides are in elements, so
 * z / out may be column slices of wider buffers ([x|z], [y_fwd|y_bwd]).
 * B/C rows must be readable up to the next multiple of 16 steps past seqlen (pad the allocation).
 * ------------------------------------------------------------------------------------- */
typedef struct cm_scan_cl_dir {
    const void  *u;          /* (batch, seqlen, dim)  conv+SiLU output                      */
    const void  *delta;      /* (batch, seqlen, dim)  pre-bias, pre-softplus; ignored when dt_low is set */
    const float *A;          /* (dim, dstate)                                               */
    const float *B;          /* (dstate, batch, seqlen) fp32                                */
    const float *C;
    const float *dt_low;     /* optional (dt_rank, batch, seqlen) fp32, same strides as B/C: the low-rank
                                time-step features x_dbl[:, :dt_rank]; when given the kernel forms
                                delta = dt_weight @ dt_low itself (the reference's dt_proj GEMM,
                                selective_scan_interface.py:187) and no delta tensor is read          */
    const float *dt_weight;  /* (dim, dt_rank) fp32, required with dt_low; dt_rank <= 16           */
    const float *D;          /* (dim) or NULL                                               */
    const float *delta_bias; /* (dim) or NULL                                               */
    void        *out;        /* (batch, seqlen, dim)  gated output                          */
    int64_t u_bs, u_ts, delta_bs, delta_ts, out_bs, out_ts;
    int64_t bc_ns, bc_bs;    /* state / batch strides of B and C                            */
    int32_t reverse_time;
    int32_t dt_rank;
    const void *xdbl;        /* optional (batch, seqlen, P + 32) in the I/O dtype: the x_proj GEMM's output rows as it
                                wrote them, columns [0,P) = dt features zero-padded to P, [P,P+16) = B_t, [P+16,P+32)
                                = C_t (selective_scan_interface.py:186, 192-215 without the transposed copies), where
                                P = 16 when dt_rank <= 16 and P = 32 when 16 < dt_rank <= 32 (bf16 only; the S2S-large
                                encoder, d_model 512).  When set for every direction, B / C / dt_low / delta are
                                ignored, dt_weight must be (dim, P) fp32 zero-padded, dim a multiple of 8 (bf16) /
                                4 (fp32), and the row-group kernel (csrc/scan_rows_fwd.hip) runs                    */
    int64_t xdbl_bs, xdbl_ts;/* batch / step strides of xdbl in elements (multiples of 8 for bf16, 4 for fp32)       */
    /* xdbl mode only, all optional, (batch, dim, 16) fp32 contiguous -- the carry interface of the time-split scan
       (no reference counterpart): */
    const float *h0;         /* state the recurrence starts from (in this direction's scan order) instead of zero     */
    float       *h_last;     /* out: state after the last processed step                                               */
    float       *decay;      /* out: product over the sequence of exp(delta' A), i.e. what a state entering this shard
                                is multiplied by on its way through it                                                */
    /* xdbl mode only, optional: what the training forward saves for cm_scan_cl_bwd (the role of the checkpoint tensor `x`
       of selective_scan_cuda.fwd, selective_scan_interface.py:42, on this kernel's 16-step blocks) */
    float       *ckpt;       /* out: (batch, 2 ceil(seqlen / 16), dim, 16) fp32: the state the recurrence ENTERS each half
                                block of 8 steps [8 m, 8 m + 8) with, in this direction's scan order (m < 2 ceil(seqlen/16);
                                steps past seqlen pass the state through)                                            */
    void        *ypre;       /* out: (batch, seqlen, dim), I/O dtype: the pre-gate output sum_n C h + D u              */
    int64_t ypre_bs, ypre_ts;
} cm_scan_cl_dir;

typedef struct cm_scan_cl_args {
    int32_t batch, seqlen, dim, dstate;
    int32_t io_dtype;        /* CM_BF16 or CM_F32: u, delta, z, out                         */
    int32_t delta_softplus;
    int32_t ndir;            /* 1 or 2                                                      */
    int32_t time_chunks;     /* xdbl mode only: 0 / 1 = one workgroup per (sequence, 64 channels); C > 1 = cut every
                                sequence into C chunks of whole 16-step blocks that run in parallel -- a summary pass
                                (per-chunk decay product and zero-state end state), a carry fold in scan order, and an
                                output pass from each chunk's entry state: SURVEY.md §8f row 3's algebra inside one
                                GPU, for batches too small to fill the chip (cm_scan_cl_fwd_auto_chunks).  Outputs
                                differ from the unchunked launch by fp32 rounding only.  Needs `workspace`.           */
    const void *z;           /* (batch, seqlen, dim) or NULL                                */
    int64_t z_bs, z_ts;
    cm_scan_cl_dir dir[2];
    void *stream;
    void   *workspace;       /* time_chunks > 1: cm_scan_cl_fwd_workspace_bytes(args) bytes, 16-byte aligned, caller owned */
    int64_t workspace_bytes;
    int32_t lanes_per_channel; /* tuning: lanes a channel's 16 states are spread over.  State-split kernel (no xdbl): 4, 8 or
                                  16; xdbl mode: 4 (scan_rows_fwd.hip) or 8 (scan_rows_fwd2.hip: z + softplus, dt_rank <= 16,
                                  unchunked; measured slower at every size, never chosen automatically); 0 = automatic  */
    int32_t pad5_;
} cm_scan_cl_args;

int cm_scan_cl_fwd(const cm_scan_cl_args *args);
/* bytes of workspace the launch described by args needs (0 unless time_chunks > 1) */
int64_t cm_scan_cl_fwd_workspace_bytes(const cm_scan_cl_args *args);
/* the chunk count that fills 256 CUs for this problem size (1 = do not chunk); a pure function of the sizes */
int32_t cm_scan_cl_fwd_auto_chunks(int32_t batch, int32_t seqlen, int32_t dim, int32_t ndir);

/* ---------------------------------------------------------------------------------------
 * Channels-last selective scan BACKWARD, both BiMamba directions in one launch (csrc/scan_rows_bwd.hip): the gradient
 * of cm_scan_cl_fwd's xdbl mode with z and softplus -- selective_scan_cuda.bwd (selective_scan_interface.py:252-256)
 * together with the dt_proj part of MambaInnerFnNoOutProj.backward (:258-283: ddelta -> d dt_proj.weight and the dt
 * columns of dx_dbl), on the forward's layout: rows (batch * time, channels), x_dbl rows as the x_proj GEMM wrote them,
 * no transposed copy.  Per direction:
 *   in : u, xdbl, A, dt_weight (dim, P) fp32 zero padded, D, delta_bias (as the forward), ckpt and ypre written by the
 *        forward, dout = gradient of the gated output; shared z
 *   out: du (gradient w.r.t. u through the scan only: the x_proj path is added by the caller), dz (THIS direction's
 *        share of dz: the caller adds the two), dxdbl (batch, seqlen, P + 32) in the I/O dtype = [d dt (P) | dB | dC],
 *        and fp32 dA (dim, 16), ddt_weight (dim, P), dD (dim), ddelta_bias (dim), ACCUMULATED into.
 * Every cross-workgroup sum (dxdbl over the channel groups, the parameter gradients over the batch) goes through the
 * caller-owned workspace and a fixed-order second pass: bit-identical results from run to run.
 * dim: multiple of 8 (bf16) / 4 (fp32); dstate 16; dt_rank <= 16, or <= 32 with bf16 I/O; strides in elements.
 * ------------------------------------------------------------------------------------- */
typedef struct cm_scan_cl_bwd_dir {
    const void  *u;          /* (batch, seqlen, dim)                                         */
    const void  *xdbl;       /* (batch, seqlen, P + 32)                                      */
    const float *A;          /* (dim, 16)                                                    */
    const float *dt_weight;  /* (dim, P) fp32, zero padded                                   */
    const float *D;          /* (dim) or NULL                                                */
    const float *delta_bias; /* (dim) or NULL                                                */
    const float *ckpt;       /* (batch, 2 ceil(seqlen / 16), dim, 16) from the forward      */
    const void  *ypre;       /* (batch, seqlen, dim) from the forward                       */
    const void  *dout;       /* (batch, seqlen, dim)                                         */
    void        *du, *dz;    /* (batch, seqlen, dim)                                         */
    void        *dxdbl;      /* (batch, seqlen, P + 32)                                      */
    float       *dA, *ddt_weight, *dD, *ddelta_bias;   /* dD / ddelta_bias may be NULL      */
    int64_t u_bs, u_ts, xdbl_bs, xdbl_ts, ypre_bs, ypre_ts, dout_bs, dout_ts, du_bs, du_ts, dz_bs, dz_ts, dxdbl_bs, dxdbl_ts;
    int32_t reverse_time;
    int32_t dt_rank;         /* P: 16 or 32                                                  */
} cm_scan_cl_bwd_dir;

typedef struct cm_scan_cl_bwd_args {
    int32_t batch, seqlen, dim, dstate;
    int32_t io_dtype;        /* CM_BF16 or CM_F32                                            */
    int32_t ndir;            /* 1 or 2                                                       */
    const void *z;           /* (batch, seqlen, dim), required                               */
    int64_t z_bs, z_ts;
    int32_t time_chunks;     /* 0: automatic (launches under 512 workgroups are cut along time: adjoint summaries per chunk,
                                a carry fold, then the full pass per chunk from its carried-in adjoint); 1: never; n: n chunks */
    int32_t overwrite;       /* 1: dA / ddt_weight / dD / ddelta_bias are written, not accumulated into */
    cm_scan_cl_bwd_dir dir[2];
    void   *stream;
    void   *workspace;       /* cm_scan_cl_bwd_workspace_bytes(args) bytes, 16-byte aligned  */
    int64_t workspace_bytes;
    int32_t da_log;          /* 1: dA is the gradient w.r.t. A_log of the reference's A = -exp(A_log) (bimamba.py:116-128), i.e. dA * A,
                                formed in the reduce pass (one element-wise launch per direction less in the caller)      */
    int32_t reserved0;
} cm_scan_cl_bwd_args;

int64_t cm_scan_cl_bwd_workspace_bytes(const cm_scan_cl_bwd_args *args);
/* the chunk count time_chunks = 0 resolves to */
int cm_scan_cl_bwd_auto_chunks(int batch, int seqlen, int dim, int ndir);
int cm_scan_cl_bwd(const cm_scan_cl_bwd_args *args);

/* ---------------------------------------------------------------------------------------
 * Channels-last causal conv, both BiMamba directions in one pass over x.
 *   y_fwd[b,t,c] = silu(bias_f[c] + sum_k w_f[c,k] x[b,t-(W-1)+k,c])      (causal,  bimamba.py:223-235)
 *   y_bwd[b,t,c] = silu(bias_b[c] + sum_k w_b[c,k] x[b,t+(W-1)-k,c])      (the reference's flipped branch,
 *                                                                           bimamba.py:236-248, un-flipped)
 * x, y_fwd, y_bwd: (batch, seqlen, dim), channel axis contiguous, strides in elements; width 4.
 * y_bwd / weight_b / bias_b may be NULL for a single (causal) direction.
 * ------------------------------------------------------------------------------------- */
typedef struct cm_conv_cl_args {
    int32_t batch, seqlen, dim, width;
    int32_t io_dtype;            /* CM_BF16 or CM_F32 */
    int32_t silu;
    const void  *x;
    const float *weight_f, *bias_f, *weight_b, *bias_b;   /* (dim, width) / (dim); bias may be NULL */
    void *y_fwd, *y_bwd;
    int64_t x_bs, x_ts, yf_bs, yf_ts, yb_bs, yb_ts;
    void *stream;
} cm_conv_cl_args;

int cm_conv_cl_fwd(const cm_conv_cl_args *args);

/* Backward of cm_conv_cl_fwd / the conv half of cm_conv_xproj (both directions in one pass; replaces
 * causal_conv1d_cuda.causal_conv1d_bwd, selective_scan_interface.py:286-288, run once per direction by the reference):
 *   dx = dx_fwd + dx_bwd (the two directions share x: reference bimamba.py:223-248), pre-activations recomputed from x;
 *   dz = dz_f + dz_b (the two directions' shares of the gate gradient, cm_scan_cl_bwd) when dz != NULL;
 *   dweight_* (dim, 4), dbias_* (dim) fp32, ACCUMULATED into, summed in a fixed order through `workspace`.
 * du_b == NULL: one direction (the unidirectional mixer of the Mamba decoder).  Strides in elements. */
typedef struct cm_conv_cl_bwd_args {
    int32_t batch, seqlen, dim, width;
    int32_t io_dtype, pad_;
    const void  *x;                                       /* (batch, seqlen, dim) view                       */
    const float *weight_f, *bias_f, *weight_b, *bias_b;   /* (dim, 4) / (dim); biases may be NULL           */
    const void  *du_f, *du_b;                             /* gradients of u_fwd / u_bwd                      */
    const void  *dz_f, *dz_b;                             /* optional                                        */
    void *dx, *dz;                                        /* outputs; dz optional                            */
    float *dweight_f, *dbias_f, *dweight_b, *dbias_b;
    int64_t x_bs, x_ts, duf_bs, duf_ts, dub_bs, dub_ts, dzf_bs, dzf_ts, dzb_bs, dzb_ts, dx_bs, dx_ts, dz_bs, dz_ts;
    void *stream;
    float *workspace;                                     /* cm_conv_cl_bwd_workspace_floats() floats       */
    int64_t workspace_floats;
    int32_t overwrite;                                    /* 1: the parameter gradients are written, not accumulated into */
    int32_t reserved0;
} cm_conv_cl_bwd_args;

/* CTC loss + gradient (csrc/ctc.hip; replaces torch.nn.functional.ctc_loss behind speechbrain.nnet.losses.ctc_loss, reference
 * train_CTC.py:405): log_probs (batch, T, V) fp32 contiguous (log-softmax outputs), targets (batch, S) int64 padded, lengths int32.
 * nll[b] = negative log-likelihood (0 where no alignment exists: zero_infinity); grad (batch, T, V) fp32 = exp(lp) - posteriors
 * for t < input_lengths[b], 0 elsewhere -- the gradient torch returns for a unit upstream gradient per utterance (the caller
 * scales by grad_out / batch).  S <= 511.  Deterministic (fixed summation order). */
typedef struct cm_ctc_args {
    int32_t batch, T, V, S, blank, Sx_max;   /* Sx_max: set by the library (2 S + 1)                          */
    const float   *log_probs;
    const int64_t *targets;
    const int32_t *input_lengths, *target_lengths;
    float *nll, *grad;
    float *workspace;                         /* cm_ctc_workspace_floats(batch, T, S) floats                    */
    int64_t workspace_floats;
    float *alpha, *beta;                      /* set by the library (views of workspace)                        */
    void *stream;
} cm_ctc_args;

int64_t cm_ctc_workspace_floats(int32_t batch, int32_t T, int32_t S);
int cm_ctc_loss(const cm_ctc_args *args);

/* Element-wise stages of a feed-forward / convolution module's training step on (rows, dim) tensors (csrc/ffn_train.hip; the
 * reference leaves them to torch: reference modules/Conmamba.py:597-617):
 *   cm_bias_act_dropout_fwd   y = dropout(act(a + bias))  [I/O dtype]      or, with res:  y = res + alpha * dropout(a + bias)  [fp32]
 *   cm_bias_act_dropout_bwd   da = alpha * dy * mask / (1 - p) * act'(a + bias)  [I/O dtype];  dbias += column sums of da (fixed order)
 * act: 0 none, 1 GELU (erf form), 2 GLU (a and da are (rows, 2 dim), bias / dbias (2 dim); no dropout, no residual).  mask: one byte per element, NULL = no dropout (eval, or p == 0); the forward draws it from a
 * counter hash of (seed, element index).  dim: multiple of 8, <= 2048; tensors contiguous, 16-byte aligned. */
#define CM_SEED_EPOCH_MUL 0x9E3779B97F4A7C15ull
typedef struct cm_ffn_elem_args {
    int64_t rows;
    int32_t dim, io_dtype, act, dy_f32;      /* dy_f32: backward's dy is fp32 although io_dtype is bf16 (the residual stream)   */
    const void  *a;                           /* (rows, dim) I/O dtype: the GEMM output without bias (backward: needed when act)  */
    const float *bias;                        /* (dim) or NULL                                                                     */
    const float *res;                         /* forward, optional: (rows, dim) fp32 residual                                      */
    void        *y;                           /* forward out: I/O dtype, or fp32 with res                                          */
    uint8_t     *mask;                        /* (rows, dim) bytes or NULL.  p > 0 drops with or without it: forward stores the
                                                 decisions there when given; backward reads them, or re-derives them from seed     */
    const void  *dy;                          /* backward in                                                                       */
    void        *da;                          /* backward out, I/O dtype                                                           */
    float       *dbias;                       /* backward, optional: (dim) fp32, ACCUMULATED into                                  */
    float       *dbias_part;                  /* cm_bias_act_dropout_bwd_workspace_floats(rows, dim) floats, required with dbias   */
    float p, alpha;
    uint64_t seed;                            /* dropout stream (cm_dropout.h): element e of the (rows, dim) tensor                */
    void *stream;
    void        *act_out;                     /* backward, optional (act 1, bf16): dropout(GELU(a + bias)) recomputed -- what the
                                                 training forward of cm_ffn_fused fed to its second GEMM                            */
    int32_t overwrite;                        /* 1: dbias is written, not accumulated into (no memset in front of the call)        */
    int32_t reserved0;
    const uint64_t *seed_epoch;               /* optional DEVICE word: the stream's seed is seed + *seed_epoch * CM_SEED_EPOCH_MUL
                                                 (mod 2^64), read when the kernel runs -- a captured hipGraph whose first node
                                                 advances the word draws fresh decisions at every replay                            */
} cm_ffn_elem_args;

int64_t cm_bias_act_dropout_bwd_workspace_floats(int64_t rows, int32_t dim);
int cm_bias_act_dropout_fwd(const cm_ffn_elem_args *args);
int cm_bias_act_dropout_bwd(const cm_ffn_elem_args *args);

/* out[j] = sum over b < nbatch of in[b * n + j], fp32 accumulation in a fixed order: folds the per-utterance weight-gradient
 * products of the training step (the reference forms them as one K = batch * time GEMM inside autograd).  in / out dtype:
 * CM_F32 or CM_BF16; n a multiple of 8 (bf16 in) / 4 (fp32 in); 16-byte aligned. */
int cm_sum_leading(const void *in, void *out, int32_t nbatch, int64_t n, int32_t in_dtype, int32_t out_dtype, void *stream);

/* Reflect padding by `pad` of the time and frequency axes of a dense channels-last (batch, time, freq, channels) tensor -- the 'same'
 * padding in front of every Conv2d of the reference's front end (speechbrain ConvolutionFrontEnd, hparams/CTC/conmamba_large.yaml:187-194).
 *   backward = 0: dst (batch, time + 2 pad, freq + 2 pad, channels) = padded src
 *   backward = 1: src is the padded tensor's gradient, dst (batch, time, freq, channels) its fold back onto the source positions
 * dtype CM_F32 / CM_BF16; 1 <= pad, 2 pad < time, freq. */
int cm_reflect_pad_tf(const void *src, void *dst, int32_t batch, int32_t time, int32_t freq, int32_t channels, int32_t pad,
                      int32_t dtype, int32_t backward, void *stream);

/* ---------------------------------------------------------------------------------------
 * Weight gradient of a Linear: out (m, n) fp32 = sum over rows k of a[k, :m]^T b[k, :n]  (dW = dY^T X; the reference leaves it to
 * autograd: one GEMM with K = batch x time per Linear, modules/Conmamba.py:597-650, selective_scan_interface.py:262-284).
 * a (rows, m), b (rows, n) bf16, row strides lda / ldb in elements (multiples of 8); m, n multiples of 128.  Split over row chunks into
 * workspace (cm_wgrad_workspace_floats floats), folded in a fixed order: deterministic.  out is WRITTEN.
 * ------------------------------------------------------------------------------------- */
typedef struct cm_wgrad_args {
    int32_t rows, m, n;
    int32_t variant;             /* 0: the library picks; 1: tiles travel through registers (2 buffers); 2: global -> LDS directly
                                    (LDS-DMA) through a 4-stage ring, three 32-row steps in flight; 3: LDS-DMA, 2 buffers of 64 rows */
    const void *a, *b;
    int64_t lda, ldb;
    float *out;
    float *workspace;
    int64_t workspace_floats;
    void *stream;
} cm_wgrad_args;

int cm_wgrad_supported(int32_t rows, int32_t m, int32_t n);
int64_t cm_wgrad_workspace_floats(int32_t rows, int32_t m, int32_t n);
int cm_wgrad_bf16(const cm_wgrad_args *args);

int64_t cm_conv_cl_bwd_workspace_floats(int32_t batch, int32_t seqlen, int32_t dim);
int cm_conv_cl_bwd(const cm_conv_cl_bwd_args *args);

/* ---------------------------------------------------------------------------------------
 * cm_conv_cl_fwd for both directions fused with both directions' x_proj GEMMs (bf16 only):
 *   y_fwd, y_bwd as cm_conv_cl_fwd (SiLU applied), and
 *   xdbl[b,t, 0:48]  = y_fwd[b,t,:] @ Wx_f^T      xdbl[b,t,48:96] = y_bwd[b,t,:] @ Wx_b^T
 * where Wx_* is the direction's x_proj.weight re-rowed to (48, dim) = [dt rows zero-padded to 16 | B rows | C rows]
 * (reference selective_scan_interface.py:186 followed by the split of :187-215), bf16, in the fragment-tiled image
 * cm_ffn_pack_weights produces.  xdbl (batch, seqlen, 96) bf16 is what cm_scan_cl_fwd's xdbl mode reads.
 * dim: multiple of 32.  Strides in elements, multiples of 4.
 * ------------------------------------------------------------------------------------- */
typedef struct cm_conv_xproj_args {
    int32_t batch, seqlen, dim, width;
    const void  *x;
    const float *weight_f, *bias_f, *weight_b, *bias_b;   /* (dim, 4) / (dim); biases may be NULL */
    const void  *wx_f, *wx_b;                             /* packed (dt_pad + 32, dim) bf16       */
    void *y_fwd, *y_bwd, *xdbl;
    int64_t x_bs, x_ts, yf_bs, yf_ts, yb_bs, yb_ts, xdbl_bs, xdbl_ts;
    void *stream;
    int32_t dt_pad;                                       /* 0 / 16: x_dbl rows [dt16 | B | C] per direction (96 columns in all);
                                                             32: [dt32 | B | C] (128 columns), the layout cm_scan_cl_fwd's xdbl
                                                             mode takes for 16 < dt_rank <= 32                                   */
    int32_t variant;                                      /* tuning: 0 = tile shape chosen from the problem size; 1 = always the
                                                             16-step tiles (the shape small launches get)                        */
} cm_conv_xproj_args;

int cm_conv_xproj(const cm_conv_xproj_args *args);

/* ---------------------------------------------------------------------------------------
 * Residual add + LayerNorm(s) over the last axis (rows x dim), fused:
 *   r   = x + alpha * y                       (y optional; x fp32 residual stream)
 *   if norm1: r1 = LN(r; g1, b1, eps1) else r1 = r
 *   if x_out: x_out = r1 (fp32)               -- new residual stream (may alias x)
 *   if norm2: out = LN(r1; g2, b2, eps2) else out = r1;  written in out_dtype if out != NULL
 * Covers the four residual/normalisation seams of ConmambaEncoderLayer.forward
 * (reference modules/Conmamba.py:638-649) and the encoder's final norm (:725).
 * ------------------------------------------------------------------------------------- */
typedef struct cm_add_ln_args {
    int64_t rows;
    int32_t dim;
    int32_t y_dtype;             /* dtype of y (CM_BF16 / CM_F32)        */
    int32_t out_dtype;           /* dtype of out                          */
    int32_t out_act;             /* activation applied to `out`: 0 none, 1 LeakyReLU(0.01) */
    const float *x;              /* (rows, dim) fp32, or NULL (treated as zeros)            */
    const void  *y;              /* (rows, dim) or NULL                   */
    float alpha;
    float eps1, eps2;
    int32_t pad2_;
    const float *g1, *b1;        /* first LN (applied to the residual) or NULL  */
    const float *g2, *b2;        /* second LN (produces `out`) or NULL          */
    float *x_out;                /* (rows, dim) fp32 or NULL                    */
    void  *out;                  /* (rows, dim) out_dtype or NULL               */
    void  *stream;
} cm_add_ln_args;

int cm_add_layernorm(const cm_add_ln_args *args);

/* ---------------------------------------------------------------------------------------
 * ConvolutionModule core, channels-last (reference modules/Conmamba.py:441-449 between the two
 * GEMMs): GLU over the channel axis -> depthwise conv over time (kernel K odd, 'same' zero padding)
 * + bias -> LayerNorm(dim) -> GELU.   in: (batch, seqlen, 2*dim)  out: (batch, seqlen, dim)
 * ------------------------------------------------------------------------------------- */
typedef struct cm_glu_dwconv_args {
    int32_t batch, seqlen, dim, ksize;
    int32_t io_dtype;
    int32_t glu_done;            /* 0: in is (batch, seqlen, 2*dim) and the GLU is applied here;
                                    1: in is (batch, seqlen, dim), already gated (cm_ln_pw_glu)      */
    const void  *in;             /* (batch, seqlen, 2*dim), contiguous               */
    const float *weight;         /* (dim, ksize) depthwise taps                       */
    const float *bias;           /* (dim) or NULL                                      */
    const float *ln_g, *ln_b;    /* (dim)                                              */
    float eps;
    int32_t variant;             /* tuning: 0 = kernel chosen from dtype / dim; 1 = always the generic 16-step kernel  */
    void *out;                   /* (batch, seqlen, dim), contiguous                   */
    void *stream;
    const float *weight_t;       /* optional (ksize, dim) copy of the taps: per-channel reads become coalesced */
    const void  *lin_w;          /* optional: the module's closing Linear(dim, dim) (reference Conmamba.py:156-158) applied to
                                    the tile before it is stored: (dim, dim) bf16 in cm_ffn_pack_weights' image; bf16 rows of
                                    dim 256 only.  out = GELU(LayerNorm(conv)) @ lin_w^T + lin_b                          */
    const float *lin_b;          /* (dim) fp32, required with lin_w                                                       */
} cm_glu_dwconv_args;

int cm_glu_dwconv_ln_gelu(const cm_glu_dwconv_args *args);

/* ---------------------------------------------------------------------------------------
 * Single-step (decode-time) updates of a Mamba mixer: the two calls of bimamba.Mamba.step (reference
 * modules/mamba/bimamba.py:320-365).  All tensors contiguous; states are fp32 and updated IN PLACE.
 *   cm_causal_conv1d_update (causal_conv1d.causal_conv1d_update, :337-343 / fallback :331-336):
 *     conv_state (batch, dim, width) <- shifted left by one with x (batch, dim) appended;
 *     out[b,c] = [silu](bias[c] + sum_k weight[c,k] * conv_state[b,c,k])
 *   cm_selective_state_update (mamba_ssm selective_state_update, :360-362 / fallback :350-358):
 *     dt' = [softplus](dt + dt_bias);  state[b,c,n] <- state * exp(dt' * A[c,n]) + dt' * B[b,n] * x[b,c];
 *     out[b,c] = (sum_n state[b,c,n] * C[b,n] + D[c] * x[b,c]) * [silu(z[b,c])]
 * ------------------------------------------------------------------------------------- */
typedef struct cm_conv_update_args {
    int32_t batch, dim, width;
    int32_t io_dtype;            /* x, out                                          */
    int32_t silu;
    int32_t pad_;
    const void  *x;              /* (batch, dim)                                    */
    float       *conv_state;     /* (batch, dim, width) fp32, in place              */
    const float *weight;         /* (dim, width)                                    */
    const float *bias;           /* (dim) or NULL                                   */
    void        *out;            /* (batch, dim)                                    */
    void *stream;
} cm_conv_update_args;

typedef struct cm_state_update_args {
    int32_t batch, dim, dstate;
    int32_t io_dtype;            /* x, dt, B, C, z, out                             */
    int32_t dt_softplus;
    int32_t pad_;
    float       *state;          /* (batch, dim, dstate) fp32, in place             */
    const void  *x, *dt;         /* (batch, dim)                                    */
    const float *A;              /* (dim, dstate)                                   */
    const void  *B, *C;          /* (batch, dstate)                                 */
    const float *D;              /* (dim) or NULL                                   */
    const void  *z;              /* (batch, dim) or NULL                            */
    const float *dt_bias;        /* (dim) or NULL                                   */
    void        *out;            /* (batch, dim)                                    */
    void *stream;
} cm_state_update_args;

int cm_causal_conv1d_update(const cm_conv_update_args *args);
int cm_selective_state_update(const cm_state_update_args *args);

/* ---------------------------------------------------------------------------------------
 * Depthwise Conv1d over time, forward and backward, on the module API's (batch, dim, seqlen) time-contiguous layout:
 * the ConvolutionModule's depthwise stage (reference modules/Conmamba.py:271-284, called at :443; nn.Conv1d with
 * groups = dim, stride 1, zero padding).  Output length = seqlen.
 *   y[b,c,t]  = bias[c] + sum_k weight[c,k] * x[b,c,t + k - pad_left]        (x = 0 outside [0, seqlen))
 *   dx[b,c,s] = sum_k weight[c,k] * dy[b,c,s - k + pad_left]
 *   dweight[c,k] += sum_{b,t} dy[b,c,t] * x[b,c,t + k - pad_left];   dbias[c] += sum_{b,t} dy[b,c,t]
 * pad_left = ksize/2 for 'same' padding, ksize-1 for the causal variant (padding then chomp, :445-446).
 * ksize <= 32.  dweight / dbias are fp32 and accumulated into (caller zero-initialises); deterministic (no atomics).
 * ------------------------------------------------------------------------------------- */
typedef struct cm_dwconv1d_args {
    int32_t batch, dim, seqlen, ksize, pad_left;
    int32_t io_dtype;            /* CM_BF16 or CM_F32: x, y, dy, dx                     */
    const void  *x;
    const float *weight;         /* (dim, ksize)                                        */
    const float *bias;           /* (dim) or NULL                                       */
    void        *y;              /* forward only                                        */
    const void  *dy;             /* backward only                                       */
    void        *dx;
    float       *dweight;        /* (dim, ksize)                                        */
    float       *dbias;          /* (dim) or NULL                                       */
    int64_t x_bs, x_ds, y_bs, y_ds, dy_bs, dy_ds, dx_bs, dx_ds;   /* batch / channel strides in elements */
    void *stream;
} cm_dwconv1d_args;

int cm_dwconv1d_fwd(const cm_dwconv1d_args *args);
int cm_dwconv1d_bwd(const cm_dwconv1d_args *args);

/* The same operator on CHANNELS-LAST rows (batch, seqlen, dim), channel axis contiguous, strides in elements: with it
 * the ConvolutionModule stays in the (batch, time, channel) layout end to end (pointwise conv = GEMM on rows, GLU over
 * the last axis), without transposing copies.  Same formulas as cm_dwconv1d_*.  The backward writes per-workgroup
 * partial tap gradients to `partial` (cm_dwconv_cl_workspace_floats(batch, seqlen, dim) fp32, caller-owned) and sums
 * them in a fixed order (deterministic); dweight / dbias are accumulated into. */
typedef struct cm_dwconv_cl_args {
    int32_t batch, dim, seqlen, ksize, pad_left;
    int32_t io_dtype;
    const void  *x;
    const float *weight;         /* (dim, ksize) */
    const float *bias;           /* (dim) or NULL */
    void        *y;              /* forward only  */
    const void  *dy;             /* backward only */
    void        *dx;
    float       *dweight;
    float       *dbias;
    float       *partial;        /* backward workspace */
    int64_t x_bs, x_ts, y_bs, y_ts, dy_bs, dy_ts, dx_bs, dx_ts;
    void *stream;
    int32_t overwrite;           /* backward: 1 = dweight / dbias are written, not accumulated into */
    int32_t reserved0;
} cm_dwconv_cl_args;

int64_t cm_dwconv_cl_workspace_floats(int32_t batch, int32_t seqlen, int32_t dim);
int cm_dwconv_cl_fwd(const cm_dwconv_cl_args *args);
int cm_dwconv_cl_bwd(const cm_dwconv_cl_args *args);

/* ---------------------------------------------------------------------------------------
 * LayerNorm over the last axis with saved statistics, forward and backward (training path).  Replaces torch's
 * LayerNorm behind the reference's normalisations: modules/Conmamba.py:262, :287 (ConvolutionModule), :597-620
 * (feed-forward pre-norms, norm1, norm2), :687 (final norm) -- under autocast torch runs them in fp32 whatever the
 * input type, so x may be bf16 while y is fp32.
 *   fwd: y = (x - mean) * rstd * gamma + beta over rows of `dim` (multiple of 4, <= 4096: the CNN front end normalises
 *        over (frequency, channel) = 2560), contiguous; mean / rstd
 *        (rows) fp32 are written when given (both or neither).
 *   bwd: dx (x_dtype, may be NULL) and dgamma / dbeta (dim, OVERWRITTEN) from dy (y_dtype), x, mean, rstd, gamma;
 *        workspace: cm_layernorm_bwd_workspace_floats(rows, dim) fp32, caller-owned; fixed summation order.
 * ------------------------------------------------------------------------------------- */
typedef struct cm_layernorm_args {
    int64_t rows;
    int32_t dim;
    int32_t x_dtype, y_dtype;    /* CM_F32 or CM_BF16 */
    float   eps;
    const void  *x;
    const float *gamma, *beta;
    void  *y;                    /* forward  */
    float *mean, *rstd;          /* forward: written (optional); backward: read */
    const void *dy;              /* backward */
    void  *dx;
    float *dgamma, *dbeta;
    float *workspace;
    void  *stream;
    const float *dres;           /* backward, optional: (rows, dim) fp32 added to dx (the residual branch's gradient: the
                                    pre-norm block's dx = dres + LayerNorm'(dy) in one pass); fp32 x and dim <= 1024 only */
    /* optional epilogue y = act(LN(x)) * chan_mask[row / mask_rows][col % mask_c] -- the front end's Conv2d block tail (LayerNorm over
       (freq, channel) -> LeakyReLU -> Dropout2d, one keep / (1 - p) factor per (sample, channel)); the backward recomputes LN's
       output for the activation's derivative and needs beta then */
    int32_t act;                 /* 0 none, 1 LeakyReLU(act_slope)                                                          */
    float   act_slope;
    const float *chan_mask;      /* (rows / mask_rows, mask_c) fp32 or NULL; mask_c a multiple of 4 dividing dim            */
    int32_t mask_rows, mask_c;
} cm_layernorm_args;

int64_t cm_layernorm_bwd_workspace_floats(int64_t rows, int32_t dim);
int cm_layernorm_fwd(const cm_layernorm_args *args);
int cm_layernorm_bwd(const cm_layernorm_args *args);

/* ---------------------------------------------------------------------------------------
 * Mixer -> convolution-module seam in one kernel (bf16 GEMM operands, d_model 256):
 *   x_out = x + alpha * y;  h = LayerNorm(x_out; ln_g, ln_b, eps);  pw = h @ W^T + bias  (W: (2*dim, dim));
 *   out = pw[:, :dim] * sigmoid(pw[:, dim:])                  (reference modules/Conmamba.py:639-640, 441-443)
 * x (rows, 256) fp32; y (rows, 256) bf16 or NULL; w: the pointwise Conv1d weight (512, 256) bf16 packed with
 * cm_ffn_pack_weights; bias (512) fp32; x_out (rows, 256) fp32 (may alias x) or NULL; out (rows, 256) bf16, which
 * cm_glu_dwconv_ln_gelu takes with glu_done = 1.
 * ------------------------------------------------------------------------------------- */
typedef struct cm_ln_pw_glu_args {
    int32_t rows, dim;
    const float *x;
    const void  *y;
    const float *ln_g, *ln_b;
    const void  *w;
    const float *bias;
    float *x_out;
    void  *out;
    float alpha, eps;
    void *stream;
} cm_ln_pw_glu_args;

int cm_ln_pw_glu(const cm_ln_pw_glu_args *args);

/* ---------------------------------------------------------------------------------------
 * First block of the CNN front end (speechbrain ConvolutionFrontEnd as configured at reference
 * hparams/CTC/conmamba_large.yaml:187-194): Conv2d(1 -> C, 3x3, stride 2 in time and frequency, reflect
 * 'same' padding) -> LayerNorm over (freq, channel) -> LeakyReLU(0.01), fused, channels-last.
 *   in : feats (batch, T, F) fp32
 *   out: (batch, T1 + 2*pad_out, F1 + 2*pad_out, C) io_dtype, T1 = ceil(T/2), F1 = ceil(F/2); with pad_out = 1 the
 *        reflect border the NEXT 3x3 stride-2 block needs is written as well, so that block runs unpadded.
 * ------------------------------------------------------------------------------------- */
typedef struct cm_cnn_block1_args {
    int32_t batch, T, F, C;
    int32_t io_dtype;            /* output dtype                                           */
    int32_t pad_out;             /* 0 or 1                                                 */
    const float *feats;          /* (batch, T, F)                                          */
    const float *weight;         /* (C, 1, 3, 3)                                           */
    const float *bias;           /* (C)                                                    */
    const float *ln_g, *ln_b;    /* (F1, C)                                                */
    float eps, slope;
    void *out;
    void *stream;
} cm_cnn_block1_args;

int cm_cnn_block1(const cm_cnn_block1_args *args);

/* ---------------------------------------------------------------------------------------
 * Second block of the CNN front end (same reference lines as cm_cnn_block1): Conv2d(64 -> 32, 3x3, stride 2,
 * no padding -- the reflect border is already in the input) -> LayerNorm over (freq, channel) -> LeakyReLU, bf16.
 *   in    : (batch, T_in, F_in, 64) bf16 channels-last  (cm_cnn_block1's output with pad_out = 1)
 *   weight: (32, 3, 3, 64) bf16 -- output channel, tap row, tap column, input channel (a torch Conv2d weight in
 *           channels_last memory format)
 *   out   : (batch, T2, F2 * 32) bf16,  T2 = (T_in - 3) / 2 + 1,  F2 = (F_in - 3) / 2 + 1
 * ------------------------------------------------------------------------------------- */
typedef struct cm_cnn_block2_args {
    int32_t batch, T_in, F_in, C_in, C_out;
    int32_t pad_;
    const void  *in;
    const void  *weight;
    const float *bias;           /* (32) or NULL                                           */
    const float *ln_g, *ln_b;    /* (F2, 32)                                               */
    float eps, slope;
    void *out;
    void *stream;
} cm_cnn_block2_args;

int cm_cnn_block2(const cm_cnn_block2_args *args);

/* ---------------------------------------------------------------------------------------
 * Both CNN front-end blocks in one kernel: cm_cnn_block1 (pad_out = 1) followed by cm_cnn_block2, without the
 * (batch, T/2 + 2, F/2 + 2, 64) intermediate ever reaching memory.  F = 80 bins, channels (64, 32).
 *   feats: (batch, T, 80) fp32      out: (batch, T2, 20 * 32) bf16,  T1 = ceil(T/2),  T2 = (T1 - 1) / 2 + 1
 *   w1 (64, 1, 3, 3) fp32, b1 (64), ln1_g / ln1_b (40, 64);  w2 (32, 3, 3, 64) bf16 (OHWI), b2 (32) fp32, ln2 (20, 32)
 * ------------------------------------------------------------------------------------- */
typedef struct cm_cnn_front_args {
    int32_t batch, T, F, C1, C2;
    int32_t pad_;
    const float *feats;
    const float *w1, *b1, *ln1_g, *ln1_b;
    const void  *w2;
    const float *b2, *ln2_g, *ln2_b;
    float eps1, eps2, slope;
    int32_t pad2_;
    void *out;
    void *stream;
} cm_cnn_front_args;

int cm_cnn_front(const cm_cnn_front_args *args);

/* ---------------------------------------------------------------------------------------
 * bf16 MFMA GEMM with fused epilogues for the ConMamba layer's projections:
 *     acc[m, n] = sum_k A[m, k] * W[n, k]            (A: activations, W: nn.Linear weight layout)
 *   epilogue 0:  out = acc + bias                                          -> bf16   (in_proj, pointwise conv)
 *   epilogue 1:  out = gelu(acc + bias)   (erf form, nn.GELU default)      -> bf16   (FFN up-projection)
 *   epilogue 2:  r = x + alpha * (acc + bias);  r1 = LN1(r) if g1 else r;  x = r1 (fp32, in place);
 *                out = LN2(r1) if g2 (bf16, optional)                                (FFN down-projection, out_proj and
 *                the conv-module Linear: the residual/LayerNorm seams of reference modules/Conmamba.py:638-649)
 * Requirements: K % 64 == 0, N % 256 == 0 (epilogue 2: N == 256 so that one workgroup owns whole rows),
 * A/W/out 16-byte aligned with leading dimensions multiples of 8 elements.
 * ------------------------------------------------------------------------------------- */
typedef struct cm_gemm_args {
    int32_t M, N, K;
    int32_t epilogue;
    const void  *A;  int64_t lda;      /* (M, K) bf16                                       */
    const void  *W;  int64_t ldw;      /* (N, K) bf16                                       */
    const float *bias;                 /* (N) fp32 or NULL                                  */
    void        *out; int64_t ldo;     /* (M, N) bf16; optional for epilogue 2              */
    float       *x;                    /* epilogue 2: (M, N) fp32 residual stream, in place */
    float alpha, eps1, eps2;
    int32_t pad_;
    const float *g1, *b1, *g2, *b2;    /* epilogue 2 LayerNorm parameters (N) or NULL       */
    void *stream;
} cm_gemm_args;

int cm_gemm_bf16(const cm_gemm_args *args);

/* ---------------------------------------------------------------------------------------
 * Position-wise feed-forward module of a ConMamba layer, one kernel (reference modules/Conmamba.py:631-650 around
 * speechbrain's PositionalwiseFeedForward: LayerNorm -> Linear(256, hidden) -> GELU -> Linear(hidden, 256), scaled
 * residual, then the layer's next LayerNorm).  d_model is 256; hidden a multiple of 256.
 *     xin   = x + add_scale * addend                (addend optional: the previous module's bf16 output)
 *     r     = xin + alpha * (W2 gelu(W1 LN_pre(xin) + b1) + b2)
 *     r     = LN_1(r)          if n1_g             (the layer's closing norm2)
 *     x_out = r                if x_out            (fp32 residual stream; may alias x)
 *     h_out = LN_2(r)          if h_out            (LN_2 skipped when n2_g is NULL), dtype h_dtype (bf16 or f32)
 * The hidden activations stay in LDS; GEMM operands are bf16 with fp32 accumulation.
 * w1 / w2 are passed in the PACKED layout produced by cm_ffn_pack_weights (once per weight update): the (R, K) matrix
 * cut into 16-row x 32-column tiles, each stored as the 1 KB register image of an MFMA operand fragment -- element
 * (r, k) of tile (r/16, k/32) sits at 16-bit index  tile*512 + ((k%32)/8*16 + r%16)*8 + k%8,  tile = (r/16)*(K/32) + k/32.
 * ------------------------------------------------------------------------------------- */
int cm_ffn_pack_weights(const void *w /* (rows, cols) bf16 row-major */, int32_t rows, int32_t cols,
                        void *out /* rows*cols bf16 */, void *stream);

typedef struct cm_ffn_args {
    int32_t rows, dim, hidden;
    int32_t h_dtype;                    /* CM_BF16 or CM_F32                                 */
    const float *x;                     /* (rows, 256) fp32                                  */
    const void  *addend;                /* (rows, 256) bf16 or NULL                          */
    const float *pre_g, *pre_b;         /* (256) LayerNorm in front of the first Linear      */
    const void  *w1;                    /* (hidden, 256) bf16, packed                        */
    const float *b1;                    /* (hidden) fp32                                     */
    const void  *w2;                    /* (256, hidden) bf16, packed                        */
    const float *b2;                    /* (256) fp32                                        */
    const float *n1_g, *n1_b;           /* optional LayerNorm applied to the stream          */
    const float *n2_g, *n2_b;           /* optional LayerNorm producing h_out                */
    float       *x_out;                 /* (rows, 256) fp32 or NULL                          */
    void        *h_out;                 /* (rows, 256) h_dtype or NULL                       */
    float add_scale, alpha, pre_eps, n1_eps, n2_eps;
    int32_t proj_dim;                   /* rows of proj_w (a multiple of 256, <= 4096); 0 without a projection            */
    void *stream;
    /* optional: the Linear that consumes h, applied to the tile in the same kernel -- the BiMamba in_proj behind the
       layer's first feed-forward module (reference bimamba.py:192-200):  proj_out = LN2(r) @ proj_w^T (+ proj_b), bf16;
       h_out must be NULL then (h is not stored) */
    const void  *proj_w;                /* (proj_dim, 256) bf16, cm_ffn_pack_weights' image                                */
    const float *proj_b;                /* (proj_dim) fp32 or NULL                                                          */
    void        *proj_out;              /* (rows, proj_dim) bf16                                                            */
    /* optional, the TRAINING forward of the module (reference modules/Conmamba.py:597-617 with its two Dropouts live):
         x_out = x + alpha * drop_p2( W2 drop_p1( GELU( bf16(W1 LN(x) + b1) ) ) + b2 )
       with what the backward needs stored on the way: pre_out = bf16(W1 LN(x) + b1) and xn_out = bf16(LN(x)).  Dropout
       decisions are cm_dropout.h's function of (seed, element index in the (rows, hidden) / (rows, 256) tensor): no mask is
       stored, cm_bias_act_dropout_bwd re-derives it.  Requires x_out, no addend / n1 / n2 / projection / h_out. */
    void        *pre_out;               /* (rows, hidden) bf16 or NULL                                                      */
    void        *xn_out;                /* (rows, 256) bf16 or NULL                                                         */
    float p1, p2;                       /* dropout probabilities behind the activation / behind the second Linear           */
    uint64_t seed1, seed2;
    float       *stats_out;             /* (2, rows) fp32 or NULL: mean, 1/std of LN(x)'s rows, as cm_layernorm_bwd reads them */
    int32_t layout;                     /* 0: w1 / w2 / proj_w in cm_ffn_pack_weights' image (16-row x 32-column tiles, v_mfma_f32_16x16x32_bf16);
                                           1: in cm_ffn_pack_weights32's (32 x 16 tiles, v_mfma_f32_32x32x16_bf16; inference forward only) */
    int32_t tokens;                     /* layout 1: tokens per workgroup, 0 / 64 or 32 (32: for launches of fewer than ~400 64-token
                                           workgroups, which leave CUs idle) */
    const uint64_t *seed_epoch;         /* optional device word added into seed1 / seed2 (cm_ffn_elem_args.seed_epoch)       */
} cm_ffn_args;

int cm_ffn_fused(const cm_ffn_args *args);
/* row-major (rows, cols) bf16 -> the 32-row x 16-column fragment-tile image of cm_ffn_args.layout = 1 (rows % 32 == 0, cols % 16 == 0) */
int cm_ffn_pack_weights32(const void *w, int32_t rows, int32_t cols, void *out, void *stream);

/* The data-gradient chain of the same module's backward (what autograd does over modules/Conmamba.py:597-617 with two Dropout, two
 * addmm and one GELU backward), for the forward cm_ffn_fused's training variant ran (same dropout seeds):
 *   da2 = alpha * dout * keep2 / (1 - p2);  dg = da2 @ W2;  da1 = dg * keep1 / (1 - p1) * GELU'(pre);  act = drop1(GELU(pre));
 *   dh = da1 @ W1;  db2 = column sums of da2;  db1 = column sums of da1 (fixed order: deterministic)
 * w2t = W2^T (hidden, 256), w1t = W1^T (256, hidden): bf16 in cm_ffn_pack_weights' image.  da2 / da1 / act are stored for the weight
 * gradients (cm_wgrad_bf16), dh (bf16) for the LayerNorm backward in front of the module.  d_model 256, hidden a multiple of 256. */
typedef struct cm_ffn_bwd_args {
    int32_t rows, dim, hidden, reserved0;
    const float *dout;                  /* (rows, 256) fp32: gradient of the module's output (the residual stream)          */
    const void  *w2t, *w1t;
    const void  *pre;                   /* (rows, hidden) bf16: cm_ffn_args.pre_out of the forward                          */
    void *da2;                          /* (rows, 256) bf16 out                                                             */
    void *da1, *act;                    /* (rows, hidden) bf16 out                                                          */
    void *dh;                           /* (rows, 256) bf16 out                                                             */
    float *db1, *db2;                   /* (hidden), (256) fp32 out (written); db2 = db1 + hidden: ONE (hidden + 256) buffer     */
    float alpha, p1, p2;
    float reserved1;
    uint64_t seed1, seed2;
    float *workspace;                   /* cm_ffn_bwd_workspace_floats(rows, hidden) floats                                  */
    int64_t workspace_floats;
    void *stream;
    float *db1_part, *db2_part;         /* internal (set by the library)                                                    */
    const uint64_t *seed_epoch;         /* the forward's (cm_ffn_args.seed_epoch): same device word, same value at run time */
} cm_ffn_bwd_args;

int64_t cm_ffn_bwd_workspace_floats(int32_t rows, int32_t hidden);
int cm_ffn_bwd_fused(const cm_ffn_bwd_args *args);

/* ---------------------------------------------------------------------------------------
 * Fbank back end (speechbrain Fbank semantics as used at reference train_CTC.py:285 and configured at
 * hparams/CTC/conmamba_large.yaml:322-326): STFT (re, im) -> power -> triangular mel filterbank -> 10*log10 with
 * floor `amin`, plus the per-utterance maximum needed by the top_db clamp.
 *   spec : (batch, n_freq, frames, 2) fp32  -- torch.stft(..., return_complex=True) viewed as real
 *   fbank: (n_freq, n_mels) fp32
 *   db   : (batch, frames, n_mels) fp32  (out)      umax : (batch) fp32, initialised to -inf by the caller (out)
 * cm_fbank_finish then applies  db = max(db, umax[b] - top_db)  and optionally the global normalisation
 * (db - mean[m]) / std[m]  (speechbrain InputNormalization, train_CTC.py:287) in place.
 * ------------------------------------------------------------------------------------- */
typedef struct cm_fbank_args {
    int32_t batch, n_freq, frames, n_mels;
    const float *spec;
    const float *fbank;
    float *db;
    float *umax;
    float amin, top_db;
    const float *mean, *std;     /* (n_mels) or NULL: used by cm_fbank_finish            */
    const int32_t *band_lo, *band_hi;   /* optional (n_mels): filter m is non-zero only on bins [lo, hi) */
    const int32_t *band_off;            /* optional (n_mels + 1): offsets into band_w                    */
    const float   *band_w;              /* optional: filter m's weights for bins lo..hi-1, packed (sum of band widths
                                           <= 4096 floats); staged in LDS instead of reading the dense matrix */
    int64_t spec_bs, spec_fs, spec_ts;  /* strides of spec in complex elements (batch, frequency, frame); all 0 =
                                           contiguous (batch, n_freq, frames).  torch.stft hands out a transposed view
                                           of a (batch, frames, n_freq) buffer: passing its strides avoids a 260 MB copy */
    void *stream;
    float *umax_part;            /* optional (batch, ceil(frames / 16)) scratch: cm_fbank_mel_db stores per-tile maxima
                                    there instead of issuing atomics on umax, and cm_fbank_finish reduces them into
                                    umax; pass the same args to both calls */
    /* cm_fbank_wav only: the STFT is computed in-kernel from the waveform (spec / spec_* are ignored) */
    const float *wav;            /* (batch, samples) fp32                                                        */
    const float *window;         /* (n_fft) fp32: the analysis window zero-padded (centred) to n_fft             */
    const float *twiddle;        /* (n_fft, 2) fp32: exp(-2 pi i k / n_fft), k = 0 .. n_fft - 1                  */
    int64_t wav_bs;              /* batch stride of wav in elements (even)                                        */
    int32_t samples, hop, n_fft; /* frames must equal 1 + samples / hop (center=True, zero padding); n_fft 512    */
    int32_t pad3_;
} cm_fbank_args;

int cm_fbank_mel_db(const cm_fbank_args *args);
int cm_fbank_finish(const cm_fbank_args *args);
/* waveform -> log-mel (before top_db / normalisation: follow with cm_fbank_finish on the same args).  Requires
 * umax_part and the packed band tables (band_lo, band_hi, band_off, band_w).  In-LDS radix-4 real FFT, n_fft = 512. */
int cm_fbank_wav(const cm_fbank_args *args);

/* ---------------------------------------------------------------------------------------
 * SpecAugment masking (speechbrain SpectrogramDrop, reference hparams/CTC/conmamba_large.yaml:273-320 and
 * train_CTC.py:291): feats[b, t, f] = fill for every (b, t, f) covered by one of the utterance's masks along
 * `dim` (1 = time, 2 = frequency).  Mask starts/lengths are drawn by the host RNG: int32 (batch, n_masks).
 * ------------------------------------------------------------------------------------- */
typedef struct cm_spec_drop_args {
    int32_t batch, frames, n_mels, n_masks;
    int32_t dim;                 /* 1 = time masks, 2 = frequency masks                   */
    int32_t pad_;
    float *feats;                /* (batch, frames, n_mels) fp32, in place                */
    const int32_t *start;        /* (batch, n_masks)                                      */
    const int32_t *length;       /* (batch, n_masks)                                      */
    const float *fill;           /* device scalar: the replacement value (tensor mean)    */
    void *stream;
} cm_spec_drop_args;

int cm_spec_drop(const cm_spec_drop_args *args);

#ifdef __cplusplus
}
#endif
#endif /* CONMAMBA_HIP_H */
