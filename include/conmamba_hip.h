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

#define CM_ABI_VERSION 1

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
/* Tuning override for the scan kernels' lane split (lanes per channel): 1, 2, 4, 8 or 16;
 * 0 restores the automatic choice (also settable with the CM_SCAN_SPLIT environment variable).
 * Returns the previous value.  Process-wide; meant for tests and benchmarks. */
int         cm_scan_set_split(int lanes_per_channel);

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
} cm_scan_bwd_args;

int cm_selective_scan_bwd(const cm_scan_bwd_args *args);

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
} cm_conv_args;

int cm_causal_conv1d_fwd(const cm_conv_args *args);
int cm_causal_conv1d_bwd(const cm_conv_args *args);

#ifdef __cplusplus
}
#endif
#endif /* CONMAMBA_HIP_H */
