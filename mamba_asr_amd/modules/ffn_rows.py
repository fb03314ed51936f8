"""A feed-forward module of the ConMamba layer as ONE autograd node on (batch * time, features) rows:

    out = x + alpha * Dropout_p2( W2 Dropout_p1( GELU( W1 LayerNorm(x) + b1 ) ) + b2 )

(reference modules/Conmamba.py:597-617 ffn_module1 / ffn_module2 = Sequential(LayerNorm, PositionalwiseFeedForward, Dropout) and
the `x + 0.5 * ffn(x)` of :638, :647; MambaDecoderLayer's norm3 / pos_ffn / dropout3 of :946-949 with alpha 1).  The reference
leaves it to torch autograd: 8 element-wise launches forward and about 10 backward around the two GEMMs.  Here: native
LayerNorm forward / backward (csrc/layernorm_train.hip), bias + GELU + dropout and bias + dropout + scaled residual add as one
kernel each (csrc/ffn_train.hip), their gradients with the bias-gradient column sums inside (deterministic), weight gradients
as per-utterance batched GEMMs folded by cm_sum_leading.  The GEMMs themselves stay library calls (MFMA only for the
projections, BASELINE.json north_star).

bf16, d_model 256 (ConMamba-large): the FORWARD is cm_ffn_fused's training variant -- one kernel, the hidden activations stay in
LDS; it stores the pre-activation (bf16), the normalised input and the LayerNorm statistics for the backward and no dropout
mask (csrc/cm_dropout.h: the decisions are a function of (seed, element index), re-derived in the backward, which also
recomputes the activation the second GEMM saw for its weight gradient).  CM_FFN_FUSED_TRAIN=0 keeps the kernel-per-stage forward."""
from __future__ import annotations

import os

import torch
import torch.nn as nn
from torch.amp import custom_bwd, custom_fwd

from .. import ops

# CM_FFN_ROWS=0: the module tree runs as written (torch element-wise kernels between the GEMMs)
ENABLED = os.environ.get("CM_FFN_ROWS", "1") == "1"
FUSED_TRAIN = os.environ.get("CM_FFN_FUSED_TRAIN", "1") == "1"
# the backward's data-gradient chain as one kernel (cm_ffn_bwd_fused); CM_FFN_FUSED_BWD=0 = dgrad GEMMs + element-wise kernels
FUSED_BWD = os.environ.get("CM_FFN_FUSED_BWD", "1") == "1"


def _fused_ok(cdt, D, F_, rows):
    return FUSED_TRAIN and cdt == torch.bfloat16 and ops.ffn_supported(D, F_, cdt) and F_ <= 2048 and rows * F_ * 2 < 2 ** 32


def supported(x, ln, lin1, act, lin2) -> bool:
    if not (ENABLED and x.is_cuda and x.dim() == 3 and x.dtype == torch.float32 and isinstance(act, nn.GELU)
            and getattr(act, "approximate", "none") == "none"):
        return False
    cdt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
    d, f = lin1.weight.shape[1], lin1.weight.shape[0]
    return (cdt in (torch.float32, torch.bfloat16) and d % 8 == 0 and f % 8 == 0 and f <= 2048 and d <= 1024 and lin1.bias is not None
            and lin2.bias is not None and ln.weight is not None and ln.bias is not None and len(ln.normalized_shape) == 1)


class FfnRowsFn(torch.autograd.Function):
    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, x, lnw, lnb, w1, b1, w2, b2, eps, p1, p2, alpha):
        cdt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
        B, T, D = x.shape
        x2 = x.detach().reshape(B * T, D)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        if _fused_ok(cdt, D, w1.shape[0], B * T):
            s1 = ops.draw_seed() if p1 > 0.0 else 0
            s2 = ops.draw_seed() if p2 > 0.0 else 0
            out = torch.empty_like(x2)
            f32 = lambda t: t.detach().float().contiguous()
            _, (pre, h, stats) = ops.ffn_fused(x2, (f32(lnw), f32(lnb), eps), ops.pack_cached(w1), f32(b1), ops.pack_cached(w2), f32(b2),
                                               alpha=alpha, x_out=out, train=(p1, p2, s1, s2))
            ctx.save_for_backward(x2, stats, h, pre, lnw, w1, w2)
            ctx.cfg = (eps, p1, p2, alpha, cdt, (B, T, D))
            ctx.seeds = (s1, s2)
            return out.view(B, T, D)
        ctx.seeds = None
        h, x2s, stats = ops.layernorm_fwd(x2, lnw, lnb, eps, cdt)
        w1c, w2c = ops.cast_cached(w1, cdt), ops.cast_cached(w2, cdt)
        a1 = torch.mm(h, w1c.t())
        g, m1 = ops.bias_act_dropout_fwd(a1, b1, act=1, p=p1)
        a2 = torch.mm(g, w2c.t())
        out, m2 = ops.bias_act_dropout_fwd(a2, b2, act=0, p=p2, res=x2, alpha=alpha)
        ctx.save_for_backward(x2s, stats, h, a1, g, m1, m2, lnw, w1, b1, w2)
        ctx.cfg = (eps, p1, p2, alpha, cdt, (B, T, D))
        return out.view(B, T, D)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, dout):
        if ctx.seeds is not None:
            return FfnRowsFn._backward_fused(ctx, dout)
        x2s, stats, h, a1, g, m1, m2, lnw, w1, b1, w2 = ctx.saved_tensors
        eps, p1, p2, alpha, cdt, (B, T, D) = ctx.cfg
        F_ = a1.shape[1]
        dout2 = dout.reshape(B * T, D)
        dout2 = dout2 if dout2.is_contiguous() else dout2.contiguous()
        if dout2.dtype != torch.float32:
            dout2 = dout2.float()
        w1c, w2c = ops.cast_cached(w1, cdt), ops.cast_cached(w2, cdt)
        da2, db2 = ops.bias_act_dropout_bwd(dout2, m2, p2, act=0, alpha=alpha, out_dtype=cdt)
        dg = torch.mm(da2, w2c)
        dw2 = ops.wgrad(da2.view(B * T, D), g.view(B * T, F_), nbatch=B)
        da1, db1 = ops.bias_act_dropout_bwd(dg, m1, p1, a=a1, bias=b1, act=1)
        dh = torch.mm(da1, w1c)
        dw1 = ops.wgrad(da1.view(B * T, F_), h.view(B * T, D), nbatch=B)
        dx, dlnw, dlnb = ops.layernorm_bwd(dh, x2s, stats, lnw, eps, dres=dout2)       # dx = dout + LayerNorm'(dh) in one pass
        dx = dx.view(B, T, D)
        return dx, dlnw, dlnb, dw1, db1, dw2, db2, None, None, None, None


def _backward_fused(ctx, dout):
    x2, stats, h, pre, lnw, w1, w2 = ctx.saved_tensors
    eps, p1, p2, alpha, cdt, (B, T, D) = ctx.cfg
    s1, s2 = ctx.seeds
    F_ = pre.shape[1]
    dout2 = dout.reshape(B * T, D)
    dout2 = dout2 if dout2.is_contiguous() else dout2.contiguous()
    if dout2.dtype != torch.float32:
        dout2 = dout2.float()
    if FUSED_BWD:
        # dout -> da2 -> dg -> da1 (+ the recomputed activation) -> dh and both bias gradients: one kernel, the hidden tile stays in LDS
        da2, da1, g, dh, db1, db2 = ops.ffn_bwd_fused(dout2, ops.pack_cached_t(w2), ops.pack_cached_t(w1), pre, alpha, p1, p2, s1, s2)
    else:
        w1c, w2c = ops.cast_cached(w1, cdt), ops.cast_cached(w2, cdt)
        da2, db2 = ops.bias_act_dropout_bwd(dout2, None, p2, act=0, alpha=alpha, out_dtype=cdt, seed=s2 if p2 > 0.0 else None)
        dg = torch.mm(da2, w2c)
        # pre already holds the bias; one pass gives the first GEMM's output gradient AND the activation the forward's second GEMM saw
        da1, db1, g = ops.bias_act_dropout_bwd(dg, None, p1, a=pre, act=1, seed=s1 if p1 > 0.0 else None, want_act=True)
        dh = torch.mm(da1, w1c)
    dw2 = ops.wgrad(da2.view(B * T, D), g.view(B * T, F_), nbatch=B)
    dw1 = ops.wgrad(da1.view(B * T, F_), h.view(B * T, D), nbatch=B)
    dx, dlnw, dlnb = ops.layernorm_bwd(dh, x2, stats, lnw, eps, dres=dout2)
    return dx.view(B, T, D), dlnw, dlnb, dw1, db1, dw2, db2, None, None, None, None


FfnRowsFn._backward_fused = staticmethod(_backward_fused)


def ffn_rows(x, ln, lin1, drop1, lin2, drop2, alpha):
    """x (B, T, D) fp32 residual stream -> x + alpha * drop2(lin2(drop1(gelu(lin1(ln(x))))))."""
    p1 = float(drop1.p) if (drop1 is not None and drop1.training) else 0.0
    p2 = float(drop2.p) if (drop2 is not None and drop2.training) else 0.0
    return FfnRowsFn.apply(x, ln.weight, ln.bias, lin1.weight, lin1.bias, lin2.weight, lin2.bias, ln.eps, p1, p2, float(alpha))
