"""The convolution module of a ConMamba layer as ONE autograd node on (batch * time, features) rows:

    out = x + Dropout( Linear( GELU( LayerNorm( DepthwiseConv_k( GLU( PointwiseConv( LayerNorm(x) ) ) ) ) ) ) )

(reference modules/Conmamba.py:182-454, non-chunked path :439-449, and the `x + convolution_module(x)` of :645).  Native
LayerNorms (layernorm_train.hip), bias + GLU and GELU and bias + dropout + residual as one kernel each and direction
(ffn_train.hip), the depthwise conv and its gradients on channels-last rows (dwconv_cl.hip), weight gradients as per-utterance
batched GEMMs folded by cm_sum_leading; the pointwise conv (kernel 1) and the closing Linear are library GEMMs on rows.  Every
reduction is deterministic."""
from __future__ import annotations

import os

import torch
import torch.nn as nn
from torch.amp import custom_bwd, custom_fwd

from .. import ops

ENABLED = os.environ.get("CM_CONVMOD_ROWS", "1") == "1"


def supported(cm, x) -> bool:
    if not (ENABLED and x.is_cuda and x.dim() == 3 and x.dtype == torch.float32):
        return False
    act = cm.after_conv[1]
    cdt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
    d = x.shape[-1]
    pw, lin = cm.bottleneck[0], cm.after_conv[2]
    return (isinstance(act, nn.GELU) and getattr(act, "approximate", "none") == "none" and cdt in (torch.float32, torch.bfloat16)
            and cm.dilation == 1 and cm.kernel_size <= 32 and d % 8 == 0 and d <= 1024 and pw.bias is not None and lin.bias is not None
            and cm.conv.bias is not None)


class ConvModuleRowsFn(torch.autograd.Function):
    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, x, ln1w, ln1b, pww, pwb, cw, cb, ln2w, ln2b, lw, lb, eps1, eps2, pad_left, p):
        cdt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
        B, T, D = x.shape
        x2 = x.detach().reshape(B * T, D)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        h, x2s, st1 = ops.layernorm_fwd(x2, ln1w, ln1b, eps1, cdt)
        wpw = ops.cast_cached(pww, cdt).view(2 * D, D)
        a_pw = torch.mm(h, wpw.t())
        gl = ops.bias_glu_fwd(a_pw, pwb)
        cv = ops.dwconv_cl_fwd(gl.view(B, T, D), cw, cb, pad_left)
        y2, cvs, st2 = ops.layernorm_fwd(cv.view(B * T, D), ln2w, ln2b, eps2, cdt)
        g, _ = ops.bias_act_dropout_fwd(y2, None, act=1)
        a_l = torch.mm(g, ops.cast_cached(lw, cdt).t())
        out, m = ops.bias_act_dropout_fwd(a_l, lb, act=0, p=p, res=x2, alpha=1.0)
        ctx.save_for_backward(x2s, st1, h, a_pw, gl, cvs, st2, y2, g, m, ln1w, pww, pwb, cw, ln2w, lw)
        ctx.cfg = (eps1, eps2, pad_left, p, cdt, (B, T, D))
        return out.view(B, T, D)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, dout):
        x2s, st1, h, a_pw, gl, cvs, st2, y2, g, m, ln1w, pww, pwb, cw, ln2w, lw = ctx.saved_tensors
        eps1, eps2, pad_left, p, cdt, (B, T, D) = ctx.cfg
        dout2 = dout.reshape(B * T, D)
        dout2 = dout2 if dout2.is_contiguous() else dout2.contiguous()
        if dout2.dtype != torch.float32:
            dout2 = dout2.float()
        da_l, db_l = ops.bias_act_dropout_bwd(dout2, m, p, act=0, out_dtype=cdt)
        dg = torch.mm(da_l, ops.cast_cached(lw, cdt))
        dlw = ops.wgrad(da_l.view(B * T, D), g.view(B * T, D), nbatch=B)
        dy2, _ = ops.bias_act_dropout_bwd(dg, None, 0.0, a=y2, act=1, want_dbias=False)
        dcv, dln2w, dln2b = ops.layernorm_bwd(dy2, cvs, st2, ln2w, eps2)
        dgl, dcw, dcb = ops.dwconv_cl_bwd(gl.view(B, T, D), cw, dcv.view(B, T, D), True, pad_left)
        da_pw, dpwb = ops.bias_glu_bwd(dgl.reshape(B * T, D), a_pw, pwb)
        wpw = ops.cast_cached(pww, cdt).view(2 * D, D)
        dh = torch.mm(da_pw, wpw)
        dpww = ops.wgrad(da_pw.view(B * T, 2 * D), h.view(B * T, D), nbatch=B).view(pww.shape)
        dx, dln1w, dln1b = ops.layernorm_bwd(dh, x2s, st1, ln1w, eps1, dres=dout2)     # dx = dout + LayerNorm'(dh) in one pass
        dx = dx.view(B, T, D)
        return (dx, dln1w, dln1b, dpww, dpwb, dcw.reshape(cw.shape), dcb, dln2w, dln2b, dlw, db_l, None, None, None, None)


def convmod_rows(cm, x):
    """x (B, T, D) fp32 residual stream -> x + convolution_module(x)."""
    drop = cm.after_conv[3]
    p = float(drop.p) if drop.training else 0.0
    pw, lin = cm.bottleneck[0], cm.after_conv[2]
    pad_left = cm.kernel_size - 1 if cm.causal else cm.kernel_size // 2
    return ConvModuleRowsFn.apply(x, cm.layer_norm.weight, cm.layer_norm.bias, pw.weight, pw.bias, cm.conv.weight, cm.conv.bias,
                                  cm.after_conv[0].weight, cm.after_conv[0].bias, lin.weight, lin.bias, cm.layer_norm.eps,
                                  cm.after_conv[0].eps, pad_left, p)
