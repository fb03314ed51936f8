"""Operator API of the Mamba mixer on MI355X — same callables, argument order and return
conventions as the reference's modules/mamba/selective_scan_interface.py, backed by the HIP
library (C ABI include/conmamba_hip.h) instead of selective_scan_cuda / causal_conv1d_cuda.

    selective_scan_fn            reference :82-88   (autograd wrapper :19-79)
    mamba_inner_fn_no_out_proj   reference :632-638 (autograd :160-294)  <- BiMamba v2 hot op
    mamba_inner_fn               reference :611-619 (autograd :297-439)
    bimamba_inner_fn             reference :621-629 (autograd :442-608; v1, unreachable from ASR)
    mamba_inner_ref / bimamba_inner_ref   reference :641-714 (unfused compositions of the ops)

Extensions (keyword-only, default off): ``reverse_time`` runs conv + scan against the time axis in
place of the reference's ``.flip(-1)`` copies (modules/mamba/bimamba.py:237, 253).

``selective_scan_ref`` (reference :91-157, pure torch) is deliberately NOT here: it is the
oracle and lives in oracle/conmamba_oracle.py; nothing in this package computes on the CPU.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch.amp import custom_bwd, custom_fwd

from ... import ops

__all__ = ["selective_scan_fn", "mamba_inner_fn", "mamba_inner_fn_no_out_proj", "bimamba_inner_fn",
           "mamba_inner_ref", "bimamba_inner_ref", "causal_conv1d_fn", "SelectiveScanFn"]


# ------------------------------------------------------------------------------------------
# selective scan
# ------------------------------------------------------------------------------------------
class SelectiveScanFn(torch.autograd.Function):
    """reference :19-79.  Saves the chunk checkpoints ``x``; the pre-gate ``out`` is recomputed by
    the backward kernel instead of being kept."""

    @staticmethod
    def forward(ctx, u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                return_last_state=False, reverse_time=False):
        ctx.squeeze_B, ctx.squeeze_C = B.dim() == 3, C.dim() == 3
        need_grad = any(t is not None and t.requires_grad for t in (u, delta, A, B, C, D, z, delta_bias))
        out, x, out_z = ops.selective_scan_fwd(u, delta, A, B, C, D, z, delta_bias, delta_softplus,
                                               reverse=reverse_time, need_out=False,
                                               need_x=need_grad or return_last_state)
        ctx.delta_softplus, ctx.reverse_time, ctx.has_z = delta_softplus, reverse_time, z is not None
        ctx.save_for_backward(u, delta, A, B, C, D, z, delta_bias, x)
        result = out_z if z is not None else out
        if not return_last_state:
            return result
        last = x[:, :, 0 if reverse_time else -1, 1::2]       # (batch, dim, dstate), reference :45
        ctx.mark_non_differentiable(last)
        return result, last

    @staticmethod
    def backward(ctx, dout, *unused):
        u, delta, A, B, C, D, z, delta_bias, x = ctx.saved_tensors
        du, ddelta, dA, dB, dC, dD, dbias, dz, _ = ops.selective_scan_bwd(
            u, delta, A, B, C, D, z, delta_bias, dout, x, ctx.delta_softplus, reverse=ctx.reverse_time)
        dB = dB.squeeze(1) if ctx.squeeze_B else dB
        dC = dC.squeeze(1) if ctx.squeeze_C else dC
        return (du, ddelta, dA.to(A.dtype), dB.to(B.dtype), dC.to(C.dtype),
                None if D is None else dD.to(D.dtype), dz,
                None if delta_bias is None else dbias.to(delta_bias.dtype), None, None, None)


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                      return_last_state=False, *, reverse_time=False):
    """Same contract as the reference's selective_scan_fn (:82-88): returns out, or (out, last_state)
    with last_state (batch, dim, dstate) excluded from autograd."""
    return SelectiveScanFn.apply(u, delta, A, B, C, D, z, delta_bias, delta_softplus, return_last_state,
                                 reverse_time)


# ------------------------------------------------------------------------------------------
# causal depthwise conv (python face of K1/K2; the reference imports it from the causal_conv1d wheel)
# ------------------------------------------------------------------------------------------
class _CausalConv1dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, silu, reverse_time):
        ctx.silu, ctx.reverse_time = silu, reverse_time
        ctx.save_for_backward(x, weight, bias)
        return ops.causal_conv1d_fwd(x, weight, bias, silu, reverse=reverse_time)

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        dx, dw, db = ops.causal_conv1d_bwd(x, weight, bias, dy, ctx.silu, reverse=ctx.reverse_time)
        return dx, dw.to(weight.dtype), None if bias is None else db.to(bias.dtype), None, None


def causal_conv1d_fn(x, weight, bias=None, activation=None, *, reverse_time=False):
    """causal_conv1d.causal_conv1d_fn(x (b,d,l), weight (d,w), bias, activation in {None,'silu','swish'})
    as the reference calls it (modules/mamba/bimamba.py:282-287)."""
    if activation not in (None, "silu", "swish"):
        raise NotImplementedError("activation must be None, silu, or swish")
    return _CausalConv1dFn.apply(x, weight, bias, activation is not None, reverse_time)


# ------------------------------------------------------------------------------------------
# fused Mamba inner op.  One autograd node covers the reference's three Function classes:
#   out_proj_weight is None            -> MambaInnerFnNoOutProj (:160-294)
#   out_proj_weight given, A_b is None -> MambaInnerFn          (:297-439)
#   A_b given                          -> BiMambaInnerFn (v1)   (:442-608)
# ------------------------------------------------------------------------------------------
def _cast_autocast(*ws):
    if torch.is_autocast_enabled("cuda"):
        dt = torch.get_autocast_dtype("cuda")
        return tuple(None if w is None else ops.cast_cached(w, dt) for w in ws)
    return ws


def _split_bc(x_dbl, rank, nstate, batch, length):
    """x_dbl ((b l), rank+2n) -> B, C as contiguous (b, 1, n, l)  (reference :192-215)."""
    bc = x_dbl[:, rank:].reshape(batch, length, 2, nstate).permute(2, 0, 3, 1).contiguous()   # (2, b, n, l)
    return bc[0].unsqueeze(1), bc[1].unsqueeze(1)


def _ebt(e, batch, length, like):
    """(e, batch, length) storage and its (batch, e, length) time-contiguous view: every kernel takes batch / channel
    strides, so tensors that a GEMM reads or writes as (e, batch*length) matrices never need a transposing copy."""
    st = torch.empty((e, batch, length), dtype=like.dtype, device=like.device)
    return st, st.transpose(0, 1)


class _MambaInner(torch.autograd.Function):
    """conv -> x_proj -> dt_proj -> selective scan [-> out_proj] with the reference's checkpointing (:171-229, 233-294).
    The skinny GEMMs run in the (feature, batch*time) orientation on (feature, batch, time) storage whose
    (batch, feature, time) views feed the HIP kernels: no transposing copies (the first version made ~10 per direction
    per layer, forward + backward: 20 % of a training step)."""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, xz, conv_w, conv_b, x_proj_w, dt_proj_w, out_proj_w, out_proj_b, A, A_b, D, delta_bias,
                delta_softplus, checkpoint_lvl, reverse_time):
        assert checkpoint_lvl in (0, 1)                                         # reference :170
        x_proj_w, dt_proj_w, out_proj_w, out_proj_b = _cast_autocast(x_proj_w, dt_proj_w, out_proj_w, out_proj_b)
        if xz.stride(-1) != 1:
            xz = xz.contiguous()
        batch, two_e, length = xz.shape
        e = two_e // 2
        rank, nstate = dt_proj_w.shape[1], A.shape[-1]
        w2 = conv_w.reshape(conv_w.shape[0], conv_w.shape[-1])                  # "d 1 w -> d w", :179
        x, z = xz[:, :e], xz[:, e:]
        u_s, u = _ebt(e, batch, length, xz)
        ops.causal_conv1d_fwd(x, w2, conv_b, True, reverse=reverse_time, out=u)  # K1, :182
        x_dblT = x_proj_w.to(u_s.dtype) @ u_s.view(e, batch * length)           # (rank + 2n, b*l) = x_dbl^T, :186
        delta = (dt_proj_w.to(u_s.dtype) @ x_dblT[:rank]).view(e, batch, length).transpose(0, 1)   # :187, (b, e, l) view
        Bm = x_dblT[rank:rank + nstate].view(nstate, batch, length).permute(1, 0, 2).unsqueeze(1)   # (b, 1, n, l) views
        Cm = x_dblT[rank + nstate:].view(nstate, batch, length).permute(1, 0, 2).unsqueeze(1)
        need_x = any(t is not None and t.requires_grad for t in
                     (xz, conv_w, conv_b, x_proj_w, dt_proj_w, out_proj_w, A, A_b, D, delta_bias))
        oz_s, oz = _ebt(e, batch, length, xz)
        _, ck, _ = ops.selective_scan_fwd(u, delta, A, Bm, Cm, D, z, delta_bias, delta_softplus, reverse=reverse_time,
                                          need_out=False, need_x=need_x, out_z_buf=oz)     # K3, :218
        ck_b = None
        if A_b is not None:                      # v1 bidirectional: second scan against time, summed (:504-512)
            ozb_s, ozb = _ebt(e, batch, length, xz)
            _, ck_b, _ = ops.selective_scan_fwd(u, delta, A_b, Bm, Cm, D, z, delta_bias, delta_softplus,
                                                reverse=not reverse_time, need_out=False, need_x=need_x, out_z_buf=ozb)
            oz_s.add_(ozb_s)
        ctx.delta_softplus, ctx.checkpoint_lvl, ctx.reverse_time = delta_softplus, checkpoint_lvl, reverse_time
        ctx.has_out_proj, ctx.has_out_bias = out_proj_w is not None, out_proj_b is not None
        ctx.has_A_b = A_b is not None
        if checkpoint_lvl >= 1:                  # recompute conv output and delta in backward (:223-224)
            u_keep, delta_keep = None, None
        else:
            u_keep, delta_keep = u_s, delta
        ctx.save_for_backward(xz, w2, conv_b, x_dblT, x_proj_w, dt_proj_w, out_proj_w, u_keep, delta_keep, A, A_b, D,
                              delta_bias, ck, ck_b)
        ctx.conv_w_shape = conv_w.shape
        if out_proj_w is None:
            return oz                                                           # (b, e, l) view, :229
        y = torch.mm(oz_s.view(e, batch * length).t(), out_proj_w.t())          # (b*l, d_model), :370
        if out_proj_b is not None:
            y = y + out_proj_b
        return y.view(batch, length, -1)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, dout):
        (xz, w2, conv_b, x_dblT, x_proj_w, dt_proj_w, out_proj_w, u_s, delta, A, A_b, D, delta_bias, ck,
         ck_b) = ctx.saved_tensors
        batch, two_e, length = xz.shape
        e = two_e // 2
        rank, nstate = dt_proj_w.shape[1], A.shape[-1]
        rev, bl = ctx.reverse_time, batch * length
        x, z = xz[:, :e], xz[:, e:]
        if ctx.checkpoint_lvl == 1:                                              # :243-246
            u_s, u = _ebt(e, batch, length, xz)
            ops.causal_conv1d_fwd(x, w2, conv_b, True, reverse=rev, out=u)
            delta = (dt_proj_w.to(u_s.dtype) @ x_dblT[:rank]).view(e, batch, length).transpose(0, 1)
        else:
            u = u_s.transpose(0, 1)
        Bm = x_dblT[rank:rank + nstate].view(nstate, batch, length).permute(1, 0, 2).unsqueeze(1)
        Cm = x_dblT[rank + nstate:].view(nstate, batch, length).permute(1, 0, 2).unsqueeze(1)
        dout_proj_w = dout_proj_b = None
        if ctx.has_out_proj:
            dflat = dout.reshape(bl, -1)                                         # (b l) d_model
            dy = (out_proj_w.t() @ dflat.t().to(out_proj_w.dtype)).view(e, batch, length).transpose(0, 1)   # :392-393
        else:
            dflat = None
            dy = dout if dout.stride(-1) == 1 else dout.contiguous()
        dy = dy.to(u_s.dtype)
        dxz = torch.empty_like(xz)                                               # dx | dz side by side, :249-250
        dx, dz = dxz[:, :e], dxz[:, e:]
        du_s, du = _ebt(e, batch, length, xz)
        dd_s, ddelta = _ebt(e, batch, length, xz)
        oz_s, oz = _ebt(e, batch, length, xz) if ctx.has_out_proj else (None, None)
        _, _, dA, dB, dC, dD, dbias, _, _ = ops.selective_scan_bwd(
            u, delta, A, Bm, Cm, D, z, delta_bias, dy, ck, ctx.delta_softplus, reverse=rev, dz=dz,
            recompute_out_z=ctx.has_out_proj, du_buf=du, ddelta_buf=ddelta, out_z_buf=oz)     # K4, :252-256
        dA_b = None
        if ctx.has_A_b:                                                          # :552-566
            du2, dd2, dA_b, dB2, dC2, dD2, dbias2, dz2, oz2 = ops.selective_scan_bwd(
                u, delta, A_b, Bm, Cm, D, z, delta_bias, dy, ck_b, ctx.delta_softplus, reverse=not rev,
                recompute_out_z=ctx.has_out_proj)
            du.add_(du2), ddelta.add_(dd2)
            dB, dC = dB + dB2, dC + dC2
            dz.add_(dz2)
            dD = None if dD is None else dD + dD2
            dbias = None if dbias is None else dbias + dbias2
            if oz is not None:
                oz.add_(oz2)
        if ctx.has_out_proj:
            dout_proj_w = (oz_s.view(e, bl) @ dflat.to(oz_s.dtype)).t()          # :399
            if ctx.has_out_bias:
                dout_proj_b = dflat.sum(0)                                       # :400
        # gradients through the two skinny projections (:258-283), all on (feature, b*l) matrices
        dx_dblT = torch.empty_like(x_dblT)
        dx_dblT[rank:rank + nstate] = dB[:, 0].permute(1, 0, 2).reshape(nstate, bl)
        dx_dblT[rank + nstate:] = dC[:, 0].permute(1, 0, 2).reshape(nstate, bl)
        dd_flat = dd_s.view(e, bl)
        # the two weight gradients reduce over b*l with a 16- / 48-wide output: as ONE GEMM with K = b*l the library runs
        # them at ~170 us each (skinny output, no split-K); as per-utterance products summed over the batch they are
        # bandwidth-bound reads of dd / u (strided batched GEMM on the (feature, batch, time) storage: no copies)
        ddt_proj_w = torch.bmm(dd_s.permute(1, 0, 2), x_dblT[:rank].view(rank, batch, length).permute(1, 2, 0)).sum(0)   # :278
        dx_dblT[:rank] = dt_proj_w.t().to(dd_flat.dtype) @ dd_flat               # :279
        dx_proj_w = torch.bmm(dx_dblT.view(-1, batch, length).permute(1, 0, 2), u_s.permute(1, 2, 0)).sum(0)            # :281
        du_s.view(e, bl).addmm_(x_proj_w.t().to(du_s.dtype), dx_dblT)            # du + x_proj^T dx_dbl^T, :282
        _, dconv_w, dconv_b = ops.causal_conv1d_bwd(x, w2, conv_b, du, True, reverse=rev, dx=dx)   # K2, :286
        return (dxz, dconv_w.reshape(ctx.conv_w_shape), dconv_b, dx_proj_w, ddt_proj_w, dout_proj_w, dout_proj_b,
                dA, dA_b, dD, dbias, None, None, None)


def _inner(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias, A, A_b,
           B, C, D, delta_bias, B_proj_bias, C_proj_bias, delta_softplus, reverse_time):
    if B is not None or C is not None or B_proj_bias is not None or C_proj_bias is not None:
        raise NotImplementedError("only input-dependent B/C without projection bias (what the ConMamba recipes "
                                  "use, reference bimamba.py:230-231) have a HIP path")
    if A.is_complex():
        raise NotImplementedError("complex A has no HIP path (unused by the ASR recipes)")
    return _MambaInner.apply(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight,
                             out_proj_bias, A, A_b, D, delta_bias, delta_softplus, 1, reverse_time)


def mamba_inner_fn_no_out_proj(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B=None, C=None,
                               D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None, delta_softplus=True, *,
                               reverse_time=False):
    """xz (batch, 2*d_inner, seqlen) -> out_z (batch, d_inner, seqlen); reference :632-638."""
    return _inner(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, None, None, A, None, B, C, D,
                  delta_bias, B_proj_bias, C_proj_bias, delta_softplus, reverse_time)


def mamba_inner_fn(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias,
                   A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None,
                   delta_softplus=True, *, reverse_time=False):
    """-> (batch, seqlen, d_model); reference :611-619."""
    return _inner(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias,
                  A, None, B, C, D, delta_bias, B_proj_bias, C_proj_bias, delta_softplus, reverse_time)


def bimamba_inner_fn(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight,
                     out_proj_bias, A, A_b, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None,
                     C_proj_bias=None, delta_softplus=True):
    """v1 bidirectional op (shared conv/projections, second scan with A_b against time); reference :621-629."""
    return _inner(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias,
                  A, A_b, B, C, D, delta_bias, B_proj_bias, C_proj_bias, delta_softplus, False)


# ------------------------------------------------------------------------------------------
# unfused compositions (reference :641-714): conv op -> projections -> scan op [-> out_proj]
# ------------------------------------------------------------------------------------------
def _projections(x, x_proj_weight, delta_proj_weight, nstate):
    batch, e, length = x.shape
    rank = delta_proj_weight.shape[1]
    x_dbl = F.linear(x.transpose(1, 2).reshape(batch * length, e), x_proj_weight)
    delta = (delta_proj_weight @ x_dbl[:, :rank].t()).reshape(e, batch, length).transpose(0, 1)
    Bm = x_dbl[:, rank:rank + nstate].reshape(batch, length, nstate).transpose(1, 2).contiguous()
    Cm = x_dbl[:, -nstate:].reshape(batch, length, nstate).transpose(1, 2).contiguous()
    return delta, Bm, Cm


def mamba_inner_ref(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight,
                    out_proj_bias, A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None,
                    delta_softplus=True):
    assert B is None and C is None and B_proj_bias is None and C_proj_bias is None
    x, z = xz.chunk(2, dim=1)
    x = causal_conv1d_fn(x, conv1d_weight.reshape(conv1d_weight.shape[0], -1), conv1d_bias, "silu")
    delta, Bm, Cm = _projections(x, x_proj_weight, delta_proj_weight, A.shape[-1])
    y = selective_scan_fn(x, delta, A, Bm, Cm, D, z=z, delta_bias=delta_bias, delta_softplus=True)
    return F.linear(y.transpose(1, 2), out_proj_weight, out_proj_bias)


def bimamba_inner_ref(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight,
                      out_proj_bias, A, A_b, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None,
                      C_proj_bias=None, delta_softplus=True):
    assert B is None and C is None and B_proj_bias is None and C_proj_bias is None
    x, z = xz.chunk(2, dim=1)
    x = causal_conv1d_fn(x, conv1d_weight.reshape(conv1d_weight.shape[0], -1), conv1d_bias, "silu")
    delta, Bm, Cm = _projections(x, x_proj_weight, delta_proj_weight, A.shape[-1])
    y = selective_scan_fn(x, delta, A, Bm, Cm, D, z=z, delta_bias=delta_bias, delta_softplus=True)
    y = y + selective_scan_fn(x, delta, A_b, Bm, Cm, D, z=z, delta_bias=delta_bias, delta_softplus=True,
                              reverse_time=True)
    return F.linear(y.transpose(1, 2), out_proj_weight, out_proj_bias)
