"""The Mamba mixer of a ConMamba layer as ONE autograd node on channels-last rows — MI355X-first counterpart of the
reference's MambaInnerFnNoOutProj (modules/mamba/selective_scan_interface.py:160-294) run twice by bimamba.Mamba.forward
(modules/mamba/bimamba.py:176-253), and of MambaInnerFn (:297-439) for the unidirectional mixer of the Mamba decoder.

Forward and backward stay on the layout the in_proj GEMM writes, (batch * time, channels) rows, end to end:

    forward   xz = hidden W_in^T  ->  cm_conv_xproj (both directions' conv + SiLU and x_proj; cm_conv_cl_fwd + a GEMM where
              that kernel has no instantiation)  ->  cm_scan_cl_fwd (row-group scan, both directions in one launch; writes the
              half-block checkpoints and the pre-gate output)  ->  out = [y_f | y_b] [s W_out | s W_out]^T
    backward  dmix = dY (s W_out)  ->  cm_scan_cl_bwd (both directions: du, dz shares, dx_dbl incl. the dt columns, dA, dD,
              d dt_proj)  ->  du += dx_dbl W_x, dW_x  ->  cm_conv_cl_bwd (dx of both directions summed, dz summed, conv
              gradients)  ->  dhidden = dxz W_in, dW_in

against the reference's per-direction (B, E, T) tensors: no flips, no transposes, no `.contiguous()` copies, 1 scan launch
instead of 2 in each pass, and nothing recomputed but the conv pre-activation (the reference recomputes conv and delta,
checkpoint_lvl 1, :243-246; here u is kept: 288 GB of HBM).  Numerics: what the operator API computes, with the same bf16
rounding points under autocast (GEMM operands; delta stays fp32 inside the scan).
"""
from __future__ import annotations

import os

import torch
from torch.amp import custom_bwd, custom_fwd

from ... import ops

# CM_ROWS_TRAIN=0: the mixers run on the (B, E, T) operator API (selective_scan_interface._MambaInner) as in rounds 1-2
ENABLED = os.environ.get("CM_ROWS_TRAIN", "1") == "1"


def supported(m, hidden: torch.Tensor) -> bool:
    """m: bimamba.Mamba (v2) or bimamba.UniMamba."""
    if not (ENABLED and hidden.is_cuda and hidden.dim() == 3):
        return False
    cdt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else hidden.dtype
    if cdt not in (torch.float32, torch.bfloat16):
        return False
    return (m.d_state == 16 and m.d_conv == 4 and m.d_inner % 8 == 0 and m.in_proj.bias is None and m.out_proj.bias is None
            and getattr(m, "init_layer_scale", None) is None and (m.dt_rank <= 16 or (m.dt_rank <= 32 and cdt == torch.bfloat16))
            and m.conv1d.bias is not None)


class _Derived:
    """Per-module operands derived from the parameters, rebuilt when a parameter changes (optimizer step, load_state_dict):
    compute-dtype copies, x_proj re-rowed to [dt (zero padded to P) | B | C] as cm_scan_cl_fwd reads it, its packed image
    for cm_conv_xproj, the zero-padded dt_proj weight (rounded to the compute dtype like the reference's GEMM operand,
    selective_scan_interface.py:187), A = -exp(A_log), the K-concatenated out_proj weight."""

    def __init__(self, m, sfx, cdt, scale):
        self.key = self.make_key(m, cdt)
        self.cdt, self.nsfx = cdt, len(sfx)
        E, R = m.d_inner, m.dt_rank
        P = ops.rows_dt_pad(R)
        self.P, self.RW = P, P + 32
        self.w_in = m.in_proj.weight.detach().to(cdt)
        self.w_out = (m.out_proj.weight.detach() * scale).to(cdt)                       # (D, E), scaled
        self.w_out_cat = torch.cat([self.w_out] * len(sfx), dim=1).contiguous()          # (D, ndir * E)
        self.xr, self.xr_packed, self.dtw, self.A = [], [], [], []
        for s in sfx:
            xp, dtp = getattr(m, "x_proj" + s), getattr(m, "dt_proj" + s)
            xr = torch.zeros(self.RW, E, dtype=cdt, device=xp.weight.device)
            xr[:R] = xp.weight.detach()[:R].to(cdt)
            xr[P:] = xp.weight.detach()[R:].to(cdt)
            self.xr.append(xr)
            self.dtw.append(ops.pad_dt_weight(dtp.weight.detach().to(cdt)))
            self.A.append((-torch.exp(getattr(m, "A_b_log" if s else "A_log").detach().float())).contiguous())
        if cdt == torch.bfloat16 and len(sfx) == 2 and E % 32 == 0 and E <= 2048:
            self.xr_packed = [ops.PackedWeight(x) for x in self.xr]
        # both directions' x_proj as ONE block-diagonal (2 RW, 2 E) matrix: the backward's input gradient (du += dx_dbl W_x) and weight
        # gradient are one GEMM / one batched GEMM for the pair instead of one per direction (the step is host-bound: launches count)
        self.xr_bd = torch.block_diag(*self.xr) if len(sfx) == 2 else None

    def rebuild_(self, m, sfx, cdt, scale):
        """The same operands from the parameters' current values, written into the tensors this object already holds (their
        addresses are what a captured hipGraph reads: ops.CACHE_INPLACE)."""
        E, R, P = m.d_inner, m.dt_rank, self.P
        self.w_in.copy_(m.in_proj.weight.detach())
        self.w_out.copy_(m.out_proj.weight.detach() * scale)
        for i, s in enumerate(sfx):
            xp, dtp = getattr(m, "x_proj" + s), getattr(m, "dt_proj" + s)
            self.w_out_cat[:, i * E:(i + 1) * E].copy_(self.w_out)
            self.xr[i][:R].copy_(xp.weight.detach()[:R])
            self.xr[i][P:].copy_(xp.weight.detach()[R:])
            self.dtw[i].copy_(ops.pad_dt_weight(dtp.weight.detach().to(cdt)))
            self.A[i].copy_(-torch.exp(getattr(m, "A_b_log" if s else "A_log").detach().float()))
            if self.xr_packed:
                self.xr_packed[i].repack_(self.xr[i])
            if self.xr_bd is not None:
                self.xr_bd[i * self.RW:(i + 1) * self.RW, i * E:(i + 1) * E].copy_(self.xr[i])
        self.key = self.make_key(m, cdt)
        return self

    @staticmethod
    def make_key(m, cdt):
        # the module's parameter OBJECTS, listed once (Module.parameters() walks the module tree: 3 ms of host time per training step,
        # which is host-bound); ops.invalidate_caches drops the list
        pl = m.__dict__.get("_cm_plist")
        if pl is None:
            pl = list(m.parameters())
            m.__dict__["_cm_plist"] = pl
        return (cdt,) + tuple((p._version, p.data_ptr()) for p in pl)


def _derived(m, sfx, cdt, scale) -> _Derived:
    d = getattr(m, "_cm_rows_derived", None)
    if d is not None and ops._cache_hit(m, "_cm_rows_derived", d.key == _Derived.make_key(m, cdt)):
        return d
    if ops.CACHE_INPLACE and d is not None and d.cdt == cdt and d.nsfx == len(sfx) and d.w_in.device == m.in_proj.weight.device:
        d.rebuild_(m, sfx, cdt, scale)
    else:
        d = _Derived(m, sfx, cdt, scale)
        m._cm_rows_derived = d
        ops._cache_new_storage()
    ops._cache_note(m, "_cm_rows_derived")
    return d


class MixerRowsFn(torch.autograd.Function):
    """hidden (B, T, D) -> (B, T, D).  Parameters per direction: conv weight (E, 1, 4), conv bias, x_proj weight (R + 32, E),
    dt_proj weight (E, R), dt_proj bias, A_log (E, 16), D (E); then in_proj weight (2E, D), out_proj weight (D, E)."""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, hidden, m, sfx, scale, ln, *params):
        cdt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else hidden.dtype
        dv = _derived(m, sfx, cdt, scale)
        ndir = len(sfx)
        B, T, D = hidden.shape
        E, R, P, RW = m.d_inner, m.dt_rank, dv.P, dv.RW
        dev = hidden.device
        x2s = stats = None
        if ln is not None:
            # pre-norm block: out = hidden + mixer(LayerNorm(hidden)) (reference Conmamba.py:641-643); hidden = the fp32 residual stream
            x2 = hidden.detach().reshape(B * T, D)
            x2 = x2 if x2.is_contiguous() else x2.contiguous()
            h2, x2s, stats = ops.layernorm_fwd(x2, ln.weight, ln.bias, ln.eps, cdt)
            h2 = h2.view(B, T, D)
        else:
            h2 = hidden.detach().to(cdt)
            h2 = h2 if h2.is_contiguous() else h2.contiguous()
        xz = torch.mm(h2.view(B * T, D), dv.w_in.t()).view(B, T, 2 * E)
        x, z = xz[:, :, :E], xz[:, :, E:]
        convs = [getattr(m, "conv1d" + s) for s in sfx]
        cw = [c.weight.detach().float().reshape(E, -1) for c in convs]
        cb = [c.bias.detach().float() for c in convs]
        ucat = torch.empty((B, T, ndir * E), dtype=cdt, device=dev)
        if dv.xr_packed:
            xdbl = ops.conv_xproj(x, cw[0], cb[0], cw[1], cb[1], dv.xr_packed[0], dv.xr_packed[1], out_f=ucat[:, :, :E], out_b=ucat[:, :, E:])
        else:
            if ndir == 2:
                ops.conv_cl_fwd(x, cw[0], cb[0], cw[1], cb[1], out_f=ucat[:, :, :E], out_b=ucat[:, :, E:])
            else:
                ops.conv_cl_fwd(x, cw[0], cb[0], out_f=ucat)
            xdbl = torch.empty((B, T, ndir * RW), dtype=cdt, device=dev)
            for i in range(ndir):
                xdbl.view(B * T, ndir * RW)[:, RW * i:RW * (i + 1)] = torch.mm(ucat.view(B * T, ndir * E)[:, E * i:E * (i + 1)], dv.xr[i].t())
        need_grad = any(ctx.needs_input_grad)
        ycat = torch.empty((B, T, ndir * E), dtype=cdt, device=dev)
        pcat = torch.empty((B, T, ndir * E), dtype=cdt, device=dev) if need_grad else None
        cks, dirs = [], []
        for i, s in enumerate(sfx):
            dtp = getattr(m, "dt_proj" + s)
            dd = dict(u=ucat[:, :, E * i:E * (i + 1)], xdbl=xdbl[:, :, RW * i:RW * (i + 1)], A=dv.A[i], D=getattr(m, "D_b" if s else "D").detach().float(),
                      delta_bias=dtp.bias.detach().float(), dt_weight=dv.dtw[i], reverse=bool(s), out=ycat[:, :, E * i:E * (i + 1)])
            if need_grad:
                ck = torch.empty(ops.scan_ckpt_shape(B, T, E), dtype=torch.float32, device=dev)
                cks.append(ck)
                dd.update(ypre=pcat[:, :, E * i:E * (i + 1)], ckpt=ck)
            dirs.append(dd)
        ops.scan_cl_fwd(dirs, z=z, delta_softplus=True)
        out = torch.mm(ycat.view(B * T, ndir * E), dv.w_out_cat.t()).view(B, T, D)
        if ln is not None:
            out = x2s.view(B, T, D) + out                                               # fp32 stream
        if need_grad:
            ctx.m, ctx.sfx, ctx.scale, ctx.dv, ctx.cdt, ctx.ln = m, sfx, scale, dv, cdt, ln
            ctx.save_for_backward(h2, xz, ucat, xdbl, pcat, ycat, x2s, stats, *cks)
        return out

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, dY):
        m, sfx, scale, dv, cdt = ctx.m, ctx.sfx, ctx.scale, ctx.dv, ctx.cdt
        h2, xz, ucat, xdbl, pcat, ycat, x2s, stats, *cks = ctx.saved_tensors
        ln = ctx.ln
        dres = None
        if ln is not None:
            dres = dY.reshape(-1, dY.shape[-1])
            dres = (dres if dres.dtype == torch.float32 else dres.float())
            dres = dres if dres.is_contiguous() else dres.contiguous()
        ndir = len(sfx)
        B, T, D = h2.shape
        E, R, P, RW = m.d_inner, m.dt_rank, dv.P, dv.RW
        dev = h2.device
        dY = dY.to(cdt)
        dY = dY if dY.is_contiguous() else dY.contiguous()
        x, z = xz[:, :, :E], xz[:, :, E:]
        # out_proj: every direction sees the same gradient, dmix = dY (scale W_out)
        dmix = torch.mm(dY.view(B * T, D), dv.w_out).view(B, T, E)
        d_out_cat = ops.wgrad(dY.view(B * T, D), ycat.view(B * T, ndir * E), nbatch=B)    # (D, ndir * E), fp32, fixed summation order
        d_out_w = d_out_cat[:, :E] if ndir == 1 else d_out_cat[:, :E] + d_out_cat[:, E:]
        d_out_w = d_out_w * scale
        # scan backward, all directions in one launch
        ducat = torch.empty((B, T, ndir * E), dtype=cdt, device=dev)
        dxdbl = torch.empty((B, T, ndir * RW), dtype=cdt, device=dev)
        dxz = torch.empty((B, T, 2 * E), dtype=cdt, device=dev)
        dzs = [dxz[:, :, E:]] if ndir == 1 else [torch.empty((B, T, E), dtype=cdt, device=dev) for _ in range(ndir)]
        dirs = []
        for i, s in enumerate(sfx):
            dtp = getattr(m, "dt_proj" + s)
            dirs.append(dict(u=ucat[:, :, E * i:E * (i + 1)], xdbl=xdbl[:, :, RW * i:RW * (i + 1)], A=dv.A[i],
                             D=getattr(m, "D_b" if s else "D").detach().float(), delta_bias=dtp.bias.detach().float(), dt_weight=dv.dtw[i],
                             reverse=bool(s), ckpt=cks[i], ypre=pcat[:, :, E * i:E * (i + 1)], dout=dmix, du=ducat[:, :, E * i:E * (i + 1)], dz=dzs[i],
                             dxdbl=dxdbl[:, :, RW * i:RW * (i + 1)]))
        res = ops.scan_cl_bwd(dirs, z, da_log=True)                                    # dA arrives as the gradient of A_log (= dA * A)
        # x_proj: weight gradient per utterance + sum (K = batch * time GEMMs with a 48-row output fill few workgroups), input
        # gradient added to du in place
        grads = []
        du2, dx2 = ducat.view(B * T, ndir * E), dxdbl.view(B * T, ndir * RW)
        dxr = []
        if ndir == 2 and dv.xr_bd is not None:
            full = ops.sum_leading(torch.bmm(dxdbl.transpose(1, 2), ucat))                # (2 RW, 2 E): the diagonal blocks are the two directions'
            dxr = [full[:RW, :E], full[RW:, E:]]
            du2.addmm_(dx2, dv.xr_bd)
        else:
            for i in range(ndir):
                dxr.append(ops.sum_leading(torch.bmm(dxdbl[:, :, RW * i:RW * (i + 1)].transpose(1, 2), ucat[:, :, E * i:E * (i + 1)])))   # (RW, E)
                du2[:, E * i:E * (i + 1)].addmm_(dx2[:, RW * i:RW * (i + 1)], dv.xr[i])
        convs = [getattr(m, "conv1d" + s) for s in sfx]
        cw = [c.weight.detach().float().reshape(E, -1) for c in convs]
        cb = [c.bias.detach().float() for c in convs]
        if ndir == 2:
            _, _, dwf, dbf, dwb, dbb = ops.conv_cl_bwd(x, cw[0], cb[0], ducat[:, :, :E], cw[1], cb[1], ducat[:, :, E:], dz_f=dzs[0], dz_b=dzs[1],
                                                       dx=dxz[:, :, :E], dz=dxz[:, :, E:])
            dconv = [(dwf, dbf), (dwb, dbb)]
        else:
            _, _, dwf, dbf, _, _ = ops.conv_cl_bwd(x, cw[0], cb[0], ducat, dx=dxz[:, :, :E])
            dconv = [(dwf, dbf)]
        d_hidden = torch.mm(dxz.view(B * T, 2 * E), dv.w_in).view(B, T, D) if (ctx.needs_input_grad[0] or ln is not None) else None
        dln = (None, None)
        if ln is not None:                                                              # dx = dY + LayerNorm'(dh) in one pass
            d_hidden, dlw, dlb = ops.layernorm_bwd(d_hidden.view(B * T, D), x2s, stats, ln.weight, ln.eps, dres=dres)
            d_hidden, dln = d_hidden.view(B, T, D), (dlw, dlb)
        d_in_w = ops.wgrad(dxz.view(B * T, 2 * E), h2.view(B * T, D), nbatch=B)            # (2E, D)
        for i, s in enumerate(sfx):
            r = res[i]
            dxw = torch.cat([dxr[i][:R], dxr[i][P:]], dim=0)                            # back to x_proj's (R + 32, E) rows
            grads += [dconv[i][0].reshape(convs[i].weight.shape), dconv[i][1], dxw, r["ddt_weight"][:, :R], r["ddelta_bias"],
                      r["dA"], r["dD"]]
        grads += [d_in_w, d_out_w]
        if ln is not None:
            grads += [dln[0], dln[1]]
        return (d_hidden, None, None, None, None, *grads)


def mixer_rows(m, hidden, ln=None):
    """Run module ``m`` (bimamba.Mamba v2 or bimamba.UniMamba) on the rows node; caller checked ``supported``.
    With ``ln`` (an nn.LayerNorm over the last axis): the pre-norm block hidden + m(ln(hidden)) on the fp32 residual stream."""
    bidir = hasattr(m, "A_b_log")
    sfx = ("", "_b") if bidir else ("",)
    scale = 0.5 if (bidir and m.if_devide_out) else 1.0
    params = []
    for s in sfx:
        conv, xp, dtp = getattr(m, "conv1d" + s), getattr(m, "x_proj" + s), getattr(m, "dt_proj" + s)
        params += [conv.weight, conv.bias, xp.weight, dtp.weight, dtp.bias, getattr(m, "A_b_log" if s else "A_log"), getattr(m, "D_b" if s else "D")]
    params += [m.in_proj.weight, m.out_proj.weight]
    if ln is not None:
        params += [ln.weight, ln.bias]
    return MixerRowsFn.apply(hidden, m, sfx, scale, ln, *params)


def block_supported(m, ln, x) -> bool:
    return (supported(m, x) and x.dtype == torch.float32 and ln.weight is not None and ln.bias is not None
            and len(ln.normalized_shape) == 1 and x.shape[-1] <= 1024)
