"""BiMamba mixer module — drop-in for the reference's modules/mamba/bimamba.py ``Mamba`` class
(constructor :40-174, forward :176-318) and ``Block`` (:409-465), running on the HIP operator API.

state_dict keys/shapes are identical to the reference: in_proj.weight, conv1d.{weight,bias},
x_proj.weight, dt_proj.{weight,bias}, A_log, D, [A_b_log, conv1d_b.*, x_proj_b.weight, dt_proj_b.*, D_b],
out_proj.weight — so reference checkpoints load unchanged.

MI355X-first differences (results identical):
  * the backward direction is computed with ``reverse_time=True`` kernels on the same xz tensor
    instead of xz.flip(-1) / out_b.flip(-1) copies (reference :237, :253);
  * the two directions are independent until the average, so they are issued on two HIP streams.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from .selective_scan_interface import (bimamba_inner_fn, causal_conv1d_fn, mamba_inner_fn,
                                       mamba_inner_fn_no_out_proj, selective_scan_fn)


class _InProjFn(torch.autograd.Function):
    """xz = in_proj(hidden) delivered as the (batch, 2e, seqlen) view of (2e, batch, seqlen) storage (reference
    bimamba.py:192-198 without its rearrange copies); weight gradient per utterance + sum (see sb_compat._LinearRowsFn)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, hidden, weight):
        batch, seqlen, _ = hidden.shape
        w = ops.cast_cached(weight, torch.get_autocast_dtype("cuda")) if torch.is_autocast_enabled("cuda") else weight
        hidden = hidden.to(w.dtype)                         # cast once; the backward's weight gradient reuses it
        ctx.save_for_backward(hidden, weight)
        xz = (w @ hidden.reshape(batch * seqlen, -1).t()).reshape(-1, batch, seqlen)
        return xz.transpose(0, 1)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dxz):
        hidden, weight = ctx.saved_tensors
        batch, seqlen, _ = hidden.shape
        flat = dxz.permute(1, 0, 2).reshape(dxz.shape[1], batch * seqlen)       # a view when dxz has xz's layout
        dh = dw = None
        if ctx.needs_input_grad[0]:
            dh = (flat.t() @ ops.cast_cached(weight, flat.dtype)).view(batch, seqlen, -1)
        if ctx.needs_input_grad[1]:
            dw = torch.bmm(dxz, hidden.to(dxz.dtype)).sum(0)
        return dh, dw


class _OutProjFn(torch.autograd.Function):
    """y = out_proj(mix) with mix the (batch, e, seqlen) view of (e, batch, seqlen) storage, read as the transposed
    (e, batch*seqlen) matrix; weight gradient per utterance + sum."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, mix, weight, bias):
        batch, e, seqlen = mix.shape
        ctx.save_for_backward(mix, weight)
        ctx.has_bias = bias is not None
        w = ops.cast_cached(weight, mix.dtype)
        y = torch.mm(mix.permute(1, 0, 2).reshape(e, batch * seqlen).t(), w.t())
        if bias is not None:
            y = y + bias.to(y.dtype)
        return y.view(batch, seqlen, -1)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        mix, weight = ctx.saved_tensors
        batch, e, seqlen = mix.shape
        dy = dy.to(mix.dtype)
        dmix = dw = db = None
        if ctx.needs_input_grad[0]:
            dmix = (ops.cast_cached(weight, dy.dtype).t() @ dy.reshape(batch * seqlen, -1).t()).view(e, batch, seqlen).transpose(0, 1)
        if ctx.needs_input_grad[1]:
            dw = torch.bmm(dy.transpose(1, 2), mix.transpose(1, 2)).sum(0)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.reshape(batch * seqlen, -1).sum(0)
        return dmix, dw, db


def _init_dt_proj(dt_proj: nn.Linear, d_inner, dt_rank, dt_init, dt_scale, dt_min, dt_max, dt_init_floor, fk):
    """reference bimamba.py:101-120."""
    std = dt_rank ** -0.5 * dt_scale
    if dt_init == "constant":
        nn.init.constant_(dt_proj.weight, std)
    elif dt_init == "random":
        nn.init.uniform_(dt_proj.weight, -std, std)
    else:
        raise NotImplementedError
    dt = torch.exp(torch.rand(d_inner, **fk) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min))
    dt = dt.clamp(min=dt_init_floor)
    with torch.no_grad():
        dt_proj.bias.copy_(dt + torch.log(-torch.expm1(-dt)))      # inverse softplus
    dt_proj.bias._no_reinit = True


def _s4d_real_log(d_inner, d_state, device):
    """A_log = log(1..N) per channel, kept fp32 (reference :122-130)."""
    a = torch.arange(1, d_state + 1, dtype=torch.float32, device=device).repeat(d_inner, 1).contiguous()
    p = nn.Parameter(torch.log(a))
    p._no_weight_decay = True
    return p


class Mamba(nn.Module):
    """``Mamba(d_model, d_state=16, d_conv=4, expand=2, ..., bimamba_type='v2')``; forward (B, L, D) -> (B, L, D)."""

    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, dt_rank="auto", dt_min=0.001, dt_max=0.1,
                 dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, conv_bias=True, bias=False,
                 use_fast_path=True, layer_idx=None, device=None, dtype=None, bimamba_type="none",
                 if_devide_out=True, init_layer_scale=None):
        fk = {"device": device, "dtype": dtype}
        super().__init__()
        self.d_model, self.d_state, self.d_conv, self.expand = d_model, d_state, d_conv, expand
        self.d_inner = int(expand * d_model)
        self.dt_rank = math.ceil(d_model / 16) if dt_rank == "auto" else dt_rank
        self.use_fast_path, self.layer_idx = use_fast_path, layer_idx
        self.bimamba_type, self.if_devide_out = bimamba_type, if_devide_out
        assert bimamba_type == "v2"                                   # reference :75
        self.init_layer_scale = init_layer_scale
        if init_layer_scale is not None:
            self.gamma = nn.Parameter(init_layer_scale * torch.ones(d_model), requires_grad=True)
        self.activation = "silu"
        self.act = nn.SiLU()

        def conv():
            return nn.Conv1d(self.d_inner, self.d_inner, d_conv, groups=self.d_inner, padding=d_conv - 1,
                             bias=conv_bias, **fk)

        # construction order mirrors the reference so that a seeded init draws the same numbers
        self.in_proj = nn.Linear(d_model, 2 * self.d_inner, bias=bias, **fk)
        self.conv1d = conv()
        self.x_proj = nn.Linear(self.d_inner, self.dt_rank + 2 * d_state, bias=False, **fk)
        self.dt_proj = nn.Linear(self.dt_rank, self.d_inner, bias=True, **fk)
        _init_dt_proj(self.dt_proj, self.d_inner, self.dt_rank, dt_init, dt_scale, dt_min, dt_max, dt_init_floor, fk)
        self.A_log = _s4d_real_log(self.d_inner, d_state, device)
        self.D = nn.Parameter(torch.ones(self.d_inner, device=device))
        self.D._no_weight_decay = True
        # second direction (v2: fully separate parameters, reference :146-172)
        self.A_b_log = _s4d_real_log(self.d_inner, d_state, device)
        self.conv1d_b = conv()
        self.x_proj_b = nn.Linear(self.d_inner, self.dt_rank + 2 * d_state, bias=False, **fk)
        self.dt_proj_b = nn.Linear(self.dt_rank, self.d_inner, bias=True, **fk)
        self.D_b = nn.Parameter(torch.ones(self.d_inner, device=device))
        self.D_b._no_weight_decay = True
        self.out_proj = nn.Linear(self.d_inner, d_model, bias=bias, **fk)
        self._side_stream: Optional[torch.cuda.Stream] = None

    # -- helpers ---------------------------------------------------------------------------
    def _direction(self, xz, backward: bool):
        sfx = "_b" if backward else ""
        conv, xp, dtp = getattr(self, "conv1d" + sfx), getattr(self, "x_proj" + sfx), getattr(self, "dt_proj" + sfx)
        A = -torch.exp(getattr(self, "A_b_log" if backward else "A_log").float())       # reference :200, :222
        Dp = getattr(self, "D_b" if backward else "D")
        return mamba_inner_fn_no_out_proj(xz, conv.weight, conv.bias, xp.weight, dtp.weight, A, None, None,
                                          Dp.float(), delta_bias=dtp.bias.float(), delta_softplus=True,
                                          reverse_time=backward)

    def forward(self, hidden_states, inference_params=None):
        if inference_params is not None:
            raise NotImplementedError("stateful decoding (reference bimamba.py:184-189, 320-406) is not on the "
                                      "training/encoder path and has no HIP kernel yet")
        from . import mixer_rows
        if mixer_rows.supported(self, hidden_states):
            # the whole mixer as one autograd node on channels-last rows (mixer_rows.py): one scan launch for both directions
            # in each pass, no (B, E, T) round trip
            return mixer_rows.mixer_rows(self, hidden_states)
        batch, seqlen, _ = hidden_states.shape
        # in_proj with the (b l d) -> (b d l) transpose folded in (reference :192-198)
        if hidden_states.is_cuda:
            xz = _InProjFn.apply(hidden_states, self.in_proj.weight)
        else:
            xz = (self.in_proj.weight @ hidden_states.reshape(batch * seqlen, -1).t()).reshape(-1, batch, seqlen).transpose(0, 1)
        if self.in_proj.bias is not None:
            xz = xz + self.in_proj.bias.to(xz.dtype)[None, :, None]
        # xz stays the (b, 2e, l) VIEW of the GEMM's (2e, b, l) output: time-contiguous, which is all the kernels need
        use_side = xz.is_cuda and not torch.is_grad_enabled()
        if use_side:
            # forward-only: the two directions share nothing but the input -> two HIP streams
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=xz.device)
            cur = torch.cuda.current_stream(xz.device)
            self._side_stream.wait_stream(cur)
            with torch.cuda.stream(self._side_stream):
                out_b = self._direction(xz, backward=True)
            out = self._direction(xz, backward=False)
            cur.wait_stream(self._side_stream)
            out_b.record_stream(cur)
        else:
            out = self._direction(xz, backward=False)
            out_b = self._direction(xz, backward=True)
        mix = 0.5 * out + 0.5 * out_b if self.if_devide_out else out + out_b      # reference :250-253
        # out / out_b are (b, e, l) views of (e, b, l) storage (selective_scan_interface._ebt), and so is mix: out_proj
        # reads it as the transposed (e, b*l) matrix instead of copying it to (b, l, e)
        if mix.is_cuda and mix.permute(1, 0, 2).is_contiguous():
            y = _OutProjFn.apply(mix, self.out_proj.weight, self.out_proj.bias)
        else:
            y = F.linear(mix.transpose(1, 2), self.out_proj.weight, self.out_proj.bias)
        if self.init_layer_scale is not None:
            y = y * self.gamma
        return y

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        dev = self.out_proj.weight.device
        conv_state = torch.zeros(batch_size, self.d_inner, self.d_conv, device=dev,
                                 dtype=self.conv1d.weight.dtype if dtype is None else dtype)
        ssm_state = torch.zeros(batch_size, self.d_inner, self.d_state, device=dev,
                                dtype=self.dt_proj.weight.dtype if dtype is None else dtype)
        return conv_state, ssm_state


class UniMamba(nn.Module):
    """Unidirectional mixer = what the reference imports as ``mamba_ssm.Mamba`` (modules/Conmamba.py:124;
    used by the Mamba decoder :854-862 and by causal encoders :580-584): in_proj -> mamba_inner_fn.
    state_dict keys: in_proj.weight, conv1d.{weight,bias}, x_proj.weight, dt_proj.{weight,bias}, A_log, D,
    out_proj.weight."""

    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, dt_rank="auto", dt_min=0.001, dt_max=0.1,
                 dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, conv_bias=True, bias=False,
                 use_fast_path=True, layer_idx=None, device=None, dtype=None):
        fk = {"device": device, "dtype": dtype}
        super().__init__()
        self.d_model, self.d_state, self.d_conv, self.expand = d_model, d_state, d_conv, expand
        self.d_inner = int(expand * d_model)
        self.dt_rank = math.ceil(d_model / 16) if dt_rank == "auto" else dt_rank
        self.use_fast_path, self.layer_idx = use_fast_path, layer_idx
        self.in_proj = nn.Linear(d_model, 2 * self.d_inner, bias=bias, **fk)
        self.conv1d = nn.Conv1d(self.d_inner, self.d_inner, d_conv, groups=self.d_inner, padding=d_conv - 1,
                                bias=conv_bias, **fk)
        self.activation = "silu"
        self.act = nn.SiLU()
        self.x_proj = nn.Linear(self.d_inner, self.dt_rank + 2 * d_state, bias=False, **fk)
        self.dt_proj = nn.Linear(self.dt_rank, self.d_inner, bias=True, **fk)
        _init_dt_proj(self.dt_proj, self.d_inner, self.dt_rank, dt_init, dt_scale, dt_min, dt_max, dt_init_floor, fk)
        self.A_log = _s4d_real_log(self.d_inner, d_state, device)
        self.D = nn.Parameter(torch.ones(self.d_inner, device=device))
        self.D._no_weight_decay = True
        self.out_proj = nn.Linear(self.d_inner, d_model, bias=bias, **fk)

    def allocate_inference_cache(self, batch_size, max_seqlen=None, dtype=None, **kwargs):
        """(conv_state (batch, d_inner, d_conv), ssm_state (batch, d_inner, d_state)), zero, fp32 (reference :367-383)."""
        dev = self.in_proj.weight.device
        return (torch.zeros(batch_size, self.d_inner, self.d_conv, device=dev, dtype=torch.float32),
                torch.zeros(batch_size, self.d_inner, self.d_state, device=dev, dtype=torch.float32))

    def step(self, hidden_states, conv_state, ssm_state):
        """One decoding step (reference bimamba.py:320-365): hidden_states (batch, 1, d_model) -> (out (batch, 1,
        d_model), conv_state, ssm_state); both states are updated in place by cm_causal_conv1d_update /
        cm_selective_state_update.  Feeding a sequence step by step from zero states reproduces ``forward``."""
        assert hidden_states.shape[1] == 1, "Only support decoding with 1 token at a time for now"
        from ... import ops
        xz = self.in_proj(hidden_states.squeeze(1))                                     # (B, 2E)
        x, z = xz.chunk(2, dim=-1)
        x = ops.causal_conv1d_update(x, conv_state, self.conv1d.weight, self.conv1d.bias, silu=True)
        x_db = self.x_proj(x)
        dt, Bm, Cm = torch.split(x_db, [self.dt_rank, self.d_state, self.d_state], dim=-1)
        dt = F.linear(dt, self.dt_proj.weight)                                          # bias is added in the kernel
        y = ops.selective_state_update(ssm_state, x, dt, -torch.exp(self.A_log.float()), Bm, Cm, self.D.float(), z=z,
                                       dt_bias=self.dt_proj.bias.float(), dt_softplus=True)
        return self.out_proj(y).unsqueeze(1), conv_state, ssm_state

    def forward(self, hidden_states, inference_params=None):
        if inference_params is not None:
            raise NotImplementedError("inference_params caches are not used by the ConMamba recipes: call step()")
        from . import mixer_rows
        if mixer_rows.supported(self, hidden_states):
            return mixer_rows.mixer_rows(self, hidden_states)
        batch, seqlen, _ = hidden_states.shape
        xz = (self.in_proj.weight @ hidden_states.reshape(batch * seqlen, -1).t()).reshape(-1, batch, seqlen)
        xz = xz.transpose(0, 1)
        if self.in_proj.bias is not None:
            xz = xz + self.in_proj.bias.to(xz.dtype)[None, :, None]
        A = -torch.exp(self.A_log.float())
        return mamba_inner_fn(xz, self.conv1d.weight, self.conv1d.bias, self.x_proj.weight,
                              self.dt_proj.weight, self.out_proj.weight, self.out_proj.bias, A, None, None,
                              self.D.float(), delta_bias=self.dt_proj.bias.float(), delta_softplus=True)


class Block(nn.Module):
    """Add -> LayerNorm -> mixer pre-norm block (reference :409-465, non-fused branch :445-449)."""

    def __init__(self, dim, mixer_cls, norm_cls=nn.LayerNorm, fused_add_norm=False, residual_in_fp32=False):
        super().__init__()
        if fused_add_norm:
            raise NotImplementedError("fused_add_norm (Triton in the reference, :451-460) is not on the ConMamba path")
        self.residual_in_fp32, self.fused_add_norm = residual_in_fp32, fused_add_norm
        self.mixer, self.norm = mixer_cls(dim), norm_cls(dim)

    def forward(self, hidden_states, residual=None, inference_params=None):
        residual = hidden_states + residual if residual is not None else hidden_states
        hidden_states = self.norm(residual.to(dtype=self.norm.weight.dtype))
        if self.residual_in_fp32:
            residual = residual.to(torch.float32)
        return self.mixer(hidden_states, inference_params=inference_params), residual

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        return self.mixer.allocate_inference_cache(batch_size, max_seqlen, dtype=dtype, **kwargs)
