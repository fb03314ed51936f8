"""``modules.mamba.mamba_blocks`` of the reference (mamba_blocks.py:22-49, 111-251): a plain stack of pre-norm Mamba
blocks.  No ASR recipe instantiates it; the operator-API names are kept so that code importing
them keeps working on this package.  Built on this package's mixers (``bimamba.Mamba`` / ``UniMamba``, HIP kernels
underneath); the reference's Triton "fused add + norm" is the same arithmetic done with torch ops here, so
``fused_add_norm`` only selects where the residual add happens, not what is computed."""
from __future__ import annotations

import math
from functools import partial

import torch
import torch.nn as nn

from .bimamba import Mamba as BiMamba
from .bimamba import UniMamba as Mamba


class RMSNorm(nn.Module):
    """y = x / sqrt(mean(x^2) + eps) * weight  (the norm the reference takes from mamba_ssm's Triton ops, :16-19)."""

    def __init__(self, dim, eps=1e-5, device=None, dtype=None):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim, device=device, dtype=dtype))
        self.register_parameter("bias", None)

    def forward(self, x):
        xf = x.float()
        return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.eps)).to(x.dtype) * self.weight


class _PreNormBlock(nn.Module):
    """Add -> norm -> mixer, returning (mixer output, residual) as bimamba.Block does (reference bimamba.py:409-465)."""

    def __init__(self, dim, mixer_cls, norm_cls, residual_in_fp32=False):
        super().__init__()
        self.residual_in_fp32 = residual_in_fp32
        self.mixer, self.norm = mixer_cls(dim), norm_cls(dim)

    def forward(self, hidden_states, residual=None, inference_params=None):
        residual = hidden_states if residual is None else hidden_states + residual
        hidden_states = self.norm(residual.to(dtype=self.norm.weight.dtype))
        if self.residual_in_fp32:
            residual = residual.to(torch.float32)
        return self.mixer(hidden_states), residual


def create_block(d_model, ssm_cls=None, ssm_cfg=None, norm_epsilon=1e-5, rms_norm=False, residual_in_fp32=False,
                 fused_add_norm=True, layer_idx=None, device=None, dtype=None):
    """One pre-norm block around ``ssm_cls`` (reference signature, mamba_blocks.py:22-33)."""
    ssm_cfg = dict(ssm_cfg or {})
    mixer_cls = partial(ssm_cls, **ssm_cfg)
    norm_cls = partial(RMSNorm if rms_norm else nn.LayerNorm, eps=norm_epsilon, device=device, dtype=dtype)
    block = _PreNormBlock(d_model, mixer_cls, norm_cls, residual_in_fp32=residual_in_fp32)
    block.layer_idx, block.fused_add_norm = layer_idx, fused_add_norm
    return block


def _init_weights(module, n_layer, initializer_range=0.02, rescale_prenorm_residual=True, n_residuals_per_layer=1):
    """GPT-2 style initialisation the reference applies to the stack (:53-82): zero Linear biases, N(0, range)
    embeddings, out_proj / fc2 weights rescaled by 1/sqrt(residual branches)."""
    if isinstance(module, nn.Linear):
        if module.bias is not None and not getattr(module.bias, "_no_reinit", False):
            nn.init.zeros_(module.bias)
    elif isinstance(module, nn.Embedding):
        nn.init.normal_(module.weight, std=initializer_range)
    if rescale_prenorm_residual:
        for name, p in module.named_parameters():
            if name in ("out_proj.weight", "fc2.weight"):
                nn.init.kaiming_uniform_(p, a=math.sqrt(5))
                with torch.no_grad():
                    p /= math.sqrt(n_residuals_per_layer * n_layer)


class MambaBlocksSequential(nn.Module):
    """``n_mamba`` pre-norm (Bi)Mamba blocks and a final norm (reference :132-251); input and output (batch, time, d_model)."""

    def __init__(self, n_mamba: int, bidirectional: bool, d_model: int, d_state: int = 16, expand: int = 2, d_conv: int = 4,
                 dt_rank="auto", conv_bias: bool = True, bias: bool = False, fused_add_norm: bool = True, rms_norm: bool = False,
                 norm_epsilon: float = 1e-5, initializer_cfg=None, residual_in_fp32=False, use_simple_block=False):
        super().__init__()
        if use_simple_block:
            raise NotImplementedError("use_simple_block (LnMambaAdd, reference :85-108) is not provided")
        self.residual_in_fp32, self.bidirectional, self.fused_add_norm = residual_in_fp32, bidirectional, fused_add_norm
        ssm_cfg = dict(d_state=d_state, expand=expand, d_conv=d_conv, dt_rank=dt_rank, conv_bias=conv_bias, bias=bias)
        if bidirectional:
            ssm_cfg["bimamba_type"] = "v2"
        self.layers = nn.Sequential(*[
            create_block(d_model, ssm_cls=BiMamba if bidirectional else Mamba, ssm_cfg=ssm_cfg, norm_epsilon=norm_epsilon,
                         rms_norm=rms_norm, residual_in_fp32=residual_in_fp32, fused_add_norm=fused_add_norm, layer_idx=i)
            for i in range(n_mamba)])
        self.norm_f = (RMSNorm if rms_norm else nn.LayerNorm)(d_model, eps=norm_epsilon)
        self.apply(partial(_init_weights, n_layer=n_mamba, **(initializer_cfg or {})))

    def forward(self, x, inference_params=None):
        if inference_params is not None:
            raise NotImplementedError("stateful decoding is not on the ConMamba path")
        hidden_states, residual = x, None
        for layer in self.layers:
            hidden_states, residual = layer(hidden_states, residual)
        residual = hidden_states if residual is None else hidden_states + residual
        return self.norm_f(residual.to(dtype=self.norm_f.weight.dtype))
