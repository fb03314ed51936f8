"""ConMamba encoder / Mamba decoder modules — drop-in for the reference's modules/Conmamba.py:
same class names, constructor signatures, forward signatures/returns and state_dict keys
(SURVEY.md §8b), with the Mamba mixers running on the HIP operator API.

    ConvolutionModule      reference :182-454  (non-chunked path :439-449)
    ConmambaEncoderLayer   reference :457-650
    ConmambaEncoder        reference :653-727
    MambaDecoderLayer      reference :730-953
    MambaDecoder           reference :956-1031
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

import os

from .. import ops
from ..sb_compat import LayerNorm, PositionalwiseFeedForward, RowsLayerNorm, RowsLinear, Swish, linear_rows
from .mamba.bimamba import Mamba as BiMamba
from .mamba.bimamba import UniMamba as Mamba

FFN_RESIDUAL_SCALE = 0.5          # reference ConMambaConstants :161
FINAL_NORM_EPS = 1e-6             # reference ConMambaConstants :166


# depthwise stage of the ConvolutionModule on cm_dwconv1d_fwd / _bwd when on the GPU; CM_NATIVE_DWCONV=0 = nn.Conv1d
USE_NATIVE_DWCONV = os.environ.get("CM_NATIVE_DWCONV", "1") == "1"


class ConvolutionModule(nn.Module):
    """LayerNorm -> pointwise conv (D -> 2D) -> GLU -> depthwise conv k -> [chomp if causal] -> LayerNorm ->
    activation -> Linear -> Dropout.  The dynamic-chunk branch of the reference (:314-437) is not reached by
    ConMamba (the encoder layer has no streaming config) and raises here."""

    def __init__(self, input_size, kernel_size=31, bias=True, activation=Swish, dropout=0.0, causal=False,
                 dilation=1):
        super().__init__()
        self.kernel_size, self.causal, self.dilation = kernel_size, causal, dilation
        span = (kernel_size - 1) * 2 ** (dilation - 1)
        self.padding = span if causal else span // 2
        self.layer_norm = RowsLayerNorm(input_size)
        self.bottleneck = nn.Sequential(nn.Conv1d(input_size, 2 * input_size, kernel_size=1, stride=1, bias=bias),
                                        nn.GLU(dim=1))
        self.conv = nn.Conv1d(input_size, input_size, kernel_size=kernel_size, stride=1, padding=self.padding,
                              dilation=dilation, groups=input_size, bias=bias)
        self.after_conv = nn.Sequential(RowsLayerNorm(input_size), activation(),
                                        RowsLinear(input_size, input_size, bias=bias), nn.Dropout(dropout))

    def forward(self, x, mask: Optional[torch.Tensor] = None, dynchunktrain_config=None):
        if dynchunktrain_config is not None:
            raise NotImplementedError("dynamic chunk training is not used on the ConMamba path")
        if x.is_cuda and USE_NATIVE_DWCONV and self.dilation == 1 and self.kernel_size <= 32:
            # channels-last all the way: the pointwise conv (kernel 1) is a GEMM on (batch, time, channel) rows, GLU runs
            # over the last axis and the depthwise conv + its autograd are cm_dwconv_cl_fwd / _bwd -- no transposing
            # copies.  'same' zero padding, or all padding in front for the causal variant (= the reference's
            # pad-then-chomp).
            pw = self.bottleneck[0]
            out = F.glu(linear_rows(self.layer_norm(x), pw.weight.squeeze(-1), pw.bias), dim=-1)
            if out.dtype in (torch.bfloat16, torch.float32):
                out = ops.DepthwiseConvClFn.apply(out, self.conv.weight, self.conv.bias,
                                                  self.kernel_size - 1 if self.causal else self.kernel_size // 2)
            else:
                out = self.conv(out.transpose(1, 2))
                out = (out[..., : -self.padding] if self.causal else out).transpose(1, 2)
            out = self.after_conv(out)
            if mask is not None:
                out.masked_fill_(mask, 0.0)
            return out
        out = self.conv(self.bottleneck(self.layer_norm(x).transpose(1, 2)))
        if self.causal:
            out = out[..., : -self.padding]
        out = self.after_conv(out.transpose(1, 2))
        if mask is not None:
            out.masked_fill_(mask, 0.0)
        return out


class ConmambaEncoderLayer(nn.Module):
    def __init__(self, d_model, d_ffn, kernel_size=31, activation=Swish, bias=True, dropout=0.0, causal=False,
                 mamba_config=None):
        super().__init__()
        assert mamba_config is not None
        bidirectional = mamba_config.pop("bidirectional")           # shared mutable dict, reference :579-591
        try:
            if causal or not bidirectional:
                self.mamba = Mamba(d_model=d_model, **mamba_config)
            else:
                self.mamba = BiMamba(d_model=d_model, bimamba_type="v2", **mamba_config)
        finally:
            mamba_config["bidirectional"] = bidirectional
        self.convolution_module = ConvolutionModule(d_model, kernel_size, bias, activation, dropout, causal=causal)

        def ffn():
            return nn.Sequential(RowsLayerNorm(d_model),
                                 PositionalwiseFeedForward(d_ffn=d_ffn, input_size=d_model, dropout=dropout,
                                                           activation=activation),
                                 nn.Dropout(dropout))

        self.ffn_module1 = ffn()
        self.ffn_module2 = ffn()
        self.norm1 = LayerNorm(d_model)
        self.norm2 = LayerNorm(d_model)
        # these four norms feed a projection and nothing else (feed-forward Linear, BiMamba in_proj, pointwise conv): under
        # autocast their kernel stores the projection's operand dtype itself (sb_compat.RowsLayerNorm.low_out)
        for ln in (self.ffn_module1[0], self.ffn_module2[0], self.norm1.norm, self.convolution_module.layer_norm):
            ln.low_out = True
        self.drop = nn.Dropout(dropout)

    def _ffn(self, mod, x):
        """x + 0.5 * ffn_module(x): one autograd node on rows where its kernels apply (modules/ffn_rows.py), else the module tree."""
        from . import ffn_rows
        ln, pff, drop = mod[0], mod[1], mod[2]
        if ffn_rows.supported(x, ln, pff.ffn[0], pff.ffn[1], pff.ffn[3]):
            return ffn_rows.ffn_rows(x, ln, pff.ffn[0], pff.ffn[2], pff.ffn[3], drop, FFN_RESIDUAL_SCALE)
        return x + FFN_RESIDUAL_SCALE * mod(x)

    def forward(self, x, src_mask=None, src_key_padding_mask=None, pos_embs=None, dynchunktrain_config=None):
        # the reference computes a conv mask and then discards it (:631-635): padding is NOT masked
        x = self._ffn(self.ffn_module1, x)
        from .mamba import mixer_rows
        if mixer_rows.block_supported(self.mamba, self.norm1.norm, x):
            x = mixer_rows.mixer_rows(self.mamba, x, ln=self.norm1.norm)                # x + mamba(norm1(x)) as one node on rows
        else:
            x = self.mamba(self.norm1(x)) + x
        from . import convmod_rows
        if dynchunktrain_config is None and convmod_rows.supported(self.convolution_module, x):
            x = convmod_rows.convmod_rows(self.convolution_module, x)         # x + convolution_module(x) as one node on rows
        else:
            x = x + self.convolution_module(x, None, dynchunktrain_config=dynchunktrain_config)
        return self.norm2(self._ffn(self.ffn_module2, x))


class ConmambaEncoder(nn.Module):
    def __init__(self, num_layers, d_model, d_ffn, kernel_size=31, activation=Swish, bias=True, dropout=0.0,
                 causal=False, mamba_config=None):
        super().__init__()
        self.layers = nn.ModuleList([
            ConmambaEncoderLayer(d_model=d_model, d_ffn=d_ffn, dropout=dropout, activation=activation,
                                 kernel_size=kernel_size, bias=bias, causal=causal, mamba_config=mamba_config)
            for _ in range(num_layers)])
        self.norm = LayerNorm(d_model, eps=FINAL_NORM_EPS)

    def forward(self, src, src_mask=None, src_key_padding_mask=None, pos_embs=None, dynchunktrain_config=None):
        if (not torch.is_grad_enabled()) and (not self.training) and src.is_cuda and dynchunktrain_config is None:
            from .. import fused
            if all(fused.supports(layer) for layer in self.layers):
                return fused.encoder_forward(self, src), None            # fused channels-last inference path
        out = src
        for layer in self.layers:
            out = layer(out, src_mask=src_mask, src_key_padding_mask=src_key_padding_mask, pos_embs=pos_embs,
                        dynchunktrain_config=dynchunktrain_config)
        return self.norm(out), None


class MambaDecoderLayer(nn.Module):
    def __init__(self, d_model, d_ffn, activation=nn.ReLU, dropout=0.0, normalize_before=False, mamba_config=None):
        super().__init__()
        assert mamba_config is not None
        bidirectional = mamba_config.pop("bidirectional")
        try:
            self.self_mamba = Mamba(d_model=d_model, **mamba_config)
            self.cross_mamba = Mamba(d_model=d_model, **mamba_config)
        finally:
            mamba_config["bidirectional"] = bidirectional
        self.pos_ffn = PositionalwiseFeedForward(d_ffn=d_ffn, input_size=d_model, dropout=dropout,
                                                 activation=activation)
        self.norm1 = LayerNorm(d_model, eps=FINAL_NORM_EPS)
        self.norm2 = LayerNorm(d_model, eps=FINAL_NORM_EPS)
        self.norm3 = LayerNorm(d_model, eps=FINAL_NORM_EPS)
        self.dropout1, self.dropout2, self.dropout3 = nn.Dropout(dropout), nn.Dropout(dropout), nn.Dropout(dropout)
        self.normalize_before = normalize_before

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None, pos_embs_tgt=None, pos_embs_src=None):
        pre = self.normalize_before
        h = self.norm1(tgt) if pre else tgt
        tgt = tgt + self.dropout1(self.self_mamba(h))
        if not pre:
            tgt = self.norm1(tgt)
        h = self.norm2(tgt) if pre else tgt
        # "cross attention": one causal scan over [memory ; tgt], keep the tgt positions (reference :934)
        tgt = tgt + self.dropout2(self.cross_mamba(torch.cat([memory, h], dim=1))[:, -h.shape[1]:])
        if not pre:
            tgt = self.norm2(tgt)
        from . import ffn_rows
        if pre and ffn_rows.supported(tgt, self.norm3.norm, self.pos_ffn.ffn[0], self.pos_ffn.ffn[1], self.pos_ffn.ffn[3]):
            return ffn_rows.ffn_rows(tgt, self.norm3.norm, self.pos_ffn.ffn[0], self.pos_ffn.ffn[2], self.pos_ffn.ffn[3], self.dropout3, 1.0), None, None
        h = self.norm3(tgt) if pre else tgt
        tgt = tgt + self.dropout3(self.pos_ffn(h))
        if not pre:
            tgt = self.norm3(tgt)
        return tgt, None, None


class MambaDecoder(nn.Module):
    def __init__(self, num_layers, d_model, d_ffn, activation=nn.ReLU, dropout=0.0, normalize_before=False,
                 mamba_config=None):
        super().__init__()
        self.layers = nn.ModuleList([
            MambaDecoderLayer(d_model=d_model, d_ffn=d_ffn, activation=activation, dropout=dropout,
                              normalize_before=normalize_before, mamba_config=mamba_config)
            for _ in range(num_layers)])
        self.norm = LayerNorm(d_model, eps=FINAL_NORM_EPS)

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None, pos_embs_tgt=None, pos_embs_src=None):
        out = tgt
        for layer in self.layers:
            out, _, _ = layer(out, memory, tgt_mask=tgt_mask, memory_mask=memory_mask,
                              tgt_key_padding_mask=tgt_key_padding_mask,
                              memory_key_padding_mask=memory_key_padding_mask, pos_embs_tgt=pos_embs_tgt,
                              pos_embs_src=pos_embs_src)
        return self.norm(out), [None], [None]
