"""ASR wrapper around the ConMamba encoder (and Mamba decoder) — the ConMamba/Mamba branches of the
reference's modules/TransformerASR.py (:674-743 constructor, :745-819 forward, :868-929 encode,
:1051-1054 Xavier re-init, :1057-1105 EncoderWrapper) and of the factory in modules/Transformer.py
(:740-758 encoder_module == 'conmamba', :778-787 decoder_module == 'mamba').

Only these branches are provided: the attention model families (transformer / conformer / branchformer
encoders, transformer decoder) are out of the hot path and raise NotImplementedError.
state_dict keys follow the reference: custom_src_module.layers.0.w.{weight,bias}, encoder.*, decoder.*,
custom_tgt_module.layers.0.emb.Embedding.weight.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from ..sb_compat import Linear, ModuleList, Swish
from .Conmamba import ConmambaEncoder, MambaDecoder


class PositionalEncoding(nn.Module):
    """Fixed sinusoidal table (reference modules/Transformer.py:796-1022); returns (1, T, D)."""

    def __init__(self, input_size, max_len=2500):
        super().__init__()
        pe = torch.zeros(max_len, input_size)
        pos = torch.arange(0, max_len).unsqueeze(1).float()
        den = torch.exp(torch.arange(0, input_size, 2).float() * -(math.log(10000.0) / input_size))
        pe[:, 0::2] = torch.sin(pos * den)
        pe[:, 1::2] = torch.cos(pos * den)
        self.register_buffer("pe", pe.unsqueeze(0))

    def forward(self, x):
        return self.pe[:, : x.size(1)].clone().detach()


class _Embedding(nn.Module):
    def __init__(self, num_embeddings, embedding_dim, blank_id=0):
        super().__init__()
        self.Embedding = nn.Embedding(num_embeddings, embedding_dim, padding_idx=blank_id)

    def forward(self, x):
        return self.Embedding(x.long())


class NormalizedEmbedding(nn.Module):
    """Embedding scaled by sqrt(d_model) (reference modules/Transformer.py:1650-1860)."""

    def __init__(self, d_model, vocab):
        super().__init__()
        self.emb = _Embedding(num_embeddings=vocab, embedding_dim=d_model, blank_id=0)
        self.d_model = d_model

    def forward(self, x):
        return self.emb(x) * math.sqrt(self.d_model)


def length_to_mask(length, max_len=None):
    max_len = int(length.max()) if max_len is None else max_len
    return torch.arange(max_len, device=length.device)[None, :] < length[:, None]


class TransformerASR(nn.Module):
    def __init__(self, tgt_vocab, input_size, d_model=512, nhead=8, num_encoder_layers=6, num_decoder_layers=6,
                 d_ffn=2048, dropout=0.1, activation=nn.ReLU, positional_encoding="fixed_abs_sine",
                 normalize_before=False, kernel_size: Optional[int] = 31, bias: Optional[bool] = True,
                 encoder_module: Optional[str] = "transformer", decoder_module: Optional[str] = "transformer",
                 conformer_activation=Swish, branchformer_activation=nn.GELU, attention_type: Optional[str] = "regularMHA",
                 max_length: Optional[int] = 2500, causal: Optional[bool] = True, csgu_linear_units=3072,
                 gate_activation=nn.Identity, use_linear_after_conv=False, mamba_config=None):
        super().__init__()
        assert num_encoder_layers + num_decoder_layers > 0
        if attention_type != "RelPosMHAXL":
            # every ConMamba recipe sets RelPosMHAXL (hparams/CTC/conmamba_large.yaml:164), whose branch adds no
            # positional encoding to src (reference :777-778 computes RelPosEncXL and ConMamba discards it); the
            # fixed_abs_sine branch (reference :779-781, 797-800) would add one and is not built: fail loudly
            raise NotImplementedError(f"attention_type={attention_type!r}: only 'RelPosMHAXL' (the ConMamba recipes' setting) "
                                      "is implemented; other types add sinusoidal positions to src in the reference")
        self.causal, self.attention_type, self.positional_encoding_type = causal, attention_type, positional_encoding
        self.num_decoder_layers = num_decoder_layers
        if encoder_module != "conmamba":
            raise NotImplementedError(f"encoder_module={encoder_module!r}: only 'conmamba' is on the MI355X hot path")
        assert normalize_before, "normalize_before must be True for Conmamba"          # reference Transformer.py:752
        # the factory hands branchformer_activation (GELU by default) to ConMamba, reference Transformer.py:746
        self.encoder = ConmambaEncoder(num_layers=num_encoder_layers, d_model=d_model, d_ffn=d_ffn, dropout=dropout,
                                       activation=branchformer_activation, kernel_size=kernel_size, bias=bias,
                                       causal=causal, mamba_config=mamba_config)
        if num_decoder_layers > 0:
            if decoder_module != "mamba":
                raise NotImplementedError(f"decoder_module={decoder_module!r}: only 'mamba' has a HIP path")
            self.decoder = MambaDecoder(num_layers=num_decoder_layers, d_ffn=d_ffn, d_model=d_model,
                                        activation=activation, dropout=dropout, normalize_before=normalize_before,
                                        mamba_config=mamba_config)
            self.positional_encoding_decoder = PositionalEncoding(d_model, max_length)
            self.custom_tgt_module = ModuleList(NormalizedEmbedding(d_model, tgt_vocab))
        self.custom_src_module = ModuleList(Linear(input_size=input_size, n_neurons=d_model, bias=True,
                                                   combine_dims=False), nn.Dropout(dropout))
        self._init_params()

    def _init_params(self):
        # reference :1051-1054 — overwrites every >=2-D parameter, including A_log / conv / dt_proj weights
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_normal_(p)

    def _prep_src(self, src):
        if src.dim() == 4:
            b, t, c1, c2 = src.shape
            src = src.reshape(b, t, c1 * c2)
        return self.custom_src_module(src)

    def encode(self, src, wav_len=None, pad_idx=0, dynchunktrain_config=None):
        """(B, T, F[, C]) -> encoder_out (B, T, D).  ConMamba ignores masks and positional embeddings (the
        reference computes RelPosEncXL here and discards it, :915-916; that dead work is skipped)."""
        out, _ = self.encoder(src=self._prep_src(src), src_mask=None, src_key_padding_mask=None, pos_embs=None,
                              dynchunktrain_config=dynchunktrain_config)
        return out

    def forward(self, src, tgt, wav_len=None, pad_idx=0):
        encoder_out = self.encode(src, wav_len, pad_idx)
        if self.num_decoder_layers == 0:
            return encoder_out, None
        t = self.custom_tgt_module(tgt)
        t = t + self.positional_encoding_decoder(t)                      # attention_type RelPosMHAXL branch, :793-796
        decoder_out, _, _ = self.decoder(tgt=t, memory=encoder_out)
        return encoder_out, decoder_out

    @torch.no_grad()
    def decode(self, tgt, encoder_out, enc_len=None):
        t = self.custom_tgt_module(tgt)
        t = t + self.positional_encoding_decoder(t)
        prediction, _, attn = self.decoder(t, encoder_out)
        return prediction, attn[-1]


class EncoderWrapper(nn.Module):
    def __init__(self, transformer, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.transformer = transformer

    def forward(self, x, wav_lens=None, pad_idx=0, **kwargs):
        return self.transformer.encode(x, wav_lens, pad_idx, **kwargs)
