#!/usr/bin/env python3
"""Round-3 golden vectors, produced by running the REFERENCE's own Python (build container only):

    python tests/golden/make_golden_r3.py

Same mechanism as make_golden.py / make_golden_r2.py (whose loaders this imports): the reference's modules are imported
from /root/reference with empty stubs for the CUDA wheels and restated stand-ins for the speechbrain names; nothing of
the reference is copied, fixtures hold tensors only.

  g_s2s_forward   reference TransformerASR (modules/TransformerASR.py:428-1054: constructor :674-743, forward :745-819,
                  decode :821-866, encode :868-929) with encoder_module 'conmamba' + decoder_module 'mamba' — i.e. the
                  reference's own TransformerInterface factory (modules/Transformer.py:629-789), PositionalEncoding
                  (:796-1022) and NormalizedEmbedding (:1650-1860) — at d_model 128, 2 ConMamba encoder layers + 2 Mamba
                  decoder layers, vocabulary 53, attention_type 'RelPosMHAXL' (what every ConMamba recipe sets,
                  hparams/S2S/conmambamamba_large.yaml:251-257), eval mode.  src (3, 41, 20, 32) 4-D as the CNN front end
                  hands it over, tgt (3, 11) token ids with padding.  Stored: encoder_out, decoder_out of forward(),
                  decode()'s prediction, encode()'s output, and the gradients of sum(decoder_out * w) w.r.t. src and three
                  parameters (embedding, first decoder in_proj, src Linear) from a train-mode pass with dropout 0.
                  Parameters come from tests/golden/synth.py (seeded; not committed).

speechbrain stand-ins used by this file (speechbrain 1.0.0 is neither in the reference tree nor installable: their
semantics are restated, "parity unpinned" for them as DESIGN.md says): nnet.linear.Linear (nn.Linear under .w),
nnet.containers.ModuleList (sequential, .layers), nnet.embedding.Embedding (nn.Embedding under .Embedding, padding_idx =
blank_id), dataio.dataio.length_to_mask, nnet.attention.RelPosEncXL (its output is computed and DISCARDED by the ConMamba
encoder, modules/TransformerASR.py:777-778 / Conmamba.py: pos_embs unused — the stand-in returns None).
"""
import importlib
import os
import sys

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _stub, load_reference, load_reference_conmamba, save   # noqa: E402
from synth import synth_input, synth_like                                       # noqa: E402

CFG = {"d_state": 16, "expand": 2, "d_conv": 4, "bidirectional": True}


def load_reference_transformer_asr(ssi, bim):
    cm = load_reference_conmamba(ssi, bim)
    sb = sys.modules["speechbrain"]
    nnet = sys.modules["speechbrain.nnet"]

    class Linear(nn.Module):                      # speechbrain.nnet.linear.Linear, restated
        def __init__(self, n_neurons, input_shape=None, input_size=None, bias=True, combine_dims=False):
            super().__init__()
            self.combine_dims = combine_dims
            self.w = nn.Linear(input_size, n_neurons, bias=bias)

        def forward(self, x):
            if x.ndim == 4 and self.combine_dims:
                x = x.reshape(x.shape[0], x.shape[1], x.shape[2] * x.shape[3])
            return self.w(x)

    class ModuleList(nn.Module):                  # speechbrain.nnet.containers.ModuleList, restated
        def __init__(self, *layers):
            super().__init__()
            self.layers = nn.ModuleList(layers)

        def forward(self, x):
            for layer in self.layers:
                x = layer(x)
            return x

    class Embedding(nn.Module):                   # speechbrain.nnet.embedding.Embedding, restated
        def __init__(self, num_embeddings, embedding_dim=128, consider_as_one_hot=False, blank_id=0):
            super().__init__()
            self.Embedding = nn.Embedding(num_embeddings, embedding_dim, padding_idx=blank_id)

        def forward(self, x):
            return self.Embedding(x.long())

    class RelPosEncXL(nn.Module):                 # output unused by ConMamba (see the header)
        def __init__(self, emb_dim):
            super().__init__()

        def forward(self, x):
            return None

    def length_to_mask(length, max_len=None, dtype=None, device=None):
        max_len = int(length.max().item()) if max_len is None else max_len
        mask = torch.arange(max_len, device=length.device)[None, :] < length[:, None]
        return mask.to(dtype if dtype is not None else length.dtype)

    nnet.linear = _stub("speechbrain.nnet.linear", Linear=Linear)
    nnet.containers = _stub("speechbrain.nnet.containers", ModuleList=ModuleList)
    nnet.embedding = _stub("speechbrain.nnet.embedding", Embedding=Embedding)
    nnet.CNN = _stub("speechbrain.nnet.CNN", Conv1d=object)
    sys.modules["speechbrain.nnet.attention"].RelPosEncXL = RelPosEncXL
    sb.dataio = _stub("speechbrain.dataio")
    sb.dataio.dataio = _stub("speechbrain.dataio.dataio", length_to_mask=length_to_mask)
    importlib.import_module("modules.Conformer")
    return importlib.import_module("modules.TransformerASR"), cm


def synth_params(module, seed):
    """synth.synth_like for the PARAMETERS only: the sinusoidal table of PositionalEncoding is a registered buffer ('.pe'
    keys of the state_dict) and must stay what the reference's constructor computed."""
    return {k: v for k, v in synth_like(module, seed).items() if not k.endswith(".pe")}


def make_s2s_forward(tasr):
    kw = dict(tgt_vocab=53, input_size=640, d_model=128, nhead=4, num_encoder_layers=2, num_decoder_layers=2, d_ffn=256,
              activation=nn.GELU, encoder_module="conmamba", decoder_module="mamba", attention_type="RelPosMHAXL",
              normalize_before=True, causal=False, mamba_config=dict(CFG))
    model = tasr.TransformerASR(dropout=0.1, **kw)
    missing = model.load_state_dict(synth_params(model, 1280), strict=False)
    assert not missing.unexpected_keys and all(k.endswith(".pe") for k in missing.missing_keys), missing
    model.eval()
    src = synth_input("g_s2s.src", (3, 41, 20, 32), 1280)
    gen = torch.Generator().manual_seed(1281)
    tgt = torch.randint(1, 53, (3, 11), generator=gen)
    tgt[1, 8:] = 0                                             # padding
    tgt[2, 5:] = 0
    wav_len = torch.tensor([1.0, 0.8, 0.6])
    with torch.no_grad():
        enc, dec = model(src, tgt, wav_len)
        pred, attn = model.decode(tgt, enc)
        enc_only = model.encode(src, wav_len)
    assert attn is None
    # gradients: train mode with dropout 0 (same parameters)
    mt = tasr.TransformerASR(dropout=0.0, **kw)
    mt.load_state_dict(synth_params(mt, 1280), strict=False)
    mt.train()
    s = src.clone().requires_grad_(True)
    _, d2 = mt(s, tgt, wav_len)
    w = synth_input("g_s2s.w", tuple(d2.shape), 1280)
    names = ["custom_tgt_module.layers.0.emb.Embedding.weight", "decoder.layers.0.self_mamba.in_proj.weight",
             "custom_src_module.layers.0.w.weight"]
    pd = dict(mt.named_parameters())
    grads = torch.autograd.grad((d2 * w).sum(), [s] + [pd[n] for n in names])
    cases = dict(tgt=tgt.to(torch.int32), wav_len=wav_len, encoder_out=enc, decoder_out=dec, decode_prediction=pred,
                 encode_out=enc_only, decoder_out_train=d2, dsrc=grads[0])
    for n, g in zip(names, grads[1:]):
        cases["g." + n] = g
    save("g_s2s_forward", **cases)


if __name__ == "__main__":
    ssi, bim = load_reference()
    tasr, cm = load_reference_transformer_asr(ssi, bim)
    make_s2s_forward(tasr)
