#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own Python.

Run in the build container only (needs /root/reference; the GPU box has neither the
reference nor this need — it consumes the committed .npz files):

    python tests/golden/make_golden.py

What is executed from the reference (nothing is copied; fixtures hold tensors only):
  * modules/mamba/selective_scan_interface.py : selective_scan_ref (:91-157) for G1/G2 and
    mamba_inner_ref (:641-675) for G3, imported by file path with EMPTY stub modules standing
    in for the three CUDA extensions it hard-imports (:14-16).  The stub ``causal_conv1d_fn``
    implements the conv semantics the reference itself defines at modules/mamba/bimamba.py:278-279
    (``act(conv1d(x)[..., :seqlen])``, padding d_conv-1, :83-91); ``selective_scan_fn`` is rebound
    to the reference's own ``selective_scan_ref``.
  * modules/mamba/bimamba.py : the reference ``Mamba`` (BiMamba v2) class, its initialisers and
    its ``forward`` (:176-253), with ``mamba_inner_fn_no_out_proj`` rebound to a wrapper over the
    reference's ``mamba_inner_ref`` (identity out_proj) because the shipped fast path needs CUDA.
  * modules/Conmamba.py : ConmambaEncoderLayer / ConmambaEncoder (:457-727), imported with stub
    speechbrain names.  The stub classes (PositionalwiseFeedForward, LayerNorm) are OUR restatement
    of speechbrain==1.0.0 (not in the reference tree) — G4 is flagged "speechbrain restated".

Golden groups: G1 scan fwd, G2 scan bwd (autograd through selective_scan_ref), G3 BiMamba-v2
mixer fwd + param grads, G4 encoder layer / 2-layer encoder, G5 CTC loss, K conv fwd/bwd.
"""
import importlib
import importlib.util
import os
import sys
import types
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

warnings.filterwarnings("ignore")
REF = os.environ.get("CONMAMBA_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _ref_conv_fn(x, weight, bias=None, activation=None):
    # semantics of reference bimamba.py:278-279 with the layer of :83-91
    w = weight.shape[-1]
    y = F.conv1d(x, weight.unsqueeze(1), bias, padding=w - 1, groups=x.shape[1])[..., : x.shape[-1]]
    return F.silu(y) if activation in ("silu", "swish") else y


def load_reference():
    _stub("causal_conv1d", causal_conv1d_fn=_ref_conv_fn, causal_conv1d_update=None)
    _stub("causal_conv1d_cuda")
    _stub("selective_scan_cuda")
    sys.path.insert(0, REF)
    ssi = importlib.import_module("modules.mamba.selective_scan_interface")
    ssi.selective_scan_fn = ssi.selective_scan_ref           # CUDA op -> the reference's CPU ref

    def inner_no_out_proj(xz, conv_w, conv_b, xw, dtw, A, B=None, C=None, D=None, delta_bias=None,
                          B_proj_bias=None, C_proj_bias=None, delta_softplus=True):
        e = xz.shape[1] // 2
        eye = torch.eye(e, dtype=xz.dtype)
        y = ssi.mamba_inner_ref(xz, conv_w, conv_b, xw, dtw, eye, None, A, B, C, D, delta_bias,
                                B_proj_bias, C_proj_bias, delta_softplus)
        return y.transpose(1, 2)                             # (b l e) -> (b e l)

    ssi.mamba_inner_fn_no_out_proj = inner_no_out_proj
    bim = importlib.import_module("modules.mamba.bimamba")
    bim.mamba_inner_fn_no_out_proj = inner_no_out_proj
    bim.selective_scan_fn = ssi.selective_scan_ref
    return ssi, bim


def load_reference_conmamba(ssi, bim):
    """Import modules/Conmamba.py with stub speechbrain / mamba_ssm names."""

    class PositionalwiseFeedForward(nn.Module):   # speechbrain 1.0.0 semantics, restated
        def __init__(self, d_ffn, input_size=None, dropout=0.0, activation=nn.ReLU):
            super().__init__()
            self.ffn = nn.Sequential(nn.Linear(input_size, d_ffn), activation(), nn.Dropout(dropout),
                                     nn.Linear(d_ffn, input_size))

        def forward(self, x):
            return self.ffn(x)

    class LayerNorm(nn.Module):                   # speechbrain.nnet.normalization.LayerNorm, restated
        def __init__(self, input_size=None, eps=1e-05, elementwise_affine=True):
            super().__init__()
            self.norm = nn.LayerNorm(input_size, eps=eps, elementwise_affine=elementwise_affine)

        def forward(self, x):
            return self.norm(x)

    class UniMamba(nn.Module):                    # mamba_ssm.Mamba twin: in_proj -> mamba_inner_ref
        def __init__(self, d_model, d_state=16, d_conv=4, expand=2):
            super().__init__()
            tmpl = bim.Mamba(d_model, d_state=d_state, d_conv=d_conv, expand=expand, bimamba_type="v2")
            for k in ("in_proj", "conv1d", "x_proj", "dt_proj", "out_proj"):
                setattr(self, k, getattr(tmpl, k))
            self.A_log, self.D = tmpl.A_log, tmpl.D
            self.dt_rank = tmpl.dt_rank

        def forward(self, h):
            b, l, d = h.shape
            xz = (self.in_proj.weight @ h.reshape(b * l, d).t()).reshape(-1, b, l).permute(1, 0, 2)
            A = -torch.exp(self.A_log.float())
            return ssi.mamba_inner_ref(xz, self.conv1d.weight, self.conv1d.bias, self.x_proj.weight,
                                       self.dt_proj.weight, self.out_proj.weight, self.out_proj.bias,
                                       A, None, None, self.D.float(), self.dt_proj.bias.float(), None, None, True)

    sb = _stub("speechbrain")
    nnet = _stub("speechbrain.nnet")
    sb.nnet = nnet
    nnet.activations = _stub("speechbrain.nnet.activations", Swish=nn.SiLU)
    nnet.attention = _stub("speechbrain.nnet.attention", MultiheadAttention=object,
                           PositionalwiseFeedForward=PositionalwiseFeedForward, RelPosMHAXL=object)
    nnet.hypermixing = _stub("speechbrain.nnet.hypermixing", HyperMixing=object)
    nnet.normalization = _stub("speechbrain.nnet.normalization", LayerNorm=LayerNorm)
    _stub("speechbrain.utils")
    _stub("speechbrain.utils.dynamic_chunk_training", DynChunkTrainConfig=object)
    _stub("mamba_ssm", Mamba=UniMamba)
    return importlib.import_module("modules.Conmamba")


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if v is None:
            continue
        if isinstance(v, torch.Tensor):
            v = v.detach().to(torch.float64 if v.dtype == torch.float64 else torch.float32).numpy()
        out[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB  keys={len(out)}")


def bf16_round(t):
    return t.to(torch.bfloat16).float()


def scan_inputs(gen, b, e, l, n, scale=1.0):
    u = torch.randn(b, e, l, generator=gen) * scale
    delta = torch.randn(b, e, l, generator=gen) * 0.5
    A = -torch.exp(torch.randn(e, n, generator=gen) * 0.3)
    Bm = torch.randn(b, n, l, generator=gen)
    Cm = torch.randn(b, n, l, generator=gen)
    D = torch.randn(e, generator=gen)
    z = torch.randn(b, e, l, generator=gen)
    bias = torch.randn(e, generator=gen) * 0.5 - 1.0
    return u, delta, A, Bm, Cm, D, z, bias


def make_g1(ssi):
    """G1: selective scan forward through the reference's selective_scan_ref."""
    gen = torch.Generator().manual_seed(3402)
    cases = {}
    shapes = {"tiny": (2, 8, 37, 16), "mid": (2, 16, 300, 16), "long": (1, 8, 2100, 16), "n8": (1, 4, 65, 8)}
    for tag, (b, e, l, n) in shapes.items():
        u, delta, A, Bm, Cm, D, z, bias = scan_inputs(gen, b, e, l, n)
        cases[f"{tag}_u"], cases[f"{tag}_delta"], cases[f"{tag}_A"] = u, delta, A
        cases[f"{tag}_B"], cases[f"{tag}_C"], cases[f"{tag}_D"] = Bm, Cm, D
        cases[f"{tag}_z"], cases[f"{tag}_bias"] = z, bias
        out, last = ssi.selective_scan_ref(u, delta, A, Bm, Cm, D, z, bias, True, True)
        cases[f"{tag}_out_full"], cases[f"{tag}_last_full"] = out, last
        if tag in ("tiny", "mid"):
            cases[f"{tag}_out_noz"] = ssi.selective_scan_ref(u, delta, A, Bm, Cm, D, None, bias, True)
            cases[f"{tag}_out_noD"] = ssi.selective_scan_ref(u, delta, A, Bm, Cm, None, z, bias, True)
            cases[f"{tag}_out_nobias"] = ssi.selective_scan_ref(u, delta, A, Bm, Cm, D, z, None, True)
            cases[f"{tag}_out_nosoftplus"] = ssi.selective_scan_ref(u, delta.abs() * 0.1, A, Bm, Cm, D, z, None, False)
            cases[f"{tag}_out_bare"] = ssi.selective_scan_ref(u, delta, A, Bm, Cm, None, None, None, True)
            # 4-D (b, g=1, n, l) B/C give the same numbers through a different code path (ssi.py:133-136)
            cases[f"{tag}_out_4d"] = ssi.selective_scan_ref(u, delta, A, Bm[:, None], Cm[:, None], D, z, bias, True)
            # reverse-time case: inputs flipped through the oracle, as bimamba.py:236-253 does
            f = lambda t: t.flip(-1)
            cases[f"{tag}_out_rev"] = f(ssi.selective_scan_ref(f(u), f(delta), A, f(Bm), f(Cm), D, f(z), bias, True))
            # bf16-input run: reference computes in fp32 and casts the output back to bf16 (ssi.py:106-108,156)
            ub, db, zb, Bb, Cb = (t.to(torch.bfloat16) for t in (u, delta, z, Bm, Cm))
            cases[f"{tag}_out_bf16"] = ssi.selective_scan_ref(ub, db, A, Bb, Cb, D, zb, bias, True).float()
    save("g1_scan_fwd", **cases)


def make_g2(ssi):
    """G2: backward = autograd through the reference's selective_scan_ref.  The reference casts its
    inputs to fp32 internally (ssi.py:107-108,122-123), so the golden gradients carry fp32 round-off;
    tests compare the fp64 analytic oracle and the HIP kernels against them at rtol 2e-3."""
    gen = torch.Generator().manual_seed(3403)
    cases = {}
    for tag, (b, e, l, n) in {"tiny": (2, 8, 37, 16), "mid": (2, 16, 150, 16), "long": (1, 8, 600, 16)}.items():
        u, delta, A, Bm, Cm, D, z, bias = (t.requires_grad_(True) for t in scan_inputs(gen, b, e, l, n))
        dout = torch.randn(b, e, l, generator=gen)
        out = ssi.selective_scan_ref(u, delta, A, Bm, Cm, D, z, bias, True)
        grads = torch.autograd.grad(out, (u, delta, A, Bm, Cm, D, z, bias), dout)
        for k, v in zip(("u", "delta", "A", "B", "C", "D", "z", "bias"), (u, delta, A, Bm, Cm, D, z, bias)):
            cases[f"{tag}_{k}"] = v.detach().float()
        cases[f"{tag}_dout"] = dout.float()
        cases[f"{tag}_out"] = out.detach().float()
        for k, gk in zip(("du", "ddelta", "dA", "dB", "dC", "dD", "dz", "dbias"), grads):
            cases[f"{tag}_{k}"] = gk.float()
        if tag == "tiny":   # no-z / no-D variant
            out2 = ssi.selective_scan_ref(u, delta, A, Bm, Cm, None, None, None, True)
            g2 = torch.autograd.grad(out2, (u, delta, A, Bm, Cm), dout)
            for k, gk in zip(("du", "ddelta", "dA", "dB", "dC"), g2):
                cases[f"{tag}_bare_{k}"] = gk.float()
    save("g2_scan_bwd", **cases)


def make_conv():
    """K: causal depthwise conv (+SiLU) fwd/bwd with the semantics of reference bimamba.py:83-91,278-279."""
    gen = torch.Generator().manual_seed(3404)
    cases = {}
    for tag, (b, e, l, w) in {"tiny": (2, 8, 37, 4), "mid": (2, 48, 333, 4), "w3": (1, 8, 20, 3), "short": (1, 8, 2, 4)}.items():
        x = torch.randn(b, e, l, generator=gen, dtype=torch.float64, requires_grad=True)
        wt = torch.randn(e, w, generator=gen, dtype=torch.float64, requires_grad=True)
        bs = torch.randn(e, generator=gen, dtype=torch.float64, requires_grad=True)
        dout = torch.randn(b, e, l, generator=gen, dtype=torch.float64)
        y = _ref_conv_fn(x, wt, bs, "silu")
        dx, dw, db = torch.autograd.grad(y, (x, wt, bs), dout)
        y_lin = _ref_conv_fn(x, wt, None, None)
        cases.update({f"{tag}_x": x, f"{tag}_w": wt, f"{tag}_b": bs, f"{tag}_dout": dout, f"{tag}_y": y,
                      f"{tag}_y_lin": y_lin, f"{tag}_dx": dx, f"{tag}_dw": dw, f"{tag}_db": db})
        cases = {k: v.detach().float() for k, v in cases.items()}
    save("k_conv", **cases)


def xavier_reinit(module):
    # reference modules/TransformerASR.py:1051-1054
    for p in module.parameters():
        if p.dim() > 1:
            nn.init.xavier_normal_(p)


def make_g3(ssi, bim):
    """G3: BiMamba-v2 mixer (reference class, reference forward) + grads."""
    cases = {}
    for tag, (d_model, b, l) in {"d144": (144, 2, 50)}.items():
        torch.manual_seed(3402)
        m = bim.Mamba(d_model, d_state=16, d_conv=4, expand=2, bimamba_type="v2")
        xavier_reinit(m)
        x = torch.randn(b, l, d_model, requires_grad=True)
        y = m(x)
        dy = torch.randn_like(y)
        params = dict(m.named_parameters())
        grads = torch.autograd.grad(y, [x] + list(params.values()), dy)
        cases[f"{tag}_x"], cases[f"{tag}_y"], cases[f"{tag}_dy"], cases[f"{tag}_dx"] = x, y, dy, grads[0]
        for (k, p), gk in zip(params.items(), grads[1:]):
            cases[f"{tag}_p.{k}"] = p
            cases[f"{tag}_g.{k}"] = gk
        # the fused inner op alone (forward direction parameters), fwd + input grad
        e = m.d_inner
        xz = torch.randn(b, 2 * e, l, requires_grad=True)
        A = -torch.exp(m.A_log.float())
        oz = ssi.mamba_inner_fn_no_out_proj(xz, m.conv1d.weight, m.conv1d.bias, m.x_proj.weight, m.dt_proj.weight,
                                            A, None, None, m.D.float(), m.dt_proj.bias.float(), None, None, True)
        doz = torch.randn_like(oz)
        cases[f"{tag}_inner_xz"], cases[f"{tag}_inner_out"], cases[f"{tag}_inner_dout"] = xz, oz, doz
        cases[f"{tag}_inner_dxz"] = torch.autograd.grad(oz, xz, doz)[0]
    save("g3_bimamba", **cases)


def make_g4(cm):
    """G4: ConmambaEncoderLayer and a 2-layer ConmambaEncoder (eval mode).  'speechbrain restated'."""
    cases = {}
    torch.manual_seed(3402)
    cfg = {"d_state": 16, "expand": 2, "d_conv": 4, "bidirectional": True}
    enc = cm.ConmambaEncoder(num_layers=2, d_model=80, d_ffn=192, kernel_size=31, activation=nn.GELU,
                             bias=True, dropout=0.1, causal=False, mamba_config=cfg)
    xavier_reinit(enc)
    enc.eval()
    x = torch.randn(2, 40, 80)
    with torch.no_grad():
        y_layer = enc.layers[0](x)
        y_enc, _ = enc(x)
    cases["x"], cases["y_layer0"], cases["y_enc"] = x, y_layer, y_enc
    for k, v in enc.state_dict().items():
        cases["p." + k] = v
    # training-mode-free gradient check of one layer (dropout=0 instance sharing the weights)
    enc.train()
    for mod in enc.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    xg = x.clone().requires_grad_(True)
    yl = enc.layers[0](xg)
    dy = torch.randn_like(yl)
    names = [k for k, _ in enc.layers[0].named_parameters()]
    grads = torch.autograd.grad(yl, [xg] + [p for _, p in enc.layers[0].named_parameters()], dy)
    cases["dy_layer0"], cases["dx_layer0"] = dy, grads[0]
    for k, gk in zip(names, grads[1:]):
        cases["g.layers.0." + k] = gk
    save("g4_encoder", **cases)

    # Mamba decoder layer (Conmamba.py:730-953), normalize_before=True as Transformer.py:778-787 passes it
    torch.manual_seed(3405)
    dec = cm.MambaDecoderLayer(d_model=64, d_ffn=128, activation=nn.ReLU, dropout=0.0, normalize_before=True,
                               mamba_config=dict(cfg))
    xavier_reinit(dec)
    dec.eval()
    tgt, mem = torch.randn(2, 7, 64), torch.randn(2, 29, 64)
    with torch.no_grad():
        out, _, _ = dec(tgt, mem)
    dcases = {"tgt": tgt, "memory": mem, "out": out}
    for k, v in dec.state_dict().items():
        dcases["p." + k] = v
    save("g4_decoder_layer", **dcases)


def make_g5():
    """G5: CTC loss in the form speechbrain's ctc_loss(reduction='batchmean') reduces to."""
    gen = torch.Generator().manual_seed(3406)
    logits = torch.randn(2, 40, 31, generator=gen, dtype=torch.float64, requires_grad=True)
    lp = logits.log_softmax(-1)
    targets = torch.randint(3, 31, (2, 9), generator=gen)
    in_rel = torch.tensor([1.0, 0.8])
    tg_rel = torch.tensor([1.0, 0.67])
    il = torch.round(in_rel * 40).int()
    tl = torch.round(tg_rel * 9).int()
    loss = F.ctc_loss(lp.transpose(0, 1), targets, il, tl, 0, reduction="sum", zero_infinity=True) / 2
    (g,) = torch.autograd.grad(loss, logits)
    save("g5_ctc", logits=logits.detach().float(), targets=targets.numpy(), in_rel=in_rel, tg_rel=tg_rel,
         loss=loss.detach().float(), dlogits=g.float())


if __name__ == "__main__":
    ssi, bim = load_reference()
    make_g1(ssi)
    make_g2(ssi)
    make_conv()
    make_g3(ssi, bim)
    cm = load_reference_conmamba(ssi, bim)
    make_g4(cm)
    make_g5()
