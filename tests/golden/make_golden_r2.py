#!/usr/bin/env python3
"""Round-2 golden vectors, produced by running the REFERENCE's own Python (build container only):

    python tests/golden/make_golden_r2.py

Same mechanism as make_golden.py (whose loaders this imports): the reference's modules are imported from
/root/reference with empty stubs for the CUDA wheels; nothing of the reference is copied, fixtures hold tensors.

  g4_large        reference ConmambaEncoder (modules/Conmamba.py:653-727), 2 layers at the BENCHMARK's dims
                  (d_model 256, d_ffn 1024, E 512, R 16), input (16, 100, 256): output of layer 0 and of the
                  encoder.  Parameters and input come from tests/golden/synth.py (seeded; 3.5 M parameters are
                  not committed), only the reference's outputs are stored.
  g3_d256         reference bimamba.Mamba (BiMamba v2, modules/mamba/bimamba.py:176-253) at d_model 256:
                  output, input gradient and every parameter gradient (seeded parameters as above).
  g3_inner_outproj  reference mamba_inner_ref WITH out_proj (selective_scan_interface.py:641-675; the math of
                  MambaInnerFn :297-439): output + gradients of every argument.
  g3_bimamba_v1   reference bimamba_inner_ref (v1: shared conv / projections, A_b scan on flipped tensors,
                  selective_scan_interface.py:678-714) on g3_inner_outproj's inputs: output + gradients.
  g4_decoder_stack  reference MambaDecoderLayer gradients and a 2-layer MambaDecoder incl. its final norm
                  (modules/Conmamba.py:730-1031): outputs + gradients w.r.t. tgt, memory and every parameter.
  g_step          reference bimamba.Mamba.step (modules/mamba/bimamba.py:320-365) in its pure-torch fallback
                  (causal_conv1d_update / selective_state_update are None under the stubs): token-by-token
                  outputs and the final conv / ssm states.
"""
import os
import sys

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import load_reference, load_reference_conmamba, save, xavier_reinit   # noqa: E402
from synth import synth_input, synth_like                                               # noqa: E402

CFG = {"d_state": 16, "expand": 2, "d_conv": 4, "bidirectional": True}


def make_g4_large(cm):
    enc = cm.ConmambaEncoder(num_layers=2, d_model=256, d_ffn=1024, kernel_size=31, activation=nn.GELU, bias=True,
                             dropout=0.1, causal=False, mamba_config=dict(CFG))
    enc.load_state_dict(synth_like(enc, 256), strict=True)
    enc.eval()
    x = synth_input("g4_large.x", (16, 100, 256), 256)
    with torch.no_grad():
        y0 = enc.layers[0](x)
        y, _ = enc(x)
    save("g4_large", y_layer0=y0, y_enc=y)


def make_g3_d256(bim):
    m = bim.Mamba(256, d_state=16, d_conv=4, expand=2, bimamba_type="v2")
    m.load_state_dict(synth_like(m, 2560), strict=True)
    x = synth_input("g3_d256.x", (2, 50, 256), 2560).requires_grad_(True)
    y = m(x)
    dy = synth_input("g3_d256.dy", tuple(y.shape), 2560)
    named = list(m.named_parameters())
    grads = torch.autograd.grad(y, [x] + [p for _, p in named], dy)
    cases = {"y": y, "dx": grads[0]}
    for (k, _), g in zip(named, grads[1:]):
        cases["g." + k] = g
    save("g3_d256", **cases)


def make_inner_outproj(ssi):
    gen = torch.Generator().manual_seed(3411)
    d_model, e, r, n, b, l = 64, 128, 4, 16, 2, 45
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).requires_grad_(True)
    xz = rn(b, 2 * e, l)
    conv_w, conv_b = rn(e, 1, 4, sc=0.5), rn(e, sc=0.1)
    xw, dtw = rn(r + 2 * n, e, sc=e ** -0.5), rn(e, r, sc=r ** -0.5)
    ow, ob = rn(d_model, e, sc=e ** -0.5), rn(d_model, sc=0.1)
    A = (-torch.exp(torch.randn(e, n, generator=gen) * 0.3)).requires_grad_(True)
    D, bias = rn(e), (torch.randn(e, generator=gen) * 0.5 - 1.0).requires_grad_(True)
    out = ssi.mamba_inner_ref(xz, conv_w, conv_b, xw, dtw, ow, ob, A, None, None, D, bias, None, None, True)
    dout = torch.randn(out.shape, generator=gen)
    args = dict(xz=xz, conv_w=conv_w, conv_b=conv_b, x_proj_w=xw, dt_proj_w=dtw, out_proj_w=ow, out_proj_b=ob, A=A, D=D,
                delta_bias=bias)
    grads = torch.autograd.grad(out, list(args.values()), dout)
    cases = {k: v for k, v in args.items()}
    cases.update(out=out, dout=dout)
    for k, g in zip(args, grads):
        cases["d_" + k] = g
    save("g3_inner_outproj", **cases)
    # v1 bidirectional composition (selective_scan_interface.py:678-714; the math of BiMambaInnerFn :442-608): shared conv /
    # projections, second scan on flipped tensors with A_b, summed, out_proj
    A_b = (-torch.exp(torch.randn(e, n, generator=gen) * 0.3)).requires_grad_(True)
    out = ssi.bimamba_inner_ref(xz, conv_w, conv_b, xw, dtw, ow, ob, A, A_b, None, None, D, bias, None, None, True)
    args["A_b"] = A_b
    grads = torch.autograd.grad(out, list(args.values()), dout)
    cases = {"A_b": A_b, "out": out}
    for k, g in zip(args, grads):
        cases["d_" + k] = g
    save("g3_bimamba_v1", **cases)


def make_decoder_stack(cm):
    torch.manual_seed(3412)
    dec = cm.MambaDecoder(num_layers=2, d_model=64, d_ffn=128, activation=nn.ReLU, dropout=0.0, normalize_before=True,
                          mamba_config=dict(CFG))
    xavier_reinit(dec)
    for k, p in dec.named_parameters():                       # make 1-D parameters non-trivial
        if p.dim() == 1 and "dt_proj" not in k:
            with torch.no_grad():
                p.add_(0.1 * torch.randn_like(p))
    dec.train()                                               # dropout p = 0
    tgt = torch.randn(2, 9, 64, requires_grad=True)
    mem = torch.randn(2, 31, 64, requires_grad=True)
    cases = {"tgt": tgt, "memory": mem}
    for k, v in dec.state_dict().items():
        cases["p." + k] = v
    # one layer: output + gradients of inputs and parameters (Conmamba.py:914-953)
    layer = dec.layers[0]
    out_l, _, _ = layer(tgt, mem)
    d_l = torch.randn_like(out_l)
    named = list(layer.named_parameters())
    g_l = torch.autograd.grad(out_l, [tgt, mem] + [p for _, p in named], d_l)
    cases.update(layer_out=out_l, layer_dout=d_l, layer_dtgt=g_l[0], layer_dmemory=g_l[1])
    for (k, _), g in zip(named, g_l[2:]):
        cases["layer_g." + k] = g
    # the stack incl. final norm (Conmamba.py:1017-1031)
    out, a, b = dec(tgt, mem)
    assert a == [None] and b == [None]
    d_o = torch.randn_like(out)
    named = list(dec.named_parameters())
    g_o = torch.autograd.grad(out, [tgt, mem] + [p for _, p in named], d_o)
    cases.update(out=out, dout=d_o, dtgt=g_o[0], dmemory=g_o[1])
    for (k, _), g in zip(named, g_o[2:]):
        cases["g." + k] = g
    save("g4_decoder_stack", **cases)


def make_step(bim):
    assert bim.causal_conv1d_update is None and bim.selective_state_update is None     # the pure-torch fallback runs
    torch.manual_seed(3413)
    m = bim.Mamba(64, d_state=16, d_conv=4, expand=2, bimamba_type="v2")
    xavier_reinit(m)
    with torch.no_grad():
        m.D.add_(0.1 * torch.randn_like(m.D))
        m.conv1d.bias.add_(0.1 * torch.randn_like(m.conv1d.bias))
    x = torch.randn(3, 21, 64)
    conv_state, ssm_state = m.allocate_inference_cache(3, 21)
    outs = []
    with torch.no_grad():
        for t in range(x.shape[1]):
            o, conv_state, ssm_state = m.step(x[:, t:t + 1], conv_state, ssm_state)
            outs.append(o)
    keys = ("in_proj.weight", "conv1d.weight", "conv1d.bias", "x_proj.weight", "dt_proj.weight", "dt_proj.bias", "A_log", "D",
            "out_proj.weight")
    cases = {"x": x, "out": torch.cat(outs, 1), "conv_state": conv_state, "ssm_state": ssm_state}
    sd = m.state_dict()
    for k in keys:
        cases["p." + k] = sd[k]
    save("g_step", **cases)


if __name__ == "__main__":
    ssi, bim = load_reference()
    make_inner_outproj(ssi)
    make_g3_d256(bim)
    make_step(bim)
    cm = load_reference_conmamba(ssi, bim)
    make_g4_large(cm)
    make_decoder_stack(cm)
