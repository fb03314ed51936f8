"""The CPU oracle (oracle/) checked against golden vectors produced by the reference's own Python
(tests/golden/make_golden.py).  This is what pins the oracle; runs on CPU."""
import subprocess
import os

import pytest
import torch

from oracle import conmamba_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def close(a, b, rtol, atol):
    torch.testing.assert_close(a.double(), b.double(), rtol=rtol, atol=atol)


@pytest.fixture(scope="module")
def c_oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return O.load_c_oracle()


@pytest.mark.parametrize("tag", ["tiny", "mid", "long", "n8"])
def test_scan_fwd_full(golden, tag):
    g = golden("g1_scan_fwd")
    args = [g[f"{tag}_{k}"] for k in ("u", "delta", "A", "B", "C", "D", "z", "bias")]
    out, last = O.selective_scan(*args, delta_softplus=True, return_last_state=True, work_dtype=torch.float64)
    close(out, g[f"{tag}_out_full"], 1e-4, 2e-5)
    close(last, g[f"{tag}_last_full"], 1e-4, 2e-5)
    out32 = O.selective_scan(*args, delta_softplus=True)
    close(out32, g[f"{tag}_out_full"], 1e-4, 2e-5)


@pytest.mark.parametrize("tag", ["tiny", "mid"])
def test_scan_fwd_variants(golden, tag):
    g = golden("g1_scan_fwd")
    u, dl, A, B, C, D, z, bias = [g[f"{tag}_{k}"] for k in ("u", "delta", "A", "B", "C", "D", "z", "bias")]
    close(O.selective_scan(u, dl, A, B, C, D, None, bias, True), g[f"{tag}_out_noz"], 1e-4, 2e-5)
    close(O.selective_scan(u, dl, A, B, C, None, z, bias, True), g[f"{tag}_out_noD"], 1e-4, 2e-5)
    close(O.selective_scan(u, dl, A, B, C, D, z, None, True), g[f"{tag}_out_nobias"], 1e-4, 2e-5)
    close(O.selective_scan(u, dl.abs() * 0.1, A, B, C, D, z, None, False), g[f"{tag}_out_nosoftplus"], 1e-4, 2e-5)
    close(O.selective_scan(u, dl, A, B, C, None, None, None, True), g[f"{tag}_out_bare"], 1e-4, 2e-5)
    close(O.selective_scan(u, dl, A, B[:, None], C[:, None], D, z, bias, True), g[f"{tag}_out_4d"], 1e-4, 2e-5)
    f = lambda t: t.flip(-1)
    close(f(O.selective_scan(f(u), f(dl), A, f(B), f(C), D, f(z), bias, True)), g[f"{tag}_out_rev"], 1e-4, 2e-5)
    bf = lambda t: t.to(torch.bfloat16)
    ob = O.selective_scan(bf(u), bf(dl), A, bf(B), bf(C), D, bf(z), bias, True)
    assert ob.dtype == torch.bfloat16
    close(ob.float(), g[f"{tag}_out_bf16"], 1.6e-2, 1e-2)   # one bf16 ulp of slack


@pytest.mark.parametrize("tag", ["tiny", "mid", "long"])
def test_scan_fwd_c_oracle(golden, c_oracle, tag):
    g = golden("g1_scan_fwd")
    args = [g[f"{tag}_{k}"] for k in ("u", "delta", "A", "B", "C", "D", "z", "bias")]
    out, last = O.selective_scan_c(*args, delta_softplus=True, return_last_state=True)
    close(out, g[f"{tag}_out_full"], 2e-4, 5e-5)
    close(last, g[f"{tag}_last_full"], 2e-4, 5e-5)


@pytest.mark.parametrize("tag", ["tiny", "mid", "long"])
def test_scan_bwd(golden, tag):
    g = golden("g2_scan_bwd")
    u, dl, A, B, C, D, z, bias, dout = [g[f"{tag}_{k}"] for k in ("u", "delta", "A", "B", "C", "D", "z", "bias", "dout")]
    r = O.selective_scan_bwd(u, dl, A, B, C, D, z, bias, dout, True)
    for k, gk in (("du", "du"), ("ddelta", "ddelta"), ("dA", "dA"), ("dB", "dB"), ("dC", "dC"), ("dD", "dD"),
                  ("dz", "dz"), ("ddelta_bias", "dbias")):
        ref = g[f"{tag}_{gk}"]
        scale = ref.abs().max().item()
        close(r[k], ref, 2e-3, 2e-4 * max(scale, 1.0))
    if tag == "tiny":
        r = O.selective_scan_bwd(u, dl, A, B, C, None, None, None, dout, True)
        for k in ("du", "ddelta", "dA", "dB", "dC"):
            ref = g[f"tiny_bare_{k}"]
            close(r[k], ref, 2e-3, 2e-4 * max(ref.abs().max().item(), 1.0))


@pytest.mark.parametrize("tag", ["tiny", "mid", "w3", "short"])
def test_conv(golden, c_oracle, tag):
    g = golden("k_conv")
    x, w, b, dout = g[f"{tag}_x"], g[f"{tag}_w"], g[f"{tag}_b"], g[f"{tag}_dout"]
    close(O.causal_conv1d(x, w, b, True), g[f"{tag}_y"], 1e-5, 1e-5)
    close(O.causal_conv1d(x, w, None, False), g[f"{tag}_y_lin"], 1e-5, 1e-5)
    dx, dw, db = O.causal_conv1d_bwd(x, w, b, dout, True)
    close(dx, g[f"{tag}_dx"], 1e-4, 1e-5)
    close(dw, g[f"{tag}_dw"], 1e-4, 1e-4)
    close(db, g[f"{tag}_db"], 1e-4, 1e-4)
    out = torch.empty_like(x)
    c_oracle.oracle_causal_conv1d_fwd_f32(O._fptr(x.contiguous()), O._fptr(w.contiguous()), O._fptr(b.contiguous()),
                                          x.shape[0], x.shape[1], x.shape[2], w.shape[1], 1, O._fptr(out))
    close(out, g[f"{tag}_y"], 1e-5, 1e-5)


def _params(g, prefix):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


def test_bimamba_layer(golden):
    g = golden("g3_bimamba")
    p = _params(g, "d144_p.")
    y = O.bimamba_v2(p, g["d144_x"])
    close(y, g["d144_y"], 1e-3, 2e-5)
    m = g  # inner op with the forward-direction parameters
    A = -torch.exp(p["A_log"].float())
    oz = O.mamba_inner_no_out_proj(m["d144_inner_xz"], p["conv1d.weight"], p["conv1d.bias"], p["x_proj.weight"],
                                   p["dt_proj.weight"], A, p["D"], p["dt_proj.bias"])
    close(oz, m["d144_inner_out"], 1e-3, 2e-5)


def test_encoder_layer_and_stack(golden):
    g = golden("g4_encoder")
    p = _params(g, "p.")
    close(O.encoder_layer(p, g["x"], "layers.0."), g["y_layer0"], 1e-3, 1e-4)
    close(O.encoder(p, g["x"], 2), g["y_enc"], 1e-3, 1e-4)
    # same result with the C scan plugged in
    close(O.encoder(p, g["x"], 2, scan=O.selective_scan_c), g["y_enc"], 1e-3, 1e-4)


def test_decoder_layer(golden):
    g = golden("g4_decoder_layer")
    p = _params(g, "p.")
    close(O.decoder_layer(p, g["tgt"], g["memory"]), g["out"], 1e-3, 1e-4)


def test_ctc(golden):
    g = golden("g5_ctc")
    lp = g["logits"].double().log_softmax(-1)
    loss = O.ctc_loss_batchmean(lp, g["targets"], g["in_rel"], g["tg_rel"])
    close(loss, g["loss"], 1e-5, 1e-5)


# ---- round-2 fixtures (tests/golden/make_golden_r2.py): benchmark dims, decoder stack + gradients, step decode ----
def _synth():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("golden_synth", os.path.join(os.path.dirname(__file__), "golden", "synth.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _large_encoder_shapes():
    """state_dict shapes of a 2-layer ConmambaEncoder at the benchmark's dims, from this package's own module (the
    reference's keys/shapes are identical: tests/test_hip_modules.py loads reference state_dicts strict=True)."""
    import torch.nn as nn
    from mamba_asr_amd.modules.Conmamba import ConmambaEncoder
    enc = ConmambaEncoder(num_layers=2, d_model=256, d_ffn=1024, kernel_size=31, activation=nn.GELU, bias=True, dropout=0.0,
                          causal=False, mamba_config={"d_state": 16, "expand": 2, "d_conv": 4, "bidirectional": True})
    return {k: tuple(v.shape) for k, v in enc.state_dict().items()}


def test_encoder_at_benchmark_dims(golden):
    S = _synth()
    g = golden("g4_large")
    p = S.synth_state(_large_encoder_shapes(), 256)
    x = S.synth_input("g4_large.x", (16, 100, 256), 256)
    close(O.encoder_layer(p, x[:4], "layers.0.", scan=O.selective_scan_c), g["y_layer0"][:4], 1e-3, 1e-4)
    close(O.encoder(p, x[:4], 2, scan=O.selective_scan_c), g["y_enc"][:4], 1e-3, 1e-4)


def test_bimamba_d256(golden):
    from mamba_asr_amd.modules.mamba.bimamba import Mamba
    S = _synth()
    g = golden("g3_d256")
    m = Mamba(256, d_state=16, d_conv=4, expand=2, bimamba_type="v2")
    p = S.synth_like(m, 2560)
    x = S.synth_input("g3_d256.x", (2, 50, 256), 2560)
    close(O.bimamba_v2(p, x), g["y"], 1e-3, 2e-5)


def test_inner_with_out_proj(golden):
    g = golden("g3_inner_outproj")
    out = O.mamba_inner(g["xz"], g["conv_w"], g["conv_b"], g["x_proj_w"], g["dt_proj_w"], g["out_proj_w"], g["out_proj_b"],
                        g["A"], g["D"], g["delta_bias"])
    close(out, g["out"], 1e-3, 2e-5)


def test_decoder_stack(golden):
    g = golden("g4_decoder_stack")
    p = _params(g, "p.")
    close(O.decoder_layer(p, g["tgt"], g["memory"], "layers.0."), g["layer_out"], 1e-3, 1e-4)
    close(O.decoder(p, g["tgt"], g["memory"], 2), g["out"], 1e-3, 1e-4)


def test_step_decode(golden):
    g = golden("g_step")
    p = _params(g, "p.")
    x = g["x"]
    conv_state, ssm_state = torch.zeros(3, 128, 4), torch.zeros(3, 128, 16)
    outs = [O.mamba_step(p, x[:, t:t + 1], conv_state, ssm_state) for t in range(x.shape[1])]
    close(torch.cat(outs, 1), g["out"], 1e-4, 1e-5)
    close(conv_state, g["conv_state"], 1e-6, 1e-6)
    close(ssm_state, g["ssm_state"], 1e-4, 1e-5)
    # and the token-by-token decode equals the full-sequence unidirectional forward (what f2 promises)
    close(torch.cat(outs, 1), O.mamba_uni(p, x), 1e-3, 1e-4)


def _s2s_shapes():
    from mamba_asr_amd.modules.TransformerASR import TransformerASR
    m = TransformerASR(tgt_vocab=53, input_size=640, d_model=128, nhead=4, num_encoder_layers=2, num_decoder_layers=2, d_ffn=256,
                       dropout=0.0, activation=torch.nn.GELU, encoder_module="conmamba", decoder_module="mamba",
                       attention_type="RelPosMHAXL", normalize_before=True, causal=False,
                       mamba_config={"d_state": 16, "expand": 2, "d_conv": 4, "bidirectional": True})
    return m, {k: tuple(v.shape) for k, v in m.state_dict().items() if not k.endswith(".pe")}     # parameters, not the sinusoid buffer


def test_s2s_forward_restatement_and_host_modules(golden):
    """The reference's own TransformerASR.forward / decode / encode (ConMamba encoder + Mamba decoder; golden made by
    tests/golden/make_golden_r3.py): (i) the oracle's restatement, (ii) this package's host-side modules
    (NormalizedEmbedding, PositionalEncoding, the state_dict key names) on the CPU — the reference's state_dict keys are
    exactly this module's keys, so synth_like gives both the same parameters."""
    S = _synth()
    g = golden("g_s2s_forward")
    m, shapes = _s2s_shapes()
    p = S.synth_state(shapes, 1280)
    src = S.synth_input("g_s2s.src", (3, 41, 20, 32), 1280)
    tgt = g["tgt"].long()
    enc, dec = O.transformer_asr_forward(p, src, tgt, 2, 2, scan=O.selective_scan_c)
    close(enc, g["encoder_out"], 1e-3, 1e-4)
    close(enc, g["encode_out"], 1e-3, 1e-4)
    close(dec, g["decoder_out"], 1e-3, 1e-4)
    close(dec, g["decode_prediction"], 1e-3, 1e-4)
    # host-side pieces of the product, no kernels involved: embedding * sqrt(d) + positional table
    miss = m.load_state_dict(p, strict=False)
    assert not miss.unexpected_keys and all(k.endswith(".pe") for k in miss.missing_keys)      # same key names as the reference
    t = m.custom_tgt_module(tgt)
    t = t + m.positional_encoding_decoder(t)
    emb = p["custom_tgt_module.layers.0.emb.Embedding.weight"]
    want = torch.nn.functional.embedding(tgt, emb) * 128 ** 0.5 + O.positional_encoding(11, 128)
    assert torch.equal(t, want)
