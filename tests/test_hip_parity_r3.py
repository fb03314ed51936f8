"""GPU parity, round 3 (VERDICT r2 "Next round" item 1): the TRAINING step and config 5 under oracle-pinned tests.

  * whole-model gradient parity at ConMamba-large dims (D 256, 2 layers, 4 x 4 s): wav -> Fbank -> CNN -> encoder ->
    ctc_lin -> CTC loss -> backward(), fp32: the loss and EVERY parameter gradient against the oracle's autograd in
    fp64; the same under bf16 autocast with a looser bound                                           [config 4, per GPU]
  * the identical step in a SPAWNED CHILD under a world-1 `nccl` (RCCL) process group with
    GradAllReducer(always_exchange=True), all-reduce and mesh, fp32 and bf16 transport                [config 4, exchange]
  * MambaDecoder layer + stack gradients at D 512 / E 1024 / R 32 (S2S-large dims) vs the oracle      [config 5]
  * the reference's own TransformerASR.forward / decode / encode (golden g_s2s_forward, make_golden_r3.py) vs this
    package's TransformerASR on the GPU, forward and gradients; forward_s2s (frontend included) vs the oracle  [f1]
  * full-size properties for config 5 (4 x 160 s, D 512: utterance independence, chunked scan vs time_chunks = 1) and
    config 2 (64 x 10 s, D 144)                                                                      [configs 2, 5]

Tolerances: fp32 gradients rtol 3e-3..5e-3 / atol 3e-4..5e-4 x max|ref| (G3 / G4's); CTC loss |delta| <= 1e-3
absolute in fp32 (BASELINE.json north_star); bf16 autocast gradients: relative L2 error per tensor <= 6e-2.
"""
import importlib.util
import os
import subprocess
import sys

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = {"d_state": 16, "expand": 2, "d_conv": 4, "bidirectional": True}

_spec = importlib.util.spec_from_file_location("golden_synth", os.path.join(os.path.dirname(__file__), "golden", "synth.py"))
S = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(S)


def close(a, b, rtol=2e-3, atol=2e-4):
    scale = max(1.0, float(b.abs().max()))
    torch.testing.assert_close(a.detach().double().cpu(), b.detach().double().cpu(), rtol=rtol, atol=atol * scale)


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


# ----------------------------------------------------------------------------------------------------------
# (i) whole-model gradients: wav -> CTC loss -> backward, against the oracle's fp64 autograd
# ----------------------------------------------------------------------------------------------------------
def _train_case(layers=2, batch=4, frames=400, seed=3402):
    from dataclasses import replace
    from mamba_asr_amd.asr import CONFIGS, ConMambaASR, samples_for_frames, synthetic_wavs
    cfg = replace(CONFIGS["conmamba_large_ctc"], num_encoder_layers=layers, transformer_dropout=0.0)
    model = ConMambaASR(cfg).to(DEV)
    model.eval()                 # dropout off (the CNN front end's Dropout2d too), normalisation statistics frozen; with grad
    wavs, _ = synthetic_wavs(batch, samples_for_frames(frames), seed, DEV)          # enabled this IS the autograd (training) path
    lens = torch.tensor([1.0, 0.9, 0.75, 0.6][:batch], device=DEV)
    for i, r in enumerate(lens.tolist()):
        wavs[i, int(round(r * wavs.shape[1])):] = 0.0
    gen = torch.Generator().manual_seed(seed + 1)
    tokens = torch.randint(1, cfg.output_neurons, (batch, 12), generator=gen)
    tok_lens = torch.tensor([1.0, 0.75, 0.5, 0.9][:batch])
    with torch.no_grad():
        model.calibrate(wavs, lens)
    return cfg, model, wavs, lens, tokens, tok_lens


def _gpu_loss_and_grads(model, wavs, lens, tokens, tok_lens, autocast=False):
    from mamba_asr_amd import ops
    for p in model.parameters():
        p.grad = None
    ops.LAUNCH_LOG = []
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            logp = model.forward_ctc(wavs, lens)
            loss = model.ctc_objective(logp.float(), tokens.to(DEV), lens, tok_lens.to(DEV))
        loss.backward()
        torch.cuda.synchronize()
        names = {e[0] for e in ops.LAUNCH_LOG}
    finally:
        ops.LAUNCH_LOG = None
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    return loss.detach(), grads, names


def _oracle_loss_and_grads(cfg, model, wavs, lens, tokens, tok_lens, layers):
    """The same step on the CPU through the oracle, parameters and activations in fp64 (the Fbank stays fp32 as in the
    product: it has no parameters), torch autograd for the gradients."""
    from oracle import conmamba_oracle as O
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    p = {k: (v.double().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    feats = O.fbank(wavs.cpu(), n_fft=cfg.n_fft, win_ms=cfg.win_length)
    feats = ((feats - sd["normalize.glob_mean"]) / sd["normalize.glob_std"]).double()
    src = O.cnn_frontend(p, feats, "CNN.")
    src = src.reshape(src.shape[0], src.shape[1], -1)
    src = F.linear(src, p["Transformer.custom_src_module.layers.0.w.weight"], p["Transformer.custom_src_module.layers.0.w.bias"])
    enc = O.encoder(p, src, layers, "Transformer.encoder.")
    logp = torch.log_softmax(F.linear(enc, p["ctc_lin.w.weight"], p["ctc_lin.w.bias"]), -1)
    b, t, _ = logp.shape
    il, tl = torch.round(lens.cpu() * t).int(), torch.round(tok_lens * tokens.shape[1]).int()
    loss = F.ctc_loss(logp.transpose(0, 1), tokens, il, tl, 0, reduction="sum", zero_infinity=True) / b   # O.ctc_loss_batchmean in fp64
    names = [k for k, _ in model.named_parameters()]
    grads = torch.autograd.grad(loss, [p[k] for k in names], allow_unused=True)
    return loss.detach(), {k: g for k, g in zip(names, grads) if g is not None}


def test_training_step_loss_and_every_gradient_vs_oracle():
    """SURVEY §8d config 4's per-GPU workload at ConMamba-large dims, fp32 and bf16 autocast."""
    cfg, model, wavs, lens, tokens, tok_lens = _train_case()
    ref_loss, ref = _oracle_loss_and_grads(cfg, model, wavs, lens, tokens, tok_lens, 2)
    loss, grads, names = _gpu_loss_and_grads(model, wavs, lens, tokens, tok_lens)
    # the mixers run as the channels-last rows node (modules/mamba/mixer_rows.py): row-group scan forward / backward, both directions per launch
    for k in ("cm_scan_cl_fwd", "cm_scan_cl_bwd", "cm_conv_cl_bwd", "cm_layernorm_bwd", "cm_dwconv_cl_bwd"):
        assert k in names, f"{k} did not run in the training step (ran: {sorted(names)})"
    d = abs(float(loss) - float(ref_loss))
    print(f"training step, fp32: CTC oracle(fp64) {float(ref_loss):.6f} gpu {float(loss):.6f} |delta| {d:.2e}")
    assert d <= 1e-3
    assert set(grads) == set(ref), sorted(set(grads) ^ set(ref))
    # every gradient: relative L2 error <= 2e-3 and every element within 2e-3 of the tensor's largest |gradient| (G3 / G4's
    # bounds; the elementwise bound is relative to the tensor's scale because single elements of the input-most tensors -- the
    # CNN front end's taps, behind two LeakyReLU kinks and 2 x 18 kernels -- sit next to an activation kink in fp32 vs fp64)
    stats = {k: (rel_l2(grads[k], ref[k]), float((grads[k].double().cpu() - ref[k]).abs().max() / ref[k].abs().max().clamp_min(1e-30)))
             for k in ref}
    top = sorted(stats.items(), key=lambda kv: -kv[1][0])[:5]
    print(f"  {len(ref)} parameter gradients; worst relative L2 errors: " + ", ".join(f"{k} {v[0]:.1e}" for k, v in top))
    bad = {k: v for k, v in stats.items() if v[0] > 2e-3 or v[1] > 2e-3}
    assert not bad, bad
    # bf16 autocast (the recipe's precision, conmamba_large.yaml:86): looser
    loss_bf, grads_bf, _ = _gpu_loss_and_grads(model, wavs, lens, tokens, tok_lens, autocast=True)
    dbf = abs(float(loss_bf) - float(ref_loss))
    print(f"training step, bf16 autocast: gpu {float(loss_bf):.6f} |delta| {dbf:.2e} (rel {dbf / abs(float(ref_loss)):.2e})")
    assert dbf <= 2e-3 * abs(float(ref_loss))
    errs = {k: rel_l2(grads_bf[k], ref[k]) for k in ref}
    bad = {k: round(v, 4) for k, v in errs.items() if v > 6e-2}
    print(f"  bf16 gradients: median rel L2 {sorted(errs.values())[len(errs) // 2]:.2e}, max {max(errs.values()):.2e}")
    assert not bad, bad


def _same_grads(a, b, names, cnn_tol=1e-5):
    """Bit-equal outside the CNN front end; the front end's conv2d backward is the vendor library's (MIOpen weight / data
    gradient kernels accumulate with atomics: last-bit differences from run to run -- last bf16 bits under autocast) -- its
    tensors within cnn_tol of their scale.  The encoder, the src Linear and the CTC head run on this package's deterministic
    kernels."""
    ok = True
    for k, x, y in zip(names, a, b):
        if k.startswith("CNN."):
            ok = ok and float((x - y).abs().max()) <= cnn_tol * float(y.abs().max().clamp_min(1e-30))
        else:
            ok = ok and torch.equal(x, y)
    return ok


def test_training_step_gradients_are_bit_reproducible():
    """Everything behind the CNN front end (vendor conv2d backward) is deterministic: with the gradient of the log-probabilities
    held fixed, and with the real CTC objective (cm_ctc_loss; torch's CTC backward accumulates with atomics), two passes give
    bit-identical gradients for every encoder / projection / head parameter."""
    cfg, model, wavs, lens, tokens, tok_lens = _train_case(batch=2, frames=200)
    runs = []
    g = None
    for _ in range(2):
        for p in model.parameters():
            p.grad = None
        logp = model.forward_ctc(wavs, lens)
        if g is None:
            g = torch.randn(logp.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5)) * 1e-2
        logp.backward(g)
        runs.append({k: p.grad.clone() for k, p in model.named_parameters()})
    names = list(runs[0])
    assert _same_grads([runs[0][k] for k in names], [runs[1][k] for k in names], names)
    assert sum(not k.startswith("CNN.") for k in names) >= 90
    # round 3: with the CTC loss on cm_ctc_loss (fixed-order posterior sums) the real objective is reproducible as well
    full = []
    for _ in range(2):
        loss, grads, _ = _gpu_loss_and_grads(model, wavs, lens, tokens, tok_lens)
        full.append((loss, grads))
    assert torch.equal(full[0][0], full[1][0])
    assert _same_grads([full[0][1][k] for k in names], [full[1][1][k] for k in names], names)


# ----------------------------------------------------------------------------------------------------------
# (ii) the same step under a world-1 RCCL group, in a spawned child
# ----------------------------------------------------------------------------------------------------------
_CHILD = r'''
import os, sys, json
sys.path.insert(0, os.environ["CM_ROOT"])
sys.path.insert(0, os.path.join(os.environ["CM_ROOT"], "tests"))
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("CM_PORT", "29533"), RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from test_hip_parity_r3 import _train_case, _same_grads, DEV
from mamba_asr_amd.ddp import GradAllReducer
cfg, model, wavs, lens, tokens, tok_lens = _train_case(batch=2, frames=200)
named = [(k, p) for k, p in model.named_parameters() if p.requires_grad]
names, params = [k for k, _ in named], [p for _, p in named]
gfix = None
def step(autocast):
    global gfix
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        logp = model.forward_ctc(wavs, lens)
    if gfix is None:
        gfix = torch.randn(logp.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5)) * 1e-2
    return logp, gfix.to(logp.dtype)
out = {}
for autocast in (False, True):
    for p in params: p.grad = None
    logp, g = step(autocast)
    logp.backward(g)                                   # plain autograd, no reducer: the gradients of test (i)
    plain = [p.grad.detach().clone() for p in params]
    for algo in ("allreduce", "mesh"):
        for cdt in (None, torch.bfloat16):
            for p in params: p.grad = None
            red = GradAllReducer(params, algo=algo, comm_dtype=cdt, always_exchange=True, broadcast_from=None)
            assert red.active and red.world == 1
            red.prepare()
            logp, g = step(autocast)
            logp.backward(g)
            red.finish()
            torch.cuda.synchronize()
            key = f"{'bf16' if autocast else 'fp32'}-{algo}-{'bf16' if cdt else 'fp32'}"
            got = [p.grad for p in params]
            tol = 3e-2 if (autocast or cdt is not None) else 1e-5     # the CNN front end's vendor conv backward: see _same_grads
            if cdt is None:
                out[key] = _same_grads(got, plain, names, tol)
            else:                                      # bf16 transport: the gradient rounded to bf16 once
                out[key] = _same_grads([x.bfloat16().float() for x in got], [q.bfloat16().float() for q in plain], names, tol)
            out[key + "-views"] = all(p.grad.data_ptr() == red._view[p].data_ptr() for p in params)
            out[key + "-bytes"] = red.bytes_per_step()
            red.close()                                # remove this reducer's hooks before the next one registers its own
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


def test_training_step_under_world1_rccl_group(tmp_path):
    """GradAllReducer(always_exchange=True) over backend 'nccl' (= RCCL) at world size 1, in a child process (a process
    group in the pytest process would outlive the test): all-reduce and mesh (all-to-all + fixed-order sum + all-gather),
    fp32 and bf16 transport, under fp32 and bf16-autocast compute.  fp32 transport: gradients BIT-EQUAL to plain autograd's
    (encoder / projections / head; the vendor conv2d backward of the CNN front end is not bit-reproducible by itself)."""
    script = tmp_path / "child.py"
    script.write_text(_CHILD)
    env = dict(os.environ, CM_ROOT=ROOT, CM_PORT=str(29500 + os.getpid() % 400), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    print(out)
    flags = {k: v for k, v in out.items() if not k.endswith("-bytes")}
    assert len(flags) == 16 and all(flags.values()), flags
    assert out["fp32-allreduce-fp32-bytes"] == 2 * out["fp32-allreduce-bf16-bytes"] > 0


# ----------------------------------------------------------------------------------------------------------
# (iii) decoder gradients at the S2S-large dims
# ----------------------------------------------------------------------------------------------------------
def test_decoder_gradients_at_s2s_large_dims():
    """MambaDecoderLayer and a 2-layer MambaDecoder at D 512 / E 1024 / dt_rank 32 / d_ffn 2048, GELU (hparams/S2S/
    conmambamamba_large.yaml:251-259): outputs and gradients w.r.t. tgt, memory and every parameter vs the oracle's fp64
    autograd (the d_model-64 golden g4_decoder_stack pins the same code against the reference)."""
    from mamba_asr_amd.modules.Conmamba import MambaDecoder
    from oracle import conmamba_oracle as O
    dec = MambaDecoder(num_layers=2, d_model=512, d_ffn=2048, activation=nn.GELU, dropout=0.0, normalize_before=True,
                       mamba_config=dict(CFG))
    sd = S.synth_like(dec, 5120)
    dec.load_state_dict(sd, strict=True)
    dec = dec.to(DEV).train()
    assert dec.layers[0].self_mamba.dt_rank == 32 and dec.layers[0].self_mamba.d_inner == 1024
    tgt0 = S.synth_input("dec512.tgt", (2, 13, 512), 5120)
    mem0 = S.synth_input("dec512.mem", (2, 70, 512), 5120)
    dout = S.synth_input("dec512.dout", (2, 13, 512), 5120)
    p64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    t64, m64 = tgt0.double().requires_grad_(True), mem0.double().requires_grad_(True)
    names = [k for k, _ in dec.named_parameters()]
    for what in ("layer", "stack"):
        tgt, mem = tgt0.to(DEV).requires_grad_(True), mem0.to(DEV).requires_grad_(True)
        if what == "layer":
            out, _, _ = dec.layers[0](tgt, mem)
            want = O.decoder_layer(p64, t64, m64, "layers.0.", act=F.gelu)
            ks = [k for k in names if k.startswith("layers.0.")]
        else:
            out, _, _ = dec(tgt, mem)
            want = O.decoder(p64, t64, m64, 2, act=F.gelu)
            ks = names
        pd = dict(dec.named_parameters())
        got = torch.autograd.grad(out, [tgt, mem] + [pd[k] for k in ks], dout.to(DEV))
        ref = torch.autograd.grad(want, [t64, m64] + [p64[k] for k in ks], dout.double())
        close(out, want.detach(), rtol=3e-3, atol=3e-4)
        close(got[0], ref[0], rtol=3e-3, atol=3e-4)
        close(got[1], ref[1], rtol=3e-3, atol=3e-4)
        worst = 0.0
        for k, a, b in zip(ks, got[2:], ref[2:]):
            close(a, b, rtol=5e-3, atol=5e-4)
            worst = max(worst, rel_l2(a, b))
        print(f"decoder {what} at D 512: {len(ks)} parameter gradients, worst relative L2 error {worst:.2e}")
        assert worst < 3e-3


# ----------------------------------------------------------------------------------------------------------
# (v) the reference's own TransformerASR.forward (ConMamba encoder + Mamba decoder)
# ----------------------------------------------------------------------------------------------------------
def _s2s_model(dropout=0.0):
    from mamba_asr_amd.modules.TransformerASR import TransformerASR
    m = TransformerASR(tgt_vocab=53, input_size=640, d_model=128, nhead=4, num_encoder_layers=2, num_decoder_layers=2, d_ffn=256,
                       dropout=dropout, activation=nn.GELU, encoder_module="conmamba", decoder_module="mamba",
                       attention_type="RelPosMHAXL", normalize_before=True, causal=False, mamba_config=dict(CFG))
    sd = {k: v for k, v in S.synth_like(m, 1280).items() if not k.endswith(".pe")}     # parameters; the sinusoid table is a buffer
    miss = m.load_state_dict(sd, strict=False)
    assert not miss.unexpected_keys and all(k.endswith(".pe") for k in miss.missing_keys)
    return m.to(DEV)


def test_transformer_asr_forward_decode_encode_vs_reference(golden):
    """Golden g_s2s_forward = outputs of the REFERENCE's TransformerASR.forward / .decode / .encode
    (modules/TransformerASR.py:745-929, Transformer.py:796-1022, 1650-1860)."""
    g = golden("g_s2s_forward")
    m = _s2s_model(dropout=0.1).eval()
    src = S.synth_input("g_s2s.src", (3, 41, 20, 32), 1280).to(DEV)
    tgt = g["tgt"].long().to(DEV)
    wav_len = g["wav_len"].to(DEV)
    for mode in ("fused-nograd", "module-grad"):
        with torch.set_grad_enabled(mode == "module-grad"):       # no_grad + eval: the fused inference kernels where supported
            enc, dec = m(src, tgt, wav_len)
            enc_only = m.encode(src, wav_len)
        close(enc, g["encoder_out"])
        close(enc_only, g["encode_out"])
        close(dec, g["decoder_out"])
    pred, attn = m.decode(tgt, enc.detach())
    assert attn is None
    close(pred, g["decode_prediction"])
    # bf16 autocast, SURVEY's encoder-output tolerance
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        enc_bf, dec_bf = m(src, tgt, wav_len)
    torch.testing.assert_close(enc_bf.float().cpu(), g["encoder_out"], rtol=3e-2, atol=5e-2)
    torch.testing.assert_close(dec_bf.float().cpu(), g["decoder_out"], rtol=3e-2, atol=6e-2)


def test_transformer_asr_gradients_vs_reference(golden):
    g = golden("g_s2s_forward")
    m = _s2s_model(dropout=0.0).train()
    src = S.synth_input("g_s2s.src", (3, 41, 20, 32), 1280).to(DEV).requires_grad_(True)
    _, dec = m(src, g["tgt"].long().to(DEV), g["wav_len"].to(DEV))
    close(dec, g["decoder_out_train"])
    w = S.synth_input("g_s2s.w", tuple(dec.shape), 1280).to(DEV)
    names = [k[2:] for k in g if k.startswith("g.")]
    pd = dict(m.named_parameters())
    grads = torch.autograd.grad((dec * w).sum(), [src] + [pd[n] for n in names])
    close(grads[0], g["dsrc"], rtol=3e-3, atol=3e-4)
    for n, gk in zip(names, grads[1:]):
        close(gk, g["g." + n], rtol=5e-3, atol=5e-4)


def test_forward_s2s_with_frontend_vs_oracle():
    """asr.ConMambaASR.forward_s2s (train_S2S.py:285-320: Fbank -> normalise -> CNN -> TransformerASR(src, <bos> tokens) ->
    ctc_lin / seq_lin log-probabilities) and the 0.3 CTC + 0.7 KL objective (:518-529) vs the oracle composition; the
    TransformerASR part of that composition is pinned to the reference by g_s2s_forward, the frontend and the KL loss are
    speechbrain restatements (parity unpinned)."""
    from dataclasses import replace
    from mamba_asr_amd.asr import CONFIGS, ConMambaASR, samples_for_frames, synthetic_wavs
    from oracle import conmamba_oracle as O
    cfg = replace(CONFIGS["conmambamamba_large_s2s"], d_model=128, d_ffn=256, num_encoder_layers=2, num_decoder_layers=2,
                  output_neurons=60, transformer_dropout=0.0)
    model = ConMambaASR(cfg).to(DEV).eval()
    wavs, lens = synthetic_wavs(3, samples_for_frames(240), 9, DEV)
    gen = torch.Generator().manual_seed(10)
    bos = torch.cat([torch.ones(3, 1, dtype=torch.long), torch.randint(3, 60, (3, 8), generator=gen)], 1)
    with torch.no_grad():
        model.calibrate(wavs, lens)
        p_ctc, p_seq = model.forward_s2s(wavs, lens, bos.to(DEV))
    p = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    feats = (O.fbank(wavs.cpu(), n_fft=cfg.n_fft, win_ms=cfg.win_length) - p["normalize.glob_mean"]) / p["normalize.glob_std"]
    src = O.cnn_frontend(p, feats, "CNN.")
    tp = {k[len("Transformer."):]: v for k, v in p.items() if k.startswith("Transformer.")}
    enc, dec = O.transformer_asr_forward(tp, src, bos, 2, 2, scan=O.selective_scan)
    want_ctc = torch.log_softmax(F.linear(enc, p["ctc_lin.w.weight"], p["ctc_lin.w.bias"]), -1)
    want_seq = torch.log_softmax(F.linear(dec, p["seq_lin.w.weight"], p["seq_lin.w.bias"]), -1)
    close(p_ctc, want_ctc, rtol=2e-3, atol=5e-4)
    close(p_seq, want_seq, rtol=2e-3, atol=5e-4)


# ----------------------------------------------------------------------------------------------------------
# (iv) full-size properties: config 5 (4 x 160 s, D 512) and config 2 (64 x 10 s, D 144)
# ----------------------------------------------------------------------------------------------------------
def _independence(cfg_name, batch, frames, layers, pick, **over):
    from dataclasses import replace
    from mamba_asr_amd.asr import CONFIGS, ConMambaASR, samples_for_frames, synthetic_wavs
    cfg = replace(CONFIGS[cfg_name], num_encoder_layers=layers, num_decoder_layers=0, **over)
    model = ConMambaASR(cfg).to(DEV).eval()
    wavs, lens = synthetic_wavs(batch, samples_for_frames(frames), cfg.seed, DEV)
    with torch.no_grad():
        model.calibrate(wavs, lens)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            full = model.encode(wavs, lens)
            perm = torch.randperm(batch, generator=torch.Generator().manual_seed(1)).to(DEV)
            shuffled = model.encode(wavs[perm], lens[perm])
            alone = model.encode(wavs[pick:pick + 1], lens[pick:pick + 1])
    assert full.shape[:2] == (batch, frames // 4) and torch.isfinite(full.float()).all()
    return model, wavs, lens, full, perm, shuffled, alone


def test_config5_full_size_properties():
    """BASELINE.json config 5's encoder at its full per-GPU size: S2S-large dims (D 512, E 1024, dt_rank 32, 12 layers),
    4 x 160 s (L = 16000 frames -> T = 4000 scan steps).  (a) utterances are independent: a permuted batch gives the permuted
    output bit for bit, on the time-chunked scan launch the size policy picks here (auto_chunks(4, 4000, 1024, 2) == 8);
    (b) the chunked launch agrees with the unchunked one (time_chunks = 1) within bf16 rounding through all 12 layers;
    (c) an utterance encoded alone (different chunk count) agrees within the same bound."""
    from mamba_asr_amd import _native, ops
    assert _native.lib().cm_scan_cl_fwd_auto_chunks(4, 4000, 1024, 2) == 8
    model, wavs, lens, full, perm, shuffled, alone = _independence("conmambamamba_large_s2s", 4, 16000, 12, pick=2)
    assert torch.equal(shuffled, full[perm])
    old = ops.SCAN_CHUNKS
    try:
        ops.SCAN_CHUNKS = "1"
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            unchunked = model.encode(wavs, lens)
    finally:
        ops.SCAN_CHUNKS = old
    err = (unchunked.float() - full.float()).abs()
    print(f"config 5, 4 x 160 s x 12 layers: chunked vs unchunked scan max|diff| {float(err.max()):.3e} mean {float(err.mean()):.3e}")
    torch.testing.assert_close(full.float(), unchunked.float(), rtol=3e-2, atol=5e-2)
    assert float(err.mean()) < 4e-3
    torch.testing.assert_close(alone[0].float(), full[2].float(), rtol=3e-2, atol=5e-2)


def test_config2_full_size_properties():
    """BASELINE.json config 2 at its full size: ConMamba-small CTC (D 144, 12 layers, n_fft 400), 64 x 10 s.  Utterance
    independence bit for bit (permuted batch; one utterance alone)."""
    model, wavs, lens, full, perm, shuffled, alone = _independence("conmamba_small_ctc", 64, 1000, 12, pick=37)
    assert torch.equal(shuffled, full[perm])
    from mamba_asr_amd import _native
    if _native.lib().cm_scan_cl_fwd_auto_chunks(1, 250, 288, 2) == 1:
        assert torch.equal(alone[0], full[37])
    else:                                                  # a lone utterance is cut into time chunks: fp32 summation order differs
        torch.testing.assert_close(alone[0].float(), full[37].float(), rtol=3e-2, atol=5e-2)
