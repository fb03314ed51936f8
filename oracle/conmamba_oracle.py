"""CPU oracle for the ConMamba encoder hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  The product path (``mamba_asr_amd``)
never imports anything from ``oracle/`` and fails loudly when its HIP library
is missing.

Every function restates, in plain torch-CPU / numpy arithmetic, the algorithm
of one reference function; the reference location is cited per function
(paths relative to the reference checkout).  Parity status:

* selective scan fwd/bwd, causal conv, Mamba inner op, BiMamba-v2 mixer,
  ConvolutionModule, ConmambaEncoderLayer / ConmambaEncoder, MambaDecoderLayer:
  PINNED.  ``tests/golden/*.npz`` were produced by importing the reference's own
  Python (``tests/golden/make_golden.py``) and ``tests/test_oracle_golden.py``
  checks this file against them.
* Fbank / InputNormalization / SpectrogramDrop / ConvolutionFrontEnd / ctc_loss
  wrapper: these live in speechbrain==1.0.0, which is neither in the reference
  tree nor installed here -> "parity unpinned": restated from the published
  speechbrain semantics, cross-checked only against an
  independent numpy STFT/mel computation in the tests.

All math is done in the dtype of the inputs after an explicit ``.double()`` or
``.float()`` by the caller; nothing here touches a GPU.
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))

# --------------------------------------------------------------------------
# optional C restatement of the scan (oracle/scan_oracle.c), used for speed in
# the cpu_baseline leg and cross-checked against the torch loop in tests.
# --------------------------------------------------------------------------
_C_LIB = None


def c_oracle_path() -> str:
    return os.path.join(_HERE, "_build", "libscan_oracle.so")


def load_c_oracle():
    """dlopen oracle/_build/libscan_oracle.so (built by oracle/Makefile)."""
    global _C_LIB
    if _C_LIB is None:
        lib = ctypes.CDLL(c_oracle_path())
        fp = ctypes.POINTER(ctypes.c_float)
        lib.oracle_selective_scan_fwd_f32.argtypes = [
            fp, fp, fp, fp, fp, fp, fp, fp,  # u delta A B C D z delta_bias
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,  # batch dim len state
            ctypes.c_int,  # softplus
            fp, fp,  # out last_state
        ]
        lib.oracle_selective_scan_fwd_f32.restype = None
        lib.oracle_causal_conv1d_fwd_f32.argtypes = [
            fp, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp]
        lib.oracle_causal_conv1d_fwd_f32.restype = None
        lib.oracle_set_threads.argtypes = [ctypes.c_int]
        lib.oracle_set_threads.restype = None
        _C_LIB = lib
    return _C_LIB


def host_threads(cap: int = 16) -> int:
    """Threads the CPU baseline may use: the process's CPU affinity, capped (a one-GPU box share is 16 cores)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, cap))


def set_threads(n: int) -> None:
    torch.set_num_threads(n)
    try:
        load_c_oracle().oracle_set_threads(n)
    except OSError:
        pass


def _fptr(t: Optional[torch.Tensor]):
    if t is None:
        return ctypes.cast(None, ctypes.POINTER(ctypes.c_float))
    assert t.dtype == torch.float32 and t.is_contiguous()
    return ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_float))


# --------------------------------------------------------------------------
# a1. selective scan  (reference: modules/mamba/selective_scan_interface.py:91-157)
# --------------------------------------------------------------------------
def softplus(x: torch.Tensor) -> torch.Tensor:
    """torch softplus, beta=1, threshold=20 (ssi.py:112 uses F.softplus defaults)."""
    return torch.where(x > 20.0, x, torch.log1p(torch.exp(torch.clamp(x, max=20.0))))


def selective_scan(u, delta, A, B, C, D=None, z=None, delta_bias=None,
                   delta_softplus=False, return_last_state=False, work_dtype=None, initial_state=None):
    """Restates selective_scan_ref (ssi.py:91-157), real A, variable B/C.

    u, delta, z: (b, d, l); A: (d, n); B, C: (b, n, l) or (b, g, n, l); D, delta_bias: (d).
    Computation dtype = fp32 (like ssi.py:107-123) unless work_dtype is given (tests use
    float64 to get a tighter oracle).  Output is cast back to u.dtype (ssi.py:156).
    ``initial_state`` (b, d, n) is NOT in the reference (its x starts at zero, ssi.py:124): it is the one-line
    generalisation the time-split tests need -- with it, scanning a sequence in two pieces equals scanning it whole.
    """
    wd = work_dtype or torch.float32
    in_dtype = u.dtype
    uu = u.to(wd)
    dt = delta.to(wd)
    if delta_bias is not None:                       # ssi.py:109-110
        dt = dt + delta_bias.to(wd)[None, :, None]
    if delta_softplus:                               # ssi.py:111-112
        dt = softplus(dt)
    bsz, dim, length = uu.shape
    nstate = A.shape[1]
    Aw = A.to(wd)
    Bw, Cw = B.to(wd), C.to(wd)
    if Bw.dim() == 4:                                # grouped (b g n l) -> per-channel (ssi.py:133)
        Bw = Bw.repeat_interleave(dim // Bw.shape[1], dim=1)
    if Cw.dim() == 4:                                # ssi.py:136
        Cw = Cw.repeat_interleave(dim // Cw.shape[1], dim=1)
    h = torch.zeros(bsz, dim, nstate, dtype=wd) if initial_state is None else initial_state.to(wd).clone()   # ssi.py:124
    y = torch.empty(bsz, dim, length, dtype=wd)
    for t in range(length):                          # ssi.py:138-151
        dt_t = dt[:, :, t]                           # (b d)
        decay = torch.exp(dt_t[:, :, None] * Aw[None])          # ssi.py:126
        if Bw.dim() == 3:
            inj = (dt_t * uu[:, :, t])[:, :, None] * Bw[:, None, :, t]   # ssi.py:131
        else:
            inj = (dt_t * uu[:, :, t])[:, :, None] * Bw[:, :, :, t]     # ssi.py:134
        h = decay * h + inj                          # ssi.py:139
        if Cw.dim() == 3:
            y[:, :, t] = (h * Cw[:, None, :, t]).sum(-1)         # ssi.py:144
        else:
            y[:, :, t] = (h * Cw[:, :, :, t]).sum(-1)            # ssi.py:146
    out = y if D is None else y + uu * D.to(wd)[None, :, None]   # ssi.py:153
    if z is not None:
        out = out * F.silu(z.to(wd))                 # ssi.py:155
    out = out.to(in_dtype)                           # ssi.py:156
    return (out, h) if return_last_state else out


def selective_scan_c(u, delta, A, B, C, D=None, z=None, delta_bias=None,
                     delta_softplus=False, return_last_state=False):
    """Same contract as selective_scan, through oracle/scan_oracle.c (fp32, B/C (b,n,l))."""
    lib = load_c_oracle()
    in_dtype = u.dtype
    f = lambda t: None if t is None else t.detach().float().contiguous()
    u_, d_, A_, B_, C_, D_, z_, db_ = map(f, (u, delta, A, B, C, D, z, delta_bias))
    if B_.dim() == 4:
        assert B_.shape[1] == 1
        B_ = B_[:, 0].contiguous()
    if C_.dim() == 4:
        assert C_.shape[1] == 1
        C_ = C_[:, 0].contiguous()
    bsz, dim, length = u_.shape
    n = A_.shape[1]
    out = torch.empty_like(u_)
    last = torch.empty(bsz, dim, n)
    lib.oracle_selective_scan_fwd_f32(_fptr(u_), _fptr(d_), _fptr(A_), _fptr(B_), _fptr(C_),
                                      _fptr(D_), _fptr(z_), _fptr(db_), bsz, dim, length, n,
                                      int(bool(delta_softplus)), _fptr(out), _fptr(last))
    out = out.to(in_dtype)
    return (out, last) if return_last_state else out


def selective_scan_bwd(u, delta, A, B, C, D, z, delta_bias, dout, delta_softplus=True,
                       work_dtype=torch.float64):
    """Analytic gradients of selective_scan (what selective_scan_cuda.bwd returns to
    ssi.py:67-79).  Derived from the recurrence; pinned by
    tests/golden G2 = autograd through the reference's selective_scan_ref.

    Returns dict(du, ddelta, dA, dB, dC, dD, dz, ddelta_bias); B, C are (b, n, l).
    """
    wd = work_dtype
    uu, dl, Aw, Bw, Cw = (t.to(wd) for t in (u, delta, A, B, C))
    g_out = dout.to(wd)
    bsz, dim, length = uu.shape
    n = Aw.shape[1]
    pre = dl if delta_bias is None else dl + delta_bias.to(wd)[None, :, None]
    dt = softplus(pre) if delta_softplus else pre
    # forward pass storing all states
    hs = torch.zeros(bsz, dim, length + 1, n, dtype=wd)
    decay = torch.exp(dt[..., None] * Aw[None, :, None, :])                    # (b d l n)
    for t in range(length):
        hs[:, :, t + 1] = decay[:, :, t] * hs[:, :, t] + (dt[:, :, t] * uu[:, :, t])[..., None] * Bw[:, None, :, t]
    y = (hs[:, :, 1:] * Cw.permute(0, 2, 1)[:, None]).sum(-1)                    # (b d l)
    if D is not None:
        y = y + uu * D.to(wd)[None, :, None]
    res = {}
    if z is not None:
        zz = z.to(wd)
        sig = torch.sigmoid(zz)
        gate = zz * sig
        res["dz"] = g_out * y * (sig * (1 + zz * (1 - sig)))
        g = g_out * gate
    else:
        res["dz"] = None
        g = g_out
    res["dD"] = None if D is None else (g * uu).sum((0, 2))
    du = torch.zeros_like(uu) if D is None else g * D.to(wd)[None, :, None]
    ddt = torch.zeros_like(uu)
    dA = torch.zeros_like(Aw)
    dB = torch.zeros_like(Bw)
    dC = torch.zeros_like(Cw)
    lam = torch.zeros(bsz, dim, n, dtype=wd)
    for t in range(length - 1, -1, -1):
        lam = Cw[:, None, :, t] * g[:, :, t, None] + (decay[:, :, t + 1] * lam if t + 1 < length else 0)
        dC[:, :, t] = (g[:, :, t, None] * hs[:, :, t + 1]).sum(1)
        da = lam * hs[:, :, t]                                               # dL/da_t
        dbv = lam                                                            # dL/db_t
        du[:, :, t] += (dbv * Bw[:, None, :, t]).sum(-1) * dt[:, :, t]
        dB[:, :, t] = (dbv * (dt[:, :, t] * uu[:, :, t])[..., None]).sum(1)
        ddt[:, :, t] = (dbv * Bw[:, None, :, t]).sum(-1) * uu[:, :, t] + (da * decay[:, :, t] * Aw[None]).sum(-1)
        dA += (da * decay[:, :, t] * dt[:, :, t, None]).sum(0)
    ddelta = ddt * torch.sigmoid(pre) if delta_softplus else ddt
    if delta_softplus:
        ddelta = torch.where(pre > 20.0, ddt, ddelta)
    res.update(du=du, ddelta=ddelta, dA=dA, dB=dB, dC=dC,
               ddelta_bias=None if delta_bias is None else ddelta.sum((0, 2)))
    return res


# --------------------------------------------------------------------------
# K1/K2. causal depthwise conv + SiLU
# (reference semantics: modules/mamba/bimamba.py:83-91 layer definition and
#  :278-279 ``act(conv1d(x)[..., :seqlen])`` with padding = d_conv - 1)
# --------------------------------------------------------------------------
def causal_conv1d(x, weight, bias=None, silu=True, work_dtype=None):
    """x: (b, d, l); weight: (d, w); out[b,d,t] = act(bias[d] + sum_k weight[d,k] x[b,d,t-(w-1)+k])."""
    wd = work_dtype or torch.float32
    xx = x.to(wd)
    width = weight.shape[1]
    length = xx.shape[-1]
    padded = F.pad(xx, (width - 1, 0))
    acc = torch.zeros_like(xx)
    for k in range(width):
        acc = acc + weight.to(wd)[None, :, k, None] * padded[:, :, k:k + length]
    if bias is not None:
        acc = acc + bias.to(wd)[None, :, None]
    if silu:
        acc = acc * torch.sigmoid(acc)
    return acc.to(x.dtype)


def causal_conv1d_bwd(x, weight, bias, dout, silu=True, work_dtype=torch.float64):
    """Gradients (dx, dweight, dbias) of causal_conv1d — contract of
    causal_conv1d_cuda.causal_conv1d_bwd as called at ssi.py:286-288."""
    wd = work_dtype
    xx, ww, go = x.to(wd), weight.to(wd), dout.to(wd)
    width = ww.shape[1]
    length = xx.shape[-1]
    padded = F.pad(xx, (width - 1, 0))
    pre = torch.zeros_like(xx)
    for k in range(width):
        pre = pre + ww[None, :, k, None] * padded[:, :, k:k + length]
    if bias is not None:
        pre = pre + bias.to(wd)[None, :, None]
    if silu:
        s = torch.sigmoid(pre)
        go = go * (s * (1 + pre * (1 - s)))
    gp = F.pad(go, (0, width - 1))
    dx = torch.zeros_like(xx)
    dw = torch.zeros_like(ww)
    for k in range(width):
        shift = width - 1 - k
        dx = dx + ww[None, :, k, None] * gp[:, :, shift:shift + length]
        dw[:, k] = (go * padded[:, :, k:k + length]).sum((0, 2))
    return dx, dw, go.sum((0, 2))


# --------------------------------------------------------------------------
# a3/a4. fused Mamba inner op (reference: ssi.py:171-229 fwd; :641-675 mamba_inner_ref)
# --------------------------------------------------------------------------
def mamba_inner_no_out_proj(xz, conv_w, conv_b, x_proj_w, dt_proj_w, A, D, delta_bias,
                            scan=selective_scan, work_dtype=None):
    """xz: (b, 2e, l) -> out_z (b, e, l).  Order of operations = ssi.py:180-229."""
    wd = work_dtype or torch.float32
    e = xz.shape[1] // 2
    length = xz.shape[-1]
    rank = dt_proj_w.shape[1]
    nstate = A.shape[1]
    x, z = xz[:, :e], xz[:, e:]                                          # ssi.py:180
    cw = conv_w.reshape(conv_w.shape[0], conv_w.shape[-1])               # ssi.py:179
    u = causal_conv1d(x, cw, conv_b, True, work_dtype=wd)                # ssi.py:182
    tok = u.permute(0, 2, 1).reshape(-1, e).to(wd)                       # (b l) e
    x_dbl = tok @ x_proj_w.to(wd).t()                                    # ssi.py:186
    delta = (dt_proj_w.to(wd) @ x_dbl[:, :rank].t()).reshape(e, -1, length).permute(1, 0, 2)  # ssi.py:187
    Bm = x_dbl[:, rank:rank + nstate].reshape(-1, length, nstate).permute(0, 2, 1)   # ssi.py:193-198
    Cm = x_dbl[:, -nstate:].reshape(-1, length, nstate).permute(0, 2, 1)             # ssi.py:205-210
    return scan(u, delta.contiguous(), A, Bm.contiguous(), Cm.contiguous(), D, z=z,
                delta_bias=delta_bias, delta_softplus=True)              # ssi.py:218-220


def mamba_inner(xz, conv_w, conv_b, x_proj_w, dt_proj_w, out_proj_w, out_proj_b, A, D, delta_bias,
                scan=selective_scan, work_dtype=None):
    """ssi.py:297-370 / mamba_inner_ref :641-675 — inner op followed by out_proj."""
    oz = mamba_inner_no_out_proj(xz, conv_w, conv_b, x_proj_w, dt_proj_w, A, D, delta_bias, scan, work_dtype)
    return F.linear(oz.permute(0, 2, 1).to(out_proj_w.dtype), out_proj_w, out_proj_b)   # ssi.py:370


# --------------------------------------------------------------------------
# a7. BiMamba v2 mixer (reference: modules/mamba/bimamba.py:176-253)
# --------------------------------------------------------------------------
def bimamba_v2(p: Dict[str, torch.Tensor], hidden, scan=selective_scan, prefix="", work_dtype=None):
    """hidden (b, l, d) -> (b, l, d).  ``p`` uses the reference state_dict key names
    (in_proj.weight, conv1d.weight, ..., A_b_log, D_b, out_proj.weight)."""
    g = lambda k: p[prefix + k]
    wd = work_dtype or hidden.dtype
    hs = hidden.to(wd)
    b, l, d = hs.shape
    xz = (g("in_proj.weight").to(wd) @ hs.reshape(b * l, d).t()).reshape(-1, b, l).permute(1, 0, 2)  # :192-196
    A = -torch.exp(g("A_log").float())                                   # :200
    A_b = -torch.exp(g("A_b_log").float())                               # :222
    fwd = mamba_inner_no_out_proj(xz, g("conv1d.weight"), g("conv1d.bias"), g("x_proj.weight"),
                                  g("dt_proj.weight"), A, g("D").float(), g("dt_proj.bias").float(),
                                  scan, work_dtype)                      # :223-235
    bwd = mamba_inner_no_out_proj(xz.flip(-1), g("conv1d_b.weight"), g("conv1d_b.bias"),
                                  g("x_proj_b.weight"), g("dt_proj_b.weight"), A_b, g("D_b").float(),
                                  g("dt_proj_b.bias").float(), scan, work_dtype)   # :236-248
    mix = 0.5 * fwd + 0.5 * bwd.flip(-1)                                 # :253 (if_devide_out=True)
    return F.linear(mix.permute(0, 2, 1).to(wd), g("out_proj.weight").to(wd))


def mamba_uni(p: Dict[str, torch.Tensor], hidden, scan=selective_scan, prefix="", work_dtype=None):
    """Unidirectional mamba_ssm.Mamba forward = in_proj -> mamba_inner_fn (ssi.py:297-370);
    used by MambaDecoderLayer (Conmamba.py:854-862)."""
    g = lambda k: p[prefix + k]
    wd = work_dtype or hidden.dtype
    hs = hidden.to(wd)
    b, l, d = hs.shape
    xz = (g("in_proj.weight").to(wd) @ hs.reshape(b * l, d).t()).reshape(-1, b, l).permute(1, 0, 2)
    A = -torch.exp(g("A_log").float())
    return mamba_inner(xz, g("conv1d.weight"), g("conv1d.bias"), g("x_proj.weight"), g("dt_proj.weight"),
                       g("out_proj.weight").to(wd), None, A, g("D").float(), g("dt_proj.bias").float(),
                       scan, work_dtype)


# --------------------------------------------------------------------------
# a10-a12. ConvolutionModule / ConmambaEncoderLayer / ConmambaEncoder
# (reference: modules/Conmamba.py:439-449, :631-650, :716-727)
# --------------------------------------------------------------------------
def _ln(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w.to(x.dtype), b.to(x.dtype), eps)


def conv_module(p, x, prefix="", kernel_size=31):
    """Conmamba.py:439-449 (non-causal, non-chunked branch), GELU activation
    (Transformer.py:746 passes branchformer_activation = nn.GELU)."""
    g = lambda k: p[prefix + k].to(x.dtype)
    out = _ln(x, g("layer_norm.weight"), g("layer_norm.bias"), 1e-5)            # :439
    out = out.transpose(1, 2)                                                   # :440
    out = F.glu(F.conv1d(out, g("bottleneck.0.weight"), g("bottleneck.0.bias")), dim=1)   # :441
    out = F.conv1d(out, g("conv.weight"), g("conv.bias"), padding=(kernel_size - 1) // 2,
                   groups=out.shape[1])                                         # :442
    out = out.transpose(1, 2)                                                   # :448
    out = _ln(out, g("after_conv.0.weight"), g("after_conv.0.bias"), 1e-5)      # :449 (LN)
    out = F.gelu(out)
    return F.linear(out, g("after_conv.2.weight"), g("after_conv.2.bias"))


def ffn_module(p, x, prefix):
    """Conmamba.py:597-617: LayerNorm -> PositionalwiseFeedForward (Linear, GELU, Dropout, Linear)."""
    g = lambda k: p[prefix + k].to(x.dtype)
    h = _ln(x, g("0.weight"), g("0.bias"), 1e-5)
    h = F.gelu(F.linear(h, g("1.ffn.0.weight"), g("1.ffn.0.bias")))
    return F.linear(h, g("1.ffn.3.weight"), g("1.ffn.3.bias"))


def encoder_layer(p, x, prefix="", scan=selective_scan, kernel_size=31):
    """ConmambaEncoderLayer.forward, Conmamba.py:631-650 (eval mode: dropout off)."""
    g = lambda k: p[prefix + k].to(x.dtype)
    x = x + 0.5 * ffn_module(p, x, prefix + "ffn_module1.")                     # :638
    skip = x
    h = _ln(x, g("norm1.norm.weight"), g("norm1.norm.bias"), 1e-5)              # :641
    x = bimamba_v2(p, h, scan, prefix + "mamba.") + skip                        # :642-643
    x = x + conv_module(p, x, prefix + "convolution_module.", kernel_size)      # :645
    y = x + 0.5 * ffn_module(p, x, prefix + "ffn_module2.")
    return _ln(y, g("norm2.norm.weight"), g("norm2.norm.bias"), 1e-5)           # :649


def encoder(p, x, num_layers, prefix="", scan=selective_scan, kernel_size=31):
    """ConmambaEncoder.forward, Conmamba.py:716-727 (final LayerNorm eps 1e-6, :687)."""
    for i in range(num_layers):
        x = encoder_layer(p, x, f"{prefix}layers.{i}.", scan, kernel_size)
    return _ln(x, p[prefix + "norm.norm.weight"].to(x.dtype), p[prefix + "norm.norm.bias"].to(x.dtype), 1e-6)


def decoder_layer(p, tgt, memory, prefix="", scan=selective_scan, act=F.relu):
    """MambaDecoderLayer.forward with normalize_before=True, Conmamba.py:914-953.  ``act`` = the layer's activation
    (class default nn.ReLU, Conmamba.py:846; the S2S recipes pass GELU, hparams/S2S/conmambamamba_large.yaml:259)."""
    g = lambda k: p[prefix + k].to(tgt.dtype)
    t1 = _ln(tgt, g("norm1.norm.weight"), g("norm1.norm.bias"), 1e-6)
    tgt = tgt + mamba_uni(p, t1, scan, prefix + "self_mamba.")                       # :920-923
    t1 = _ln(tgt, g("norm2.norm.weight"), g("norm2.norm.bias"), 1e-6)
    cat = torch.cat([memory, t1], dim=1)
    tgt = tgt + mamba_uni(p, cat, scan, prefix + "cross_mamba.")[:, -t1.shape[1]:]   # :934-937
    t1 = _ln(tgt, g("norm3.norm.weight"), g("norm3.norm.bias"), 1e-6)
    h = act(F.linear(t1, g("pos_ffn.ffn.0.weight"), g("pos_ffn.ffn.0.bias")))
    return tgt + F.linear(h, g("pos_ffn.ffn.3.weight"), g("pos_ffn.ffn.3.bias"))     # :946-949


def decoder(p, tgt, memory, num_layers, prefix="", scan=selective_scan, act=F.relu):
    """MambaDecoder.forward, Conmamba.py:1017-1031: the layers, then the final LayerNorm (eps 1e-6)."""
    for i in range(num_layers):
        tgt = decoder_layer(p, tgt, memory, f"{prefix}layers.{i}.", scan, act)
    return _ln(tgt, p[prefix + "norm.norm.weight"].to(tgt.dtype), p[prefix + "norm.norm.bias"].to(tgt.dtype), 1e-6)


def positional_encoding(length: int, d_model: int, dtype=torch.float32) -> torch.Tensor:
    """PositionalEncoding, Transformer.py:992-1022: pe[t, 2i] = sin(t / 10000^(2i/d)), pe[t, 2i+1] = cos(...); the table
    is built in fp32 (positions * exp(arange * -(ln 10000 / d))) exactly as the reference builds it; returns (1, length, d)."""
    pe = torch.zeros(length, d_model)
    pos = torch.arange(0, length).unsqueeze(1).float()
    den = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(pos * den)
    pe[:, 1::2] = torch.cos(pos * den)
    return pe.unsqueeze(0).to(dtype)


def transformer_asr_forward(p, src, tgt, num_enc, num_dec, prefix="", scan=selective_scan, act=F.gelu):
    """TransformerASR.forward with encoder_module 'conmamba', decoder_module 'mamba', attention_type 'RelPosMHAXL'
    (TransformerASR.py:745-819), eval mode: 4-D src is flattened (:760-762), custom_src_module = Linear (+ Dropout)
    (:727-735), no positional encoding on src in the RelPosMHAXL branch (:777-778: computed, unused by ConMamba), the
    ConMamba encoder ignores every mask (Conmamba.py:631-635); tgt -> NormalizedEmbedding = Embedding * sqrt(d_model)
    (Transformer.py:1851-1860) + PositionalEncoding (TransformerASR.py:793-796) -> MambaDecoder (masks unused,
    Conmamba.py:914-953).  -> (encoder_out, decoder_out)."""
    if src.dim() == 4:
        src = src.reshape(src.shape[0], src.shape[1], -1)
    x = F.linear(src, p[prefix + "custom_src_module.layers.0.w.weight"].to(src.dtype),
                 p[prefix + "custom_src_module.layers.0.w.bias"].to(src.dtype))
    enc = encoder(p, x, num_enc, prefix + "encoder.", scan)
    if num_dec == 0:
        return enc, None
    emb = p[prefix + "custom_tgt_module.layers.0.emb.Embedding.weight"].to(src.dtype)
    d_model = emb.shape[1]
    t = F.embedding(tgt.long(), emb) * math.sqrt(d_model)
    t = t + positional_encoding(t.shape[1], d_model, t.dtype)
    return enc, decoder(p, t, enc, num_dec, prefix + "decoder.", scan, act)


def mamba_step(p, hidden, conv_state, ssm_state, prefix=""):
    """One decoding step, restating the pure-torch fallback of reference modules/mamba/bimamba.py:320-365.
    hidden (b, 1, d); conv_state (b, e, w) and ssm_state (b, e, n) are updated in place.  -> out (b, 1, d)."""
    g = lambda k: p[prefix + k]
    rank, n = g("dt_proj.weight").shape[1], g("A_log").shape[1]
    xz = F.linear(hidden[:, 0], g("in_proj.weight"))                                # :323
    e = xz.shape[-1] // 2
    x, z = xz[:, :e], xz[:, e:]                                                     # :324
    conv_state.copy_(torch.roll(conv_state, shifts=-1, dims=-1))                    # :328
    conv_state[:, :, -1] = x                                                        # :329
    x = (conv_state * g("conv1d.weight").reshape(e, -1)).sum(-1) + g("conv1d.bias")  # :330-332
    x = F.silu(x)                                                                   # :333
    x_db = F.linear(x, g("x_proj.weight"))                                          # :343
    dt, Bm, Cm = x_db[:, :rank], x_db[:, rank:rank + n], x_db[:, rank + n:]         # :344
    dt = F.linear(dt, g("dt_proj.weight"))                                          # :346 (no bias here)
    A = -torch.exp(g("A_log").float())                                              # :347
    dt = F.softplus(dt + g("dt_proj.bias"))                                         # :352
    dA = torch.exp(dt[:, :, None] * A[None])                                        # :353
    dB = dt[:, :, None] * Bm[:, None, :]                                            # :354
    ssm_state.copy_(ssm_state * dA + x[:, :, None] * dB)                            # :355
    y = (ssm_state * Cm[:, None, :]).sum(-1) + g("D") * x                           # :356-357
    y = y * F.silu(z)                                                               # :358
    return F.linear(y, g("out_proj.weight"))[:, None]                               # :364-365


# --------------------------------------------------------------------------
# a15. frontend (speechbrain 1.0.0 semantics restated; PARITY UNPINNED — see header)
# --------------------------------------------------------------------------
def mel_filterbank(n_mels=80, n_fft=512, sample_rate=16000, f_min=0.0, f_max=8000.0) -> torch.Tensor:
    """speechbrain.processing.features.Filterbank (triangular, mel scale 2595*log10(1+f/700)):
    returns (n_fft//2+1, n_mels)."""
    to_mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)
    mel = torch.linspace(to_mel(f_min), to_mel(f_max), n_mels + 2)
    hz = 700.0 * (10.0 ** (mel / 2595.0) - 1.0)
    band = (hz[1:] - hz[:-1])[:-1]
    f_central = hz[1:-1]
    all_freqs = torch.linspace(0, sample_rate // 2, n_fft // 2 + 1)
    slope = (all_freqs[None, :] - f_central[:, None]) / band[:, None]
    left, right = slope + 1.0, -slope + 1.0
    fb = torch.clamp(torch.minimum(left, right), min=0.0)
    return fb.t().contiguous()


def fbank(wav: torch.Tensor, n_fft=512, win_ms=25, hop_ms=10, n_mels=80, sample_rate=16000,
          top_db=80.0, amin=1e-10) -> torch.Tensor:
    """Fbank = STFT (hamming, center, constant pad) -> power spectrum -> mel -> dB with
    per-utterance top_db clamp.  wav (b, samples) -> (b, frames, n_mels), fp32
    (call site: train_CTC.py:285; config conmamba_large.yaml:322-326)."""
    win = int(round(sample_rate / 1000.0 * win_ms))
    hop = int(round(sample_rate / 1000.0 * hop_ms))
    window = torch.hamming_window(win, dtype=torch.float32)
    spec = torch.stft(wav.float(), n_fft, hop, win, window, center=True, pad_mode="constant",
                      normalized=False, onesided=True, return_complex=True)
    power = (spec.real ** 2 + spec.imag ** 2).transpose(1, 2)                   # (b, t, f)
    mel = power @ mel_filterbank(n_mels, n_fft, sample_rate, 0.0, sample_rate / 2)
    db = 10.0 * torch.log10(torch.clamp(mel, min=amin))                         # multiplier 10 (power)
    floor = db.amax(dim=(-2, -1), keepdim=True) - top_db
    return torch.maximum(db, floor)


def global_norm_stats(feats: torch.Tensor, lens: torch.Tensor):
    """InputNormalization(norm_type='global') first-batch statistics: mean over utterances of
    the per-utterance (valid-frame) mean / std per mel bin."""
    means, stds = [], []
    for i in range(feats.shape[0]):
        n = int(torch.round(lens[i] * feats.shape[1]))
        means.append(feats[i, :n].mean(0))
        stds.append(feats[i, :n].std(0).clamp(min=1e-10))
    return torch.stack(means).mean(0), torch.stack(stds).mean(0)


def cnn_frontend(p, feats, prefix=""):
    """ConvolutionFrontEnd(2 blocks, 1 layer each, channels (64, 32), k=3, stride 2 in time and frequency,
    'same' reflect padding, LayerNorm over (freq, ch), LeakyReLU(0.01)) — conmamba_large.yaml:187-194; eval
    mode.  feats (b, t, 80) -> (b, ceil(t/4), 20, 32).  Parameter names: blocks.{i}.conv.{weight,bias},
    blocks.{i}.norm.norm.{weight,bias} (the build's naming; speechbrain's own key names are unpinned)."""
    x = feats[:, :, :, None]                                                    # b t f 1 (channels-last)
    for blk in range(2):
        w = p[f"{prefix}blocks.{blk}.conv.weight"].to(x.dtype)                  # (co, ci, 3, 3)
        bb = p[f"{prefix}blocks.{blk}.conv.bias"].to(x.dtype)
        xin = F.pad(x.permute(0, 3, 1, 2), (1, 1, 1, 1), mode="reflect")        # b c t f, pad k//2
        y = F.conv2d(xin, w, bb, stride=2).permute(0, 2, 3, 1)                  # b t f c
        lw = p[f"{prefix}blocks.{blk}.norm.norm.weight"].to(x.dtype)
        lb = p[f"{prefix}blocks.{blk}.norm.norm.bias"].to(x.dtype)
        x = F.leaky_relu(F.layer_norm(y, tuple(y.shape[-2:]), lw, lb, 1e-5), 0.01)
    return x


def asr_encode(p, wav, wav_lens, num_layers, norm_mean, norm_std, scan=selective_scan, n_fft=512, win_ms=25):
    """Full CTC encoder forward in eval mode, train_CTC.py:285-298: Fbank -> global normalisation -> CNN front
    end -> reshape + Linear (TransformerASR.py:760-773) -> ConmambaEncoder.  ``p`` holds 'CNN.*' and
    'Transformer.*' parameters under the reference state_dict names."""
    feats = fbank(wav, n_fft=n_fft, win_ms=win_ms)
    feats = (feats - norm_mean) / norm_std
    src = cnn_frontend(p, feats, "CNN.")
    b, t = src.shape[:2]
    src = src.reshape(b, t, -1)
    src = F.linear(src, p["Transformer.custom_src_module.layers.0.w.weight"],
                   p["Transformer.custom_src_module.layers.0.w.bias"])
    return encoder(p, src, num_layers, "Transformer.encoder.", scan)


def ctc_loss_batchmean(log_probs, targets, in_lens_rel, tgt_lens_rel, blank=0):
    """speechbrain.nnet.losses.ctc_loss(reduction='batchmean') (call site train_CTC.py:405):
    F.ctc_loss(sum, zero_infinity) / batch with lengths = round(rel * max_len)."""
    b, t, _ = log_probs.shape
    il = torch.round(in_lens_rel * t).int()
    tl = torch.round(tgt_lens_rel * targets.shape[1]).int()
    loss = F.ctc_loss(log_probs.transpose(0, 1).float(), targets, il, tl, blank, reduction="sum",
                      zero_infinity=True)
    return loss / b
