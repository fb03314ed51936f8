"""Tensor-level wrappers over the C ABI (include/conmamba_hip.h).

These play the role of the reference's binary extension modules
``selective_scan_cuda`` / ``causal_conv1d_cuda`` (reference
modules/mamba/selective_scan_interface.py:15-16): torch tensors in, torch tensors out, all
arithmetic in the HIP library.  PyTorch is used for device memory and the stream only.
"""
from __future__ import annotations

import ctypes as ct
from typing import Optional, Tuple

import os
import contextlib

import torch

from . import _native as N

_DT = {torch.float32: N.CM_F32, torch.bfloat16: N.CM_BF16, torch.float16: N.CM_F16}


# Gradient reductions across workgroups (scan backward: dA / dB / dC / dD / ddelta_bias; causal conv backward: dweight /
# dbias) run as per-workgroup partials + a fixed-order second pass: bit-identical gradients from run to run (what
# SURVEY.md §8d config 4's 1-GPU vs 8-GPU comparison needs to be meaningful at fp32).  CM_DETERMINISTIC=0 switches to
# the fp32 atomics the CUDA kernels behind the reference use (no workspace, order-dependent last bits).
DETERMINISTIC = os.environ.get("CM_DETERMINISTIC", "1") == "1"


# When set to a list, every native launch is bracketed by HIP events recorded on the stream the kernel
# is launched on; entries are (kernel_name, start_event, end_event, units).  Used by bench.py's roofline leg.
LAUNCH_LOG: Optional[list] = None


def _launch(name: str, fn, args, units: int = 0):
    log = LAUNCH_LOG
    if log is None:
        N.check(fn(ct.byref(args)), name)
        return
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    N.check(fn(ct.byref(args)), name)
    e1.record(st)
    log.append((name, e0, e1, units))


# Weight-derived caches under hipGraph replay (brain.Brain graph_steps).  A captured launch reads a cached tensor by ADDRESS, so the
# caches must keep their storage for as long as a graph lives and be refreshed in place:
#   CACHE_INPLACE     a miss that finds an older entry of the same shape / dtype rewrites that entry's tensor instead of allocating
#   forced_refresh()  context: every entry's FIRST lookup is treated as a miss (while the "fresh" variant of a graph is captured: the
#                     refresh kernels land in the graph) and (owner, attribute) of every refreshed entry is collected, so that the keys
#                     of exactly those entries can be brought up to date after a replay did the refresh (rekey_caches) -- Python does
#                     not see a replay's kernels
CACHE_INPLACE = False
CACHE_GENERATION = 0                              # bumped whenever an entry gets NEW storage or entries are dropped: graphs captured
                                                  # before the bump may hold addresses the caches no longer own (brain drops them)
_FORCE = None                                     # None, or the set of (id(owner), attribute) already refreshed in this forced pass
_LOG = None


@contextlib.contextmanager
def forced_refresh():
    global _FORCE, _LOG
    old = (_FORCE, _LOG)
    _FORCE, _LOG = set(), []
    try:
        yield _LOG
    finally:
        _FORCE, _LOG = old


def _cache_hit(owner, attr, key_matches: bool) -> bool:
    if not key_matches:
        return False
    return _FORCE is None or (id(owner), attr) in _FORCE


def _cache_new_storage():
    global CACHE_GENERATION
    CACHE_GENERATION += 1
    if os.environ.get("CM_CACHE_TRACE"):                              # who allocates: one line per bump
        import traceback
        print("cache generation", CACHE_GENERATION, " <- ".join(f"{f.name}:{f.lineno}" for f in traceback.extract_stack()[-5:-1]), flush=True)


def _cache_note(owner, attr):
    if _FORCE is not None:
        _FORCE.add((id(owner), attr))
        _LOG.append((owner, attr))


def rekey_caches(entries) -> None:
    """entries: what forced_refresh() collected.  Marks each entry as holding the CURRENT version of its parameter(s): call only
    right after the kernels that refresh exactly these entries ran (a replay of the graph they were captured into)."""
    for owner, attr in entries:
        c = getattr(owner, attr, None)
        if c is None:
            continue
        if attr == "_cm_rows_derived":
            c.key = c.make_key(owner, c.cdt)
        else:
            setattr(owner, attr, ((owner._version, owner.data_ptr()), c[1]))


def cast_cached(p: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """p in `dtype`, without autograd, cached on the tensor until it is modified in place (optimizer step, load_state_dict,
    broadcast): the autograd nodes of this package use a weight's bf16 copy in forward AND backward, several times per
    step, and autocast's own cache does not reach into custom Functions."""
    if p.dtype == dtype:
        return p.detach()
    c = getattr(p, "_cm_cast", None)
    key = (p._version, p.data_ptr())             # data_ptr: `p.data = ...` swaps storage without bumping _version
    usable = c is not None and c[1].dtype == dtype and c[1].device == p.device
    if usable and _cache_hit(p, "_cm_cast", c[0] == key):
        return c[1]
    if CACHE_INPLACE and usable and c[1].shape == p.shape:
        c[1].copy_(p.detach())                   # same storage, new values
        p._cm_cast = (key, c[1])
        _cache_note(p, "_cm_cast")
        return c[1]
    t = p.detach().to(dtype)
    try:
        p._cm_cast = (key, t)
        _cache_new_storage()
        _cache_note(p, "_cm_cast")
    except (AttributeError, RuntimeError):
        pass
    return t


def reflect_pad_tf(x, pad, backward=False, shape=None):
    """Reflect padding of the time and frequency axes of a channels-last (batch, time, freq, channels) tensor (cm_reflect_pad_tf).
    backward: x is the padded tensor's gradient and ``shape`` the source's shape -> the gradient folded back."""
    _dev_check(x)
    if x.dim() != 4 or not x.is_contiguous() or x.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("reflect_pad_tf: expected a contiguous (batch, time, freq, channels) fp32 / bf16 tensor")
    if backward:
        b, t, f, c = shape
        if tuple(x.shape) != (b, t + 2 * pad, f + 2 * pad, c):
            raise RuntimeError("reflect_pad_tf: gradient shape does not match the padded source shape")
        out = torch.empty((b, t, f, c), dtype=x.dtype, device=x.device)
    else:
        b, t, f, c = x.shape
        out = torch.empty((b, t + 2 * pad, f + 2 * pad, c), dtype=x.dtype, device=x.device)
    rc = N.lib().cm_reflect_pad_tf(_ptr(x), _ptr(out), b, t, f, c, int(pad), _DT[x.dtype], int(bool(backward)), _stream())
    N.check(rc, "cm_reflect_pad_tf")
    return out


class ReflectPadTfFn(torch.autograd.Function):
    """autograd node over cm_reflect_pad_tf (torch: F.pad(..., mode='reflect') on the 5-d view, 448 / 851 us at block 2's size)."""

    @staticmethod
    def forward(ctx, x, pad):
        ctx.pad, ctx.shape = pad, tuple(x.shape)
        return reflect_pad_tf(x if x.is_contiguous() else x.contiguous(), pad)

    @staticmethod
    def backward(ctx, dy):
        return reflect_pad_tf(dy if dy.is_contiguous() else dy.contiguous(), ctx.pad, backward=True, shape=ctx.shape), None


def wgrad(a, b, nbatch=None, variant=0):
    """dW (M, N) fp32 = a^T b for a (rows, M), b (rows, N): the weight gradient of a Linear from its output gradient and its input.
    bf16 operands with M, N multiples of 128: cm_wgrad_bf16 (split over row chunks, fixed-order fold).  Otherwise ``nbatch``
    batched GEMMs over equal row chunks folded by cm_sum_leading (fp32 accumulation in a fixed order either way)."""
    _dev_check(a, b)
    rows, M = a.shape
    Nn = b.shape[1]
    if (a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and a.stride(1) == 1 and b.stride(1) == 1 and a.stride(0) % 8 == 0
            and b.stride(0) % 8 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0 and N.lib().cm_wgrad_supported(rows, M, Nn)):
        args = N.WgradArgs()
        args.rows, args.m, args.n, args.variant = rows, M, Nn, int(variant)   # 0: the library picks; 1 registers, 2 LDS-DMA staging
        args.a, args.b, args.lda, args.ldb = _ptr(a), _ptr(b), a.stride(0), b.stride(0)
        nws = int(N.lib().cm_wgrad_workspace_floats(rows, M, Nn))
        ws = torch.empty((nws,), dtype=torch.float32, device=a.device)
        out = torch.empty((M, Nn), dtype=torch.float32, device=a.device)
        args.out, args.workspace, args.workspace_floats, args.stream = _ptr(out), _ptr(ws), nws, _stream()
        _launch("cm_wgrad_bf16", N.lib().cm_wgrad_bf16, args, units=rows)
        return out
    nb = nbatch if (nbatch and rows % nbatch == 0) else 1
    return sum_leading(torch.bmm(a.view(nb, rows // nb, M).transpose(1, 2), b.view(nb, rows // nb, Nn)))


def pack_cached(p: torch.Tensor) -> "PackedWeight":
    """p (2-D parameter) as a bf16 PackedWeight (cm_ffn_fused's fragment-tiled image), cached like cast_cached."""
    c = getattr(p, "_cm_pack", None)
    key = (p._version, p.data_ptr())
    usable = c is not None and c[1].data.device == p.device
    if usable and _cache_hit(p, "_cm_pack", c[0] == key):
        return c[1]
    if CACHE_INPLACE and usable and c[1].shape == tuple(p.shape):
        c[1].repack_(cast_cached(p, torch.bfloat16))
        p._cm_pack = (key, c[1])
        _cache_note(p, "_cm_pack")
        return c[1]
    pw = PackedWeight(cast_cached(p, torch.bfloat16))
    try:
        p._cm_pack = (key, pw)
        _cache_new_storage()
        _cache_note(p, "_cm_pack")
    except (AttributeError, RuntimeError):
        pass
    return pw


def pack_cached_t(p: torch.Tensor) -> "PackedWeight":
    """p^T as a bf16 PackedWeight (the "weights" of cm_ffn_bwd_fused's two GEMMs), cached like cast_cached."""
    c = getattr(p, "_cm_pack_t", None)
    key = (p._version, p.data_ptr())
    usable = c is not None and c[1].data.device == p.device
    if usable and _cache_hit(p, "_cm_pack_t", c[0] == key):
        return c[1]
    if CACHE_INPLACE and usable and c[1].shape == tuple(p.shape)[::-1]:
        c[1].repack_(cast_cached(p, torch.bfloat16).t().contiguous())
        p._cm_pack_t = (key, c[1])
        _cache_note(p, "_cm_pack_t")
        return c[1]
    pw = PackedWeight(cast_cached(p, torch.bfloat16).t().contiguous())
    try:
        p._cm_pack_t = (key, pw)
        _cache_new_storage()
        _cache_note(p, "_cm_pack_t")
    except (AttributeError, RuntimeError):
        pass
    return pw


def ffn_bwd_fused(dout, w2t, w1t, pre, alpha, p1, p2, seed1, seed2):
    """The data-gradient chain of a feed-forward module's backward in one kernel (cm_ffn_bwd_fused), for the forward
    ffn_fused(..., train=(p1, p2, seed1, seed2)) ran.  dout (rows, 256) fp32; w2t / w1t PackedWeight of W2^T (hidden, 256) /
    W1^T (256, hidden); pre (rows, hidden) bf16.  -> (da2, da1, act, dh, db1, db2): bf16 (rows, 256) / (rows, hidden) x 2 /
    (rows, 256), fp32 (hidden) / (256)."""
    _dev_check(dout, pre)
    rows, d = dout.shape
    hidden = pre.shape[1]
    if dout.dtype != torch.float32 or not dout.is_contiguous() or pre.dtype != torch.bfloat16 or not pre.is_contiguous() or pre.shape[0] != rows:
        raise RuntimeError("ffn_bwd_fused: dout must be contiguous fp32 (rows, 256), pre contiguous bf16 (rows, hidden)")
    if w2t.shape != (hidden, d) or w1t.shape != (d, hidden):
        raise RuntimeError("ffn_bwd_fused: packed weight shapes do not match")
    dev = dout.device
    da2 = torch.empty((rows, d), dtype=torch.bfloat16, device=dev)
    dh = torch.empty((rows, d), dtype=torch.bfloat16, device=dev)
    da1 = torch.empty((rows, hidden), dtype=torch.bfloat16, device=dev)
    act = torch.empty((rows, hidden), dtype=torch.bfloat16, device=dev)
    nws = int(N.lib().cm_ffn_bwd_workspace_floats(rows, hidden))
    ws = torch.empty((nws + hidden + d,), dtype=torch.float32, device=dev)
    db1, db2 = ws[nws:nws + hidden], ws[nws + hidden:]
    a = N.FfnBwdArgs()
    a.rows, a.dim, a.hidden = rows, d, hidden
    a.dout, a.w2t, a.w1t, a.pre = _ptr(dout), _ptr(w2t.data), _ptr(w1t.data), _ptr(pre)
    a.da2, a.da1, a.act, a.dh, a.db1, a.db2 = _ptr(da2), _ptr(da1), _ptr(act), _ptr(dh), _ptr(db1), _ptr(db2)
    a.alpha, a.p1, a.p2, a.seed1, a.seed2 = float(alpha), float(p1), float(p2), int(seed1), int(seed2)
    a.seed_epoch = _seed_epoch_ptr()
    a.workspace, a.workspace_floats, a.stream = _ptr(ws), nws, _stream()
    _launch("cm_ffn_bwd_fused", N.lib().cm_ffn_bwd_fused, a, units=rows)
    return da2, da1, act, dh, db1, db2


def invalidate_caches(module: torch.nn.Module) -> None:
    """Drop every cached low-precision weight copy under ``module`` (cast_cached's per-parameter copies and the fused
    path's per-layer images).  The caches key on (parameter version, storage pointer): in-place writes made under
    torch.no_grad() on the parameter itself, optimizer steps, load_state_dict and `p.data = new` are seen; writes
    THROUGH ``p.data`` (``p.data.copy_()``, EMA / SWA code, vector_to_parameters) are not -- call this after them.
    load_state_dict calls it by itself (hook installed by asr.ConMambaASR)."""
    _cache_new_storage()
    for p in module.parameters():
        for attr in ("_cm_pack", "_cm_pack_t", "_cm_cast"):
            if hasattr(p, attr):
                try:
                    delattr(p, attr)
                except AttributeError:
                    pass
    for m in module.modules():
        m.__dict__.pop("_cm_plist", None)
        for attr in ("_cm_fused_cache", "_cm_frontend_cache", "_cm_rows_derived"):
            if hasattr(m, attr):
                delattr(m, attr)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _dev_check(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("mamba_asr_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")


def _time_contig(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    # same rule as the reference wrappers (selective_scan_interface.py:24-35): last dim must have stride 1
    if t is None:
        return None
    return t if t.stride(-1) == 1 else t.contiguous()


def _f32c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    return t.detach().to(torch.float32).contiguous()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    # the raw handle of the current stream: torch.cuda.current_stream() builds a Stream object per call (2.5 us x 220 launches per
    # forward; the training step is host-bound: 44.7 ms to enqueue 47 ms of kernels)
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def num_chunks(seqlen: int) -> int:
    return (seqlen + N.CM_SCAN_CHUNK - 1) // N.CM_SCAN_CHUNK


def _bc4(t: torch.Tensor) -> torch.Tensor:
    if t.dim() == 3:
        t = t.unsqueeze(1)
    if t.dim() != 4 or t.shape[1] != 1:
        raise RuntimeError("B/C must be (batch, dstate, seqlen) or (batch, 1, dstate, seqlen): one group only")
    return _time_contig(t)


def _fill_scan_args(a, u, delta, A, B, C_, D, z, delta_bias, delta_softplus, reverse):
    b, d, l = u.shape
    a.batch, a.dim, a.seqlen, a.dstate = b, d, l, A.shape[1]
    a.io_dtype, a.bc_dtype = _DT[u.dtype], _DT[B.dtype]
    a.delta_softplus, a.reverse_time = int(bool(delta_softplus)), int(bool(reverse))
    a.u, a.delta, a.A, a.B, a.C = _ptr(u), _ptr(delta), _ptr(A), _ptr(B), _ptr(C_)
    a.D, a.z, a.delta_bias = _ptr(D), _ptr(z), _ptr(delta_bias)
    a.u_bs, a.u_ds = u.stride(0), u.stride(1)
    a.delta_bs, a.delta_ds = delta.stride(0), delta.stride(1)
    if z is not None:
        a.z_bs, a.z_ds = z.stride(0), z.stride(1)
    a.B_bs, a.B_ns = B.stride(0), B.stride(2)
    a.C_bs, a.C_ns = C_.stride(0), C_.stride(2)
    a.stream = _stream()


def selective_scan_fwd(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                       reverse=False, need_out=True, need_x=True, out_z_buf: Optional[torch.Tensor] = None,
                       h0: Optional[torch.Tensor] = None, split: int = 0):
    """-> (out, x, out_z): what selective_scan_cuda.fwd returns (selective_scan_interface.py:42).
    ``out`` is the pre-gate output (None when z is given and need_out is False), ``x`` the
    checkpoint tensor (batch, dim, nchunks, 2*dstate) or None, ``out_z`` the gated output or None.
    ``h0`` (batch, dim, dstate) fp32: state the recurrence starts from (time-split scans, seqpar.py).
    ``split``: tuning, lanes per channel (cm_scan_fwd_args.lanes_per_channel; 0 = the library's choice)."""
    _dev_check(u, delta, A, B, C, D, z, delta_bias)
    u, delta, z = _time_contig(u), _time_contig(delta), _time_contig(z)
    if delta.dtype != u.dtype or (z is not None and z.dtype != u.dtype):
        raise RuntimeError("u, delta and z must share one dtype")
    B, C = _bc4(B), _bc4(C)
    if B.dtype != C.dtype:
        raise RuntimeError("B and C must share one dtype")
    A, D, delta_bias = _f32c(A), _f32c(D), _f32c(delta_bias)
    b, d, l = u.shape
    a = N.ScanFwdArgs()
    _fill_scan_args(a, u, delta, A, B, C, D, z, delta_bias, delta_softplus, reverse)
    out = torch.empty((b, d, l), dtype=u.dtype, device=u.device) if (z is None or need_out) else None
    out_z = None
    if z is not None:
        # out_z_buf: a pre-allocated time-contiguous (batch, dim, seqlen) view, e.g. of (dim, batch, seqlen) storage
        out_z = out_z_buf if out_z_buf is not None else torch.empty((b, d, l), dtype=u.dtype, device=u.device)
        if out_z.shape != (b, d, l) or out_z.stride(-1) != 1 or out_z.dtype != u.dtype:
            raise RuntimeError("out_z_buf must be a time-contiguous (batch, dim, seqlen) tensor of u's dtype")
    x = torch.empty((b, d, num_chunks(l), 2 * A.shape[1]), dtype=torch.float32, device=u.device) if need_x else None
    a.out, a.out_z, a.x = _ptr(out), _ptr(out_z), _ptr(x)
    if h0 is not None:
        _dev_check(h0)
        if tuple(h0.shape) != (b, d, A.shape[1]):
            raise RuntimeError(f"h0 must be (batch, dim, dstate) = {(b, d, A.shape[1])}, got {tuple(h0.shape)}")
        h0 = _f32c(h0)
        a.h0 = _ptr(h0)
    a.out_bs, a.out_ds = d * l, l
    a.lanes_per_channel = int(split)
    if out_z is not None:
        if out is not None and (out_z.stride(0), out_z.stride(1)) != (d * l, l):
            raise RuntimeError("out and a strided out_z_buf cannot be requested together (they share strides)")
        a.out_bs, a.out_ds = out_z.stride(0), out_z.stride(1)
    _launch("cm_selective_scan_fwd", N.lib().cm_selective_scan_fwd, a, units=b * l)
    return out, x, out_z


def selective_scan_bwd(u, delta, A, B, C, D, z, delta_bias, dout, x, delta_softplus=False, reverse=False,
                       dz: Optional[torch.Tensor] = None, recompute_out_z=False, du_buf: Optional[torch.Tensor] = None,
                       ddelta_buf: Optional[torch.Tensor] = None, out_z_buf: Optional[torch.Tensor] = None, split: int = 0):
    """-> (du, ddelta, dA, dB, dC, dD, ddelta_bias, dz, out_z): the tuple selective_scan_cuda.bwd
    returns (selective_scan_interface.py:67, 252).  ``dz`` may be a pre-allocated view (e.g. half of
    dxz, :249-256).  dB/dC are fp32 (batch, 1, dstate, seqlen)."""
    _dev_check(u, delta, A, B, C, D, z, delta_bias, dout, x)
    u, delta, z, dout = _time_contig(u), _time_contig(delta), _time_contig(z), _time_contig(dout)
    B, C = _bc4(B), _bc4(C)
    A, D, delta_bias = _f32c(A), _f32c(D), _f32c(delta_bias)
    b, d, l = u.shape
    n = A.shape[1]
    a = N.ScanBwdArgs()
    _fill_scan_args(a.fwd, u, delta, A, B, C, D, z, delta_bias, delta_softplus, reverse)
    if x is None:
        raise RuntimeError("selective_scan_bwd needs the forward's checkpoint tensor x")
    a.fwd.x = _ptr(x)
    a.fwd.lanes_per_channel = int(split)
    out_z = None
    def _buf(t, what):                       # optional pre-allocated time-contiguous (batch, dim, seqlen) views
        if t is None:
            return torch.empty((b, d, l), dtype=u.dtype, device=u.device)
        if t.shape != (b, d, l) or t.stride(-1) != 1 or t.dtype != u.dtype:
            raise RuntimeError(f"{what} must be a time-contiguous (batch, dim, seqlen) tensor of u's dtype")
        return t
    if z is not None and recompute_out_z:
        out_z = _buf(out_z_buf, "out_z_buf")
        a.fwd.out_z, a.fwd.out_bs, a.fwd.out_ds = _ptr(out_z), out_z.stride(0), out_z.stride(1)
    dev = u.device
    du = _buf(du_buf, "du_buf")
    ddelta = _buf(ddelta_buf, "ddelta_buf")
    if z is not None and dz is None:
        dz = torch.empty((b, d, l), dtype=u.dtype, device=dev)
    if z is not None and dz.stride(-1) != 1:
        raise RuntimeError("dz must be time-contiguous")
    dA = torch.zeros((d, n), dtype=torch.float32, device=dev)
    dB = torch.zeros((b, 1, n, l), dtype=torch.float32, device=dev)
    dC = torch.zeros((b, 1, n, l), dtype=torch.float32, device=dev)
    dD = torch.zeros((d,), dtype=torch.float32, device=dev) if D is not None else None
    dbias = torch.zeros((d,), dtype=torch.float32, device=dev) if delta_bias is not None else None
    a.dout, a.dout_bs, a.dout_ds = _ptr(dout), dout.stride(0), dout.stride(1)
    a.du, a.ddelta, a.dz = _ptr(du), _ptr(ddelta), _ptr(dz if z is not None else None)
    a.du_bs, a.du_ds, a.ddelta_bs, a.ddelta_ds = du.stride(0), du.stride(1), ddelta.stride(0), ddelta.stride(1)
    if z is not None:
        a.dz_bs, a.dz_ds = dz.stride(0), dz.stride(1)
    a.dA, a.dB, a.dC, a.dD, a.ddelta_bias = _ptr(dA), _ptr(dB), _ptr(dC), _ptr(dD), _ptr(dbias)
    if DETERMINISTIC:
        nbytes = int(N.lib().cm_selective_scan_bwd_workspace_bytes(ct.byref(a)))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)          # caching allocator: 16-byte aligned
        a.workspace, a.workspace_bytes = _ptr(ws), nbytes
    _launch("cm_selective_scan_bwd", N.lib().cm_selective_scan_bwd, a, units=b * l)
    return du, ddelta, dA, dB, dC, dD, dbias, (dz if z is not None else None), out_z


def causal_conv1d_fwd(x, weight, bias=None, silu=True, reverse=False, out: Optional[torch.Tensor] = None):
    """x (batch, dim, seqlen), weight (dim, width) -> y.  causal_conv1d_cuda.causal_conv1d_fwd
    (selective_scan_interface.py:182) with seq_idx=None."""
    _dev_check(x, weight, bias)
    x = _time_contig(x)
    w, bs = _f32c(weight), _f32c(bias)
    b, d, l = x.shape
    y = out if out is not None else torch.empty((b, d, l), dtype=x.dtype, device=x.device)
    a = N.ConvArgs()
    a.batch, a.dim, a.seqlen, a.width = b, d, l, w.shape[1]
    a.io_dtype, a.silu, a.reverse_time = _DT[x.dtype], int(bool(silu)), int(bool(reverse))
    a.x, a.weight, a.bias, a.y = _ptr(x), _ptr(w), _ptr(bs), _ptr(y)
    a.x_bs, a.x_ds, a.y_bs, a.y_ds = x.stride(0), x.stride(1), y.stride(0), y.stride(1)
    a.stream = _stream()
    _launch("cm_causal_conv1d_fwd", N.lib().cm_causal_conv1d_fwd, a, units=b * l)
    return y


def causal_conv1d_bwd(x, weight, bias, dy, silu=True, reverse=False, dx: Optional[torch.Tensor] = None):
    """-> (dx, dweight (dim, width) fp32, dbias (dim) fp32 or None); dx may be a pre-allocated view
    (selective_scan_interface.py:286-288)."""
    _dev_check(x, weight, bias, dy, dx)
    x, dy = _time_contig(x), _time_contig(dy)
    w, bs = _f32c(weight), _f32c(bias)
    b, d, l = x.shape
    if dx is None:
        dx = torch.empty((b, d, l), dtype=x.dtype, device=x.device)
    if dx.stride(-1) != 1:
        raise RuntimeError("dx must be time-contiguous")
    dw = torch.zeros_like(w)
    db = torch.zeros((d,), dtype=torch.float32, device=x.device) if bias is not None else None
    a = N.ConvArgs()
    a.batch, a.dim, a.seqlen, a.width = b, d, l, w.shape[1]
    a.io_dtype, a.silu, a.reverse_time = _DT[x.dtype], int(bool(silu)), int(bool(reverse))
    a.x, a.weight, a.bias = _ptr(x), _ptr(w), _ptr(bs)
    a.x_bs, a.x_ds = x.stride(0), x.stride(1)
    a.dy, a.dx, a.dweight, a.dbias = _ptr(dy), _ptr(dx), _ptr(dw), _ptr(db)
    a.dy_bs, a.dy_ds, a.dx_bs, a.dx_ds = dy.stride(0), dy.stride(1), dx.stride(0), dx.stride(1)
    a.stream = _stream()
    if DETERMINISTIC:
        ws = torch.empty((b * d * (w.shape[1] + 1),), dtype=torch.float32, device=x.device)
        a.workspace = _ptr(ws)
    _launch("cm_causal_conv1d_bwd", N.lib().cm_causal_conv1d_bwd, a, units=b * l)
    return dx, dw, db


# ------------------------------------------------------------------------------------------
# channels-last ops (the fused BiMamba path)
# ------------------------------------------------------------------------------------------
def _rows_ok(t: torch.Tensor, what: str):
    if t.dim() != 3 or t.stride(2) != 1:
        raise RuntimeError(f"{what} must be (batch, seqlen, dim) with the channel axis contiguous")


def alloc_bc(nrows: int, batch: int, seqlen: int, device) -> torch.Tensor:
    """(nrows, batch, seqlen) fp32 buffer whose storage is readable 16 steps past the end, as
    cm_scan_cl_fwd requires of B/C (scalar loads fetch whole 16-step groups)."""
    n = nrows * batch * seqlen
    flat = torch.empty(n + 16, dtype=torch.float32, device=device)
    flat[n:].zero_()                                     # the readable tail must be finite
    return flat[:n].view(nrows, batch, seqlen)


def rows_dt_pad(dt_rank: int) -> int:
    """Width the dt features are zero-padded to in the x_dbl rows of cm_scan_cl_fwd's xdbl mode: 16, or 32 (bf16 only)."""
    if not 1 <= dt_rank <= 32:
        raise RuntimeError(f"the xdbl mode of cm_scan_cl_fwd needs 1 <= dt_rank <= 32 (got {dt_rank})")
    return 16 if dt_rank <= 16 else 32


def pad_dt_weight(dt_weight: torch.Tensor) -> torch.Tensor:
    """(dim, dt_rank <= 32) -> (dim, 16 | 32) fp32, zero padded: the dt_weight layout of cm_scan_cl_fwd's xdbl mode."""
    d, r = dt_weight.shape
    out = torch.zeros((d, rows_dt_pad(r)), dtype=torch.float32, device=dt_weight.device)
    out[:, :r] = dt_weight.detach().float()
    return out


def _scan_cl_dir_rows(x, dd, u0, z, keep):
    """One direction descriptor in xdbl mode: x_dbl rows (batch, seqlen, P + 32) = [dt P | B16 | C16] in the I/O dtype,
    P = 16, or 32 (bf16 only) for 16 < dt_rank <= 32."""
    u, xdbl, dt_w = dd["u"], dd["xdbl"], dd["dt_weight"]
    _dev_check(u, xdbl, dt_w, dd["A"])
    _rows_ok(u, "u")
    _rows_ok(xdbl, "xdbl")
    b, l, d = u0.shape
    pad = xdbl.shape[-1] - 32
    if pad not in (16, 32) or (pad == 32 and u0.dtype != torch.bfloat16):
        raise RuntimeError("xdbl mode: xdbl rows are 48 wide, or 64 wide (dt_rank > 16) in bf16")
    if u.shape != (b, l, d) or xdbl.shape != (b, l, pad + 32) or u.dtype != u0.dtype or xdbl.dtype != u0.dtype:
        raise RuntimeError("xdbl mode: u (batch, seqlen, dim) and xdbl (batch, seqlen, 48 | 64) must share shape prefix and dtype")
    if dt_w.shape != (d, pad):
        raise RuntimeError(f"xdbl mode: dt_weight must be (dim, {pad}), zero padded (ops.pad_dt_weight)")
    A, D, bias, dt_w = _f32c(dd["A"]), _f32c(dd.get("D")), _f32c(dd.get("delta_bias")), _f32c(dt_w)
    out = dd.get("out")
    if out is None:
        out = torch.empty((b, l, d), dtype=u.dtype, device=u.device)
    _rows_ok(out, "out")
    keep += [A, D, bias, dt_w]
    x.u, x.A, x.D, x.delta_bias, x.out, x.dt_weight, x.xdbl = _ptr(u), _ptr(A), _ptr(D), _ptr(bias), _ptr(out), _ptr(dt_w), _ptr(xdbl)
    x.u_bs, x.u_ts, x.out_bs, x.out_ts = u.stride(0), u.stride(1), out.stride(0), out.stride(1)
    x.xdbl_bs, x.xdbl_ts, x.dt_rank = xdbl.stride(0), xdbl.stride(1), pad
    x.reverse_time = int(bool(dd.get("reverse", False)))
    ck, yp = dd.get("ckpt"), dd.get("ypre")                  # what the training forward saves for scan_cl_bwd
    if ck is not None:
        _dev_check(ck)
        if ck.dtype != torch.float32 or not ck.is_contiguous() or tuple(ck.shape) != (b, 2 * ((l + 15) // 16), d, 16):
            raise RuntimeError("xdbl mode: ckpt must be a contiguous fp32 (batch, 2 * ceil(seqlen / 16), dim, 16) tensor")
        keep.append(ck)
        x.ckpt = _ptr(ck)
    if yp is not None:
        _dev_check(yp)
        _rows_ok(yp, "ypre")
        if yp.shape != (b, l, d) or yp.dtype != u0.dtype:
            raise RuntimeError("xdbl mode: ypre must be (batch, seqlen, dim) in the I/O dtype")
        keep.append(yp)
        x.ypre, x.ypre_bs, x.ypre_ts = _ptr(yp), yp.stride(0), yp.stride(1)
    for key in ("h0", "h_last", "decay"):                    # carry interface of the time-split scan: (batch, dim, 16) fp32
        t = dd.get(key)
        if t is not None:
            _dev_check(t)
            if t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != (b, d, 16):
                raise RuntimeError(f"xdbl mode: {key} must be a contiguous fp32 (batch, dim, 16) tensor")
            keep.append(t)
            setattr(x, key, _ptr(t))
    return out


# time chunks of cm_scan_cl_fwd's xdbl mode: "auto" (cm_scan_cl_fwd_auto_chunks: cut sequences when the batch is too small
# to fill the chip), or a fixed count (1 = never)
SCAN_CHUNKS = os.environ.get("CM_SCAN_CHUNKS", "auto")


def scan_cl_fwd(directions, z=None, delta_softplus=True, time_chunks=None, split: int = 0):
    """Channels-last selective scan, 1 or 2 directions in one launch (cm_scan_cl_fwd).
    ``time_chunks`` (xdbl mode): None = the CM_SCAN_CHUNKS policy, else the chunk count.

    ``directions``: list of dicts with u, delta (batch, seqlen, dim), A (dim, 16), B, C (16, batch, seqlen) fp32
    time-contiguous (see alloc_bc), D, delta_bias (dim) or None, out (batch, seqlen, dim) view or None, reverse.
    Returns the list of out tensors."""
    if not 1 <= len(directions) <= 2:
        raise RuntimeError("1 or 2 directions")
    u0 = directions[0]["u"]
    _dev_check(u0, z)
    _rows_ok(u0, "u")
    b, l, d = u0.shape
    a = N.ScanClArgs()
    a.batch, a.seqlen, a.dim, a.dstate = b, l, d, 16
    a.io_dtype, a.delta_softplus, a.ndir = _DT[u0.dtype], int(bool(delta_softplus)), len(directions)
    keep = []
    if z is not None:
        _rows_ok(z, "z")
        a.z, a.z_bs, a.z_ts = _ptr(z), z.stride(0), z.stride(1)
    outs = []
    for i, dd in enumerate(directions):
        if dd.get("xdbl") is not None:
            outs.append(_scan_cl_dir_rows(a.dir[i], dd, u0, z, keep))
            continue
        u, delta = dd["u"], dd.get("delta")
        dt_low, dt_w = dd.get("dt_low"), dd.get("dt_weight")
        _dev_check(u, delta, dd["A"], dd["B"], dd["C"], dt_low, dt_w)
        _rows_ok(u, "u")
        if delta is not None:
            _rows_ok(delta, "delta")
        elif dt_low is None:
            raise RuntimeError("either delta or (dt_low, dt_weight) is required")
        if u.dtype != u0.dtype or (delta is not None and delta.dtype != u0.dtype) or (z is not None and z.dtype != u0.dtype):
            raise RuntimeError("u, delta, z must share one dtype")
        Bm, Cm = dd["B"], dd["C"]
        if Bm.dtype != torch.float32 or Bm.stride(2) != 1 or Bm.stride() != Cm.stride() or Bm.shape != (16, b, l):
            raise RuntimeError("B/C must be fp32 (16, batch, seqlen), time-contiguous, with equal strides")
        A, D, bias = _f32c(dd["A"]), _f32c(dd.get("D")), _f32c(dd.get("delta_bias"))
        dt_w = _f32c(dt_w)
        if dt_low is not None:
            if dt_low.dtype != torch.float32 or dt_low.stride(2) != 1 or dt_low.stride()[:2] != Bm.stride()[:2]:
                raise RuntimeError("dt_low must be fp32 (dt_rank, batch, seqlen) with the strides of B/C")
        out = dd.get("out")
        if out is None:
            out = torch.empty((b, l, d), dtype=u.dtype, device=u.device)
        _rows_ok(out, "out")
        keep += [A, D, bias, dt_w]
        x = a.dir[i]
        x.u, x.delta, x.A, x.B, x.C, x.D, x.delta_bias, x.out = (_ptr(u), _ptr(delta), _ptr(A), _ptr(Bm), _ptr(Cm),
                                                                   _ptr(D), _ptr(bias), _ptr(out))
        x.dt_low, x.dt_weight, x.dt_rank = _ptr(dt_low), _ptr(dt_w), (0 if dt_low is None else dt_low.shape[0])
        x.u_bs, x.u_ts = u.stride(0), u.stride(1)
        if delta is not None:
            x.delta_bs, x.delta_ts = delta.stride(0), delta.stride(1)
        x.out_bs, x.out_ts = out.stride(0), out.stride(1)
        x.bc_ns, x.bc_bs = Bm.stride(0), Bm.stride(1)
        x.reverse_time = int(bool(dd.get("reverse", False)))
        outs.append(out)
    a.stream = _stream()
    a.lanes_per_channel = int(split)                     # tuning of the state-split kernel: 4, 8, 16 lanes per channel; 0 = automatic
    if "xdbl" in directions[0]:
        if time_chunks is None:
            time_chunks = (N.lib().cm_scan_cl_fwd_auto_chunks(b, l, d, len(directions)) if SCAN_CHUNKS == "auto"
                           else int(SCAN_CHUNKS))
        a.time_chunks = int(time_chunks)
        nbytes = N.lib().cm_scan_cl_fwd_workspace_bytes(ct.byref(a))
        if nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=u0.device)
            keep.append(ws)
            a.workspace, a.workspace_bytes = _ptr(ws), nbytes
    elif time_chunks not in (None, 0, 1):
        raise RuntimeError("time_chunks needs the xdbl mode")
    _launch("cm_scan_cl_fwd", N.lib().cm_scan_cl_fwd, a, units=b * l * len(directions))
    return outs


def _elem_args(rows, dim, io_dtype, act, p, alpha):
    a = N.FfnElemArgs()
    a.rows, a.dim, a.io_dtype, a.act, a.p, a.alpha, a.stream = rows, dim, _DT[io_dtype], int(act), float(p), float(alpha), _stream()
    a.seed_epoch = _seed_epoch_ptr()
    return a


def draw_seed() -> int:
    """64-bit seed for a dropout mask from torch's (seedable) CPU generator: no device synchronisation."""
    return int(torch.randint(0, 2 ** 62, (1,)).item())


# Dropout under hipGraph replay.  A seed drawn on the host is a constant of the captured launch; with SEED_EPOCH set to a
# one-element int64 DEVICE tensor every dropout kernel adds  *SEED_EPOCH * CM_SEED_EPOCH_MUL  to its seed when it runs
# (cm_ffn_elem_args.seed_epoch), so a graph whose first node increments the word draws new decisions per replay while the
# forward and backward kernels of one replay still agree.  None (default): the host seed alone.  graphs.GraphedTrainStep sets it.
SEED_EPOCH = None
SEED_EPOCH_MUL = 0x9E3779B97F4A7C15


def _seed_epoch_ptr():
    t = SEED_EPOCH
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.int64 and t.numel() == 1):
        raise RuntimeError("ops.SEED_EPOCH must be a one-element int64 device tensor")
    return _ptr(t)


def effective_seed(seed: int, epoch: int) -> int:
    """The seed a kernel uses when the device word holds ``epoch`` (for tests and for reproducing a replay eagerly)."""
    return (int(seed) + int(epoch) * SEED_EPOCH_MUL) & (2 ** 64 - 1)


def bias_act_dropout_fwd(a, bias, act=0, p=0.0, res=None, alpha=1.0, seed=None, store_mask=True):
    """y = dropout(act(a + bias)) in a's dtype, or with ``res`` (fp32, a's shape) y = res + alpha * dropout(a + bias) in fp32
    (cm_bias_act_dropout_fwd).  a (rows, dim) contiguous bf16 / fp32; act 0 none, 1 GELU.  -> (y, mask or None).
    The keep decisions are a function of (seed, element index) (csrc/cm_dropout.h): with ``seed`` given (draw_seed()) and
    ``store_mask=False`` no mask is written and bias_act_dropout_bwd re-derives it from the same seed."""
    _dev_check(a, bias, res)
    if a.dim() != 2 or not a.is_contiguous() or a.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("bias_act_dropout_fwd: a must be a contiguous (rows, dim) bf16 / fp32 tensor")
    rows, dim = a.shape
    args = _elem_args(rows, dim, a.dtype, act, p, alpha)
    bs = _f32c(bias)
    mask = torch.empty((rows, dim), dtype=torch.uint8, device=a.device) if (p > 0.0 and store_mask) else None
    if res is not None and (res.dtype != torch.float32 or not res.is_contiguous() or res.shape != a.shape):
        raise RuntimeError("bias_act_dropout_fwd: res must be a contiguous fp32 tensor of a's shape")
    y = torch.empty((rows, dim), dtype=torch.float32 if res is not None else a.dtype, device=a.device)
    args.a, args.bias, args.res, args.y, args.mask = _ptr(a), _ptr(bs), _ptr(res), _ptr(y), _ptr(mask)
    if p > 0.0:
        args.seed = draw_seed() if seed is None else int(seed)
    _launch("cm_bias_act_dropout_fwd", N.lib().cm_bias_act_dropout_fwd, args, units=rows)
    return y, mask


def bias_act_dropout_bwd(dy, mask, p, a=None, bias=None, act=0, alpha=1.0, out_dtype=None, want_dbias=True, seed=None, want_act=False):
    """da = alpha * dy * mask / (1 - p) * act'(a + bias) in ``out_dtype`` (default a's / dy's dtype); dbias = column sums of da
    (fp32, deterministic) (cm_bias_act_dropout_bwd).  dy (rows, dim) fp32 or the I/O dtype.  -> (da, dbias or None).
    ``mask`` None with ``seed`` given: the decisions are re-derived from the seed.  ``want_act`` (act 1, bf16): also returns
    dropout(GELU(a + bias)) recomputed, as cm_ffn_fused's training forward fed it to its second GEMM -> (da, dbias, act)."""
    _dev_check(dy, mask, a, bias)
    if dy.dim() != 2 or not dy.is_contiguous():
        raise RuntimeError("bias_act_dropout_bwd: dy must be a contiguous (rows, dim) tensor")
    rows, dim = dy.shape
    io = out_dtype or (a.dtype if a is not None else dy.dtype)
    if io not in (torch.float32, torch.bfloat16) or dy.dtype not in (torch.float32, io):
        raise RuntimeError("bias_act_dropout_bwd: dtypes must be fp32 / bf16, dy fp32 or the I/O dtype")
    if a is not None and (a.dtype != io or not a.is_contiguous() or a.shape != dy.shape):
        raise RuntimeError("bias_act_dropout_bwd: a must be contiguous, of dy's shape, in the I/O dtype")
    args = _elem_args(rows, dim, io, act, p if (mask is not None or seed is not None) else 0.0, alpha)
    if seed is not None:
        args.seed = int(seed)
    bs = _f32c(bias)
    da = torch.empty((rows, dim), dtype=io, device=dy.device)
    act_out = None
    if want_act:
        act_out = torch.empty((rows, dim), dtype=io, device=dy.device)
        args.act_out = _ptr(act_out)
    args.a, args.bias, args.mask, args.dy, args.da = _ptr(a), _ptr(bs), _ptr(mask), _ptr(dy), _ptr(da)
    args.dy_f32 = int(dy.dtype == torch.float32 and io != torch.float32)
    dbias = None
    if want_dbias:
        nws = int(N.lib().cm_bias_act_dropout_bwd_workspace_floats(rows, dim))
        ws = torch.empty((nws + dim,), dtype=torch.float32, device=dy.device)
        dbias = ws[nws:]
        args.dbias, args.dbias_part, args.overwrite = _ptr(dbias), _ptr(ws), 1
    _launch("cm_bias_act_dropout_bwd", N.lib().cm_bias_act_dropout_bwd, args, units=rows)
    return (da, dbias, act_out) if want_act else (da, dbias)


def bias_glu_fwd(a, bias):
    """(rows, 2 dim) -> (rows, dim): (a[:, :dim] + b[:dim]) * sigmoid(a[:, dim:] + b[dim:])  (cm_bias_act_dropout_fwd, act 2)."""
    _dev_check(a, bias)
    rows, d2 = a.shape
    if not a.is_contiguous() or a.dtype not in (torch.float32, torch.bfloat16) or d2 % 16:
        raise RuntimeError("bias_glu_fwd: a must be a contiguous (rows, 2 dim) bf16 / fp32 tensor, dim a multiple of 8")
    args = _elem_args(rows, d2 // 2, a.dtype, 2, 0.0, 1.0)
    bs = _f32c(bias)
    y = torch.empty((rows, d2 // 2), dtype=a.dtype, device=a.device)
    args.a, args.bias, args.y = _ptr(a), _ptr(bs), _ptr(y)
    _launch("cm_bias_act_dropout_fwd", N.lib().cm_bias_act_dropout_fwd, args, units=rows)
    return y


def bias_glu_bwd(dy, a, bias):
    """-> (da (rows, 2 dim) in a's dtype, dbias (2 dim) fp32) of bias_glu_fwd (cm_bias_act_dropout_bwd, act 2; deterministic)."""
    _dev_check(dy, a, bias)
    rows, d = dy.shape
    if not dy.is_contiguous() or dy.dtype != a.dtype or a.shape != (rows, 2 * d) or not a.is_contiguous():
        raise RuntimeError("bias_glu_bwd: dy (rows, dim) and a (rows, 2 dim) must be contiguous and share a dtype")
    args = _elem_args(rows, d, a.dtype, 2, 0.0, 1.0)
    bs = _f32c(bias)
    da = torch.empty_like(a)
    nws = int(N.lib().cm_bias_act_dropout_bwd_workspace_floats(rows, d))
    ws = torch.empty((nws + 2 * d,), dtype=torch.float32, device=dy.device)
    dbias = ws[nws:]
    args.overwrite = 1
    args.a, args.bias, args.dy, args.da, args.dbias, args.dbias_part = _ptr(a), _ptr(bs), _ptr(dy), _ptr(da), _ptr(dbias), _ptr(ws)
    _launch("cm_bias_act_dropout_bwd", N.lib().cm_bias_act_dropout_bwd, args, units=rows)
    return da, dbias


def ctc_supported(log_probs, targets) -> bool:
    return log_probs.is_cuda and log_probs.dim() == 3 and targets.dim() == 2 and targets.shape[1] <= 511 and log_probs.shape[2] > 1


def ctc_loss_grad(log_probs, targets, input_lengths, target_lengths, blank=0):
    """log_probs (batch, T, V) -> (nll (batch) fp32, grad (batch, T, V) fp32) (cm_ctc_loss): per-utterance negative log-likelihood
    (0 where no alignment exists) and its gradient w.r.t. log_probs; lengths are integer tensors."""
    _dev_check(log_probs, targets, input_lengths, target_lengths)
    lp = log_probs.detach().float().contiguous()
    b, t, v = lp.shape
    tg = targets.detach().to(torch.int64).contiguous()
    il, tl = input_lengths.detach().to(torch.int32).contiguous(), target_lengths.detach().to(torch.int32).contiguous()
    s = tg.shape[1]
    nll = torch.empty((b,), dtype=torch.float32, device=lp.device)
    grad = torch.empty_like(lp)
    nws = int(N.lib().cm_ctc_workspace_floats(b, t, s))
    ws = torch.empty((nws,), dtype=torch.float32, device=lp.device)
    a = N.CtcArgs()
    a.batch, a.T, a.V, a.S, a.blank = b, t, v, s, int(blank)
    a.log_probs, a.targets, a.input_lengths, a.target_lengths, a.nll, a.grad = _ptr(lp), _ptr(tg), _ptr(il), _ptr(tl), _ptr(nll), _ptr(grad)
    a.workspace, a.workspace_floats, a.stream = _ptr(ws), nws, _stream()
    _launch("cm_ctc_loss", N.lib().cm_ctc_loss, a, units=b * t)
    return nll, grad


class CtcLossFn(torch.autograd.Function):
    """sum over the batch of the per-utterance CTC negative log-likelihoods (zero_infinity), gradient from the same call."""

    @staticmethod
    def forward(ctx, log_probs, targets, input_lengths, target_lengths, blank):
        nll, grad = ctc_loss_grad(log_probs, targets, input_lengths, target_lengths, blank)
        ctx.save_for_backward(grad)
        ctx.in_dtype = log_probs.dtype
        return nll.sum()

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        return (grad * gout).to(ctx.in_dtype), None, None, None, None


def sum_leading(t: torch.Tensor, out_dtype=torch.float32) -> torch.Tensor:
    """t (batch, ...) -> sum over the leading axis with fp32 accumulation in a fixed order (cm_sum_leading): folds per-utterance
    weight-gradient products; output in ``out_dtype`` (fp32 = a parameter's gradient dtype, no cast afterwards)."""
    _dev_check(t)
    n = t[0].numel()
    vec = 8 if t.dtype == torch.bfloat16 else 4
    if t.dtype not in (torch.float32, torch.bfloat16) or n % vec or not t.is_contiguous():
        return t.sum(0).to(out_dtype)
    out = torch.empty(t.shape[1:], dtype=out_dtype, device=t.device)
    N.check(N.lib().cm_sum_leading(_ptr(t), _ptr(out), t.shape[0], n, _DT[t.dtype], _DT[out_dtype], _stream()), "cm_sum_leading")
    return out


def conv_cl_bwd(x, weight_f, bias_f, du_f, weight_b=None, bias_b=None, du_b=None, dz_f=None, dz_b=None, dx=None, dz=None):
    """Backward of the channels-last causal depthwise conv + SiLU, one or both BiMamba directions in one pass
    (cm_conv_cl_bwd).  x, du_*, dz_* (batch, seqlen, dim) views; weights (dim, 4).  dx = dx_fwd + dx_bwd; dz = dz_f + dz_b
    when given.  Returns (dx, dz or None, dweight_f, dbias_f, dweight_b, dbias_b) with fp32 parameter gradients."""
    _dev_check(x, weight_f, bias_f, du_f, weight_b, bias_b, du_b, dz_f, dz_b, dx, dz)
    for t, nm in ((x, "x"), (du_f, "du_f"), (du_b, "du_b"), (dz_f, "dz_f"), (dz_b, "dz_b"), (dx, "dx"), (dz, "dz")):
        if t is not None:
            _rows_ok(t, nm)
            if t.shape != x.shape or t.dtype != x.dtype:
                raise RuntimeError(f"conv_cl_bwd: {nm} must match x in shape and dtype")
    b, l, d = x.shape
    two = du_b is not None
    wf, bf, wb, bb = _f32c(weight_f).reshape(d, -1), _f32c(bias_f), (_f32c(weight_b).reshape(d, -1) if two else None), (_f32c(bias_b) if two else None)
    if dx is None:
        dx = torch.empty((b, l, d), dtype=x.dtype, device=x.device)
    if dz is None and dz_f is not None:
        dz = torch.empty((b, l, d), dtype=x.dtype, device=x.device)
    dev = x.device
    kw = wf.shape[1]
    a = N.ConvClBwdArgs()
    a.batch, a.seqlen, a.dim, a.width, a.io_dtype = b, l, d, wf.shape[1], _DT[x.dtype]
    a.x, a.weight_f, a.bias_f, a.weight_b, a.bias_b = _ptr(x), _ptr(wf), _ptr(bf), _ptr(wb), _ptr(bb)
    a.du_f, a.du_b, a.dz_f, a.dz_b, a.dx, a.dz = _ptr(du_f), _ptr(du_b), _ptr(dz_f), _ptr(dz_b), _ptr(dx), _ptr(dz)
    # the kernel writes contiguous (dim, 4) / (dim) tensors: hand it contiguous staging slices of one buffer (overwrite: no memset)
    flat = torch.empty((2 * d * (kw + 1),), dtype=torch.float32, device=dev)
    a.overwrite = 1
    dwf, dwb_ = flat[:d * kw].view(d, kw), flat[d * kw:2 * d * kw].view(d, kw)
    dbf_, dbb_ = flat[2 * d * kw:2 * d * kw + d], flat[2 * d * kw + d:]
    dbf = dbf_ if bf is not None else None
    dwb = dwb_ if two else None
    dbb = dbb_ if (two and bb is not None) else None
    a.dweight_f, a.dbias_f, a.dweight_b, a.dbias_b = _ptr(dwf), _ptr(dbf), _ptr(dwb), _ptr(dbb)
    a.x_bs, a.x_ts, a.duf_bs, a.duf_ts = x.stride(0), x.stride(1), du_f.stride(0), du_f.stride(1)
    if two:
        a.dub_bs, a.dub_ts = du_b.stride(0), du_b.stride(1)
    if dz_f is not None:
        a.dzf_bs, a.dzf_ts = dz_f.stride(0), dz_f.stride(1)
    if dz_b is not None:
        a.dzb_bs, a.dzb_ts = dz_b.stride(0), dz_b.stride(1)
    a.dx_bs, a.dx_ts = dx.stride(0), dx.stride(1)
    if dz is not None:
        a.dz_bs, a.dz_ts = dz.stride(0), dz.stride(1)
    nws = int(N.lib().cm_conv_cl_bwd_workspace_floats(b, l, d))
    ws = torch.empty((nws,), dtype=torch.float32, device=dev)
    a.workspace, a.workspace_floats, a.stream = _ptr(ws), nws, _stream()
    _launch("cm_conv_cl_bwd", N.lib().cm_conv_cl_bwd, a, units=b * l)
    return dx, dz, dwf, dbf, dwb, dbb


def scan_ckpt_shape(batch: int, seqlen: int, dim: int):
    """Shape of the checkpoint tensor the training forward writes (cm_scan_cl_dir.ckpt)."""
    return (batch, 2 * ((seqlen + 15) // 16), dim, 16)


def scan_cl_bwd(directions, z, time_chunks=0, da_log=False):
    """Channels-last selective scan backward, 1 or 2 directions in one launch (cm_scan_cl_bwd): the gradient of
    scan_cl_fwd's xdbl mode (z and softplus on) including the dt_proj part.

    ``directions``: list of dicts with u (batch, seqlen, dim), xdbl (batch, seqlen, P + 32), A (dim, 16), dt_weight (dim, P)
    zero padded (pad_dt_weight), D, delta_bias (dim), ckpt and ypre as the forward wrote them, dout (batch, seqlen, dim),
    reverse; optional pre-allocated du, dz, dxdbl views.  Returns a list of dicts du, dz (this direction's share), dxdbl
    (I/O dtype, [d dt | dB | dC]), and fp32 dA (dim, 16), ddt_weight (dim, P), dD, ddelta_bias (dim).
    ``time_chunks``: 0 lets the library cut launches that would leave CUs idle along time, 1 never, n asks for n chunks.
    ``da_log``: dA is returned as the gradient w.r.t. A_log (A = -exp(A_log)): dA * A, formed in the reduce pass."""
    if not 1 <= len(directions) <= 2:
        raise RuntimeError("1 or 2 directions")
    u0 = directions[0]["u"]
    _dev_check(u0, z)
    _rows_ok(u0, "u"), _rows_ok(z, "z")
    b, l, d = u0.shape
    if z.shape != (b, l, d) or z.dtype != u0.dtype:
        raise RuntimeError("scan_cl_bwd: z must be (batch, seqlen, dim) in u's dtype")
    a = N.ScanClBwdArgs()
    a.batch, a.seqlen, a.dim, a.dstate, a.io_dtype, a.ndir = b, l, d, 16, _DT[u0.dtype], len(directions)
    a.z, a.z_bs, a.z_ts = _ptr(z), z.stride(0), z.stride(1)
    a.time_chunks = int(time_chunks)
    a.da_log = int(bool(da_log))
    keep, outs = [], []
    for i, dd in enumerate(directions):
        u, xdbl, dout, ypre, ck = dd["u"], dd["xdbl"], dd["dout"], dd["ypre"], dd["ckpt"]
        _dev_check(u, xdbl, dout, ypre, ck, dd["A"], dd["dt_weight"])
        for t, nm in ((u, "u"), (xdbl, "xdbl"), (dout, "dout"), (ypre, "ypre")):
            _rows_ok(t, nm)
        pad = xdbl.shape[-1] - 32
        if pad not in (16, 32) or (pad == 32 and u0.dtype != torch.bfloat16):
            raise RuntimeError("scan_cl_bwd: xdbl rows are 48 wide, or 64 wide (dt_rank > 16) in bf16")
        if any(t.shape != (b, l, d) or t.dtype != u0.dtype for t in (u, dout, ypre)) or xdbl.shape != (b, l, pad + 32) or xdbl.dtype != u0.dtype:
            raise RuntimeError("scan_cl_bwd: u / dout / ypre (batch, seqlen, dim) and xdbl (batch, seqlen, 48 | 64) must share shape prefix and dtype")
        if tuple(ck.shape) != scan_ckpt_shape(b, l, d) or ck.dtype != torch.float32 or not ck.is_contiguous():
            raise RuntimeError("scan_cl_bwd: ckpt must be the contiguous fp32 tensor the forward wrote (ops.scan_ckpt_shape)")
        dt_w = dd["dt_weight"]
        if dt_w.shape != (d, pad):
            raise RuntimeError(f"scan_cl_bwd: dt_weight must be (dim, {pad}), zero padded (ops.pad_dt_weight)")
        A, D, bias, dt_w = _f32c(dd["A"]), _f32c(dd.get("D")), _f32c(dd.get("delta_bias")), _f32c(dt_w)
        du = dd.get("du") if dd.get("du") is not None else torch.empty((b, l, d), dtype=u.dtype, device=u.device)
        dz = dd.get("dz") if dd.get("dz") is not None else torch.empty((b, l, d), dtype=u.dtype, device=u.device)
        dx = dd.get("dxdbl") if dd.get("dxdbl") is not None else torch.empty((b, l, pad + 32), dtype=u.dtype, device=u.device)
        for t, nm in ((du, "du"), (dz, "dz"), (dx, "dxdbl")):
            _rows_ok(t, nm)
        if i == 0:                                            # one buffer for every parameter gradient of the launch (overwrite: no memset)
            npar = d * (16 + pad + 2)
            zeros = torch.empty((len(directions) * npar,), dtype=torch.float32, device=u.device)
            a.overwrite = 1
        zp = zeros[i * npar:(i + 1) * npar]
        dA, dW = zp[:16 * d].view(d, 16), zp[16 * d:(16 + pad) * d].view(d, pad)
        dD = zp[(16 + pad) * d:(17 + pad) * d] if D is not None else None
        db = zp[(17 + pad) * d:] if bias is not None else None
        keep += [A, D, bias, dt_w]
        x = a.dir[i]
        x.u, x.xdbl, x.A, x.dt_weight, x.D, x.delta_bias, x.ckpt, x.ypre, x.dout = (_ptr(u), _ptr(xdbl), _ptr(A), _ptr(dt_w), _ptr(D),
                                                                                    _ptr(bias), _ptr(ck), _ptr(ypre), _ptr(dout))
        x.du, x.dz, x.dxdbl, x.dA, x.ddt_weight, x.dD, x.ddelta_bias = _ptr(du), _ptr(dz), _ptr(dx), _ptr(dA), _ptr(dW), _ptr(dD), _ptr(db)
        x.u_bs, x.u_ts, x.xdbl_bs, x.xdbl_ts = u.stride(0), u.stride(1), xdbl.stride(0), xdbl.stride(1)
        x.ypre_bs, x.ypre_ts, x.dout_bs, x.dout_ts = ypre.stride(0), ypre.stride(1), dout.stride(0), dout.stride(1)
        x.du_bs, x.du_ts, x.dz_bs, x.dz_ts, x.dxdbl_bs, x.dxdbl_ts = du.stride(0), du.stride(1), dz.stride(0), dz.stride(1), dx.stride(0), dx.stride(1)
        x.reverse_time, x.dt_rank = int(bool(dd.get("reverse", False))), pad
        outs.append(dict(du=du, dz=dz, dxdbl=dx, dA=dA, ddt_weight=dW, dD=dD, ddelta_bias=db))
    a.stream = _stream()
    nbytes = int(N.lib().cm_scan_cl_bwd_workspace_bytes(ct.byref(a)))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=u0.device)
    a.workspace, a.workspace_bytes = _ptr(ws), nbytes
    _launch("cm_scan_cl_bwd", N.lib().cm_scan_cl_bwd, a, units=b * l * len(directions))
    return outs


def conv_xproj(x, weight_f, bias_f, weight_b, bias_b, wx_f, wx_b, out_f, out_b, xdbl=None, variant: int = 0):
    """Both directions' depthwise conv + SiLU and x_proj GEMMs in one kernel (cm_conv_xproj, bf16).
    x (batch, seqlen, dim) view; conv weights (dim, 4) fp32; wx_f / wx_b PackedWeight of the (P + 32, dim) re-rowed x_proj
    weights [dt P | B | C], P = 16, or 32 for 16 < dt_rank <= 32; out_f / out_b (batch, seqlen, dim) views.
    Returns xdbl (batch, seqlen, 2 * (P + 32)) bf16."""
    _dev_check(x, weight_f, bias_f, weight_b, bias_b, out_f, out_b)
    _rows_ok(x, "x"), _rows_ok(out_f, "out_f"), _rows_ok(out_b, "out_b")
    b, l, d = x.shape
    if x.dtype != torch.bfloat16 or out_f.dtype != torch.bfloat16 or out_b.dtype != torch.bfloat16:
        raise RuntimeError("conv_xproj: bf16 only")
    rw = wx_f.shape[0]
    if rw not in (48, 64) or wx_f.shape != (rw, d) or wx_b.shape != (rw, d):
        raise RuntimeError("conv_xproj: wx must be PackedWeight of shape (48 | 64, dim)")
    wf, bf, wb, bb = _f32c(weight_f), _f32c(bias_f), _f32c(weight_b), _f32c(bias_b)
    if xdbl is None:
        xdbl = torch.empty((b, l, 2 * rw), dtype=torch.bfloat16, device=x.device)
    elif xdbl.shape != (b, l, 2 * rw) or xdbl.dtype != torch.bfloat16:
        raise RuntimeError(f"conv_xproj: xdbl must be bf16 (batch, seqlen, {2 * rw})")
    a = N.ConvXprojArgs()
    a.batch, a.seqlen, a.dim, a.width = b, l, d, wf.shape[1]
    a.x, a.weight_f, a.bias_f, a.weight_b, a.bias_b = _ptr(x), _ptr(wf), _ptr(bf), _ptr(wb), _ptr(bb)
    a.wx_f, a.wx_b, a.y_fwd, a.y_bwd, a.xdbl = _ptr(wx_f.data), _ptr(wx_b.data), _ptr(out_f), _ptr(out_b), _ptr(xdbl)
    a.x_bs, a.x_ts, a.yf_bs, a.yf_ts = x.stride(0), x.stride(1), out_f.stride(0), out_f.stride(1)
    a.yb_bs, a.yb_ts, a.xdbl_bs, a.xdbl_ts = out_b.stride(0), out_b.stride(1), xdbl.stride(0), xdbl.stride(1)
    a.stream, a.dt_pad, a.variant = _stream(), rw - 32, int(variant)     # variant 1: always the 16-step tiles
    _launch("cm_conv_xproj", N.lib().cm_conv_xproj, a, units=b * l)
    return xdbl


def conv_cl_fwd(x, weight_f, bias_f, weight_b=None, bias_b=None, silu=True, out_f=None, out_b=None):
    """Channels-last depthwise conv (+SiLU) for one or both BiMamba directions in one pass (cm_conv_cl_fwd).
    x (batch, seqlen, dim) view; weights (dim, 4); returns (y_fwd, y_bwd or None)."""
    _dev_check(x, weight_f, bias_f, weight_b, bias_b)
    _rows_ok(x, "x")
    b, l, d = x.shape
    wf, bf, wb, bb = _f32c(weight_f), _f32c(bias_f), _f32c(weight_b), _f32c(bias_b)
    if out_f is None:
        out_f = torch.empty((b, l, d), dtype=x.dtype, device=x.device)
    if wb is not None and out_b is None:
        out_b = torch.empty((b, l, d), dtype=x.dtype, device=x.device)
    a = N.ConvClArgs()
    a.batch, a.seqlen, a.dim, a.width, a.io_dtype, a.silu = b, l, d, wf.shape[1], _DT[x.dtype], int(bool(silu))
    a.x, a.weight_f, a.bias_f, a.weight_b, a.bias_b = _ptr(x), _ptr(wf), _ptr(bf), _ptr(wb), _ptr(bb)
    a.y_fwd, a.y_bwd = _ptr(out_f), _ptr(out_b if wb is not None else None)
    a.x_bs, a.x_ts, a.yf_bs, a.yf_ts = x.stride(0), x.stride(1), out_f.stride(0), out_f.stride(1)
    if wb is not None:
        a.yb_bs, a.yb_ts = out_b.stride(0), out_b.stride(1)
    a.stream = _stream()
    _launch("cm_conv_cl_fwd", N.lib().cm_conv_cl_fwd, a, units=b * l)
    return out_f, (out_b if wb is not None else None)


def add_layernorm(x, y=None, alpha=1.0, norm1=None, norm2=None, x_out=None, out_dtype=None, want_out=True,
                  out_act=0):
    """r = x + alpha*y; r1 = LN1(r) if norm1 else r; x_out <- r1 (fp32, may alias x); out = LN2(r1) if norm2 else r1.
    x (rows.., dim) fp32 contiguous; norm = (weight, bias, eps).  Returns (x_out or None, out or None)."""
    _dev_check(x, y)
    ref = x if x is not None else y
    if x is not None and (x.dtype != torch.float32 or not x.is_contiguous()):
        raise RuntimeError("add_layernorm: x must be contiguous fp32")
    d = ref.shape[-1]
    rows = ref.numel() // d
    a = N.AddLnArgs()
    a.rows, a.dim, a.out_act = rows, d, int(out_act)
    keep = []
    if y is not None:
        if not y.is_contiguous() or y.shape != ref.shape:
            raise RuntimeError("add_layernorm: y must be contiguous with x's shape")
        a.y, a.y_dtype = _ptr(y), _DT[y.dtype]
    a.alpha = float(alpha)
    if norm1 is not None:
        g, bt = _f32c(norm1[0]), _f32c(norm1[1])
        keep += [g, bt]
        a.g1, a.b1, a.eps1 = _ptr(g), _ptr(bt), float(norm1[2])
    if norm2 is not None:
        g, bt = _f32c(norm2[0]), _f32c(norm2[1])
        keep += [g, bt]
        a.g2, a.b2, a.eps2 = _ptr(g), _ptr(bt), float(norm2[2])
    out = None
    if want_out:
        od = out_dtype or torch.bfloat16
        out = torch.empty(ref.shape, dtype=od, device=ref.device)
        a.out, a.out_dtype = _ptr(out), _DT[od]
    a.x = _ptr(x)
    a.x_out = _ptr(x_out)
    a.stream = _stream()
    _launch("cm_add_layernorm", N.lib().cm_add_layernorm, a, units=rows)
    return x_out, out


def causal_conv1d_update(x, conv_state, weight, bias=None, silu=True):
    """Decode-time conv step (cm_causal_conv1d_update; reference bimamba.py:331-343): conv_state (batch, dim, width) fp32 is
    shifted and gets x (batch, dim) appended IN PLACE; returns out (batch, dim) in x's dtype."""
    _dev_check(x, conv_state, weight, bias)
    if conv_state.dtype != torch.float32 or not conv_state.is_contiguous():
        raise RuntimeError("causal_conv1d_update: conv_state must be a contiguous fp32 (batch, dim, width) tensor")
    x = x.contiguous()
    w, bs = _f32c(weight).reshape(conv_state.shape[1], -1), _f32c(bias)
    out = torch.empty_like(x)
    a = N.ConvUpdateArgs()
    a.batch, a.dim, a.width, a.io_dtype, a.silu = x.shape[0], x.shape[1], conv_state.shape[2], _DT[x.dtype], int(bool(silu))
    a.x, a.conv_state, a.weight, a.bias, a.out, a.stream = _ptr(x), _ptr(conv_state), _ptr(w), _ptr(bs), _ptr(out), _stream()
    _launch("cm_causal_conv1d_update", N.lib().cm_causal_conv1d_update, a, units=x.shape[0])
    return out


def selective_state_update(state, x, dt, A, B, C, D=None, z=None, dt_bias=None, dt_softplus=False):
    """Decode-time SSM step (cm_selective_state_update; reference bimamba.py:350-362): state (batch, dim, dstate) fp32 is
    updated IN PLACE; x, dt, z (batch, dim), B, C (batch, dstate) share one dtype; returns y (batch, dim)."""
    _dev_check(state, x, dt, A, B, C, D, z, dt_bias)
    if state.dtype != torch.float32 or not state.is_contiguous():
        raise RuntimeError("selective_state_update: state must be a contiguous fp32 (batch, dim, dstate) tensor")
    x = x.contiguous()
    dt, B, C = dt.to(x.dtype).contiguous(), B.to(x.dtype).contiguous(), C.to(x.dtype).contiguous()
    z = None if z is None else z.to(x.dtype).contiguous()
    A, D, dt_bias = _f32c(A), _f32c(D), _f32c(dt_bias)
    out = torch.empty_like(x)
    a = N.StateUpdateArgs()
    a.batch, a.dim, a.dstate, a.io_dtype, a.dt_softplus = x.shape[0], x.shape[1], state.shape[2], _DT[x.dtype], int(bool(dt_softplus))
    a.state, a.x, a.dt, a.A, a.B, a.C, a.D, a.z, a.dt_bias, a.out = (_ptr(state), _ptr(x), _ptr(dt), _ptr(A), _ptr(B), _ptr(C), _ptr(D),
                                                                     _ptr(z), _ptr(dt_bias), _ptr(out))
    a.stream = _stream()
    _launch("cm_selective_state_update", N.lib().cm_selective_state_update, a, units=x.shape[0])
    return out


def _dwconv_args(x, weight, bias, pad_left):
    _dev_check(x, weight, bias)
    x = _time_contig(x)
    if x.dim() != 3:
        raise RuntimeError("dwconv1d: x must be (batch, dim, seqlen)")
    w = _f32c(weight).reshape(x.shape[1], -1)
    bs = _f32c(bias)
    a = N.Dwconv1dArgs()
    a.batch, a.dim, a.seqlen, a.ksize, a.pad_left, a.io_dtype = x.shape[0], x.shape[1], x.shape[2], w.shape[1], int(pad_left), _DT[x.dtype]
    a.x, a.weight, a.bias, a.x_bs, a.x_ds = _ptr(x), _ptr(w), _ptr(bs), x.stride(0), x.stride(1)
    a.stream = _stream()
    return a, x, w, bs


def dwconv1d_fwd(x, weight, bias=None, pad_left=None):
    """Depthwise Conv1d over time (cm_dwconv1d_fwd): x (batch, dim, seqlen), weight (dim, 1, k) or (dim, k), zero
    padding with ``pad_left`` zeros in front (default k // 2 = 'same'); returns y of the same shape."""
    k = weight.shape[-1]
    a, x, w, bs = _dwconv_args(x, weight, bias, k // 2 if pad_left is None else pad_left)
    y = torch.empty_like(x, memory_format=torch.contiguous_format)
    a.y, a.y_bs, a.y_ds = _ptr(y), y.stride(0), y.stride(1)
    _launch("cm_dwconv1d_fwd", N.lib().cm_dwconv1d_fwd, a, units=x.shape[0] * x.shape[2])
    return y


def dwconv1d_bwd(x, weight, dy, has_bias=True, pad_left=None):
    """-> (dx, dweight (dim, k) fp32, dbias (dim) fp32 or None) of dwconv1d_fwd (cm_dwconv1d_bwd)."""
    k = weight.shape[-1]
    a, x, w, _ = _dwconv_args(x, weight, None, k // 2 if pad_left is None else pad_left)
    _dev_check(dy)
    dy = _time_contig(dy)
    if dy.dtype != x.dtype or dy.shape != x.shape:
        raise RuntimeError("dwconv1d_bwd: dy must match x in shape and dtype")
    dx = torch.empty_like(x, memory_format=torch.contiguous_format)
    dw = torch.zeros_like(w)
    db = torch.zeros((x.shape[1],), dtype=torch.float32, device=x.device) if has_bias else None
    a.dy, a.dy_bs, a.dy_ds, a.dx, a.dx_bs, a.dx_ds = _ptr(dy), dy.stride(0), dy.stride(1), _ptr(dx), dx.stride(0), dx.stride(1)
    a.dweight, a.dbias = _ptr(dw), _ptr(db)
    _launch("cm_dwconv1d_bwd", N.lib().cm_dwconv1d_bwd, a, units=x.shape[0] * x.shape[2])
    return dx, dw, db


class DepthwiseConv1dFn(torch.autograd.Function):
    """autograd node over cm_dwconv1d_fwd / _bwd (drop-in for the depthwise nn.Conv1d of the ConvolutionModule)."""

    @staticmethod
    def forward(ctx, x, weight, bias, pad_left):
        ctx.save_for_backward(x, weight)
        ctx.has_bias, ctx.pad_left = bias is not None, pad_left
        return dwconv1d_fwd(x, weight, bias, pad_left)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx, dw, db = dwconv1d_bwd(x, weight, dy.to(x.dtype), ctx.has_bias, ctx.pad_left)
        return dx, dw.reshape(weight.shape).to(weight.dtype), (db.to(weight.dtype) if db is not None else None), None


def _dwconv_cl_args(x, weight, bias, pad_left):
    _dev_check(x, weight, bias)
    _rows_ok(x, "x")
    w = _f32c(weight).reshape(x.shape[2], -1)
    bs = _f32c(bias)
    a = N.DwconvClArgs()
    a.batch, a.seqlen, a.dim, a.ksize, a.pad_left, a.io_dtype = x.shape[0], x.shape[1], x.shape[2], w.shape[1], int(pad_left), _DT[x.dtype]
    a.x, a.weight, a.bias, a.x_bs, a.x_ts = _ptr(x), _ptr(w), _ptr(bs), x.stride(0), x.stride(1)
    a.stream = _stream()
    return a, w, bs


def dwconv_cl_fwd(x, weight, bias=None, pad_left=None):
    """Depthwise Conv1d over time on channels-last rows (cm_dwconv_cl_fwd): x (batch, seqlen, dim), channel axis
    contiguous; weight (dim, 1, k) or (dim, k); returns y (batch, seqlen, dim)."""
    k = weight.shape[-1]
    a, w, bs = _dwconv_cl_args(x, weight, bias, k // 2 if pad_left is None else pad_left)
    y = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    a.y, a.y_bs, a.y_ts = _ptr(y), y.stride(0), y.stride(1)
    _launch("cm_dwconv_cl_fwd", N.lib().cm_dwconv_cl_fwd, a, units=x.shape[0] * x.shape[1])
    return y


def dwconv_cl_bwd(x, weight, dy, has_bias=True, pad_left=None):
    """-> (dx, dweight (dim, k) fp32, dbias (dim) fp32 or None) of dwconv_cl_fwd (cm_dwconv_cl_bwd, deterministic)."""
    k = weight.shape[-1]
    a, w, _ = _dwconv_cl_args(x, weight, None, k // 2 if pad_left is None else pad_left)
    _dev_check(dy)
    if dy.stride(2) != 1:
        dy = dy.contiguous()
    if dy.dtype != x.dtype or dy.shape != x.shape:
        raise RuntimeError("dwconv_cl_bwd: dy must match x in shape and dtype")
    dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    dw = torch.empty_like(w)
    db = torch.empty((x.shape[2],), dtype=torch.float32, device=x.device) if has_bias else None
    a.overwrite = 1
    part = torch.empty((N.lib().cm_dwconv_cl_workspace_floats(x.shape[0], x.shape[1], x.shape[2]),), dtype=torch.float32, device=x.device)
    a.dy, a.dy_bs, a.dy_ts, a.dx, a.dx_bs, a.dx_ts = _ptr(dy), dy.stride(0), dy.stride(1), _ptr(dx), dx.stride(0), dx.stride(1)
    a.dweight, a.dbias, a.partial = _ptr(dw), _ptr(db), _ptr(part)
    _launch("cm_dwconv_cl_bwd", N.lib().cm_dwconv_cl_bwd, a, units=x.shape[0] * x.shape[1])
    return dx, dw, db


class DepthwiseConvClFn(torch.autograd.Function):
    """autograd node over cm_dwconv_cl_fwd / _bwd: the ConvolutionModule's depthwise conv on (batch, time, channel) rows."""

    @staticmethod
    def forward(ctx, x, weight, bias, pad_left):
        ctx.save_for_backward(x, weight)
        ctx.has_bias, ctx.pad_left = bias is not None, pad_left
        return dwconv_cl_fwd(x, weight, bias, pad_left)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx, dw, db = dwconv_cl_bwd(x, weight, dy.to(x.dtype), ctx.has_bias, ctx.pad_left)
        return dx, dw.reshape(weight.shape).to(weight.dtype), (db.to(weight.dtype) if db is not None else None), None


def _layernorm_args(x2, weight, eps, y_dtype):
    a = N.LayerNormArgs()
    a.rows, a.dim, a.x_dtype, a.y_dtype, a.eps = x2.shape[0], x2.shape[1], _DT[x2.dtype], _DT[y_dtype], float(eps)
    a.x, a.gamma, a.stream = _ptr(x2), _ptr(weight), _stream()
    return a


def _ln_epilogue(a, x2, leaky_slope, chan_mask, mask_rows):
    """Fill the optional activation + channel-mask epilogue of cm_layernorm_args."""
    if leaky_slope is not None:
        a.act, a.act_slope = 1, float(leaky_slope)
    if chan_mask is not None:
        _dev_check(chan_mask)
        if chan_mask.dtype != torch.float32 or not chan_mask.is_contiguous() or chan_mask.dim() != 2:
            raise RuntimeError("layernorm: chan_mask must be a contiguous fp32 (groups, channels) tensor")
        c = chan_mask.shape[1]
        if c % 4 or x2.shape[1] % c or x2.shape[0] != chan_mask.shape[0] * mask_rows:
            raise RuntimeError("layernorm: chan_mask (groups, channels) needs channels % 4 == 0, dim % channels == 0, rows == groups * mask_rows")
        a.chan_mask, a.mask_rows, a.mask_c = _ptr(chan_mask), int(mask_rows), c


def layernorm_fwd(x, weight, bias, eps, out_dtype=None, leaky_slope=None, chan_mask=None, mask_rows=1):
    """LayerNorm over the last axis (cm_layernorm_fwd) -> (y, mean, rstd); x fp32 or bf16, contiguous rows; y in
    `out_dtype` (default fp32: what torch's autocast gives, reference modules/Conmamba.py:597-620)."""
    _dev_check(x, weight, bias)
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError(f"layernorm_fwd: unsupported dtype {x.dtype}")
    x2 = x.reshape(-1, x.shape[-1])
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    out_dtype = out_dtype or torch.float32
    w, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
    y = torch.empty(x2.shape, dtype=out_dtype, device=x.device)
    stats = torch.empty((2, x2.shape[0]), dtype=torch.float32, device=x.device)
    a = _layernorm_args(x2, w, eps, out_dtype)
    a.beta, a.y, a.mean, a.rstd = _ptr(b), _ptr(y), _ptr(stats[0]), _ptr(stats[1])
    _ln_epilogue(a, x2, leaky_slope, chan_mask, mask_rows)
    _launch("cm_layernorm_fwd", N.lib().cm_layernorm_fwd, a, units=x2.shape[0])
    return y.view(x.shape), x2, stats


def layernorm_bwd(dy, x2, stats, weight, eps, need_dx=True, dres=None, bias=None, leaky_slope=None, chan_mask=None, mask_rows=1):
    """-> (dx (x2's dtype and shape) or None, dgamma, dbeta (dim) fp32) of layernorm_fwd (cm_layernorm_bwd, deterministic).
    ``dres`` (fp32, x2's shape; fp32 x2, dim <= 1024): added to dx in the same pass (the residual branch's gradient)."""
    _dev_check(dy, x2, stats, weight)
    dy2 = dy.reshape(x2.shape)
    if not dy2.is_contiguous():
        dy2 = dy2.contiguous()
    w = weight.detach().float().contiguous()
    dx = torch.empty_like(x2) if need_dx else None
    dgb = torch.empty((2, x2.shape[1]), dtype=torch.float32, device=x2.device)
    ws = torch.empty((N.lib().cm_layernorm_bwd_workspace_floats(x2.shape[0], x2.shape[1]),), dtype=torch.float32, device=x2.device)
    a = _layernorm_args(x2, w, eps, dy2.dtype)
    a.mean, a.rstd, a.dy, a.dx = _ptr(stats[0]), _ptr(stats[1]), _ptr(dy2), _ptr(dx)
    a.dgamma, a.dbeta, a.workspace = _ptr(dgb[0]), _ptr(dgb[1]), _ptr(ws)
    if leaky_slope is not None or chan_mask is not None:
        bb = bias.detach().float().contiguous()                   # the activation's derivative needs LN's output: beta
        a.beta = _ptr(bb)
        _ln_epilogue(a, x2, leaky_slope, chan_mask, mask_rows)
    fused_res = dres is not None and need_dx and x2.dtype == torch.float32 and x2.shape[1] <= 1024
    if fused_res:
        _dev_check(dres)
        if dres.dtype != torch.float32 or not dres.is_contiguous() or dres.shape != x2.shape:
            raise RuntimeError("layernorm_bwd: dres must be a contiguous fp32 tensor of x2's shape")
        a.dres = _ptr(dres)
    _launch("cm_layernorm_bwd", N.lib().cm_layernorm_bwd, a, units=x2.shape[0])
    if dres is not None and need_dx and not fused_res:
        dx = dx + dres
    return dx, dgb[0], dgb[1]


class LayerNormFn(torch.autograd.Function):
    """autograd node over cm_layernorm_fwd / _bwd.  Output fp32 under autocast or for fp32 input (torch's rule), else the
    input dtype."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, low_dtype=None):
        out_dtype = torch.float32 if (x.dtype == torch.float32 or torch.is_autocast_enabled("cuda")) else x.dtype
        if low_dtype is not None:                             # the consumer is a projection that would round to low_dtype anyway
            out_dtype = low_dtype
        y, x2, stats = layernorm_fwd(x, weight, bias, eps, out_dtype)
        ctx.save_for_backward(x2, stats, weight)
        ctx.eps, ctx.shape = eps, x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, stats, weight = ctx.saved_tensors
        if dy.dtype not in (torch.float32, torch.bfloat16):
            dy = dy.float()
        dx, dg, db = layernorm_bwd(dy, x2, stats, weight, ctx.eps, need_dx=ctx.needs_input_grad[0])
        return (dx.view(ctx.shape) if dx is not None else None), dg.to(weight.dtype), db.to(weight.dtype), None, None


class LnActDropFn(torch.autograd.Function):
    """y = LeakyReLU(LayerNorm(x)) * chan_mask -- the tail of a front-end Conv2d block (LayerNorm over (freq, channel) -> LeakyReLU ->
    Dropout2d with one factor per (sample, channel)) as ONE kernel forward and ONE backward (cm_layernorm_fwd / _bwd with the
    epilogue fields).  x (batch, time, freq, channel) contiguous; weight / bias (freq * channel); chan_mask (batch, channel) fp32 with
    the 1 / keep scale folded in, or None.  ``out_dtype``: what the consumer (the next Conv2d / Linear under autocast) rounds to anyway:
    the fp32 result is rounded once, here."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, slope, chan_mask, out_dtype):
        b, t, f, c = x.shape
        x2 = x.reshape(b * t, f * c)
        y, x2s, stats = layernorm_fwd(x2, weight, bias, eps, out_dtype, leaky_slope=slope, chan_mask=chan_mask, mask_rows=t)
        ctx.save_for_backward(x2s, stats, weight, bias, chan_mask if chan_mask is not None else weight.new_empty(0))
        ctx.cfg = (eps, slope, t, chan_mask is not None, x.shape)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, stats, weight, bias, mask = ctx.saved_tensors
        eps, slope, t, has_mask, shape = ctx.cfg
        if dy.dtype not in (torch.float32, torch.bfloat16):
            dy = dy.float()
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dx, dg, db = layernorm_bwd(dy.view(x2.shape[0], -1), x2, stats, weight, eps, need_dx=ctx.needs_input_grad[0], bias=bias, leaky_slope=slope,
                                   chan_mask=mask if has_mask else None, mask_rows=t)
        return (dx.view(shape) if dx is not None else None), dg.to(weight.dtype), db.to(weight.dtype), None, None, None, None


def ln_pw_glu(x, y, alpha, norm, w, bias, x_out=None):
    """Mixer -> convolution-module seam (cm_ln_pw_glu): x_out = x + alpha*y; g = GLU(LayerNorm(x_out) @ W^T + bias).
    x (rows, 256) fp32 contiguous; y (rows, 256) bf16 or None; norm = (weight, bias, eps); w: PackedWeight of the
    (512, 256) pointwise-conv weight; bias (512) fp32.  x_out defaults to x (in place).  Returns g (rows, 256) bf16."""
    _dev_check(x, y, bias)
    rows, d = x.shape
    if x.dtype != torch.float32 or not x.is_contiguous() or d != 256:
        raise RuntimeError("ln_pw_glu: x must be a contiguous fp32 (rows, 256) tensor")
    if y is not None and (y.dtype != torch.bfloat16 or not y.is_contiguous() or y.shape != x.shape):
        raise RuntimeError("ln_pw_glu: y must be a contiguous bf16 (rows, 256) tensor")
    if not isinstance(w, PackedWeight) or w.shape != (512, 256):
        raise RuntimeError("ln_pw_glu: w must be a PackedWeight of shape (512, 256)")
    g_, b_, bs = _f32c(norm[0]), _f32c(norm[1]), _f32c(bias)
    xo = x if x_out is None else x_out
    out = torch.empty((rows, d), dtype=torch.bfloat16, device=x.device)
    a = N.LnPwGluArgs()
    a.rows, a.dim, a.x, a.y, a.ln_g, a.ln_b, a.w, a.bias = rows, d, _ptr(x), _ptr(y), _ptr(g_), _ptr(b_), _ptr(w.data), _ptr(bs)
    a.x_out, a.out, a.alpha, a.eps, a.stream = _ptr(xo), _ptr(out), float(alpha), float(norm[2]), _stream()
    _launch("cm_ln_pw_glu", N.lib().cm_ln_pw_glu, a, units=rows)
    return out


def glu_dwconv_ln_gelu(inp, weight, bias, ln_weight, ln_bias, eps=1e-5, weight_t=None, glu_done=False, lin_w=None, lin_b=None,
                       variant: int = 0):
    """(batch, seqlen, 2*dim) -> (batch, seqlen, dim): GLU, depthwise conv (k=31, same padding), LayerNorm, GELU
    (cm_glu_dwconv_ln_gelu).  weight (dim, 1, k) or (dim, k); weight_t: optional precomputed fp32 (k, dim) copy of the
    taps (made here per call otherwise) so that the kernel's per-channel tap reads are coalesced."""
    _dev_check(inp, weight, bias, ln_weight, ln_bias)
    if not inp.is_contiguous():
        inp = inp.contiguous()
    b, l, d2 = inp.shape
    d = d2 if glu_done else d2 // 2                       # glu_done: the input is already gated (cm_ln_pw_glu), dim wide
    w = _f32c(weight).reshape(d, -1)
    bs, g, bt = _f32c(bias), _f32c(ln_weight), _f32c(ln_bias)
    out = torch.empty((b, l, d), dtype=inp.dtype, device=inp.device)
    a = N.GluDwconvArgs()
    a.batch, a.seqlen, a.dim, a.ksize, a.io_dtype, a.glu_done = b, l, d, w.shape[1], _DT[inp.dtype], int(bool(glu_done))
    a.in_, a.weight, a.bias, a.ln_g, a.ln_b, a.eps, a.out = _ptr(inp), _ptr(w), _ptr(bs), _ptr(g), _ptr(bt), float(eps), _ptr(out)
    wt = w.t().contiguous() if weight_t is None else weight_t
    if wt.dtype != torch.float32 or wt.shape != (w.shape[1], d) or not wt.is_contiguous():
        raise RuntimeError("glu_dwconv_ln_gelu: weight_t must be a contiguous fp32 (k, dim) tensor")
    a.weight_t = _ptr(wt)
    if lin_w is not None:                                 # closing Linear(dim, dim) of the module, applied in the same kernel
        if not isinstance(lin_w, PackedWeight) or lin_w.shape != (d, d) or lin_b is None:
            raise RuntimeError("glu_dwconv_ln_gelu: lin_w must be a PackedWeight of shape (dim, dim), with lin_b")
        lb = _f32c(lin_b)
        a.lin_w, a.lin_b = _ptr(lin_w.data), _ptr(lb)
    a.stream, a.variant = _stream(), int(variant)        # variant 1: always the generic 16-step kernel
    _launch("cm_glu_dwconv_ln_gelu", N.lib().cm_glu_dwconv_ln_gelu, a, units=b * l)
    return out


def cnn_block1(feats, weight, bias, ln_weight, ln_bias, eps=1e-5, slope=0.01, out_dtype=torch.bfloat16, pad_out=1):
    """feats (batch, T, F) fp32 -> (batch, ceil(T/2) + 2*pad_out, ceil(F/2) + 2*pad_out, C): 3x3 stride-2 conv, LayerNorm
    over (freq, channel), LeakyReLU, with the reflect border of the next 3x3 block already in place (cm_cnn_block1)."""
    _dev_check(feats, weight, bias, ln_weight, ln_bias)
    feats = feats.float().contiguous()
    b, t, f = feats.shape
    w, bs, g, bt = _f32c(weight), _f32c(bias), _f32c(ln_weight), _f32c(ln_bias)
    cch = w.shape[0]
    t1, f1 = (t + 1) // 2, (f + 1) // 2
    out = torch.empty((b, t1 + 2 * pad_out, f1 + 2 * pad_out, cch), dtype=out_dtype, device=feats.device)
    a = N.CnnBlock1Args()
    a.batch, a.T, a.F, a.C, a.io_dtype, a.pad_out = b, t, f, cch, _DT[out_dtype], pad_out
    a.feats, a.weight, a.bias, a.ln_g, a.ln_b, a.eps, a.slope, a.out = (_ptr(feats), _ptr(w), _ptr(bs), _ptr(g), _ptr(bt),
                                                                          float(eps), float(slope), _ptr(out))
    a.stream = _stream()
    _launch("cm_cnn_block1", N.lib().cm_cnn_block1, a, units=b * t)
    return out


def cnn_front_supported(feats, w1, w2_ohwi) -> bool:
    return (feats.shape[-1] == 80 and tuple(w1.shape) == (64, 1, 3, 3) and tuple(w2_ohwi.shape) == (32, 3, 3, 64)
            and w2_ohwi.dtype == torch.bfloat16 and feats.shape[1] >= 5)


def cnn_front(feats, w1, b1, ln1_w, ln1_b, eps1, w2_ohwi, b2, ln2_w, ln2_b, eps2, slope=0.01):
    """feats (batch, T, 80) fp32 -> (batch, T2, 640) bf16: both ConvolutionFrontEnd blocks in one kernel (cm_cnn_front)."""
    _dev_check(feats, w1, b1, ln1_w, ln1_b, w2_ohwi, b2, ln2_w, ln2_b)
    feats = feats.float().contiguous()
    if not cnn_front_supported(feats, w1, w2_ohwi) or not w2_ohwi.is_contiguous():
        raise RuntimeError("cnn_front: needs 80 bins, a (64, 1, 3, 3) first conv and a contiguous bf16 (32, 3, 3, 64) second conv")
    b, t, f = feats.shape
    t1 = (t + 1) // 2
    t2 = (t1 - 1) // 2 + 1
    w1f, b1f, g1, bt1 = _f32c(w1), _f32c(b1), _f32c(ln1_w).reshape(-1), _f32c(ln1_b).reshape(-1)
    b2f, g2, bt2 = _f32c(b2), _f32c(ln2_w).reshape(-1), _f32c(ln2_b).reshape(-1)
    if g1.numel() != 40 * 64 or g2.numel() != 20 * 32:
        raise RuntimeError("cnn_front: LayerNorm parameters must have 40*64 and 20*32 elements")
    out = torch.empty((b, t2, 640), dtype=torch.bfloat16, device=feats.device)
    a = N.CnnFrontArgs()
    a.batch, a.T, a.F, a.C1, a.C2 = b, t, f, 64, 32
    a.feats, a.w1, a.b1, a.ln1_g, a.ln1_b = _ptr(feats), _ptr(w1f), _ptr(b1f), _ptr(g1), _ptr(bt1)
    a.w2, a.b2, a.ln2_g, a.ln2_b = _ptr(w2_ohwi), _ptr(b2f), _ptr(g2), _ptr(bt2)
    a.eps1, a.eps2, a.slope, a.out, a.stream = float(eps1), float(eps2), float(slope), _ptr(out), _stream()
    _launch("cm_cnn_front", N.lib().cm_cnn_front, a, units=b * t)
    return out


def cnn_block2(y1, weight_ohwi, bias, ln_weight, ln_bias, eps=1e-5, slope=0.01):
    """y1 (batch, T_in, F_in, 64) bf16 channels-last (cnn_block1 output with its reflect border) ->
    (batch, T2, F2*32) bf16: 3x3 stride-2 conv 64 -> 32, LayerNorm over (freq, channel), LeakyReLU (cm_cnn_block2).
    weight_ohwi: (32, 3, 3, 64) bf16 contiguous (a Conv2d weight permuted to output, row, column, input)."""
    _dev_check(y1, weight_ohwi, bias, ln_weight, ln_bias)
    if y1.dtype != torch.bfloat16 or not y1.is_contiguous() or y1.dim() != 4:
        raise RuntimeError("cnn_block2: input must be a contiguous bf16 (batch, T, F, C) tensor")
    if weight_ohwi.dtype != torch.bfloat16 or not weight_ohwi.is_contiguous() or weight_ohwi.dim() != 4:
        raise RuntimeError("cnn_block2: weight must be a contiguous bf16 (C_out, 3, 3, C_in) tensor")
    b, t, f, cin = y1.shape
    cout = weight_ohwi.shape[0]
    if tuple(weight_ohwi.shape[1:]) != (3, 3, cin):
        raise RuntimeError("cnn_block2: weight shape does not match the input channels / 3x3 taps")
    t2, f2 = (t - 3) // 2 + 1, (f - 3) // 2 + 1
    bs, g, bt = _f32c(bias), _f32c(ln_weight).reshape(-1), _f32c(ln_bias).reshape(-1)
    if g.numel() != f2 * cout or bt.numel() != f2 * cout:
        raise RuntimeError("cnn_block2: LayerNorm parameters must have F2 * C_out elements")
    out = torch.empty((b, t2, f2 * cout), dtype=torch.bfloat16, device=y1.device)
    a = N.CnnBlock2Args()
    a.batch, a.T_in, a.F_in, a.C_in, a.C_out = b, t, f, cin, cout
    a.in_, a.weight, a.bias, a.ln_g, a.ln_b, a.eps, a.slope, a.out = (_ptr(y1), _ptr(weight_ohwi), _ptr(bs), _ptr(g), _ptr(bt),
                                                                      float(eps), float(slope), _ptr(out))
    a.stream = _stream()
    _launch("cm_cnn_block2", N.lib().cm_cnn_block2, a, units=b * t2)
    return out


def gemm_supported(m: int, n: int, k: int, epilogue: int) -> bool:
    return k % 64 == 0 and n % 256 == 0 and (epilogue != 2 or n == 256)


def gemm_bf16(a, w, bias=None, epilogue=0, x=None, alpha=1.0, norm1=None, norm2=None, want_out=True):
    """acc = a @ w.T (a (M, K) bf16 row view, w (N, K) bf16) with a fused epilogue (cm_gemm_bf16):
    0: + bias -> bf16;  1: gelu(+ bias) -> bf16;  2: x += alpha*(acc + bias) [optional LN1 into x], out = LN2(x) bf16.
    bias / LayerNorm parameters are fp32 tensors; norm = (weight, bias, eps).  Returns out (or None)."""
    _dev_check(a, w, bias, x)
    if a.dtype != torch.bfloat16 or w.dtype != torch.bfloat16 or a.dim() != 2 or a.stride(1) != 1 or not w.is_contiguous():
        raise RuntimeError("gemm_bf16: a must be a (M, K) bf16 row-major view and w a contiguous (N, K) bf16 tensor")
    m, k = a.shape
    n = w.shape[0]
    g = N.GemmArgs()
    g.M, g.N, g.K, g.epilogue = m, n, k, epilogue
    g.A, g.lda, g.W, g.ldw, g.bias = _ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(bias)
    out = None
    if epilogue != 2 or want_out:
        out = torch.empty((m, n), dtype=torch.bfloat16, device=a.device)
        g.out, g.ldo = _ptr(out), n
    if epilogue == 2:
        if x is None or x.dtype != torch.float32 or not x.is_contiguous() or x.shape != (m, n):
            raise RuntimeError("gemm_bf16: epilogue 2 needs a contiguous fp32 residual x of shape (M, N)")
        g.x, g.alpha = _ptr(x), float(alpha)
        if norm1 is not None:
            g.g1, g.b1, g.eps1 = _ptr(norm1[0]), _ptr(norm1[1]), float(norm1[2])
        if norm2 is not None:
            g.g2, g.b2, g.eps2 = _ptr(norm2[0]), _ptr(norm2[1]), float(norm2[2])
    g.stream = _stream()
    _launch("cm_gemm_bf16", N.lib().cm_gemm_bf16, g, units=m)
    return out


def ffn_supported(d_model: int, hidden: int, dtype) -> bool:
    return dtype == torch.bfloat16 and d_model == 256 and hidden >= 256 and hidden % 256 == 0


# cm_ffn_fused's inference forward on v_mfma_f32_32x32x16_bf16 (csrc/ffn_fused32.hip, weights in the 32 x 16 tile image) instead of
# 16x16x32 (csrc/ffn_fused.hip): CM_FFN_MFMA32=1
FFN_LAYOUT = 32 if os.environ.get("CM_FFN_MFMA32", "0") == "1" else 16
# CM_FFN_SMALL_ROWS=N: launches of at most N rows run the 32x32x16 kernel with 32-token workgroups (fused.py then keeps both weight
# images).  Off by default: alone it wins below ~12 k rows (8000 rows: 28.5 / 19.4 vs 34.4 / 24.4 us; 16000: 42.8 / 31.2 vs 40.5 / 28.8),
# in the encoder at 16 x 40 s (two parts of 8000 rows whose kernels share the chip) it changes nothing (4.78 vs 4.78 ms)
SMALL_FFN_ROWS = int(os.environ.get("CM_FFN_SMALL_ROWS", "0"))


class PackedWeight:
    """A bf16 (rows, cols) matrix in a fragment-tiled image: ``layout`` 16 = 16-row x 32-column tiles (cm_ffn_pack_weights: every
    kernel that streams packed weights), 32 = 32 x 16 tiles (cm_ffn_pack_weights32: cm_ffn_fused with cm_ffn_args.layout = 1)."""

    def __init__(self, w: torch.Tensor, layout: int = 16):
        _dev_check(w)
        if w.dtype != torch.bfloat16 or w.dim() != 2:
            raise RuntimeError("PackedWeight: expected a 2-D bf16 tensor")
        if layout not in (16, 32):
            raise RuntimeError("PackedWeight: layout must be 16 or 32")
        w = w.contiguous()
        self.shape = tuple(w.shape)
        self.layout = layout
        self.data = torch.empty(w.numel(), dtype=torch.bfloat16, device=w.device)
        self.repack_(w)

    def repack_(self, w: torch.Tensor) -> "PackedWeight":
        """Rewrite the image from a new (rows, cols) bf16 matrix of the same shape, in place (the address a captured graph holds)."""
        if w.dtype != torch.bfloat16 or tuple(w.shape) != self.shape or w.device != self.data.device:
            raise RuntimeError("PackedWeight.repack_: expected a bf16 tensor of the packed shape on the image's device")
        w = w.contiguous()
        fn = N.lib().cm_ffn_pack_weights if self.layout == 16 else N.lib().cm_ffn_pack_weights32
        rc = fn(_ptr(w), w.shape[0], w.shape[1], _ptr(self.data), _stream())
        N.check(rc, "cm_ffn_pack_weights")
        return self


def ffn_fused(x, pre_norm, w1, b1, w2, b2, alpha=0.5, addend=None, add_scale=1.0, norm1=None, norm2=None,
              x_out=None, want_h=True, h_dtype=torch.bfloat16, proj_w=None, proj_b=None, proj_out=None, train=None, tokens=None):
    """Whole feed-forward module on the fp32 residual stream (cm_ffn_fused):
        xin = x + add_scale*addend;  r = xin + alpha*(W2 gelu(W1 LN_pre(xin) + b1) + b2);  r = LN1(r) if norm1;
        x_out <- r (x itself when x_out is None);  returns h = LN2(r) (or r when norm2 is None) in h_dtype if want_h.
    x (rows, 256) fp32; w1 (hidden, 256) / w2 (256, hidden) bf16 tensors or PackedWeight (pack once, reuse); b1/b2 and
    LayerNorm parameters fp32; addend (rows, 256) bf16; norm = (weight, bias, eps).
    With ``proj_w`` (PackedWeight (P, 256), P % 256 == 0; ``proj_b`` fp32 (P) or None) the Linear that consumes h runs in the
    same kernel and (rows, P) bf16 = h @ proj_w^T (+ proj_b) is returned in h's place; h itself is not stored.
    ``train`` = (p1, p2, seed1, seed2): the module's TRAINING forward, x_out = x + alpha * drop_p2(W2 drop_p1(gelu(bf16(W1 LN(x) + b1))) + b2)
    with the backward's inputs stored on the way -> (x_out, (pre, xn, stats)): pre (rows, hidden) bf16 = W1 LN(x) + b1, xn (rows, 256)
    bf16 = LN(x), stats (2, rows) fp32 = the rows' mean and 1/std (layernorm_bwd's input); the dropout decisions are cm_dropout.h's function of (seed, element index) (no mask is stored)."""
    _dev_check(x, b1, b2, addend, proj_b)
    rows, d = x.shape
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise RuntimeError("ffn_fused: x must be a contiguous fp32 (rows, 256) tensor")
    if d != 256:
        raise RuntimeError(f"ffn_fused: d_model must be 256 (got {d})")
    lay = w1.layout if isinstance(w1, PackedWeight) else (w2.layout if isinstance(w2, PackedWeight) else (16 if train is not None else FFN_LAYOUT))
    w1 = w1 if isinstance(w1, PackedWeight) else PackedWeight(w1, lay)
    w2 = w2 if isinstance(w2, PackedWeight) else PackedWeight(w2, lay)
    if w1.shape[1] != d or w2.shape != (d, w1.shape[0]):
        raise RuntimeError("ffn_fused: weight shapes do not match x")
    if w1.layout != w2.layout or (proj_w is not None and isinstance(proj_w, PackedWeight) and proj_w.layout != w1.layout):
        raise RuntimeError("ffn_fused: w1, w2 and proj_w must be packed in the same layout")
    if train is not None and w1.layout != 16:
        raise RuntimeError("ffn_fused: the training forward takes layout-16 weights")
    for t in (b1, b2) + tuple(pre_norm[:2]):
        if t.dtype != torch.float32:
            raise RuntimeError("ffn_fused: biases and LayerNorm parameters must be fp32")
    a = N.FfnArgs()
    a.rows, a.dim, a.hidden = rows, d, w1.shape[0]
    a.x, a.pre_g, a.pre_b, a.pre_eps = _ptr(x), _ptr(pre_norm[0]), _ptr(pre_norm[1]), float(pre_norm[2])
    a.w1, a.b1, a.w2, a.b2, a.alpha = _ptr(w1.data), _ptr(b1), _ptr(w2.data), _ptr(b2), float(alpha)
    a.layout = 1 if w1.layout == 32 else 0
    if w1.layout == 32:
        # 32-token workgroups when 64-token ones would not fill the chip's 512 workgroup slots
        a.tokens = int(tokens) if tokens else (32 if rows <= SMALL_FFN_ROWS else 64)
    if addend is not None:
        if addend.dtype != torch.bfloat16 or not addend.is_contiguous() or addend.shape != x.shape:
            raise RuntimeError("ffn_fused: addend must be a contiguous bf16 tensor shaped like x")
        a.addend, a.add_scale = _ptr(addend), float(add_scale)
    if norm1 is not None:
        a.n1_g, a.n1_b, a.n1_eps = _ptr(norm1[0]), _ptr(norm1[1]), float(norm1[2])
    if norm2 is not None:
        a.n2_g, a.n2_b, a.n2_eps = _ptr(norm2[0]), _ptr(norm2[1]), float(norm2[2])
    xo = x if x_out is None else x_out
    a.x_out = _ptr(xo)
    h = None
    if train is not None:
        if addend is not None or norm1 is not None or norm2 is not None or proj_w is not None:
            raise RuntimeError("ffn_fused: the training forward takes no addend / norm1 / norm2 / projection")
        pre = torch.empty((rows, w1.shape[0]), dtype=torch.bfloat16, device=x.device)
        xn = torch.empty((rows, d), dtype=torch.bfloat16, device=x.device)
        stats = torch.empty((2, rows), dtype=torch.float32, device=x.device)
        a.pre_out, a.xn_out, a.stats_out = _ptr(pre), _ptr(xn), _ptr(stats)
        a.p1, a.p2, a.seed1, a.seed2 = float(train[0]), float(train[1]), int(train[2]), int(train[3])
        a.seed_epoch = _seed_epoch_ptr()
        a.stream = _stream()
        _launch("cm_ffn_fused", N.lib().cm_ffn_fused, a, units=rows)
        return xo, (pre, xn, stats)
    if proj_w is not None:
        if not isinstance(proj_w, PackedWeight) or proj_w.shape[1] != d or proj_w.shape[0] % 256 or proj_w.shape[0] > 4096:
            raise RuntimeError("ffn_fused: proj_w must be a PackedWeight of shape (P, 256), P a multiple of 256 up to 4096")
        h = proj_out if proj_out is not None else torch.empty((rows, proj_w.shape[0]), dtype=torch.bfloat16, device=x.device)
        if h.shape != (rows, proj_w.shape[0]) or h.dtype != torch.bfloat16 or not h.is_contiguous():
            raise RuntimeError("ffn_fused: proj_out must be a contiguous bf16 (rows, P) tensor")
        _dev_check(h)
        pb = _f32c(proj_b)
        a.proj_w, a.proj_b, a.proj_out, a.proj_dim = _ptr(proj_w.data), _ptr(pb), _ptr(h), proj_w.shape[0]
    elif want_h:
        h = torch.empty((rows, d), dtype=h_dtype, device=x.device)
        a.h_out, a.h_dtype = _ptr(h), _DT[h_dtype]
    a.stream = _stream()
    _launch("cm_ffn_fused", N.lib().cm_ffn_fused, a, units=rows)
    return xo, h


_BANDS = {}


def _mel_bands(fb: torch.Tensor):
    """Per-filter [lo, hi) range of non-zero bins (triangular filters are contiguous), cached per filterbank tensor."""
    key = (fb.data_ptr(), tuple(fb.shape))
    if key not in _BANDS:
        nz = (fb != 0)
        idx = torch.arange(fb.shape[0], device=fb.device)[:, None]
        lo = torch.where(nz, idx, fb.shape[0]).amin(0).to(torch.int32)
        hi = (torch.where(nz, idx, -1).amax(0) + 1).to(torch.int32)
        lo = torch.minimum(lo, hi)
        width = (hi - lo).to(torch.int64)
        off = torch.zeros(fb.shape[1] + 1, dtype=torch.int64, device=fb.device)
        off[1:] = torch.cumsum(width, 0)
        total = int(off[-1])
        packed = None
        if 0 < total <= 4096:                              # gather each filter's band into one packed vector
            m_of = torch.repeat_interleave(torch.arange(fb.shape[1], device=fb.device), width)
            f_of = torch.arange(total, device=fb.device) - off[m_of] + lo.to(torch.int64)[m_of]
            packed = fb[f_of, m_of].contiguous()
        _BANDS[key] = (lo.contiguous(), hi.contiguous(), off.to(torch.int32).contiguous(), packed)
    return _BANDS[key]


def fbank_from_stft(spec, fbank, amin=1e-10, top_db=80.0, mean=None, std=None):
    """torch.stft(..., return_complex=True) output (batch, n_freq, frames) -> (batch, frames, n_mels) fp32 log-mel
    features with the per-utterance top_db clamp and optional global normalisation (cm_fbank_mel_db + cm_fbank_finish)."""
    _dev_check(spec, fbank, mean, std)
    sr = torch.view_as_real(spec) if spec.is_complex() else spec
    if sr.dtype != torch.float32 or sr.stride(3) != 1 or any(st % 2 for st in sr.stride()[:3]):
        sr = sr.float().contiguous()
    b, nf, t, _ = sr.shape                               # strided view accepted as is (no 260 MB transpose copy)
    fb = _f32c(fbank)
    m = fb.shape[1]
    db = torch.empty((b, t, m), dtype=torch.float32, device=sr.device)
    umax = torch.empty((b,), dtype=torch.float32, device=sr.device)
    part = torch.empty((b, (t + 15) // 16), dtype=torch.float32, device=sr.device)      # per-tile maxima (no atomics)
    a = N.FbankArgs()
    a.batch, a.n_freq, a.frames, a.n_mels = b, nf, t, m
    a.spec, a.fbank, a.db, a.umax, a.amin, a.top_db = _ptr(sr), _ptr(fb), _ptr(db), _ptr(umax), float(amin), float(top_db)
    mean, std = _f32c(mean), _f32c(std)
    a.mean, a.std = _ptr(mean), _ptr(std)
    lo, hi, off, packed = _mel_bands(fb)
    a.band_lo, a.band_hi = _ptr(lo), _ptr(hi)
    if packed is not None:
        a.band_off, a.band_w = _ptr(off), _ptr(packed)
    a.spec_bs, a.spec_fs, a.spec_ts = sr.stride(0) // 2, sr.stride(1) // 2, sr.stride(2) // 2
    a.umax_part = _ptr(part)
    a.stream = _stream()
    _launch("cm_fbank_mel_db", N.lib().cm_fbank_mel_db, a, units=b * t)
    _launch("cm_fbank_finish", N.lib().cm_fbank_finish, a, units=b * t)
    return db


_FFT_TABLES = {}


def _fft_tables(window: torch.Tensor, n_fft: int):
    """(window zero-padded and centred to n_fft, twiddles exp(-2 pi i k / n_fft) computed in float64), cached."""
    key = (window.data_ptr(), window.numel(), n_fft, str(window.device))
    if key not in _FFT_TABLES:
        win = torch.zeros(n_fft, dtype=torch.float32, device=window.device)
        left = (n_fft - window.numel()) // 2
        win[left:left + window.numel()] = window.float()
        k = torch.arange(n_fft, dtype=torch.float64)
        ang = -2.0 * torch.pi * k / n_fft
        tw = torch.stack([torch.cos(ang), torch.sin(ang)], dim=1).to(torch.float32).to(window.device).contiguous()
        _FFT_TABLES[key] = (win, tw)
    return _FFT_TABLES[key]


def fbank_wav_supported(n_fft: int, hop: int, n_mels: int) -> bool:
    return n_fft == 512 and hop % 2 == 0 and n_mels <= 128


def fbank_from_wav(wav, window, n_fft, hop, fbank, amin=1e-10, top_db=80.0, mean=None, std=None):
    """(batch, samples) fp32 waveform -> (batch, 1 + samples // hop, n_mels) fp32 log-mel features, STFT included
    (cm_fbank_wav + cm_fbank_finish): torch.stft(center=True, pad_mode='constant', window centred in n_fft) semantics."""
    _dev_check(wav, window, fbank, mean, std)
    if wav.dtype != torch.float32 or wav.dim() != 2 or wav.stride(1) != 1:
        wav = wav.float().contiguous()
    if wav.stride(0) % 2:
        wav = wav.contiguous() if wav.shape[1] % 2 == 0 else torch.nn.functional.pad(wav, (0, 1))[:, :wav.shape[1]]
    b, ns = wav.shape
    fb = _f32c(fbank)
    nf, m = fb.shape
    t = 1 + ns // hop
    win, tw = _fft_tables(window, n_fft)
    lo, hi, off, packed = _mel_bands(fb)
    if packed is None:
        raise RuntimeError("fbank_from_wav: filterbank is not banded (no packed band weights)")
    db = torch.empty((b, t, m), dtype=torch.float32, device=wav.device)
    umax = torch.empty((b,), dtype=torch.float32, device=wav.device)
    part = torch.empty((b, (t + 15) // 16), dtype=torch.float32, device=wav.device)
    a = N.FbankArgs()
    a.batch, a.n_freq, a.frames, a.n_mels = b, nf, t, m
    a.fbank, a.db, a.umax, a.amin, a.top_db = _ptr(fb), _ptr(db), _ptr(umax), float(amin), float(top_db)
    mean, std = _f32c(mean), _f32c(std)
    a.mean, a.std = _ptr(mean), _ptr(std)
    a.band_lo, a.band_hi, a.band_off, a.band_w = _ptr(lo), _ptr(hi), _ptr(off), _ptr(packed)
    a.umax_part = _ptr(part)
    a.wav, a.window, a.twiddle, a.wav_bs = _ptr(wav), _ptr(win), _ptr(tw), wav.stride(0)
    a.samples, a.hop, a.n_fft = ns, hop, n_fft
    a.stream = _stream()
    _launch("cm_fbank_wav", N.lib().cm_fbank_wav, a, units=b * t)
    _launch("cm_fbank_finish", N.lib().cm_fbank_finish, a, units=b * t)
    return db


def spec_drop_(feats, start, length, dim, fill):
    """In-place SpecAugment masking (cm_spec_drop): feats (batch, frames, n_mels) fp32; start/length int32
    (batch, n_masks); dim 1 = time, 2 = frequency; fill = 0-dim device tensor."""
    _dev_check(feats, start, length, fill)
    if feats.dtype != torch.float32 or not feats.is_contiguous():
        raise RuntimeError("spec_drop_: feats must be contiguous fp32")
    start, length = start.to(torch.int32).contiguous(), length.to(torch.int32).contiguous()
    fill = fill.reshape(1).float()
    b, t, m = feats.shape
    a = N.SpecDropArgs()
    a.batch, a.frames, a.n_mels, a.n_masks, a.dim = b, t, m, start.shape[1], dim
    a.feats, a.start, a.length, a.fill = _ptr(feats), _ptr(start), _ptr(length), _ptr(fill)
    a.stream = _stream()
    _launch("cm_spec_drop", N.lib().cm_spec_drop, a, units=b * t)
    return feats
