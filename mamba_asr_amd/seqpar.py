"""Time-parallel (sequence-split) ConMamba encoder forward across the GPUs of a node.

The reference has NO sequence parallelism: its long-form recipe (hparams/S2S/conmambamamba_large.yaml:
251-257, L ~ 16000 frames) runs every utterance on one GPU.  Here an utterance's encoder steps are cut into contiguous
time shards, one per rank (rank order = time order), and the three time-mixing stages of a ConMamba layer are stitched
with small exchanges; everything else in the layer (both feed-forward modules, the LayerNorms, in_proj / out_proj, the
pointwise conv, GLU, the convolution module's Linear) is row-local and needs nothing:

  * selective scan (reference math selective_scan_interface.py:91-157).  The recurrence h_t = a_t h_{t-1} + b_t is
    affine in the state, so a shard is summarised by (P, h_end): P = prod_t a_t over the shard (from the forward
    kernel's per-chunk products x[..., 2n]) and h_end = its last state from a zero start (x[..., 1::2] of the last
    chunk).  ONE all-gather of those 2 x (batch, E, N) fp32 tensors (64 KB per utterance and direction at E = 512) lets
    every rank fold its carry-in  H_r = P_{r-1} H_{r-1} + h_{r-1}  (against rank order for the backward direction), and
    a second pass of the same kernel from that state (cm_selective_scan_fwd's h0) gives exactly the unsplit outputs.
    The second pass is skipped on the rank whose carry is zero.  Cost: 2 scans of 1 / W of the sequence instead of 1 scan
    of all of it -- a latency win for W > 2 on long-form audio, which is what this row is for.
  * causal depthwise conv, width 4 (bimamba.py:83-91): 3-frame halo from the left neighbour (right, for the backward
    direction);
  * the convolution module's depthwise conv, k = 31, 'same' padding (Conmamba.py:284-290): 15-frame halos both sides.
Halos travel in one all-gather of each rank's first / last frames (a few KB).  Edge ranks pad with zeros, which is
the unsplit operator's own zero padding.

The exchange code is backend-agnostic: ``HipBackend`` (default) runs the HIP kernels of this package; the world-2 gloo
test on CPU plugs in the oracle (test infrastructure) to check the exchange algebra; tests/test_seqpar.py checks the
HIP path on one GPU by running the shards of a sequence through the same functions on W threads (``run_local``).
Forward / inference only.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------------------
# collectives: torch.distributed, or an in-process stand-in that walks the shards one after another
# ------------------------------------------------------------------------------------------------------------
class DistGroup:
    """all_gather over a torch.distributed process group (nccl == RCCL on the GPUs, gloo in the CPU test)."""

    def __init__(self, group=None):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def all_gather(self, t: torch.Tensor) -> List[torch.Tensor]:
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t.contiguous(), group=self.group)
        return out


class _ThreadRank:
    def __init__(self, shared, rank, world):
        self._s, self.rank, self.world = shared, rank, world

    def all_gather(self, t: torch.Tensor) -> List[torch.Tensor]:
        self._s["slots"][self.rank] = t
        self._s["barrier"].wait()
        out = list(self._s["slots"])
        self._s["barrier"].wait()
        return out


def run_local(world: int, fn):
    """W ranks emulated by W threads of one process (single-GPU tests and debugging): fn(group) runs once per rank with
    a group whose all_gather meets the other threads at a barrier; every thread launches on the same HIP stream, so a
    tensor handed over at the barrier is ordered before its readers.  -> [fn's result for rank 0, 1, ...]."""
    import threading
    shared = {"slots": [None] * world, "barrier": threading.Barrier(world)}
    results, errors = [None] * world, []
    dev = torch.cuda.current_device() if torch.cuda.is_available() else None

    def work(r):
        try:
            if dev is not None:
                torch.cuda.set_device(dev)
            results[r] = fn(_ThreadRank(shared, r, world))
        except BaseException as e:                             # noqa: BLE001 -- re-raised below
            errors.append(e)
            shared["barrier"].abort()

    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errors:
        raise errors[0]
    return results


def carry_in(Ps: Sequence[torch.Tensor], hs: Sequence[torch.Tensor], rank: int, reverse: bool) -> Optional[torch.Tensor]:
    """State entering shard ``rank`` given every shard's (P, h_end): fold the affine maps of the shards before it in
    scan order.  None when nothing precedes it."""
    order = range(len(Ps) - 1, rank, -1) if reverse else range(0, rank)
    H = None
    for r in order:
        H = hs[r] if H is None else Ps[r] * H + hs[r]
    return H


def shard_summary(x_ckpt: torch.Tensor, reverse: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """(P, h_end), each (batch, dim, dstate), from the forward kernel's checkpoints (batch, dim, nchunks, 2 * dstate):
    [..., 2n] = product of exp(delta' A) over the chunk, [..., 2n + 1] = state at the chunk's last processed step."""
    P = x_ckpt[..., 0::2].prod(dim=2)
    h_end = x_ckpt[:, :, 0 if reverse else -1, 1::2]
    return P.contiguous(), h_end.contiguous()


# ------------------------------------------------------------------------------------------------------------
# backends
# ------------------------------------------------------------------------------------------------------------
class HipBackend:
    """The package's HIP kernels (GPU tensors only)."""

    @staticmethod
    def scan(u, delta, A, B, C, D, z, delta_bias, reverse, h0):
        from . import ops
        _, x, y = ops.selective_scan_fwd(u, delta, A, B, C, D, z, delta_bias, True, reverse=reverse, need_out=False, need_x=True,
                                         h0=h0)
        P, h_end = shard_summary(x, reverse)
        return y, P, h_end

    @staticmethod
    def causal_conv(x, weight, bias, reverse):
        from . import ops
        return ops.causal_conv1d_fwd(x, weight, bias, True, reverse=reverse)

    @staticmethod
    def dwconv_rows(x, weight, bias, pad_left):
        from . import ops
        return ops.dwconv_cl_fwd(x.contiguous(), weight, bias, pad_left)


# ------------------------------------------------------------------------------------------------------------
# the three stitched operators
# ------------------------------------------------------------------------------------------------------------
def exchange_halo(t: torch.Tensor, n_left: int, n_right: int, group, dim: int = -1) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (from_left, from_right): the last ``n_left`` frames of the previous rank's ``t`` and the first ``n_right`` frames
    of the next rank's, along ``dim``; zeros at the sequence's ends.  One all-gather of (head | tail) slabs."""
    T = t.shape[dim]
    if T < max(n_left, n_right):
        raise ValueError(f"time shard of {T} frames is shorter than the halo ({max(n_left, n_right)}): use fewer ranks")
    head = t.narrow(dim, 0, n_right) if n_right else None
    tail = t.narrow(dim, T - n_left, n_left) if n_left else None
    parts = [p for p in (head, tail) if p is not None]
    slabs = group.all_gather(torch.cat(parts, dim=dim).contiguous())
    r, W = group.rank, group.world
    zeros = lambda n: torch.zeros_like(t.narrow(dim, 0, n))
    from_left = (slabs[r - 1].narrow(dim, n_right, n_left) if r > 0 else zeros(n_left)) if n_left else None
    from_right = (slabs[r + 1].narrow(dim, 0, n_right) if r + 1 < W else zeros(n_right)) if n_right else None
    return from_left, from_right


def causal_conv1d_seq_parallel(x, weight, bias, group, reverse: bool = False, backend=HipBackend):
    """Causal depthwise conv + SiLU (reference bimamba.py:278-287) on a time shard x (batch, dim, T_local)."""
    w = weight.shape[-1] - 1
    left, right = exchange_halo(x, 0 if reverse else w, w if reverse else 0, group)
    if reverse:
        return backend.causal_conv(torch.cat([x, right], dim=-1), weight, bias, True)[..., : x.shape[-1]]
    return backend.causal_conv(torch.cat([left, x], dim=-1), weight, bias, False)[..., w:]


def selective_scan_seq_parallel(u, delta, A, B, C, D, z, delta_bias, group, reverse: bool = False, backend=HipBackend):
    """Selective scan (softplus on, as the mixer calls it) over time shards: -> out_z of this shard, equal to the
    corresponding slice of the unsplit scan."""
    y, P, h_end = backend.scan(u, delta, A, B, C, D, z, delta_bias, reverse, None)          # pass 1: from a zero state
    summ = group.all_gather(torch.stack([P, h_end]))                                           # (2, batch, dim, dstate) per rank
    H = carry_in([s[0] for s in summ], [s[1] for s in summ], group.rank, reverse)
    if H is None:
        return y                                                                               # first shard in scan order
    y2, _, _ = backend.scan(u, delta, A, B, C, D, z, delta_bias, reverse, H.contiguous())     # pass 2: from the carry
    return y2


def dwconv_same_seq_parallel(x_rows, weight, bias, group, backend=HipBackend):
    """'same'-padded depthwise conv over time on channels-last rows (batch, T_local, dim) (Conmamba.py:284-290, 442)."""
    k = weight.shape[-1]
    hl, hr = k // 2, k - 1 - k // 2
    left, right = exchange_halo(x_rows, hl, hr, group, dim=1)
    ext = torch.cat([left, x_rows, right], dim=1)
    return backend.dwconv_rows(ext, weight, bias, hl)[:, hl:hl + x_rows.shape[1]]


# ------------------------------------------------------------------------------------------------------------
# the mixer, the layer, the encoder
# ------------------------------------------------------------------------------------------------------------
def bimamba_seq_parallel(m, hidden, group, backend=HipBackend):
    """BiMamba v2 forward (reference bimamba.py:192-253) on a time shard hidden (batch, T_local, d_model)."""
    batch, T, _ = hidden.shape
    E, R, N = m.d_inner, m.dt_rank, m.d_state
    xz = F.linear(hidden, m.in_proj.weight, m.in_proj.bias).transpose(1, 2)                   # (batch, 2E, T)
    x, z = xz[:, :E].contiguous(), xz[:, E:].contiguous()
    outs = []
    for sfx, rev in (("", False), ("_b", True)):
        conv, xp, dtp = getattr(m, "conv1d" + sfx), getattr(m, "x_proj" + sfx), getattr(m, "dt_proj" + sfx)
        A = -torch.exp(getattr(m, "A_b_log" if rev else "A_log").float())
        Dp = getattr(m, "D_b" if rev else "D").float()
        u = causal_conv1d_seq_parallel(x, conv.weight.reshape(E, -1), conv.bias, group, rev, backend)
        x_dbl = F.linear(u.transpose(1, 2).reshape(batch * T, E), xp.weight.to(u.dtype))    # selective_scan_interface.py:186
        delta = (dtp.weight.to(u.dtype) @ x_dbl[:, :R].t()).reshape(E, batch, T).transpose(0, 1).contiguous()   # :187
        Bm = x_dbl[:, R:R + N].reshape(batch, T, N).transpose(1, 2).contiguous()
        Cm = x_dbl[:, R + N:].reshape(batch, T, N).transpose(1, 2).contiguous()
        outs.append(selective_scan_seq_parallel(u, delta, A, Bm, Cm, Dp, z, dtp.bias.float(), group, rev, backend))
    mix = 0.5 * outs[0] + 0.5 * outs[1] if m.if_devide_out else outs[0] + outs[1]
    return F.linear(mix.transpose(1, 2), m.out_proj.weight.to(mix.dtype), m.out_proj.bias)


def conv_module_seq_parallel(cm, x, group, backend=HipBackend):
    """ConvolutionModule forward (reference Conmamba.py:439-449, non-causal) on a time shard (batch, T_local, d_model)."""
    if cm.causal or cm.dilation != 1:
        raise NotImplementedError("time-split ConvolutionModule: the non-causal, undilated module of the ASR recipes only")
    pw = cm.bottleneck[0]
    out = F.glu(F.linear(cm.layer_norm(x), pw.weight.squeeze(-1), pw.bias), dim=-1)
    out = dwconv_same_seq_parallel(out, cm.conv.weight, cm.conv.bias, group, backend)
    return cm.after_conv(out)


def encoder_layer_seq_parallel(layer, x, group, backend=HipBackend):
    """ConmambaEncoderLayer.forward (reference Conmamba.py:631-650, eval mode) on a time shard."""
    x = x + 0.5 * layer.ffn_module1(x)
    x = bimamba_seq_parallel(layer.mamba, layer.norm1(x), group, backend) + x
    x = x + conv_module_seq_parallel(layer.convolution_module, x, group, backend)
    return layer.norm2(x + 0.5 * layer.ffn_module2(x))


@torch.no_grad()
def encoder_forward_seq_parallel(encoder, src_shard, group, backend=HipBackend):
    """ConmambaEncoder.forward (reference Conmamba.py:716-727) with the TIME axis split over ``group``: src_shard
    (batch, T_local, d_model) is this rank's contiguous slice of the sequence -> this rank's slice of the output."""
    if encoder.training:
        raise RuntimeError("time-split encoder forward is an inference path: call encoder.eval() first")
    out = src_shard
    for layer in encoder.layers:
        out = encoder_layer_seq_parallel(layer, out, group, backend)
    return encoder.norm(out)


# ------------------------------------------------------------------------------------------------------------
# the same, on the channels-last fused inference kernels (the path bench.py times; bf16, d_model 256)
# ------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def encoder_forward_seq_parallel_fused(encoder, src_shard, group):
    """Time-split ConmambaEncoder.forward on the FUSED kernels of fused.py (cm_ffn_fused, cm_conv_xproj, the row-group
    cm_scan_cl_fwd, cm_ln_pw_glu, cm_glu_dwconv_ln_gelu; bf16 GEMM operands, fp32 residual stream).  Row-local kernels run
    unchanged on the shard; the three time-mixing kernels run on halo-extended rows (3 frames either side for the two
    causal convs, 15 either side for the k = 31 depthwise conv; the outputs of the halo rows are dropped), and the scan
    runs twice: from a zero state, writing (decay, h_last) -- cm_scan_cl_dir's carry outputs -- which ONE all-gather per
    layer spreads, then from the folded carry (h0).  src_shard (batch, T_local, 256) -> (batch, T_local, 256) fp32.
    Raises if a layer is not covered by the all-native bf16 route (then use encoder_forward_seq_parallel)."""
    from . import fused, ops
    dtype = torch.bfloat16
    batch, T, D = src_shard.shape
    caches = [fused._cache(layer, dtype) for layer in encoder.layers]
    for layer, c in zip(encoder.layers, caches):
        if not (fused.supports(layer) and ops.ffn_supported(D, c.ffn1["w1"].shape[0], dtype) and c.rows_mode and c.row_width == 48 and c.wx_packed is not None
                and c.pw_packed is not None and c.in_bias is None and c.out_bias is None and c.gamma is None):
            raise NotImplementedError("layer outside the fused bf16 route (d_model 256, dt_rank <= 16): use encoder_forward_seq_parallel")
    dev = src_shard.device
    fin = (encoder.norm.norm.weight.detach().float(), encoder.norm.norm.bias.detach().float(), encoder.norm.norm.eps)
    E = caches[0].d_inner
    with torch.autocast("cuda", enabled=False):
        x = src_shard.detach().float().reshape(batch * T, D).contiguous().clone()
        out = None
        for li, c in enumerate(caches):
            f1, f2 = c.ffn1, c.ffn2
            _, h = ops.ffn_fused(x, f1["ln"], f1["w1p"], f1["b1f"], f1["w2p"], f1["b2f"], alpha=0.5, norm2=c.norm1)
            xz = (h @ c.in_proj.t()).view(batch, T, 2 * E)
            # both causal convs + x_proj on rows extended by 3 frames either side
            left, right = exchange_halo(xz[:, :, :E], 3, 3, group, dim=1)
            xext = torch.cat([left, xz[:, :, :E], right], dim=1)
            uext = torch.empty((batch, T + 6, 2 * E), dtype=dtype, device=dev)
            xdbl_ext = ops.conv_xproj(xext, c.dirs[0]["conv_w"], c.dirs[0]["conv_b"], c.dirs[1]["conv_w"], c.dirs[1]["conv_b"],
                                      c.wx_packed[0], c.wx_packed[1], out_f=uext[:, :, :E], out_b=uext[:, :, E:])
            ucat, xdbl = uext[:, 3:3 + T], xdbl_ext[:, 3:3 + T]
            ycat = torch.empty((batch, T, 2 * E), dtype=dtype, device=dev)
            summ = torch.empty((2, 2, batch, E, 16), dtype=torch.float32, device=dev)      # [direction][decay | h_last]
            dirs = [dict(u=ucat[:, :, i * E:(i + 1) * E], A=d["A"], D=d["D"], delta_bias=d["dt_bias"], dt_weight=d["dt_w16"],
                         xdbl=xdbl[:, :, 48 * i:48 * (i + 1)], out=ycat[:, :, i * E:(i + 1) * E], reverse=bool(i),
                         decay=summ[i, 0], h_last=summ[i, 1]) for i, d in enumerate(c.dirs)]
            ops.scan_cl_fwd(dirs, z=xz[:, :, E:], delta_softplus=True)                     # pass 1: from a zero state
            allsum = group.all_gather(summ)
            redo = []
            for i in range(2):
                H = carry_in([s_[i, 0] for s_ in allsum], [s_[i, 1] for s_ in allsum], group.rank, bool(i))
                if H is not None:
                    dd = {k: v for k, v in dirs[i].items() if k not in ("decay", "h_last")}
                    dd["h0"] = H.contiguous()
                    redo.append(dd)
            if redo:
                ops.scan_cl_fwd(redo, z=xz[:, :, E:], delta_softplus=True)                 # pass 2: from the carry
            y = ycat.view(batch * T, 2 * E) @ c.out_cat.t()
            gl = ops.ln_pw_glu(x, y, 1.0, c.cm_ln, c.pw_packed, c.pw_bf)
            gl3 = gl.view(batch, T, D)
            left, right = exchange_halo(gl3, 15, 15, group, dim=1)
            g = ops.glu_dwconv_ln_gelu(torch.cat([left, gl3, right], dim=1), c.dw_w, c.dw_b, c.cm_ln2[0], c.cm_ln2[1], c.cm_ln2[2],
                                       weight_t=c.dw_wt, glu_done=True)[:, 15:15 + T]
            yl = torch.addmm(c.lin_b, g.reshape(-1, D), c.lin_w.t())
            if li == len(caches) - 1:
                _, out = ops.ffn_fused(x, f2["ln"], f2["w1p"], f2["b1f"], f2["w2p"], f2["b2f"], alpha=0.5, addend=yl, norm1=c.norm2,
                                       norm2=fin, h_dtype=torch.float32)
            else:
                ops.ffn_fused(x, f2["ln"], f2["w1p"], f2["b1f"], f2["w2p"], f2["b2f"], alpha=0.5, addend=yl, norm1=c.norm2, want_h=False)
        return out.view(batch, T, D)


def exchange_bytes_per_layer(batch: int, d_model: int, expand: int = 2, d_state: int = 16, d_conv: int = 4, kernel_size: int = 31,
                             itemsize: int = 4) -> int:
    """Bytes one rank contributes to the all-gathers of one layer: 2 directions x (P, h_end) fp32 + the three halos."""
    E = expand * d_model
    scan = 2 * 2 * batch * E * d_state * 4
    halos = 2 * batch * E * (d_conv - 1) * itemsize + batch * d_model * (kernel_size - 1) * itemsize
    return scan + halos
