"""Data-parallel gradient exchange for ConMamba training on one MI355X node (one process per GPU).

The reference trains with SpeechBrain's Brain, which wraps modules in torch DistributedDataParallel over NCCL
(train_CTC.py:1062 ddp_init_group; hparams/CTC/conmamba_large.yaml:90 grad_accumulation_factor 4): ONE exchange
per optimizer step, a mean all-reduce of gradients, suppressed (`no_sync`) on non-stepping micro-batches.  This
module is that exchange written directly on torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo"
in the CPU tests), laid out for 288 GB GPUs on point-to-point links:

  * gradients LIVE in persistent flat fp32 buckets (≈25 MB, parameters in REVERSE registration order, i.e. roughly
    the order backward produces them): every ``p.grad`` is a view into its bucket, so autograd accumulates straight
    into communication memory -- no per-step ``torch.cat`` into a staging tensor and no per-parameter copy-back
    (the first version moved 2 x 126 MB and launched hundreds of small kernels per optimizer step that way);
  * a bucket's exchange is launched from the autograd hook of its last gradient (``async_op=True``: RCCL runs on
    its own stream), so communication overlaps the rest of backward; the mean is ONE ``mul_`` per bucket;
  * ``algo="allreduce"`` (default) hands the bucket to RCCL's all-reduce.  ``algo="mesh"`` is reduce-scatter +
    all-gather written for the xGMI full mesh: an all-to-all sends shard j of every rank DIRECTLY to rank j (all 7
    links of a GPU carry 1/8 of the bucket each, instead of the whole bucket circulating a ring through one link per
    direction: SURVEY.md §5 prices 126 MB at ≈0.4 ms vs ≈2.9 ms), the owner sums its 8 shards in fixed rank order
    (deterministic, unlike a ring whose order depends on the rank's position), and an all-gather returns the sums;
  * gradients may travel as bf16 (halves the per-link bytes; a persistent bf16 staging buffer per bucket) or fp32;
  * accumulation without a kernel per parameter: ``prepare()`` before a backward detaches the ``.grad``s, so autograd
    hands each gradient over as the tensor that produced it, the hook collects them per bucket, and ONE multi-tensor add
    per bucket (``torch._foreach_add_``) folds them into the bucket -- instead of the ~500 in-place adds per micro-batch
    autograd issues into attached views (3.6 ms of a 89 ms step in rocprofv3's stats); ``flush()`` / ``finish()``
    re-attach the views for the optimizer.  A backward run without ``prepare()`` accumulates in place as before;
  * ``no_sync()`` skips the exchange on accumulation micro-batches; ``finish()`` waits for the exchange;
    ``zero_grad()`` clears the buckets with one memset each and keeps the views attached (the Brain loop calls it
    in place of ``optimizer.zero_grad(set_to_none=True)``, which would detach them).
"""
from __future__ import annotations

import contextlib
import os
import time
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class _Bucket:
    def __init__(self, params: List[torch.nn.Parameter], comm_dtype, world: int):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.padded = -(-self.numel // world) * world                     # shards of equal size for the mesh algorithm
        dev = params[0].device
        self.flat = torch.zeros(self.padded, dtype=torch.float32, device=dev)
        self.views = []
        off = 0
        for p in params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.comm_dtype = comm_dtype
        self.stage = None if comm_dtype in (None, torch.float32) else torch.empty(self.padded, dtype=comm_dtype, device=dev)
        self.pending = len(params)
        self.work = None
        self.launched = False
        self.inc_v: List[torch.Tensor] = []                               # views / gradients waiting for the multi-tensor add
        self.inc_g: List[torch.Tensor] = []

    def fold(self):
        """Add the collected gradients into the bucket: one multi-tensor launch for the fp32 ones."""
        if not self.inc_g:
            return
        same = [(v, g) for v, g in zip(self.inc_v, self.inc_g) if g.dtype == v.dtype and g.device == v.device]
        if same:
            torch._foreach_add_([v for v, _ in same], [g for _, g in same])
        for v, g in zip(self.inc_v, self.inc_g):
            if g.dtype != v.dtype or g.device != v.device:
                v.add_(g.to(device=v.device, dtype=v.dtype))
        self.inc_v, self.inc_g = [], []

    def attach(self):
        """(Re-)point every parameter's .grad at its view; a gradient that autograd allocated elsewhere (after a
        zero_grad(set_to_none=True) made by foreign code) is folded into the bucket first."""
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None:
                p.grad = v
            elif g.data_ptr() != v.data_ptr():
                v.copy_(g)
                p.grad = v

    def reset(self):
        self.pending = len(self.params)
        self.work = None
        self.launched = False


class GradAllReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None, bucket_mb: float = 25.0,
                 comm_dtype: Optional[torch.dtype] = None, broadcast_from: Optional[int] = 0, algo: Optional[str] = None,
                 always_exchange: bool = False):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.params = [p for p in params if p.requires_grad]
        self.comm_dtype = comm_dtype
        self.algo = algo or os.environ.get("CM_DDP_ALGO", "allreduce")
        if self.algo not in ("allreduce", "mesh"):
            raise ValueError(f"algo {self.algo!r}: 'allreduce' or 'mesh'")
        self._sync = True
        self._detached = False
        # a single rank has nothing to exchange; ``always_exchange`` runs the collectives anyway (world-1 smoke of the
        # RCCL path on a one-GPU box)
        self.active = self.world > 1 or (always_exchange and dist.is_initialized())
        self.buckets: List[_Bucket] = []
        self.exposed_s = 0.0                       # host time finish() spent waiting for the exchange (last step)
        self.steps = 0
        cap = int(bucket_mb * 1024 * 1024)
        groups, cur, size = [], [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * 4
            if cur and (size + nbytes > cap or p.device != cur[0].device):
                groups.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            groups.append(cur)
        self._owner, self._view = {}, {}
        for g in groups:
            b = _Bucket(g, comm_dtype, self.world)
            b.attach()
            self.buckets.append(b)
            for p, v in zip(g, b.views):
                self._owner[p] = b
                self._view[p] = v
                p.register_post_accumulate_grad_hook(self._hook)
        if broadcast_from is not None and self.world > 1:      # DDP broadcasts parameters when it wraps a module
            for p in self.params:
                dist.broadcast(p.detach(), src=broadcast_from, group=self.group)   # detach(): shares the version counter (ops.cast_cached)

    # ---- autograd side -------------------------------------------------------------------
    def _hook(self, p):
        b, v = self._owner[p], self._view[p]
        g = p.grad
        if g.data_ptr() != v.data_ptr():
            if self._detached:                     # prepare() ran: autograd handed the gradient over, it joins the bucket's
                b.inc_v.append(v)                  # multi-tensor add
                b.inc_g.append(g)
                p.grad = None
            else:                                  # foreign zero_grad(set_to_none=True): fold in and re-attach
                v.copy_(g)
                p.grad = v
        if not self._sync or not self.active:
            return
        b.pending -= 1
        if b.pending == 0:
            b.fold()
            self._launch(b)

    def _launch(self, b: _Bucket):
        b.launched = True
        buf = b.flat
        if b.stage is not None:
            b.stage.copy_(b.flat)
            buf = b.stage
        if self.algo == "allreduce":
            b.work = [dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
            return
        # mesh: all-to-all (shard j of every rank -> rank j), fixed-order local sum, all-gather of the sums
        n = b.padded // self.world
        recv = torch.empty_like(buf)
        w = dist.all_to_all_single(recv, buf, group=self.group, async_op=True)
        b.work = [w]
        b._mesh = (recv, buf, n)

    def _complete(self, b: _Bucket):
        for w in b.work:
            w.wait()
        buf = b.flat if b.stage is None else b.stage
        if self.algo == "mesh":
            recv, buf, n = b._mesh
            shard = recv.view(self.world, n).sum(0, dtype=torch.float32).to(buf.dtype)     # fixed rank order
            dist.all_gather_into_tensor(buf, shard, group=self.group)
            b._mesh = None
        if b.stage is not None:
            b.flat.copy_(b.stage)
        b.flat.mul_(1.0 / self.world)                                                      # the mean, one kernel per bucket

    # ---- training-loop side ----------------------------------------------------------------
    def prepare(self):
        """Call before a backward: detach every ``.grad`` so that autograd passes gradients by reference (no add kernel
        per parameter); they are folded into the buckets by ``flush()`` / ``finish()`` or when a bucket's exchange starts."""
        for p in self.params:
            p.grad = None
        self._detached = True

    def flush(self):
        """Call after the backward of an accumulation micro-batch: fold the collected gradients, re-attach the views."""
        for b in self.buckets:
            b.fold()
            b.attach()
        self._detached = False

    @contextlib.contextmanager
    def no_sync(self):
        """Accumulation micro-batch: gradients stay local (the reference's Brain uses DDP.no_sync the same way)."""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def finish(self):
        """Call after backward of a stepping micro-batch: waits for every bucket; gradients then hold the means."""
        if not self.active:
            self.flush()
            return
        t0 = time.perf_counter()
        for b in self.buckets:
            if not b.launched:                     # parameters of this bucket that produced no gradient this step: their
                b.fold()                           # slices are zero (zero_grad) and travel as zeros
                self._launch(b)
            self._complete(b)
            b.attach()
            b.reset()
        self._detached = False
        if self.buckets and self.buckets[0].flat.is_cuda:
            torch.cuda.current_stream().synchronize()
        self.exposed_s = time.perf_counter() - t0
        self.steps += 1

    def zero_grad(self):
        """Clear the gradients in place (one memset per bucket); the .grad views stay attached."""
        for b in self.buckets:
            b.flat.zero_()
            b.attach()
            b.reset()

    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        """torch.nn.utils.clip_grad_norm_ on the flat buckets (a handful of kernels instead of one per parameter;
        bucket padding is zero).  -> total norm before clipping."""
        total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(b.flat) for b in self.buckets]))
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        for b in self.buckets:
            b.flat.mul_(coef.to(b.flat.device))
        return total

    def bytes_per_step(self) -> int:
        return sum(b.padded * (2 if b.comm_dtype == torch.bfloat16 else 4) for b in self.buckets)
