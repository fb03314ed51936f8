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
  * ``no_sync()`` skips the exchange on accumulation micro-batches; ``zero_grad()`` clears the buckets with one
    memset each and keeps the views attached (the Brain loop calls it in place of
    ``optimizer.zero_grad(set_to_none=True)``, which would detach them);
  * round 3: a bucket's WHOLE exchange (bf16 staging copy, collective(s), the mesh algorithm's fixed-order sum and
    all-gather, copy back, the mean) is queued from the hook on a communication stream behind the producing kernels,
    into persistent staging / receive / shard buffers, and ends in an event; ``finish()`` makes the training stream
    wait for those events -- no host synchronisation anywhere in the exchange (on gloo / CPU tensors the collectives
    are synchronous calls and the same code runs in line);
  * parameters that produced NO gradient since the last ``zero_grad()`` are tracked: ``hide_unused()`` detaches their
    (all-zero) ``.grad`` before the optimizer step, so AdamW skips them exactly as it does after the reference loop's
    ``zero_grad(set_to_none=True)`` (no weight decay / moment update on a parameter that was not used).
"""
from __future__ import annotations

import contextlib
import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class _Bucket:
    def __init__(self, params: List[torch.nn.Parameter], comm_dtype, world: int, algo: str):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.padded = -(-self.numel // world) * world                     # shards of equal size for the mesh algorithm
        dev = params[0].device
        self.flat = torch.zeros(self.padded, dtype=params[0].dtype, device=dev)       # fp32 for every recipe (fp64 models in tests)
        self.views = []
        off = 0
        for p in params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.comm_dtype = comm_dtype
        low = comm_dtype not in (None, params[0].dtype)
        self.stage = torch.empty(self.padded, dtype=comm_dtype, device=dev) if low else None
        # mesh algorithm: persistent receive buffer (world shards) and this rank's summed shard
        tdt = comm_dtype if low else params[0].dtype
        self.recv = torch.empty(self.padded, dtype=tdt, device=dev) if algo == "mesh" else None
        self.shard = torch.empty(self.padded // world, dtype=tdt, device=dev) if algo == "mesh" else None
        self.event = torch.cuda.Event() if dev.type == "cuda" else None   # recorded when the exchange's last kernel is queued
        self.pending = len(params)
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

    def attach(self, keep_foreign: bool = True):
        """(Re-)point every parameter's .grad at its view.  A gradient that autograd allocated elsewhere (after a
        zero_grad(set_to_none=True) made by foreign code) is folded into the bucket first, or dropped (zero_grad)."""
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None:
                p.grad = v
            elif g.data_ptr() != v.data_ptr():
                if keep_foreign:
                    v.copy_(g)
                p.grad = v

    def reset(self):
        self.pending = len(self.params)
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
        self.steps = 0
        self._touched = set()                      # parameters that produced a gradient since the last zero_grad()
        self._wait_events = None                   # (start, end) CUDA events around finish()'s waits, for exposed_s
        cap = int(bucket_mb * 1024 * 1024)
        groups, cur, size = [], [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * 4
            if cur and (size + nbytes > cap or p.device != cur[0].device or p.dtype != cur[0].dtype):
                groups.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            groups.append(cur)
        self._owner, self._view, self._handles = {}, {}, []
        cuda = bool(self.params) and self.params[0].is_cuda
        # the exchange's kernels (staging casts, the mesh sum, the mean) run on this stream, next to backward
        self.comm_stream = torch.cuda.Stream(device=self.params[0].device) if (cuda and self.active) else None
        for g in groups:
            b = _Bucket(g, comm_dtype, self.world, self.algo)
            b.attach()
            self.buckets.append(b)
            for p, v in zip(g, b.views):
                self._owner[p] = b
                self._view[p] = v
                self._handles.append(p.register_post_accumulate_grad_hook(self._hook))
        if broadcast_from is not None and self.world > 1:      # DDP broadcasts parameters when it wraps a module
            for p in self.params:
                dist.broadcast(p.detach(), src=broadcast_from, group=self.group)   # detach(): shares the version counter (ops.cast_cached)

    def close(self):
        """Remove the autograd hooks (the parameters keep their bucket views as .grad until something replaces them)."""
        for h in self._handles:
            h.remove()
        self._handles = []

    # ---- autograd side -------------------------------------------------------------------
    def _hook(self, p):
        b, v = self._owner[p], self._view[p]
        self._touched.add(p)
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
        """Queue the bucket's whole exchange.  GPU: on the communication stream, ordered behind the kernels that produced
        the bucket; every `wait()` below is a stream dependency (ProcessGroupNCCL), not a host wait.  CPU (gloo): in line."""
        b.launched = True
        cs = self.comm_stream
        if cs is not None:
            cs.wait_stream(torch.cuda.current_stream(b.flat.device))
        with (torch.cuda.stream(cs) if cs is not None else contextlib.nullcontext()):
            buf = b.flat
            if b.stage is not None:
                b.stage.copy_(b.flat)
                buf = b.stage
            if self.algo == "allreduce":
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True).wait()
            else:
                # mesh: all-to-all (shard j of every rank -> rank j), fixed-order local sum, all-gather of the sums
                n = b.padded // self.world
                dist.all_to_all_single(b.recv, buf, group=self.group, async_op=True).wait()
                if b.shard.dtype in (torch.float32, torch.float64):
                    torch.sum(b.recv.view(self.world, n), 0, out=b.shard)                 # fixed rank order
                else:
                    b.shard.copy_(b.recv.view(self.world, n).sum(0, dtype=torch.float32))
                dist.all_gather_into_tensor(buf, b.shard, group=self.group, async_op=True).wait()
            if b.stage is not None:
                b.flat.copy_(b.stage)
            if self.world > 1:
                b.flat.mul_(1.0 / self.world)                                              # the mean, one kernel per bucket
            if b.event is not None:
                b.event.record(cs if cs is not None else torch.cuda.current_stream(b.flat.device))

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

    def discard_pending(self):
        """Forget gradients collected since prepare() without folding them in (a backward that did not happen: an aborted hipGraph
        capture) and re-attach the bucket views."""
        for b in self.buckets:
            b.inc_v, b.inc_g = [], []
            b.attach(keep_foreign=False)
            b.reset()
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
        """Call after backward of a stepping micro-batch: gradients hold the means for everything queued on the current
        stream after this call.  No host synchronisation: the current stream waits for each bucket's event."""
        if not self.active:
            self.flush()
            return
        cuda = self.comm_stream is not None
        cur = torch.cuda.current_stream(self.buckets[0].flat.device) if cuda else None
        if cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
        for b in self.buckets:
            if not b.launched:                     # parameters of this bucket that produced no gradient this step: their
                b.fold()                           # slices are zero (zero_grad) and travel as zeros
                self._launch(b)
            if cuda:
                cur.wait_event(b.event)
            b.attach()
            b.reset()
        if cuda:
            e1.record(cur)
            self._wait_events = (e0, e1)
        self._detached = False
        self.steps += 1

    @property
    def exposed_s(self) -> float:
        """Time the training stream spent waiting for the exchange in the last finish() (GPU time between two events on
        that stream; synchronises on the second one -- a measurement helper for bench.py, not part of the step)."""
        if self._wait_events is None:
            return 0.0
        e0, e1 = self._wait_events
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e-3

    def hide_unused(self) -> int:
        """Before optimizer.step(): detach the .grad of every parameter that produced no gradient since the last zero_grad()
        (it would be an all-zero view, which AdamW treats as a gradient: weight decay and moment decay on a parameter the
        step did not use; the reference loop's zero_grad(set_to_none=True) leaves None there and the optimizer skips it).
        zero_grad() re-attaches them.  -> number of hidden parameters."""
        n = 0
        for p in self.params:
            if p not in self._touched:
                p.grad = None
                n += 1
        return n

    def zero_grad(self):
        """Clear the gradients in place (one memset per bucket); the .grad views stay attached (a foreign .grad tensor is
        dropped, not folded in: it belongs to the step being cleared)."""
        for b in self.buckets:
            b.attach(keep_foreign=False)
            b.flat.zero_()
            b.reset()
        self._touched.clear()

    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        """torch.nn.utils.clip_grad_norm_ on the flat buckets (a handful of kernels instead of one per parameter;
        bucket padding is zero).  -> total norm before clipping."""
        total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(b.flat) for b in self.buckets]))
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        for b in self.buckets:
            b.flat.mul_(coef.to(b.flat.device))
        return total

    def bytes_per_step(self) -> int:
        return sum(b.padded * (b.stage if b.stage is not None else b.flat).element_size() for b in self.buckets)
