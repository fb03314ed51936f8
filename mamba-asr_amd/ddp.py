"""Data-parallel gradient exchange for ConMamba training on one MI355X node (one process per GPU).

The reference trains with SpeechBrain's Brain, which wraps modules in torch DistributedDataParallel over NCCL
(train_CTC.py:1062 ddp_init_group; hparams/CTC/conmamba_large.yaml:90 grad_accumulation_factor 4): ONE exchange
per optimizer step, a mean all-reduce of gradients, suppressed (`no_sync`) on non-stepping micro-batches.  This
module is that exchange written directly on torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo"
in the CPU tests):
  * parameters are bucketed (≈25 MB) in REVERSE registration order, i.e. roughly the order gradients appear;
  * a bucket's all-reduce is launched from the autograd hook of its last gradient (async_op=True: RCCL runs on its
    own stream), so communication overlaps the rest of backward;
  * gradients may be reduced in bf16 (halves the per-link bytes on xGMI's point-to-point links) or fp32;
  * `no_sync()` skips the exchange on accumulation micro-batches, `finish()` waits and writes the means back.
"""
from __future__ import annotations

import contextlib
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class _Bucket:
    def __init__(self, params: List[torch.nn.Parameter], comm_dtype):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.comm_dtype = comm_dtype
        self.flat: Optional[torch.Tensor] = None
        self.pending = 0
        self.work = None

    def reset(self):
        self.pending = len(self.params)
        self.work = None


class GradAllReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None, bucket_mb: float = 25.0,
                 comm_dtype: Optional[torch.dtype] = None, broadcast_from: Optional[int] = 0):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in params if p.requires_grad]
        self.comm_dtype = comm_dtype
        self._sync = True
        self.buckets: List[_Bucket] = []
        cap = int(bucket_mb * 1024 * 1024)
        cur, size = [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * (2 if comm_dtype == torch.bfloat16 else 4)
            if cur and size + nbytes > cap:
                self.buckets.append(_Bucket(cur, comm_dtype))
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(_Bucket(cur, comm_dtype))
        self._owner = {}
        for b in self.buckets:
            b.reset()
            for p in b.params:
                self._owner[p] = b
                p.register_post_accumulate_grad_hook(self._hook)
        if broadcast_from is not None and self.world > 1:      # DDP broadcasts parameters when it wraps a module
            for p in self.params:
                dist.broadcast(p.detach(), src=broadcast_from, group=self.group)   # detach(): shares the version counter (ops.cast_cached)

    # ---- autograd side -------------------------------------------------------------------
    def _hook(self, p):
        if not self._sync or self.world == 1:
            return
        b = self._owner[p]
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b: _Bucket):
        dt = b.comm_dtype or b.params[0].grad.dtype
        b.flat = torch.cat([p.grad.detach().reshape(-1).to(dt) for p in b.params])
        b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    # ---- training-loop side ----------------------------------------------------------------
    @contextlib.contextmanager
    def no_sync(self):
        """Accumulation micro-batch: gradients stay local (the reference's Brain uses DDP.no_sync the same way)."""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def finish(self):
        """Call after backward of a stepping micro-batch: waits for every bucket, writes mean gradients back."""
        if self.world == 1:
            return
        for b in self.buckets:
            if b.work is None:                     # params of this bucket that never produced a grad this step
                ready = [p for p in b.params if p.grad is not None]
                if len(ready) != len(b.params):
                    for p in b.params:
                        if p.grad is None:
                            p.grad = torch.zeros_like(p)
                self._launch(b)
            b.work.wait()
            off = 0
            for p in b.params:
                n = p.numel()
                p.grad.copy_(b.flat[off:off + n].view_as(p.grad).to(p.grad.dtype) / self.world)
                off += n
            b.flat = None
            b.reset()

    def bytes_per_step(self) -> int:
        return sum(b.numel * (2 if b.comm_dtype == torch.bfloat16 else 4) for b in self.buckets)
