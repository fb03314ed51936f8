"""Brain-style training loop with the hooks the reference's recipes override (train_CTC.py:164-718 subclasses
speechbrain.core.Brain: compute_forward, compute_objectives, on_stage_start/end, on_fit_batch_end) and the
fit_batch semantics SURVEY.md Appendix A records for speechbrain 1.0.0: bf16 autocast over forward+loss,
(loss / accum).backward(), gradient exchange only on stepping micro-batches, clip_grad_norm_, non-finite skip,
optimizer step, zero_grad(set_to_none=True).  speechbrain itself is not installable here; this is the part of its
surface the ConMamba recipes touch."""
from __future__ import annotations

import enum
from types import SimpleNamespace
from typing import Optional

import os

import torch

from .ddp import GradAllReducer


class Stage(enum.Enum):
    TRAIN = 1
    VALID = 2
    TEST = 3


class Brain:
    def __init__(self, modules=None, opt_class=None, hparams=None, run_opts=None, checkpointer=None):
        run_opts = dict(run_opts or {})
        self.device = torch.device(run_opts.get("device", "cuda" if torch.cuda.is_available() else "cpu"))
        self.precision = run_opts.get("precision", (hparams or {}).get("precision", "fp32"))
        self.grad_accumulation_factor = int(run_opts.get("grad_accumulation_factor",
                                                         (hparams or {}).get("grad_accumulation_factor", 1)))
        self.max_grad_norm = float(run_opts.get("max_grad_norm", (hparams or {}).get("max_grad_norm", 5.0)))
        self.modules = torch.nn.ModuleDict(modules or {}).to(self.device)
        self.hparams = SimpleNamespace(**(hparams or {}))
        self.opt_class = opt_class
        self.checkpointer = checkpointer
        self.step = 0
        self.optimizer_step = 0
        self.avg_train_loss = 0.0
        self.optimizer = None
        self.reducer: Optional[GradAllReducer] = None
        self.distributed = torch.distributed.is_available() and torch.distributed.is_initialized()
        # run_opts["graph_steps"] (or CM_GRAPH_TRAIN=1): forward + loss + backward of a micro-batch replayed as one hipGraph
        # per batch shape (see _fit_batch_graphed)
        self.graph_steps = bool(run_opts.get("graph_steps", os.environ.get("CM_GRAPH_TRAIN", "0") == "1"))
        self._graphs = {}                          # batch signature -> {"fresh" / "warm": captured micro-batch}, least recently used first
        # graphs pay when batch shapes repeat (fixed-length chunks, a bucketing sampler); a recipe whose every batch has its own
        # padded length would capture forever: at most this many signatures stay captured, the least recently used one goes
        self.graph_max_shapes = int(run_opts.get("graph_max_shapes", 16))
        self._graph_pool = None
        self._graph_generation = None              # ops.CACHE_GENERATION when the newest graph was captured
        self._caches_epoch = None                  # _weights_epoch() the in-place weight caches were last refreshed at
        self._epoch_params = None

    # ---- hooks to override ---------------------------------------------------------------
    def compute_forward(self, batch, stage):
        raise NotImplementedError

    def compute_objectives(self, predictions, batch, stage):
        raise NotImplementedError

    def on_stage_start(self, stage, epoch=None):
        pass

    def on_stage_end(self, stage, stage_loss, epoch=None):
        pass

    def on_fit_batch_end(self, batch, outputs, loss, should_step):
        pass

    def on_fit_start(self):
        params = [p for p in self.modules.parameters() if p.requires_grad]
        if self.opt_class is not None and self.optimizer is None:
            self.optimizer = self.opt_class(params)
            # the recipes build torch.optim.AdamW without `fused` (reference hparams/CTC/conmamba_large.yaml:248-252) and that
            # is what runs by default.  CM_FUSED_ADAMW=1 switches GPU parameters to torch's fused multi-tensor kernel (~1 ms
            # instead of ~6 ms of foreach kernels per step for 31.5 M parameters: 94.4 -> 91.1 ms per 32 x 40 s step); its
            # fp32 on-device bias corrections do not reproduce the foreach path bit for bit (loss after 5 warm-up steps
            # 2217.98 vs 2209.33), hence opt-in
            opt = self.optimizer
            if (isinstance(opt, (torch.optim.AdamW, torch.optim.Adam)) and os.environ.get("CM_FUSED_ADAMW", "0") == "1"
                    and params and all(p.is_cuda for p in params) and not opt.state):
                for g in opt.param_groups:
                    if g.get("fused") is None and not g.get("capturable", False) and not g.get("differentiable", False):
                        g["fused"], g["foreach"] = True, False
        # the flat gradient buckets serve a single process too (multi-tensor accumulation, clip and zero_grad on a handful of
        # kernels); CM_FLAT_GRADS=0 keeps autograd's own per-parameter gradients when nothing is exchanged
        if self.reducer is None and params and (self.distributed or os.environ.get("CM_FLAT_GRADS", "1") == "1"):
            self.reducer = GradAllReducer(params)

    # ---- the loop ---------------------------------------------------------------------------
    def _autocast(self):
        if self.precision == "bf16":
            return torch.autocast(self.device.type, dtype=torch.bfloat16)
        if self.precision == "fp16":
            return torch.autocast(self.device.type, dtype=torch.float16)
        return torch.autocast(self.device.type, enabled=False)

    def graph_prologue(self, batch):
        """Graph mode only: the part of a micro-batch that must stay eager (anything whose launches depend on host-side
        random draws or update Python state: feature extraction with running statistics, SpecAugment).  Returns the batch
        compute_forward / compute_objectives receive inside the captured region (tensors of fixed shape per signature)."""
        return batch

    def _micro_batch(self, batch):
        """forward + loss + backward of one accumulation micro-batch with the gradients folded into the flat buckets and no
        exchange: the region a graph captures (the exchange, if any, follows eagerly in fit_batch's tail)."""
        with self.reducer.no_sync():
            with self._autocast():
                outputs = self.compute_forward(batch, Stage.TRAIN)
                loss = self.compute_objectives(outputs, batch, Stage.TRAIN)
            self.reducer.prepare()
            (loss / self.grad_accumulation_factor).backward()
            self.reducer.flush()
        return outputs, loss

    def _weights_epoch(self):
        """Changes whenever a parameter is written in place (optimizer step, load_state_dict, a checkpoint recovery) or its
        storage is swapped: the sum of the parameters' version counters and storage pointers."""
        if self._epoch_params is None:
            self._epoch_params = list(self.modules.parameters())
        return sum(p._version + p.data_ptr() for p in self._epoch_params)

    def _fit_batch_graphed(self, batch):
        """fit_batch with the micro-batch's device work replayed from a hipGraph (a training micro-batch of the small
        configurations is launch-bound: ~2,500 kernels, host enqueue time above the GPU's).  Per batch signature (shapes
        and dtypes after graph_prologue): the first micro-batch runs eagerly (it warms the vendor libraries' lazy state and
        creates the weight caches), later ones copy their tensors into the captured inputs and replay.
        What makes a replay a fresh training step: dropout seeds are offset by a device word the graph's first node
        increments (ops.SEED_EPOCH), torch's own generator is graph-registered by torch.cuda.graph, and the weight-derived
        operands (bf16 copies, packed images, the mixers' derived tensors) live in storage that is refreshed IN PLACE
        (ops.CACHE_INPLACE).  Two variants are captured per signature: "fresh" contains the kernels that refresh those operands
        and is replayed for the first micro-batch after an optimizer step; "warm" reads them as they are (the other micro-batches
        of an accumulation window: 2.5 ms of small kernels fewer per 32 x 40 s micro-batch).
        The gradient exchange (world > 1) is NOT captured: it runs after the replay, un-overlapped, through reducer.finish()."""
        from . import ops
        if self.reducer is None:
            raise RuntimeError("graph_steps needs the flat gradient buckets (CM_FLAT_GRADS=1 or a GradAllReducer)")
        ops.CACHE_INPLACE = True
        if self._graph_generation != ops.CACHE_GENERATION and any(self._graphs.values()):
            # a weight cache got new storage since the last capture (ops.invalidate_caches after a load_state_dict, a cache created by
            # another code path): the captured launches may read addresses the caches no longer own -- start over
            self._graphs = {}
            self._graph_pool = None                                        # the pool dies with its last graph
            self._graph_drops = getattr(self, "_graph_drops", 0) + 1
            if self._graph_drops == 4:
                import warnings
                warnings.warn("graph_steps: the captured micro-batches were dropped four times because a weight cache got new storage; a "
                              "cache that allocates per step defeats graph mode (CM_CACHE_TRACE=1 prints who allocates)")
        should_step = (self.step + 1) % self.grad_accumulation_factor == 0
        pro = self.graph_prologue(batch)
        flat = list(pro) if isinstance(pro, (tuple, list)) else [pro]
        key = tuple((tuple(t.shape), t.dtype) if torch.is_tensor(t) else ("py", t) for t in flat)
        if key in self._graphs:
            self._graphs[key] = self._graphs.pop(key)                      # most recently used: last
        if key not in self._graphs:                                        # first sight of this shape: eager
            while len(self._graphs) >= max(self.graph_max_shapes, 1):
                self._graphs.pop(next(iter(self._graphs)))
            if not any(self._graphs.values()):
                self._graph_pool = None                                    # the pool dies with its last graph
            self._graphs[key] = {}
            outputs, loss = self._micro_batch(pro)
            self._caches_epoch = self._weights_epoch()                     # eager lookups refreshed what was stale
        else:
            variant = "warm" if self._caches_epoch == self._weights_epoch() else "fresh"
            g = self._graphs[key].get(variant)
            if g is None:
                if ops.SEED_EPOCH is None:
                    ops.SEED_EPOCH = torch.zeros(1, dtype=torch.int64, device=self.device)
                static = [t.clone() if torch.is_tensor(t) else t for t in flat]
                sbatch = type(pro)(static) if isinstance(pro, (tuple, list)) else static[0]
                before = set(self.reducer._touched)
                graph = torch.cuda.CUDAGraph()
                try:
                    with (ops.forced_refresh() if variant == "fresh" else _null()) as log:
                        # thread_local: a process group's watchdog thread polls its events with hipEventQuery, which a capture in the
                        # default "global" mode turns into an error in THAT thread (and the process dies with it)
                        with torch.cuda.graph(graph, pool=self._graph_pool, capture_error_mode="thread_local"):
                            ops.SEED_EPOCH.add_(1)
                            outputs, loss = self._micro_batch(sbatch)
                            loss = loss.detach()
                except Exception as exc:                                   # noqa: BLE001 -- whatever the recipe's hooks do that a capture forbids
                    # (a .item(), a host-side length computation, an allocation outside torch): this loop runs eagerly from here on
                    import warnings
                    warnings.warn(f"graph_steps: capturing the micro-batch failed ({type(exc).__name__}: {str(exc)[:300]}); "
                                  "continuing eagerly (keep host synchronisation out of compute_forward / compute_objectives, or move "
                                  "it into graph_prologue)")
                    torch.cuda.synchronize()
                    self.reducer.discard_pending()
                    self.reducer._touched = before
                    ops.invalidate_caches(self.modules)                    # entries created inside the failed capture point into its pool
                    self.graph_steps, self._graphs, self._graph_pool = False, {}, None
                    return self.fit_batch(batch)
                if self._graph_pool is None:
                    self._graph_pool = graph.pool()
                self._graph_generation = ops.CACHE_GENERATION              # entries created inside the capture are this graph's own
                g = self._graphs[key][variant] = SimpleNamespace(graph=graph, static=static, outputs=outputs, loss=loss,
                                                                  touched=set(self.reducer._touched),
                                                                  entries=list(log) if variant == "fresh" else [])
                self.reducer._touched = before                           # capture ran no kernel: nothing was touched yet
            else:
                for dst, src in zip(g.static, flat):
                    if torch.is_tensor(dst):
                        dst.copy_(src)
            g.graph.replay()
            if variant == "fresh":
                ops.rekey_caches(g.entries)                                # the replay refreshed exactly these entries
                self._caches_epoch = self._weights_epoch()
            self.reducer._touched |= g.touched
            outputs, loss = g.outputs, g.loss.clone()
        self._step_tail(should_step, loss)
        self.step += 1
        self.on_fit_batch_end(batch, outputs, loss, should_step)
        return loss.detach()

    def _step_tail(self, should_step, loss):
        if not should_step:
            return
        self.reducer.finish()
        norm = self.reducer.clip_grad_norm_(self.max_grad_norm)
        if torch.isfinite(norm) and torch.isfinite(loss):
            self.reducer.hide_unused()
            self.optimizer.step()
            self.optimizer_step += 1
        self.reducer.zero_grad()

    def fit_batch(self, batch):
        if self.graph_steps:
            return self._fit_batch_graphed(batch)
        should_step = (self.step + 1) % self.grad_accumulation_factor == 0
        sync = self.reducer.no_sync() if (self.reducer is not None and not should_step) else _null()
        with sync:
            with self._autocast():
                outputs = self.compute_forward(batch, Stage.TRAIN)
                loss = self.compute_objectives(outputs, batch, Stage.TRAIN)
            if self.reducer is not None:
                self.reducer.prepare()                   # gradients arrive by reference, one multi-tensor add per bucket
            (loss / self.grad_accumulation_factor).backward()
            if self.reducer is not None and not should_step:
                self.reducer.flush()
        if should_step:
            if self.reducer is not None:
                self.reducer.finish()
                norm = self.reducer.clip_grad_norm_(self.max_grad_norm)       # on the flat gradient buckets
            else:
                params = [p for p in self.modules.parameters() if p.grad is not None]
                norm = torch.nn.utils.clip_grad_norm_(params, self.max_grad_norm)
            if torch.isfinite(norm) and torch.isfinite(loss):     # the one host wait of a step (speechbrain's check_gradients does the same)
                if self.reducer is not None:
                    self.reducer.hide_unused()           # parameters without a gradient this step: .grad None, as after set_to_none
                self.optimizer.step()
                self.optimizer_step += 1
            if self.reducer is not None:
                self.reducer.zero_grad()                 # memset of the buckets; the .grad views stay attached
            else:
                self.optimizer.zero_grad(set_to_none=True)
        self.step += 1
        self.on_fit_batch_end(batch, outputs, loss, should_step)
        return loss.detach()

    def evaluate_batch(self, batch, stage):
        with torch.no_grad(), self._autocast():
            out = self.compute_forward(batch, stage)
            return self.compute_objectives(out, batch, stage).detach()

    def fit(self, epoch_counter, train_set, valid_set=None):
        self.on_fit_start()
        for epoch in epoch_counter:
            self.on_stage_start(Stage.TRAIN, epoch)
            self.modules.train()
            total, n = 0.0, 0
            for batch in train_set:
                total += float(self.fit_batch(batch))
                n += 1
            self.avg_train_loss = total / max(n, 1)
            self.on_stage_end(Stage.TRAIN, self.avg_train_loss, epoch)
            if valid_set is not None:
                self.on_stage_start(Stage.VALID, epoch)
                self.modules.eval()
                vt, vn = 0.0, 0
                for batch in valid_set:
                    vt += float(self.evaluate_batch(batch, Stage.VALID))
                    vn += 1
                self.on_stage_end(Stage.VALID, vt / max(vn, 1), epoch)

    def evaluate(self, test_set, **unused):
        self.on_stage_start(Stage.TEST, None)
        self.modules.eval()
        total, n = 0.0, 0
        for batch in test_set:
            total += float(self.evaluate_batch(batch, Stage.TEST))
            n += 1
        self.on_stage_end(Stage.TEST, total / max(n, 1), None)
        return total / max(n, 1)


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
