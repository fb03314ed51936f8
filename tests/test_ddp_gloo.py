"""World-size-2 gloo test (CPU) of the data-parallel gradient exchange (mamba_asr_amd.ddp): bucketed async
all-reduce launched from autograd hooks == mean of the two ranks' gradients; no_sync leaves gradients local and
accumulates; the Brain loop steps once per grad_accumulation_factor micro-batches."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    torch.manual_seed(7)
    return torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.GELU(), torch.nn.Linear(64, 64), torch.nn.GELU(),
                               torch.nn.Linear(64, 5))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mamba_asr_amd.ddp import GradAllReducer
    from mamba_asr_amd.brain import Brain
    model = _model()
    if rank == 1:                                   # start from different weights: the reducer must broadcast rank 0's
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    red = GradAllReducer(model.parameters(), bucket_mb=0.004)      # several buckets
    assert len(red.buckets) > 2
    ref = _model()
    for p, r in zip(model.parameters(), ref.parameters()):
        assert torch.equal(p, r)
    g = torch.Generator().manual_seed(100 + rank)
    x, y = torch.randn(8, 16, generator=g), torch.randn(8, 5, generator=g)
    # accumulation micro-batch: local only
    with red.no_sync():
        ((model(x) - y) ** 2).mean().backward()
    local = [p.grad.clone() for p in model.parameters()]
    # stepping micro-batch: exchanged
    ((model(x) - y) ** 2).mean().backward()
    red.finish()
    # expected: mean over ranks of (2 * local grad)
    both = []
    for r in range(world):
        gr = torch.Generator().manual_seed(100 + r)
        xr, yr = torch.randn(8, 16, generator=gr), torch.randn(8, 5, generator=gr)
        m = _model()
        (2 * ((m(xr) - yr) ** 2).mean()).backward()
        both.append([p.grad for p in m.parameters()])
    ok = all(torch.allclose(p.grad, (a + b) / 2, atol=1e-6) for p, a, b in zip(model.parameters(), *both))
    ok = ok and all(torch.allclose(l * 2, a, atol=1e-6) for l, a in zip(local, both[rank]))

    # Brain loop: 4 micro-batches, accumulation 2 -> 2 optimizer steps, identical weights on both ranks afterwards
    class B(Brain):
        def compute_forward(self, batch, stage):
            return self.modules["net"](batch[0])

        def compute_objectives(self, pred, batch, stage):
            return ((pred - batch[1]) ** 2).mean()

    brain = B({"net": _model()}, opt_class=lambda ps: torch.optim.SGD(ps, lr=0.1),
              hparams={"grad_accumulation_factor": 2, "max_grad_norm": 1e30}, run_opts={"device": "cpu"})
    data = [(torch.randn(4, 16, generator=g), torch.randn(4, 5, generator=g)) for _ in range(4)]
    brain.fit(range(1), data)
    flat = torch.cat([p.detach().reshape(-1) for p in brain.modules.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    ok = ok and brain.optimizer_step == 2 and torch.allclose(gathered[0], gathered[1], atol=1e-6)
    # gradients live in the persistent buckets across steps: .grad is still the bucket view after zero_grad
    red2 = brain.reducer
    ok = ok and all(p.grad is not None and p.grad.data_ptr() == red2._view[p].data_ptr() for p in red2.params)
    ok = ok and all(float(b.flat.abs().max()) == 0.0 for b in red2.buckets)

    # the mesh algorithm (all-to-all + fixed-order sum + all-gather) and bf16 transport give the same means
    for algo, cdt, tol in (("mesh", None, 1e-6), ("allreduce", torch.bfloat16, 2e-2), ("mesh", torch.bfloat16, 2e-2)):
        m = _model()
        try:
            r2 = GradAllReducer(m.parameters(), bucket_mb=0.004, algo=algo, comm_dtype=cdt)
            (2 * ((m(x) - y) ** 2).mean()).backward()
            r2.finish()
        except RuntimeError as e:                   # gloo builds without all_to_all: the algebra is still checked below
            if "all_to_all" in str(e) or "alltoall" in str(e).lower():
                continue
            raise
        ok = ok and all(torch.allclose(p.grad, (a + b) / 2, atol=tol, rtol=tol) for p, a, b in zip(m.parameters(), *both))

    # Brain.fit through the duration-bucket sampler with an ODD number of batches: every rank must run the same number of
    # micro-batches or the last all-reduce has no peer
    from mamba_asr_amd.dataio import DurationBucketBatchSampler
    dur = [1.0 + 0.1 * i for i in range(33)]
    smp = DurationBucketBatchSampler(dur, max_batch_length=30.0, num_buckets=4, seed=5, rank=rank, world=world)
    n_common = len(DurationBucketBatchSampler(dur, max_batch_length=30.0, num_buckets=4, seed=5))
    gd = torch.Generator().manual_seed(9)
    pool = [(torch.randn(16, generator=gd), torch.randn(5, generator=gd)) for _ in dur]
    batches = [(torch.stack([pool[i][0] for i in b]), torch.stack([pool[i][1] for i in b])) for b in smp]
    brain2 = B({"net": _model()}, opt_class=lambda ps: torch.optim.SGD(ps, lr=0.05),
               hparams={"grad_accumulation_factor": 1}, run_opts={"device": "cpu"})
    brain2.fit(range(1), batches)
    flat = torch.cat([p.detach().reshape(-1) for p in brain2.modules.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    ok = ok and n_common % world == 1 and brain2.optimizer_step == (n_common + 1) // world
    ok = ok and torch.allclose(gathered[0], gathered[1], atol=1e-6)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_grad_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_flat_buckets_accumulate_by_reference_single_process():
    """GradAllReducer.prepare() / flush() / finish() without a process group: gradients handed over by reference and folded
    with one multi-tensor add per bucket equal autograd's own in-place accumulation bit for bit over several micro-batches
    (same order of additions), the views are re-attached for the optimizer, a parameter that got no gradient in a
    micro-batch keeps its sum, and a backward run WITHOUT prepare() still accumulates (in place, as before)."""
    from mamba_asr_amd.ddp import GradAllReducer
    torch.manual_seed(3)

    def make():
        torch.manual_seed(5)
        return torch.nn.Sequential(torch.nn.Linear(12, 20), torch.nn.GELU(), torch.nn.Linear(20, 7), torch.nn.Linear(7, 3))

    ref, net = make(), make()
    extra_ref, extra = torch.nn.Parameter(torch.ones(4)), torch.nn.Parameter(torch.ones(4))      # used in one micro-batch only
    red = GradAllReducer(list(net.parameters()) + [extra], bucket_mb=0.001)                      # several small buckets
    assert len(red.buckets) > 1 and not red.active
    xs = [torch.randn(5, 12) for _ in range(4)]
    for i, x in enumerate(xs):
        loss_r = ref(x).square().mean() + (extra_ref.sum() * 0.5 if i == 1 else 0.0)
        loss_r.backward()
        loss = net(x).square().mean() + (extra.sum() * 0.5 if i == 1 else 0.0)
        if i != 2:
            red.prepare()
        loss.backward()                                          # micro-batch 2: no prepare(), in-place accumulation
        if i < 3:
            red.flush()
        else:
            red.finish()
        for p in list(net.parameters()) + [extra]:
            assert p.grad is not None and p.grad.data_ptr() == red._view[p].data_ptr()
    for a, b in zip(ref.parameters(), net.parameters()):
        assert torch.equal(a.grad, b.grad)
    assert torch.equal(extra_ref.grad, extra.grad)
    total = red.clip_grad_norm_(1e9)
    want = torch.linalg.vector_norm(torch.stack([p.grad.norm() for p in list(ref.parameters()) + [extra_ref]]))
    torch.testing.assert_close(total, want)
    red.zero_grad()
    assert all(float(p.grad.abs().sum()) == 0.0 for p in net.parameters())


# ------------------------------------------------------------------------------------------------------------------
# SURVEY §8d config 4's check on the real model structure: gradients of a ConMamba CTC step computed by ONE rank on the
# global batch == what TWO ranks hold after the exchange, each on its half (gloo, CPU, fp64).  The mixer's operator has no
# CPU implementation in the product (GPU only, by design), so this TEST injects the oracle's differentiable restatement
# as `mamba_inner_fn_no_out_proj`; everything else -- ConmambaEncoder modules, CTC objective, Brain.fit_batch with
# accumulation, GradAllReducer's buckets / hooks / mesh and all-reduce algorithms -- is the product code.
# ------------------------------------------------------------------------------------------------------------------
def _oracle_inner(xz, conv_w, conv_b, xw, dtw, A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None,
                  delta_softplus=True, *, reverse_time=False):
    from oracle import conmamba_oracle as O
    x = xz.flip(-1) if reverse_time else xz
    import functools
    scan = functools.partial(O.selective_scan, work_dtype=xz.dtype)              # the scan's own default is fp32
    y = O.mamba_inner_no_out_proj(x, conv_w, conv_b, xw, dtw, A, D, delta_bias, scan=scan, work_dtype=xz.dtype)
    return y.flip(-1) if reverse_time else y


def _conmamba_ctc(seed=11):
    import torch.nn as nn
    from mamba_asr_amd.modules import Conmamba as CM
    from mamba_asr_amd.modules.mamba import bimamba
    bimamba.mamba_inner_fn_no_out_proj = _oracle_inner
    torch.manual_seed(seed)
    enc = CM.ConmambaEncoder(num_layers=2, d_model=32, d_ffn=64, kernel_size=7, activation=nn.GELU, bias=True, dropout=0.0,
                             causal=False, mamba_config={"d_state": 16, "expand": 2, "d_conv": 4, "bidirectional": True})
    head = nn.Linear(32, 11)
    return nn.ModuleDict({"enc": enc, "head": head}).double()


def _ctc_batches(n_utt=8, frames=24):
    g = torch.Generator().manual_seed(21)
    x = torch.randn(n_utt, frames, 32, generator=g, dtype=torch.float64)
    tok = torch.randint(1, 11, (n_utt, 5), generator=g)
    return x, tok


def _layer_worker(rank, world, port, q):
    try:
        _layer_worker_body(rank, world, port, q)
    except BaseException as e:                      # surface the failure instead of a queue timeout
        q.put((rank, {"error": repr(e)}))
        raise


def _layer_worker_body(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mamba_asr_amd.brain import Brain
    from mamba_asr_amd.ddp import GradAllReducer
    from mamba_asr_amd import sb_compat as sb

    class B(Brain):
        def compute_forward(self, batch, stage):
            out, _ = self.modules["net"]["enc"](batch[0])
            return torch.log_softmax(self.modules["net"]["head"](out), -1)

        def compute_objectives(self, pred, batch, stage):
            ones = torch.ones(pred.shape[0], dtype=torch.float64)
            return sb.ctc_loss(pred, batch[1], ones, ones, 0, reduction="batchmean")

    x, tok = _ctc_batches()
    res = {}
    for algo in ("allreduce", "mesh"):
        # two micro-batches of 2 utterances per rank, accumulation 2: rank r sees utterances [4k + 2r, 4k + 2r + 2)
        brain = B({"net": _conmamba_ctc()}, opt_class=lambda ps: torch.optim.SGD(ps, lr=0.0),
                  hparams={"grad_accumulation_factor": 2, "max_grad_norm": 1e30}, run_opts={"device": "cpu"})
        params = [p for p in brain.modules.parameters()]
        try:
            brain.reducer = GradAllReducer(params, bucket_mb=0.02, algo=algo)
            brain.on_fit_start()
            brain.modules.train()
            kept = {}
            brain.reducer.zero_grad = lambda: kept.update({i: p.grad.clone() for i, p in enumerate(params)})   # keep the step's gradients
            for k in range(2):
                lo = 4 * k + 2 * rank
                brain.fit_batch((x[lo:lo + 2], tok[lo:lo + 2]))
        except RuntimeError as e:
            if "all_to_all" in str(e) or "alltoall" in str(e).lower():
                res[algo] = "skipped"
                continue
            raise
        # one rank, the global batch: two micro-batches of 4 utterances, same accumulation
        ref = B({"net": _conmamba_ctc()}, opt_class=lambda ps: torch.optim.SGD(ps, lr=0.0),
                hparams={"grad_accumulation_factor": 2, "max_grad_norm": 1e30}, run_opts={"device": "cpu"})
        ref.distributed = False
        os.environ["CM_FLAT_GRADS"] = "0"
        ref.on_fit_start()
        ref.modules.train()
        rp = [p for p in ref.modules.parameters()]
        for k in range(2):
            with torch.autocast("cpu", enabled=False):
                loss = ref.compute_objectives(ref.compute_forward((x[4 * k:4 * k + 4], tok[4 * k:4 * k + 4]), None), (None, tok[4 * k:4 * k + 4]), None)
            (loss / 2).backward()
        worst = max(float((kept[i] - p.grad).abs().max() / p.grad.abs().max().clamp_min(1e-30)) for i, p in enumerate(rp))
        res[algo] = worst
    q.put((rank, res))
    dist.destroy_process_group()


def test_conmamba_layer_one_rank_equals_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_layer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for rank in (0, 1):
        for algo, worst in res[rank].items():
            # fp64 model: the two sums differ by rounding only -- fp32 rounding for A / D / dt_bias, which the mixer hands to
            # its operator as .float() exactly like the reference (bimamba.py:200, 231-233): their gradients pass an fp32 node
            assert worst == "skipped" or (not isinstance(worst, str) and worst < 1e-6), (rank, algo, worst)
    assert res[0]["allreduce"] != "skipped"


def test_hide_unused_matches_set_to_none_semantics():
    """A parameter that took no part in a step keeps `.grad is None` for the optimizer (reference loop: zero_grad(set_to_none=True)):
    AdamW must not decay it.  ADVICE r2."""
    from mamba_asr_amd.ddp import GradAllReducer
    torch.manual_seed(1)
    used, unused = torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(5))
    before = unused.detach().clone()
    red = GradAllReducer([used, unused])
    opt = torch.optim.AdamW([used, unused], lr=0.1, weight_decay=0.5)
    red.prepare()
    (used * 2).sum().backward()
    red.finish()
    assert red.hide_unused() == 1 and unused.grad is None and used.grad is not None
    opt.step()
    red.zero_grad()
    assert torch.equal(unused, before)                    # untouched: no weight decay, no moments
    assert unused.grad is not None and float(unused.grad.abs().sum()) == 0.0     # re-attached for the next step
    # a stale foreign gradient does not leak into the cleared bucket (zero_grad drops it)
    used.grad = torch.ones(5)
    red.zero_grad()
    assert float(used.grad.abs().sum()) == 0.0 and used.grad.data_ptr() == red._view[used].data_ptr()
