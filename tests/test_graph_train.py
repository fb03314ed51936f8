"""The training micro-batch as a hipGraph (brain.Brain graph_steps; DESIGN §5 "host-bound training"):
  * the dropout kernels' device-side seed offset (cm_ffn_elem_args.seed_epoch / ops.SEED_EPOCH): with the word at value e every
    kernel behaves bit for bit as an eager launch with seed + e * CM_SEED_EPOCH_MUL, a captured launch draws new decisions per
    replay, and the backward kernels of a replay re-derive the forward's;
  * a graphed Brain trains like the eager one: same losses and parameters (dropout off: every kernel outside the vendor's conv2d
    backward is deterministic), weights re-cast inside the graph after optimizer steps, gradient accumulation, a second batch shape.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ffn_inputs(rows, hidden, seed):
    g = torch.Generator().manual_seed(seed)
    D = 256
    x = torch.randn(rows, D, generator=g).to(DEV)
    lnw, lnb = (1 + 0.1 * torch.randn(D, generator=g)).to(DEV), (0.1 * torch.randn(D, generator=g)).to(DEV)
    w1 = (torch.randn(hidden, D, generator=g) / 16).to(DEV).bfloat16()
    w2 = (torch.randn(D, hidden, generator=g) / (hidden ** 0.5)).to(DEV).bfloat16()
    b1, b2 = (0.1 * torch.randn(hidden, generator=g)).to(DEV), (0.1 * torch.randn(D, generator=g)).to(DEV)
    return x, (lnw, lnb, 1e-5), w1, b1, w2, b2


def test_seed_epoch_is_an_offset_of_the_host_seed():
    from mamba_asr_amd import ops
    rows, hidden, p1, p2, s1, s2 = 200, 1024, 0.1, 0.2, 424242, 171717
    x, ln, w1, b1, w2, b2 = _ffn_inputs(rows, hidden, 3)
    dout = torch.randn(rows, 256, device=DEV)
    a_el = torch.randn(rows, hidden, device=DEV).bfloat16()
    assert ops.SEED_EPOCH is None
    try:
        for epoch in (0, 1, 77):
            ops.SEED_EPOCH = torch.full((1,), epoch, dtype=torch.int64, device=DEV)
            o_dev, (pre, _, _) = ops.ffn_fused(x, ln, w1, b1, w2, b2, alpha=0.5, x_out=torch.empty_like(x), train=(p1, p2, s1, s2))
            bw_dev = ops.ffn_bwd_fused(dout, ops.PackedWeight(w2.t().contiguous()), ops.PackedWeight(w1.t().contiguous()), pre, 0.5, p1, p2, s1, s2)
            y_dev, _ = ops.bias_act_dropout_fwd(a_el, None, act=1, p=p1, seed=s1, store_mask=False)
            da_dev, _ = ops.bias_act_dropout_bwd(a_el, None, p1, a=a_el, act=1, seed=s1)
            ops.SEED_EPOCH = None
            e1, e2 = ops.effective_seed(s1, epoch), ops.effective_seed(s2, epoch)
            o_host, (pre_h, _, _) = ops.ffn_fused(x, ln, w1, b1, w2, b2, alpha=0.5, x_out=torch.empty_like(x), train=(p1, p2, e1, e2))
            bw_host = ops.ffn_bwd_fused(dout, ops.PackedWeight(w2.t().contiguous()), ops.PackedWeight(w1.t().contiguous()), pre_h, 0.5, p1, p2, e1, e2)
            y_host, _ = ops.bias_act_dropout_fwd(a_el, None, act=1, p=p1, seed=e1, store_mask=False)
            da_host, _ = ops.bias_act_dropout_bwd(a_el, None, p1, a=a_el, act=1, seed=e1)
            assert torch.equal(o_dev, o_host) and torch.equal(y_dev, y_host) and torch.equal(da_dev, da_host)
            for u, v in zip(bw_dev, bw_host):
                assert torch.equal(u, v)
            if epoch == 0:
                first = o_dev.clone()
            else:
                assert not torch.equal(o_dev, first)                   # another epoch, other decisions
    finally:
        ops.SEED_EPOCH = None


def test_captured_dropout_draws_fresh_decisions_per_replay():
    from mamba_asr_amd import ops
    rows, hidden, p1, p2, s1, s2 = 256, 1024, 0.1, 0.1, 99, 1234
    x, ln, w1, b1, w2, b2 = _ffn_inputs(rows, hidden, 5)
    w2t, w1t = ops.PackedWeight(w2.t().contiguous()), ops.PackedWeight(w1.t().contiguous())
    dout = torch.randn(rows, 256, device=DEV)
    word = torch.zeros(1, dtype=torch.int64, device=DEV)                     # the graph holds THIS tensor's address
    try:
        ops.SEED_EPOCH = word
        ops.ffn_fused(x, ln, w1, b1, w2, b2, alpha=0.5, x_out=torch.empty_like(x), train=(p1, p2, s1, s2))      # warm (weight packing)
        word.zero_()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            word.add_(1)
            out, (pre, _, _) = ops.ffn_fused(x, ln, w1, b1, w2, b2, alpha=0.5, x_out=torch.empty_like(x), train=(p1, p2, s1, s2))
            da2, da1, act, dh, db1, db2 = ops.ffn_bwd_fused(dout, w2t, w1t, pre, 0.5, p1, p2, s1, s2)
        seen = []
        for r in range(3):
            graph.replay()
            torch.cuda.synchronize()
            epoch = int(word.item())
            assert epoch == r + 1
            snap = (out.clone(), da2.clone(), act.clone())
            ops.SEED_EPOCH = None                                             # the same launches, eagerly, with the offset applied on the host
            e1, e2 = ops.effective_seed(s1, epoch), ops.effective_seed(s2, epoch)
            o_h, (pre_h, _, _) = ops.ffn_fused(x, ln, w1, b1, w2, b2, alpha=0.5, x_out=torch.empty_like(x), train=(p1, p2, e1, e2))
            bw_h = ops.ffn_bwd_fused(dout, w2t, w1t, pre_h, 0.5, p1, p2, e1, e2)
            ops.SEED_EPOCH = word
            assert torch.equal(snap[0], o_h) and torch.equal(snap[1], bw_h[0]) and torch.equal(snap[2], bw_h[2])
            # the backward's masks are the forward's: where the second dropout dropped (out == x), da2 == 0
            dropped = snap[0] == x
            assert 0.05 < float(dropped.float().mean()) < 0.15
            assert not snap[1][dropped].any()
            seen.append(snap[0])
        assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])
    finally:
        ops.SEED_EPOCH = None


def _tiny_brain(graph, dropout, accum=1, seed=0):
    from mamba_asr_amd import sb_compat as sb
    from mamba_asr_amd.asr import ASRConfig, ConMambaASR
    from mamba_asr_amd.brain import Brain, Stage
    torch.manual_seed(seed)
    cfg = ASRConfig(name="tiny", d_model=256, num_encoder_layers=2, d_ffn=1024, output_neurons=31, transformer_dropout=dropout)
    model = ConMambaASR(cfg).to(DEV)

    class ASR(Brain):
        def graph_prologue(self, batch):
            wavs, lens, tokens, tlens = batch
            with torch.no_grad():
                feats = self.modules["asr"].features(wavs, lens, epoch=0)
            return (feats, lens, tokens, tlens)

        def compute_forward(self, batch, stage):
            wavs, lens, tokens, tlens = batch
            return self.modules["asr"].forward_ctc(wavs, lens, epoch=0, feats=wavs if wavs.dim() == 3 else None)

        def compute_objectives(self, pred, batch, stage):
            wavs, lens, tokens, tlens = batch
            return self.modules["asr"].ctc_objective(pred, tokens, lens, tlens)

    brain = ASR({"asr": model}, opt_class=lambda ps: torch.optim.AdamW(ps, lr=2e-4),
                hparams={"precision": "bf16", "grad_accumulation_factor": accum, "max_grad_norm": 5.0},
                run_opts={"device": DEV, "graph_steps": graph})
    brain.on_fit_start()
    brain.modules.train()
    return brain


def _batches():
    from mamba_asr_amd.asr import samples_for_frames, synthetic_wavs
    out = []
    for b, frames, seed in ((4, 400, 1), (2, 240, 2)):
        wavs, lens = synthetic_wavs(b, samples_for_frames(frames), seed, DEV)
        g = torch.Generator().manual_seed(seed)
        tokens = torch.randint(1, 31, (b, frames // 16), generator=g).to(DEV)
        out.append((wavs, lens, tokens, lens.clone()))
    return out


@pytest.mark.parametrize("accum", [1, 2])
def test_graphed_brain_trains_like_the_eager_one(accum):
    from mamba_asr_amd import ops
    batches = _batches()
    order = [0, 0, 0, 1, 0, 1, 1, 0, 0, 0, 1, 1]    # shape 0: eager, capture, replay ...; shape 1 enters later
    try:
        runs = []
        for graph in (False, True):
            brain = _tiny_brain(graph, dropout=0.0, accum=accum)
            losses = []
            for n, i in enumerate(order):
                if n == 5:                          # weights written behind the loop's back (a checkpoint recovery, EMA swap ...)
                    with torch.no_grad():
                        for p in brain.modules.parameters():
                            p.mul_(1.002)
                if n == 6:                          # what load_state_dict's hook does: the caches lose their storage, captured graphs must go
                    ops.invalidate_caches(brain.modules)
                losses.append(float(brain.fit_batch(batches[i])))
            runs.append((losses, [p.detach().clone() for p in brain.modules.parameters()], brain))
        (l_e, p_e, _), (l_g, p_g, bg) = runs
        assert bg.optimizer_step == len(order) // accum
        assert len(bg._graphs) == 2 and all(len(v) >= 1 for v in bg._graphs.values())      # both shapes were captured (again, after the drop)
        if accum == 2:                                                        # with and without the weight-refresh kernels
            assert any(set(v) == {"fresh", "warm"} for v in bg._graphs.values())
        for a, b in zip(l_e, l_g):
            assert abs(a - b) <= 2e-3 * max(1.0, abs(a)), (l_e, l_g)
        assert l_g[-1] < l_g[0]                                               # it trains
        worst = max(float((a - b).abs().max()) for a, b in zip(p_e, p_g))
        assert worst < 5e-4, worst          # AdamW steps of 2e-4: a stale bf16 weight copy or a lost gradient would show as ~1e-3
    finally:
        ops.SEED_EPOCH = None


def test_graphed_brain_with_dropout_runs_and_differs_between_replays():
    from mamba_asr_amd import ops
    batch = _batches()[0]
    try:
        brain = _tiny_brain(True, dropout=0.1)
        brain.modules["asr"].calibrate(batch[0], batch[1])
        brain.modules["asr"].normalize.eval()                                 # fixed normalisation statistics: same features every step
        for g in brain.optimizer.param_groups:
            g["lr"] = 0.0                                                     # frozen weights: the loss varies through dropout alone
            g["weight_decay"] = 0.0
        losses = [float(brain.fit_batch(batch)) for _ in range(5)]
        assert all(l == l for l in losses)
        assert len({round(l, 4) for l in losses[2:]}) == 3, losses            # three replays, three dropout draws
        assert int(ops.SEED_EPOCH.item()) == 4                                # one increment per replay (capture itself runs nothing)
    finally:
        ops.SEED_EPOCH = None


def test_graph_signature_cap_evicts_least_recently_used():
    from mamba_asr_amd import ops
    batches = _batches()
    try:
        brain = _tiny_brain(True, dropout=0.0)
        brain.graph_max_shapes = 1
        for i in (0, 0, 0, 1, 1, 0, 0):
            l1, l2 = float(brain.fit_batch(batches[i])), float(brain.fit_batch(batches[i]))
            assert l1 == l1 and l2 == l2
            assert len(brain._graphs) == 1
        assert brain.optimizer_step == 14
    finally:
        ops.SEED_EPOCH = None


_CHILD_DIST = r'''
import os, sys, json
sys.path.insert(0, os.environ["CM_ROOT"])
sys.path.insert(0, os.path.join(os.environ["CM_ROOT"], "tests"))
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("CM_PORT", "29541"), RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
from test_graph_train import _tiny_brain, _batches
from mamba_asr_amd.ddp import GradAllReducer
batches = _batches()
order = [0, 0, 0, 0, 1, 1, 0, 1]
def run(graph, use_dist):
    brain = _tiny_brain(graph, dropout=0.0, accum=2)
    if use_dist:
        brain.reducer.close()
        params = [p for p in brain.modules.parameters() if p.requires_grad]
        brain.reducer = GradAllReducer(params, always_exchange=True, broadcast_from=None)
        assert brain.reducer.active
    losses = [float(brain.fit_batch(batches[i])) for i in order]
    return losses, [p.detach().clone() for p in brain.modules.parameters()], brain
l0, p0, _ = run(False, False)                              # eager, no process group
dist.init_process_group("nccl", rank=0, world_size=1)
l1, p1, b1 = run(True, True)                               # graphed micro-batches, RCCL exchange after the replay
out = {"losses": max(abs(a - b) / max(1.0, abs(a)) for a, b in zip(l0, l1)),
       "params": max(float((a - b).abs().max()) for a, b in zip(p0, p1)),
       "captured": sum(len(v) for v in b1._graphs.values()), "steps": b1.optimizer_step, "exchanges": b1.reducer.steps}
dist.barrier()
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


def test_graphed_brain_under_world1_rccl_group(tmp_path):
    """graph_steps with an ACTIVE GradAllReducer (world-1 `nccl` group, always_exchange): the captured micro-batch accumulates locally,
    the exchange runs eagerly behind the replay of every stepping micro-batch -- same losses and parameters as the eager
    single-process loop.  In a child process (a process group in the pytest process would outlive the test)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "child_graph_dist.py"
    script.write_text(_CHILD_DIST)
    env = dict(os.environ, CM_ROOT=root, CM_PORT=str(29500 + (os.getpid() + 7) % 400), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])
    print(out)
    assert out["losses"] < 2e-3 and out["params"] < 5e-4, out
    assert out["captured"] >= 3 and out["steps"] == 4 and out["exchanges"] == 4, out


def test_capture_failure_falls_back_to_the_eager_loop():
    """A recipe hook that synchronises with the host (here: .item() inside compute_objectives) cannot be captured: the loop warns, drops
    graph mode and carries on eagerly -- same losses as a loop that was eager from the start."""
    import warnings
    from mamba_asr_amd import ops
    batch = _batches()[0]
    try:
        runs = []
        for graph in (False, True):
            brain = _tiny_brain(graph, dropout=0.0)
            base = brain.compute_objectives

            def with_sync(pred, b, stage, base=base):
                loss = base(pred, b, stage)
                assert loss.item() == loss.item()                  # host synchronisation: illegal while a stream is capturing
                return loss
            brain.compute_objectives = with_sync
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                losses = [float(brain.fit_batch(batch)) for _ in range(4)]
            runs.append((losses, brain, [str(x.message) for x in w]))
        (l_e, _, _), (l_g, bg, msgs) = runs
        assert bg.graph_steps is False and any("capturing the micro-batch failed" in m for m in msgs)
        assert bg.optimizer_step == 4
        for a, b in zip(l_e, l_g):
            assert abs(a - b) <= 2e-3 * max(1.0, abs(a)), (l_e, l_g)
    finally:
        ops.SEED_EPOCH = None
