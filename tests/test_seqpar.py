"""Time-split (sequence-parallel) ConMamba forward, mamba_asr_amd.seqpar (the reference has no
sequence parallelism, so the reference for every check here is the UNSPLIT operator / encoder on the same inputs).

CPU (world-2 gloo, two processes): the exchange algebra -- chunk summaries (P, h_end), carry folding, conv halos --
with the oracle plugged in as the local operator backend.
GPU: cm_selective_scan_fwd's h0 (a sequence scanned in two pieces == scanned whole, both directions), and the whole
2-layer encoder cut into 2 / 4 time shards through the HIP kernels (shards run as threads of one process)."""
import os
import sys

import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = {"d_state": 16, "expand": 2, "d_conv": 4, "bidirectional": True}


class OracleBackend:
    """The CPU oracle as the local operators (test infrastructure only)."""

    @staticmethod
    def scan(u, delta, A, B, C, D, z, delta_bias, reverse, h0):
        from oracle import conmamba_oracle as O
        f = (lambda t: t.flip(-1)) if reverse else (lambda t: t)
        dt = O.softplus(delta.float() + delta_bias.float()[None, :, None])
        y, h_end = O.selective_scan(f(u), f(delta), A, f(B), f(C), D, f(z), delta_bias, True, return_last_state=True,
                                    initial_state=h0)
        P = torch.exp(A.float()[None] * dt.sum(-1)[:, :, None])           # prod_t exp(dt_t A) over the shard
        if h0 is not None:                                                # summaries are defined from a zero start
            h_end = None
        return f(y), P, h_end

    @staticmethod
    def causal_conv(x, weight, bias, reverse):
        from oracle import conmamba_oracle as O
        if reverse:
            return O.causal_conv1d(x.flip(-1), weight, bias, True).flip(-1)
        return O.causal_conv1d(x, weight, bias, True)

    @staticmethod
    def dwconv_rows(x, weight, bias, pad_left):
        k = weight.shape[-1]
        xt = torch.nn.functional.pad(x.transpose(1, 2), (pad_left, k - 1 - pad_left))
        return torch.nn.functional.conv1d(xt, weight.reshape(x.shape[2], 1, k), bias, groups=x.shape[2]).transpose(1, 2)


def _encoder(d_model=32, d_ffn=64, layers=2, seed=3):
    from mamba_asr_amd.modules.Conmamba import ConmambaEncoder
    torch.manual_seed(seed)
    enc = ConmambaEncoder(num_layers=layers, d_model=d_model, d_ffn=d_ffn, kernel_size=31, activation=nn.GELU, bias=True,
                          dropout=0.0, causal=False, mamba_config=dict(CFG))
    for p in enc.parameters():
        if p.dim() > 1:
            nn.init.xavier_normal_(p)
    return enc.eval()


def _unsplit_cpu(enc, x):
    """The unsplit encoder through the oracle (same state_dict keys as the reference)."""
    from oracle import conmamba_oracle as O
    p = {k: v.detach() for k, v in enc.state_dict().items()}
    return O.encoder(p, x, len(enc.layers))


def _worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mamba_asr_amd import seqpar
    enc = _encoder()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 96, 32, generator=g)                               # the whole sequence, the same on every rank
    T = x.shape[1] // world
    grp = seqpar.DistGroup()
    got = seqpar.encoder_forward_seq_parallel(enc, x[:, rank * T:(rank + 1) * T].contiguous(), grp, backend=OracleBackend)
    want = _unsplit_cpu(enc, x)[:, rank * T:(rank + 1) * T]
    err = (got - want).abs().max().item()
    q.put((rank, err))
    dist.destroy_process_group()


def test_time_split_encoder_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert [r for r, _ in res] == [0, 1]
    assert all(e < 2e-4 for _, e in res), res


def test_carry_folding_matches_sequential_scan():
    """carry_in folds the shards' affine maps in scan order: 4 shards, both directions, against one long scan."""
    from mamba_asr_amd import seqpar
    from oracle import conmamba_oracle as O
    g = torch.Generator().manual_seed(5)
    b, e, l, n, W = 2, 6, 64, 16, 4
    u, delta = torch.randn(b, e, l, generator=g), torch.randn(b, e, l, generator=g) * 0.5
    A = -torch.exp(torch.randn(e, n, generator=g) * 0.3)
    B, C = torch.randn(b, n, l, generator=g), torch.randn(b, n, l, generator=g)
    D, z, bias = torch.randn(e, generator=g), torch.randn(b, e, l, generator=g), torch.randn(e, generator=g) - 1
    T = l // W
    for reverse in (False, True):
        f = (lambda t: t.flip(-1)) if reverse else (lambda t: t)
        whole = f(O.selective_scan(f(u), f(delta), A, f(B), f(C), D, f(z), bias, True))
        sl = lambda t, r: t[..., r * T:(r + 1) * T].contiguous()
        summ = [OracleBackend.scan(sl(u, r), sl(delta, r), A, sl(B, r), sl(C, r), D, sl(z, r), bias, reverse, None) for r in range(W)]
        for r in range(W):
            H = seqpar.carry_in([s[1] for s in summ], [s[2] for s in summ], r, reverse)
            y = summ[r][0] if H is None else OracleBackend.scan(sl(u, r), sl(delta, r), A, sl(B, r), sl(C, r), D, sl(z, r), bias,
                                                                reverse, H)[0]
            torch.testing.assert_close(y, sl(whole, r), rtol=1e-4, atol=1e-5)
    assert seqpar.exchange_bytes_per_layer(1, 256) == 2 * 2 * 512 * 16 * 4 + 2 * 512 * 3 * 4 + 256 * 30 * 4


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("reverse", [False, True])
def test_scan_initial_state_continues_a_sequence(dtype, reverse):
    """cm_selective_scan_fwd with h0: scanning [first piece], then [second piece from the first's last state], equals
    scanning the whole sequence -- bit for bit (the recurrence visits the same values in the same order); lengths
    chosen so that the cut is NOT on a 64-step checkpoint boundary of the whole sequence."""
    from mamba_asr_amd import ops
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(3)
    b, e, l, n, cut = 3, 96, 328, 16, 200
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    u, delta, z = rnd(b, e, l).to(dtype), (rnd(b, e, l) * 0.5).to(dtype), rnd(b, e, l).to(dtype)
    A = -torch.exp(rnd(e, n) * 0.3)
    B, C = rnd(b, n, l), rnd(b, n, l)
    D, bias = rnd(e), rnd(e) - 1
    _, x_all, whole = ops.selective_scan_fwd(u, delta, A, B, C, D, z, bias, True, reverse=reverse, need_out=False)
    first, second = (slice(cut, l), slice(0, cut)) if reverse else (slice(0, cut), slice(cut, l))     # in scan order
    cs = lambda t, s: t[..., s].contiguous()
    _, x1, y1 = ops.selective_scan_fwd(cs(u, first), cs(delta, first), A, cs(B, first), cs(C, first), D, cs(z, first), bias, True,
                                       reverse=reverse, need_out=False)
    h1 = x1[:, :, 0 if reverse else -1, 1::2].contiguous()
    _, x2, y2 = ops.selective_scan_fwd(cs(u, second), cs(delta, second), A, cs(B, second), cs(C, second), D, cs(z, second), bias,
                                       True, reverse=reverse, need_out=False, h0=h1)
    assert torch.equal(y1, whole[..., first]) and torch.equal(y2, whole[..., second])
    # the last state of the continued piece is the whole sequence's last state
    torch.testing.assert_close(x2[:, :, 0 if reverse else -1, 1::2], x_all[:, :, 0 if reverse else -1, 1::2], rtol=1e-6, atol=1e-7)
    # the backward kernel refuses an initial state (forward-only feature)
    with pytest.raises(RuntimeError):
        N = ops.N
        a = N.ScanBwdArgs()
        a.fwd.h0 = h1.data_ptr()
        N.check(N.lib().cm_selective_scan_bwd(a), "cm_selective_scan_bwd")


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_time_split_encoder_on_hip_kernels(world):
    """2-layer ConMamba encoder (d_model 64) on a 256-step sequence cut into 2 / 4 time shards, every shard through the
    HIP kernels + the exchange code (shards = threads of this process), against the unsplit module forward (fp32)."""
    from mamba_asr_amd import seqpar
    enc = _encoder(d_model=64, d_ffn=128, seed=4).to("cuda")
    x = torch.randn(3, 256, 64, generator=torch.Generator().manual_seed(8)).to("cuda")
    with torch.no_grad():
        want, _ = enc(x)
    T = x.shape[1] // world
    got = seqpar.run_local(world, lambda grp: seqpar.encoder_forward_seq_parallel(
        enc, x[:, grp.rank * T:(grp.rank + 1) * T].contiguous(), grp))
    torch.testing.assert_close(torch.cat(got, dim=1), want, rtol=2e-3, atol=2e-4)
    # and against the CPU oracle on the whole sequence
    ref = _unsplit_cpu(enc.cpu(), x.cpu())
    torch.testing.assert_close(torch.cat(got, dim=1).cpu(), ref, rtol=2e-3, atol=5e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("reverse", [False, True])
def test_row_group_scan_carry_interface(reverse):
    """cm_scan_cl_fwd (row-group kernel) h0 / h_last / decay: a sequence scanned in two pieces, the second from the first's
    h_last, equals the whole; decay = exp(A * sum delta'); h_last of the continued piece = h_last of the whole."""
    from mamba_asr_amd import ops
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(4)
    b, l, e, cut = 3, 205, 64, 120
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    u, z = rnd(b, l, e).bfloat16(), rnd(b, l, e).bfloat16()
    xdbl = (rnd(b, l, 48) * 0.5).bfloat16()
    A = -torch.exp(rnd(e, 16) * 0.3)
    Wdt = ops.pad_dt_weight(rnd(e, 16) * 0.25)
    D, bias = torch.ones(e, device=dev), rnd(e) - 2
    common = dict(A=A, D=D, delta_bias=bias, dt_weight=Wdt, reverse=reverse)
    st = lambda: torch.empty(b, e, 16, device=dev)
    hw, dw = st(), st()
    (whole,) = ops.scan_cl_fwd([dict(common, u=u, xdbl=xdbl, h_last=hw, decay=dw)], z=z)
    first, second = (slice(cut, l), slice(0, cut)) if reverse else (slice(0, cut), slice(cut, l))
    cs = lambda t, s: t[:, s].contiguous()
    h1, d1, h2, d2 = st(), st(), st(), st()
    (y1,) = ops.scan_cl_fwd([dict(common, u=cs(u, first), xdbl=cs(xdbl, first), h_last=h1, decay=d1)], z=cs(z, first))
    (y2,) = ops.scan_cl_fwd([dict(common, u=cs(u, second), xdbl=cs(xdbl, second), h0=h1, h_last=h2, decay=d2)], z=cs(z, second))
    assert torch.equal(y1, whole[:, first]) and torch.equal(y2, whole[:, second])
    torch.testing.assert_close(h2, hw, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(d1 * d2, dw, rtol=2e-5, atol=1e-30)
    # decay against its definition, delta' from the bf16 dt columns and the bf16-rounded weight as the kernel forms it
    dt = xdbl[:, first, :16].float() @ Wdt.bfloat16().float().t() + bias
    want = torch.exp(A[None] * torch.nn.functional.softplus(dt).sum(1)[:, :, None])
    torch.testing.assert_close(d1, want, rtol=2e-3, atol=1e-30)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_time_split_fused_encoder_vs_reference_golden(world, golden):
    """The fused bf16 route (the kernels bench.py times) cut into 2 / 4 time shards, against the REFERENCE's 2-layer
    d_model-256 encoder (golden g4_large: 16 x 100 x 256) and against the unsplit fused route."""
    import importlib.util
    from mamba_asr_amd import fused, seqpar
    from mamba_asr_amd.modules.Conmamba import ConmambaEncoder
    spec = importlib.util.spec_from_file_location("golden_synth", os.path.join(os.path.dirname(__file__), "golden", "synth.py"))
    S = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(S)
    enc = ConmambaEncoder(num_layers=2, d_model=256, d_ffn=1024, kernel_size=31, activation=nn.GELU, bias=True, dropout=0.0,
                          causal=False, mamba_config=dict(CFG))
    enc.load_state_dict(S.synth_like(enc, 256), strict=True)
    enc = enc.to("cuda").eval()
    x = S.synth_input("g4_large.x", (16, 100, 256), 256).to("cuda")
    T = x.shape[1] // world
    got = torch.cat(seqpar.run_local(world, lambda grp: seqpar.encoder_forward_seq_parallel_fused(
        enc, x[:, grp.rank * T:(grp.rank + 1) * T].contiguous(), grp)), dim=1)
    torch.testing.assert_close(got.cpu(), golden("g4_large")["y_enc"], rtol=3e-2, atol=5e-2)
    with torch.no_grad():
        unsplit = fused.encoder_forward(enc, x, dtype=torch.bfloat16, streams=1)
    err = (got - unsplit).abs()
    print(f"time-split fused (W={world}) vs unsplit fused: max|diff| {err.max():.3e} mean {err.mean():.3e}")
    assert err.mean() < 2e-3 and err.max() < 6e-2          # same kernels; shard edges change bf16 rounding order only
