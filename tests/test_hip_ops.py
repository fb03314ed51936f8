"""GPU parity tests of the native ops, called through the C ABI (mamba_asr_amd.ops), against
 (a) the golden vectors produced by the reference's own Python and (b) the CPU oracle on seeded
inputs.  Tolerances: fp32 I/O rtol 2e-4 / atol 5e-5 on forward results (v_exp_f32 / v_log_f32 are
~1 ulp, the recurrence accumulates over seqlen); bf16 I/O within 2 bf16 ulps (rtol 1.6e-2)."""
import pytest
import torch

from oracle import conmamba_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda"


def close(a, b, rtol, atol):
    torch.testing.assert_close(a.detach().double().cpu(), b.detach().double().cpu(), rtol=rtol, atol=atol)


def gpu(*ts):
    return [None if t is None else t.to(DEV) for t in ts]


@pytest.fixture(scope="module")
def ops():
    from mamba_asr_amd import ops as _ops
    return _ops


@pytest.mark.parametrize("tag", ["tiny", "mid", "long", "n8"])
@pytest.mark.parametrize("split", [0, 1, 2, 4, 8, 16])
def test_scan_fwd_golden(ops, golden, tag, split):
    g = golden("g1_scan_fwd")
    u, dl, A, B, C, D, z, bias = gpu(*[g[f"{tag}_{k}"] for k in ("u", "delta", "A", "B", "C", "D", "z", "bias")])
    if split > A.shape[1]:
        pytest.skip("split larger than dstate")
    out, x, out_z = ops.selective_scan_fwd(u, dl, A, B, C, D, z, bias, True, split=split)      # 0 = automatic choice
    close(out_z, g[f"{tag}_out_full"], 2e-4, 5e-5)
    close(x[:, :, -1, 1::2], g[f"{tag}_last_full"], 2e-4, 5e-5)       # ssi.py:45 contract
    close(out * torch.nn.functional.silu(z), g[f"{tag}_out_full"], 2e-4, 5e-5)


@pytest.mark.parametrize("tag", ["tiny", "mid"])
def test_scan_fwd_variants(ops, golden, tag):
    g = golden("g1_scan_fwd")
    u, dl, A, B, C, D, z, bias = gpu(*[g[f"{tag}_{k}"] for k in ("u", "delta", "A", "B", "C", "D", "z", "bias")])
    t = (2e-4, 5e-5)
    close(ops.selective_scan_fwd(u, dl, A, B, C, D, None, bias, True)[0], g[f"{tag}_out_noz"], *t)
    close(ops.selective_scan_fwd(u, dl, A, B, C, None, z, bias, True)[2], g[f"{tag}_out_noD"], *t)
    close(ops.selective_scan_fwd(u, dl, A, B, C, D, z, None, True)[2], g[f"{tag}_out_nobias"], *t)
    close(ops.selective_scan_fwd(u, dl.abs() * 0.1, A, B, C, D, z, None, False)[2], g[f"{tag}_out_nosoftplus"], *t)
    close(ops.selective_scan_fwd(u, dl, A, B, C, None, None, None, True)[0], g[f"{tag}_out_bare"], *t)
    close(ops.selective_scan_fwd(u, dl, A, B[:, None], C[:, None], D, z, bias, True)[2], g[f"{tag}_out_4d"], *t)
    # reverse_time flag == flip inputs, scan, flip output (reference bimamba.py:237,253)
    close(ops.selective_scan_fwd(u, dl, A, B, C, D, z, bias, True, reverse=True)[2], g[f"{tag}_out_rev"], *t)
    bf = lambda v: v.to(torch.bfloat16)
    ob = ops.selective_scan_fwd(bf(u), bf(dl), A, bf(B), bf(C), D, bf(z), bias, True)[2]
    assert ob.dtype == torch.bfloat16
    close(ob.float(), g[f"{tag}_out_bf16"], 1.6e-2, 1e-2)
    # bf16 activations with fp32 B/C (the fused path keeps x_dbl in fp32)
    ob2 = ops.selective_scan_fwd(bf(u), bf(dl), A, bf(B).float(), bf(C).float(), D, bf(z), bias, True)[2]
    close(ob2.float(), g[f"{tag}_out_bf16"], 1.6e-2, 1e-2)


@pytest.mark.parametrize("shape", [(1, 8, 1, 16), (3, 70, 129, 16), (2, 64, 64, 16), (1, 288, 251, 16), (2, 16, 40, 8)])
@pytest.mark.parametrize("reverse", [False, True])
def test_scan_fwd_vs_oracle_ragged(ops, shape, reverse):
    """odd sizes: channel counts that do not fill a workgroup, seqlen not a multiple of the vector
    width (unaligned element-wise path), single step."""
    b, e, l, n = shape
    gen = torch.Generator().manual_seed(1234 + l)
    u = torch.randn(b, e, l, generator=gen)
    dl = torch.randn(b, e, l, generator=gen) * 0.5
    A = -torch.exp(torch.randn(e, n, generator=gen) * 0.3)
    B, C = torch.randn(b, n, l, generator=gen), torch.randn(b, n, l, generator=gen)
    D, z, bias = torch.randn(e, generator=gen), torch.randn(b, e, l, generator=gen), torch.randn(e, generator=gen) - 1
    f = (lambda t: t.flip(-1)) if reverse else (lambda t: t)
    ref, last = O.selective_scan(f(u), f(dl), A, f(B), f(C), D, f(z), bias, True, True, work_dtype=torch.float64)
    ref = f(ref)
    out, x, out_z = ops.selective_scan_fwd(*gpu(u, dl, A, B, C, D, z, bias), True, reverse=reverse)
    close(out_z, ref, 2e-4, 5e-5)
    close(x[:, :, 0 if reverse else -1, 1::2], last, 2e-4, 5e-5)


def test_scan_fwd_strided_xz_views(ops):
    """u/z as the two halves of one xz tensor (reference ssi.py:180) — dim/batch strides honoured."""
    gen = torch.Generator().manual_seed(7)
    b, e, l, n = 2, 32, 96, 16
    xz = torch.randn(b, 2 * e, l, generator=gen).to(DEV)
    u, z = xz.chunk(2, dim=1)
    dl = (torch.randn(b, e, l, generator=gen) * 0.5).to(DEV)
    A = -torch.rand(e, n, generator=gen).to(DEV) - 0.1
    B, C = torch.randn(b, n, l, generator=gen).to(DEV), torch.randn(b, n, l, generator=gen).to(DEV)
    got = ops.selective_scan_fwd(u, dl, A, B, C, None, z, None, True)[2]
    ref = O.selective_scan(u.cpu(), dl.cpu(), A.cpu(), B.cpu(), C.cpu(), None, z.cpu(), None, True, work_dtype=torch.float64)
    close(got, ref, 2e-4, 5e-5)


@pytest.mark.parametrize("tag", ["tiny", "mid", "long"])
def test_scan_bwd_golden(ops, golden, tag):
    g = golden("g2_scan_bwd")
    u, dl, A, B, C, D, z, bias, dout = gpu(*[g[f"{tag}_{k}"] for k in ("u", "delta", "A", "B", "C", "D", "z", "bias", "dout")])
    _, x, out_z = ops.selective_scan_fwd(u, dl, A, B, C, D, z, bias, True, need_out=False)
    du, dd, dA, dB, dC, dD, dbias, dz, oz = ops.selective_scan_bwd(u, dl, A, B, C, D, z, bias, dout, x, True,
                                                                   recompute_out_z=True)
    close(oz, g[f"{tag}_out"], 2e-4, 5e-5)
    for got, key in ((du, "du"), (dd, "ddelta"), (dA, "dA"), (dB[:, 0], "dB"), (dC[:, 0], "dC"), (dD, "dD"),
                     (dz, "dz"), (dbias, "dbias")):
        ref = g[f"{tag}_{key}"]
        close(got, ref, 3e-3, 3e-4 * max(ref.abs().max().item(), 1.0))
    if tag == "tiny":
        _, x, _ = ops.selective_scan_fwd(u, dl, A, B, C, None, None, None, True)
        r = ops.selective_scan_bwd(u, dl, A, B, C, None, None, None, dout, x, True)
        for got, key in ((r[0], "du"), (r[1], "ddelta"), (r[2], "dA"), (r[3][:, 0], "dB"), (r[4][:, 0], "dC")):
            ref = g[f"tiny_bare_{key}"]
            close(got, ref, 3e-3, 3e-4 * max(ref.abs().max().item(), 1.0))
        assert r[5] is None and r[6] is None and r[7] is None


@pytest.mark.parametrize("shape", [(2, 70, 129, 16), (1, 8, 3, 16), (2, 16, 200, 8)])
@pytest.mark.parametrize("reverse", [False, True])
def test_scan_bwd_vs_oracle(ops, shape, reverse):
    b, e, l, n = shape
    gen = torch.Generator().manual_seed(99 + l)
    u = torch.randn(b, e, l, generator=gen)
    dl = torch.randn(b, e, l, generator=gen) * 0.5
    A = -torch.exp(torch.randn(e, n, generator=gen) * 0.3)
    B, C = torch.randn(b, n, l, generator=gen), torch.randn(b, n, l, generator=gen)
    D, z, bias = torch.randn(e, generator=gen), torch.randn(b, e, l, generator=gen), torch.randn(e, generator=gen) - 1
    dout = torch.randn(b, e, l, generator=gen)
    f = (lambda t: t.flip(-1)) if reverse else (lambda t: t)
    r = O.selective_scan_bwd(f(u), f(dl), A, f(B), f(C), D, f(z), bias, f(dout), True)
    gu, gdl, gA, gB, gC, gD, gz, gbias, gdout = gpu(u, dl, A, B, C, D, z, bias, dout)
    _, x, _ = ops.selective_scan_fwd(gu, gdl, gA, gB, gC, gD, gz, gbias, True, reverse=reverse, need_out=False)
    du, dd, dA, dB, dC, dD, dbias, dz, _ = ops.selective_scan_bwd(gu, gdl, gA, gB, gC, gD, gz, gbias, gdout, x, True,
                                                                  reverse=reverse)
    for got, ref in ((du, f(r["du"])), (dd, f(r["ddelta"])), (dA, r["dA"]), (dB[:, 0], f(r["dB"])),
                     (dC[:, 0], f(r["dC"])), (dD, r["dD"]), (dz, f(r["dz"])), (dbias, r["ddelta_bias"])):
        close(got, ref, 2e-3, 2e-4 * max(ref.abs().max().item(), 1.0))


@pytest.mark.parametrize("tag", ["tiny", "mid", "w3", "short"])
def test_conv_golden(ops, golden, tag):
    g = golden("k_conv")
    x, w, b, dout = gpu(g[f"{tag}_x"], g[f"{tag}_w"], g[f"{tag}_b"], g[f"{tag}_dout"])
    close(ops.causal_conv1d_fwd(x, w, b, True), g[f"{tag}_y"], 1e-5, 1e-5)
    close(ops.causal_conv1d_fwd(x, w, None, False), g[f"{tag}_y_lin"], 1e-5, 1e-5)
    dx, dw, db = ops.causal_conv1d_bwd(x, w, b, dout, True)
    close(dx, g[f"{tag}_dx"], 1e-4, 1e-5)
    close(dw, g[f"{tag}_dw"], 1e-4, 1e-4)
    close(db, g[f"{tag}_db"], 1e-4, 1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("l", [64, 251])
def test_conv_reverse_and_strided(ops, dtype, l):
    gen = torch.Generator().manual_seed(5)
    xz = torch.randn(2, 96, l, generator=gen).to(dtype).to(DEV)
    x = xz[:, :48]
    w, b = torch.randn(48, 4, generator=gen).to(DEV), torch.randn(48, generator=gen).to(DEV)
    ref = O.causal_conv1d(x.cpu().float().flip(-1), w.cpu(), b.cpu(), True, work_dtype=torch.float64).flip(-1)
    got = ops.causal_conv1d_fwd(x, w, b, True, reverse=True)
    tol = (1e-5, 1e-5) if dtype == torch.float32 else (1.6e-2, 1e-2)
    close(got.float(), ref, *tol)
    dy = torch.randn(2, 48, l, generator=gen).to(dtype).to(DEV)
    dxz = torch.zeros_like(xz)
    dx, dw, db = ops.causal_conv1d_bwd(x, w, b, dy, True, reverse=True, dx=dxz[:, :48])
    rdx, rdw, rdb = O.causal_conv1d_bwd(x.cpu().float().flip(-1), w.cpu(), b.cpu(), dy.cpu().float().flip(-1), True)
    tol = (1e-4, 1e-4) if dtype == torch.float32 else (2e-2, 2e-2)
    close(dxz[:, :48].float(), rdx.flip(-1), *tol)
    close(dw, rdw, tol[0], tol[1] * max(1.0, rdw.abs().max().item()))
    close(db, rdb, tol[0], tol[1] * max(1.0, rdb.abs().max().item()))
    assert float(dxz[:, 48:].abs().max()) == 0.0        # untouched half


@pytest.mark.parametrize("shape", [(2, 37, 128), (1, 1000, 288), (3, 64, 64), (2, 5, 70)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("split", [4, 8, 16])
def test_scan_channels_last_two_directions(ops, shape, dtype, split):
    """cm_scan_cl_fwd: both BiMamba directions in one launch, channels-last, z / out as column slices."""
    from mamba_asr_amd import _native
    b, l, e = shape
    gen = torch.Generator().manual_seed(l * 7 + e)
    xz = torch.randn(b, l, 2 * e, generator=gen).to(dtype)
    z = xz[:, :, e:]
    dirs, refs = [], []
    ycat = torch.zeros(b, l, 2 * e, dtype=dtype, device=DEV)
    for i, rev in enumerate((False, True)):
        u = torch.randn(b, l, e, generator=gen).to(dtype)
        dl = (torch.randn(b, l, e, generator=gen) * 0.5).to(dtype)
        A = -torch.exp(torch.randn(e, 16, generator=gen) * 0.3)
        Bm, Cm = torch.randn(16, b, l, generator=gen), torch.randn(16, b, l, generator=gen)
        D, bias = torch.randn(e, generator=gen), torch.randn(e, generator=gen) - 1
        f = (lambda t: t.flip(-1)) if rev else (lambda t: t)
        tr = lambda t: t.float().transpose(1, 2)           # (b, l, e) -> (b, e, l)
        ref = O.selective_scan(f(tr(u)), f(tr(dl)), A, f(Bm.permute(1, 0, 2)), f(Cm.permute(1, 0, 2)), D, f(tr(z)),
                               bias, True, work_dtype=torch.float64)
        refs.append(f(ref).transpose(1, 2))
        gB, gC = ops.alloc_bc(16, b, l, DEV), ops.alloc_bc(16, b, l, DEV)
        gB.copy_(Bm), gC.copy_(Cm)
        dirs.append(dict(u=u.to(DEV), delta=dl.to(DEV), A=A.to(DEV), B=gB, C=gC, D=D.to(DEV), delta_bias=bias.to(DEV),
                         out=ycat[:, :, i * e:(i + 1) * e], reverse=rev))
    ops.scan_cl_fwd(dirs, z=xz.to(DEV)[:, :, e:], delta_softplus=True, split=split)
    tol = (2e-4, 5e-5) if dtype == torch.float32 else (1.6e-2, 1e-2)
    close(ycat[:, :, :e].float(), refs[0], *tol)
    close(ycat[:, :, e:].float(), refs[1], *tol)


@pytest.mark.parametrize("rank", [9, 16])
def test_scan_channels_last_in_kernel_dt_proj(ops, rank):
    """delta formed inside cm_scan_cl_fwd from (dt_low, dt_weight) == scan of the explicit delta tensor."""
    b, l, e = 2, 77, 128
    gen = torch.Generator().manual_seed(rank)
    u = torch.randn(b, l, e, generator=gen)
    z = torch.randn(b, l, e, generator=gen)
    feat = ops.alloc_bc(rank + 32, b, l, DEV)
    feat.copy_(torch.randn(rank + 32, b, l, generator=gen))
    Wdt = torch.randn(e, rank, generator=gen) * 0.3
    A = -torch.exp(torch.randn(e, 16, generator=gen) * 0.3)
    D, bias = torch.randn(e, generator=gen), torch.randn(e, generator=gen) - 1
    delta = torch.einsum("er,rbl->ble", Wdt, feat[:rank].cpu())
    common = dict(u=u.to(DEV), A=A.to(DEV), B=feat[rank:rank + 16], C=feat[rank + 16:], D=D.to(DEV), delta_bias=bias.to(DEV))
    (ref,) = ops.scan_cl_fwd([dict(common, delta=delta.to(DEV).contiguous())], z=z.to(DEV))
    (got,) = ops.scan_cl_fwd([dict(common, dt_low=feat[:rank], dt_weight=Wdt.to(DEV))], z=z.to(DEV))
    close(got, ref, 1e-4, 1e-5)
    want = O.selective_scan(u.transpose(1, 2), delta.transpose(1, 2), A, feat[rank:rank + 16].cpu().permute(1, 0, 2),
                            feat[rank + 16:].cpu().permute(1, 0, 2), D, z.transpose(1, 2), bias, True, work_dtype=torch.float64)
    close(got, want.transpose(1, 2), 2e-4, 5e-5)


@pytest.mark.parametrize("shape", [(2, 37, 128), (1, 1000, 288), (3, 64, 64), (2, 5, 72), (1, 16, 512), (2, 333, 512)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rank", [9, 16, 24, 32])
@pytest.mark.parametrize("lanes", [4, 8])
def test_scan_rows_two_directions(ops, shape, dtype, rank, lanes):
    """cm_scan_cl_fwd in xdbl mode (row-group kernel, csrc/scan_rows_fwd.hip): x_proj output rows [dt16 | B | C]
    (dt_rank <= 16) or [dt32 | B | C] (dt_rank <= 32, bf16: the S2S-large encoder's rank) read as written, delta
    formed in-kernel on the matrix pipe, both directions in one launch — against the fp64 oracle on the same
    (dtype-rounded) inputs.  Ragged sizes: seqlen not a multiple of 16, dim not a multiple of 64."""
    b, l, e = shape
    P = 16 if rank <= 16 else 32                                        # width the dt features are zero-padded to
    RW = P + 32
    if lanes == 8 and rank > 16:
        pytest.skip("the 2-states-per-lane kernel (scan_rows_fwd2.hip) is built for dt_rank <= 16")
    if P == 32 and dtype == torch.float32:
        u = torch.zeros(1, 16, 64, device=DEV)
        with pytest.raises(RuntimeError, match="64 wide"):
            ops.scan_cl_fwd([dict(u=u, A=-torch.ones(64, 16, device=DEV), dt_weight=torch.zeros(64, 32, device=DEV),
                                  xdbl=torch.zeros(1, 16, 64, device=DEV))])
        return
    gen = torch.Generator().manual_seed(l * 11 + e + rank)
    xz = torch.randn(b, l, 2 * e, generator=gen).to(dtype)
    z = xz[:, :, e:]
    dirs, refs = [], []
    ycat = torch.zeros(b, l, 2 * e, dtype=dtype, device=DEV)
    xcat = torch.zeros(b, l, 2 * RW, dtype=dtype)
    for i, rev in enumerate((False, True)):
        u = torch.randn(b, l, e, generator=gen).to(dtype)
        A = -torch.exp(torch.randn(e, 16, generator=gen) * 0.3)
        xd = torch.randn(b, l, RW, generator=gen)
        xd[:, :, rank:P] = 0.0
        xcat[:, :, RW * i:RW * (i + 1)] = xd.to(dtype)
        xd = xcat[:, :, RW * i:RW * (i + 1)].float()                     # what the kernel sees
        Wdt = torch.randn(e, rank, generator=gen) * 0.3
        D, bias = torch.randn(e, generator=gen), torch.randn(e, generator=gen) - 1
        # bf16 I/O: the kernel runs the dt_proj product as a bf16 MFMA (weight rounded to bf16, fp32 accumulate), which is
        # what the reference does under autocast (selective_scan_interface.py:187); the oracle sees the same rounded weight
        Wq = Wdt.to(dtype).float()
        delta = torch.einsum("er,blr->bel", Wq.double(), xd[:, :, :rank].double())         # (b, e, l), pre-bias
        Bm, Cm = xd[:, :, P:P + 16].transpose(1, 2), xd[:, :, P + 16:].transpose(1, 2)        # (b, 16, l)
        f = (lambda t: t.flip(-1)) if rev else (lambda t: t)
        tr = lambda t: t.float().transpose(1, 2)
        ref = O.selective_scan(f(tr(u)), f(delta), A, f(Bm), f(Cm), D, f(tr(z)), bias, True, work_dtype=torch.float64)
        refs.append(f(ref).transpose(1, 2))
        dirs.append(dict(u=u.to(DEV), A=A.to(DEV), D=D.to(DEV), delta_bias=bias.to(DEV), dt_weight=ops.pad_dt_weight(Wdt.to(DEV)),
                         out=ycat[:, :, i * e:(i + 1) * e], reverse=rev))
    gx = xcat.to(DEV)
    for i in range(2):
        dirs[i]["xdbl"] = gx[:, :, RW * i:RW * (i + 1)]
    # lanes 4: 4 states per lane (scan_rows_fwd.hip); lanes 8: 2 states per lane (scan_rows_fwd2.hip, what small launches get)
    ops.scan_cl_fwd(dirs, z=xz.to(DEV)[:, :, e:], delta_softplus=True, split=lanes, time_chunks=1)
    tol = (2e-4, 5e-5) if dtype == torch.float32 else (1.6e-2, 1e-2)
    close(ycat[:, :, :e].float(), refs[0], *tol)
    close(ycat[:, :, e:].float(), refs[1], *tol)
    if l > 64 or lanes == 8:
        return                      # raw (un-softplused) random time steps grow the state without bound on long inputs
    # single direction, no z, no D, no bias, no softplus
    (got,) = ops.scan_cl_fwd([dict(u=dirs[0]["u"], A=dirs[0]["A"], dt_weight=dirs[0]["dt_weight"], xdbl=dirs[0]["xdbl"])],
                             delta_softplus=False)
    xd = xcat[:, :, :RW].float()
    Wp = dirs[0]["dt_weight"].cpu().to(dtype).float()
    delta = torch.einsum("er,blr->bel", Wp.double(), xd[:, :, :P].double())
    ref = O.selective_scan(dirs[0]["u"].cpu().float().transpose(1, 2), delta, dirs[0]["A"].cpu(), xd[:, :, P:P + 16].transpose(1, 2),
                           xd[:, :, P + 16:].transpose(1, 2), None, None, None, False, work_dtype=torch.float64)
    close(got.float(), ref.transpose(1, 2), *((2e-3, 1e-4) if dtype == torch.float32 else (1.6e-2, 2e-2)))   # growing states: looser rtol


@pytest.mark.parametrize("shape,chunks", [((2, 1000, 128), 4), ((1, 333, 72), 3), ((3, 205, 64), 8), ((1, 4000, 64), 16),
                                          ((2, 50, 64), 7), ((2, 517, 136), 2)])
@pytest.mark.parametrize("dtype,rank", [(torch.float32, 16), (torch.bfloat16, 9), (torch.bfloat16, 32)])
def test_scan_rows_time_chunks(ops, shape, chunks, dtype, rank):
    """cm_scan_cl_fwd with time_chunks > 1 (summary pass + carry fold + output pass, csrc/scan_rows_fwd.hip) against the
    unchunked launch and the fp64 oracle, both directions, ragged last chunk, chunk counts that exceed the blocks there
    are; then with a caller h0 and the whole-sequence h_last / decay outputs."""
    from mamba_asr_amd import _native
    b, l, e = shape
    P = 16 if rank <= 16 else 32
    RW = P + 32
    gen = torch.Generator().manual_seed(l + 3 * e + chunks)
    z = torch.randn(b, l, e, generator=gen).to(dtype).to(DEV)
    dirs, refs = [], []
    for i, rev in enumerate((False, True)):
        u = torch.randn(b, l, e, generator=gen).to(dtype)
        A = -torch.exp(torch.randn(e, 16, generator=gen) * 0.3)
        xd = torch.randn(b, l, RW, generator=gen) * 0.5
        xd[:, :, rank:P] = 0.0
        xq = xd.to(dtype)
        Wdt = torch.randn(e, rank, generator=gen) * 0.3
        D, bias = torch.randn(e, generator=gen), torch.randn(e, generator=gen) - 1
        xf = xq.float()
        delta = torch.einsum("er,blr->bel", Wdt.to(dtype).double(), xf[:, :, :rank].double())
        f = (lambda t: t.flip(-1)) if rev else (lambda t: t)
        tr = lambda t: t.float().transpose(1, 2)
        ref = O.selective_scan(f(tr(u)), f(delta), A, f(tr(xf[:, :, P:P + 16])), f(tr(xf[:, :, P + 16:])), D, f(tr(z.cpu())), bias, True,
                               work_dtype=torch.float64)
        refs.append(f(ref).transpose(1, 2))
        dirs.append(dict(u=u.to(DEV), A=A.to(DEV), D=D.to(DEV), delta_bias=bias.to(DEV), dt_weight=ops.pad_dt_weight(Wdt.to(DEV)),
                         xdbl=xq.to(DEV), reverse=rev))
    whole = ops.scan_cl_fwd([dict(d) for d in dirs], z=z, time_chunks=1)
    ops.LAUNCH_LOG = log = []
    try:
        cut = ops.scan_cl_fwd([dict(d) for d in dirs], z=z, time_chunks=chunks)
    finally:
        ops.LAUNCH_LOG = None
    assert [n for n, *_ in log] == ["cm_scan_cl_fwd"]
    otol = (2e-4, 5e-5) if dtype == torch.float32 else (1.6e-2, 1e-2)
    ctol = (1e-5, 1e-5) if dtype == torch.float32 else (8e-3, 1e-3)       # bf16: one output rounding apart at most
    for i in range(2):
        close(cut[i].float(), whole[i].float(), *ctol)
        close(cut[i].float(), refs[i], *otol)
    # caller-side carry through a chunked launch == through the unchunked one
    st = lambda: torch.empty(b, e, 16, device=DEV)
    h0 = [torch.randn(b, e, 16, generator=gen).to(DEV) for _ in range(2)]
    outs = {}
    for c in (1, chunks):
        hl, dc = [st(), st()], [st(), st()]
        y = ops.scan_cl_fwd([dict(d, h0=h0[i], h_last=hl[i], decay=dc[i]) for i, d in enumerate(dirs)], z=z, time_chunks=c)
        outs[c] = (y, hl, dc)
    for i in range(2):
        close(outs[chunks][0][i].float(), outs[1][0][i].float(), *ctol)
        close(outs[chunks][1][i], outs[1][1][i], 2e-5, 1e-6)
        close(outs[chunks][2][i], outs[1][2][i], 2e-5, 1e-30)
    # the sizes-only policy: small batches are cut, large ones are not
    lib = _native.lib()
    assert lib.cm_scan_cl_fwd_auto_chunks(64, 1000, 512, 2) == 1
    assert lib.cm_scan_cl_fwd_auto_chunks(16, 1000, 512, 2) == 1
    assert lib.cm_scan_cl_fwd_auto_chunks(8, 1000, 512, 2) == 7      # 1024 / 128 workgroups = 8, capped at 1000 // 128 steps
    assert lib.cm_scan_cl_fwd_auto_chunks(4, 4000, 1024, 2) == 8
    assert lib.cm_scan_cl_fwd_auto_chunks(1, 100, 64, 1) == 1


@pytest.mark.parametrize("shape", [(2, 37, 128), (1, 1000, 288), (3, 64, 512), (2, 5, 96), (1, 33, 1024), (16, 1000, 512), (40, 403, 288)])
@pytest.mark.parametrize("rw", [48, 64])
def test_conv_xproj(ops, shape, rw):
    """(rw: x_dbl rows [dt16 | B | C] or, for 16 < dt_rank <= 32, [dt32 | B | C].)  cm_conv_xproj == cm_conv_cl_fwd (bit-exact u) and x_dbl rows == u @ Wx^T computed in fp32 from the bf16 u and
    bf16 weights (the products the MFMA forms), within bf16 output rounding; x a column slice of a wider [x | z].
    The last two shapes are large enough for the 32-step tiles (ragged last tile in the second); the 16-step tiles on
    the same input must give the same bits."""
    b, l, e = shape
    gen = torch.Generator().manual_seed(l + e)
    xz = torch.randn(b, l, 2 * e, generator=gen).bfloat16().to(DEV)
    x = xz[:, :, :e]
    wf, wb = torch.randn(e, 4, generator=gen).to(DEV) * 0.5, torch.randn(e, 4, generator=gen).to(DEV) * 0.5
    bf, bb = torch.randn(e, generator=gen).to(DEV) * 0.1, torch.randn(e, generator=gen).to(DEV) * 0.1
    wx = [(torch.randn(rw, e, generator=gen) * 0.1).bfloat16().to(DEV) for _ in range(2)]
    ucat = torch.zeros(b, l, 2 * e, dtype=torch.bfloat16, device=DEV)
    xdbl = ops.conv_xproj(x, wf, bf, wb, bb, ops.PackedWeight(wx[0]), ops.PackedWeight(wx[1]), out_f=ucat[:, :, :e], out_b=ucat[:, :, e:])
    rf, rb = ops.conv_cl_fwd(x, wf, bf, wb, bb, True)
    assert torch.equal(ucat[:, :, :e], rf) and torch.equal(ucat[:, :, e:], rb)
    # independent check of the conv itself against the oracle (time-contiguous layout)
    ref = O.causal_conv1d(x.float().cpu().transpose(1, 2), wf.cpu(), bf.cpu(), True, work_dtype=torch.float64).transpose(1, 2)
    close(rf.float(), ref, 1.6e-2, 1e-2)
    for i, u in enumerate((rf, rb)):
        want = u.float() @ wx[i].float().t()
        close(xdbl[:, :, rw * i:rw * (i + 1)].float(), want, 1.6e-2, 2e-2)
    assert xdbl.shape == (b, l, 2 * rw)
    if b * ((l + 31) // 32) >= 512:
        u16 = torch.zeros_like(ucat)
        x16 = ops.conv_xproj(x, wf, bf, wb, bb, ops.PackedWeight(wx[0]), ops.PackedWeight(wx[1]), out_f=u16[:, :, :e], out_b=u16[:, :, e:],
                             variant=1)                      # the 16-step tiles
        assert torch.equal(u16, ucat) and torch.equal(x16, xdbl)


def test_scan_rows_extreme_time_steps(ops):
    """Row-group scan at the edges of its arithmetic: time steps beyond softplus' linear threshold (delta = x for
    x > 20), decays that underflow to zero (delta * A << -126 in exp2), negligible time steps (softplus' e^x branch) and
    zero inputs -- against the fp64 oracle, all results finite."""
    b, l, e = 2, 70, 64
    gen = torch.Generator().manual_seed(99)
    u = torch.randn(b, l, e, generator=gen)
    z = torch.randn(b, l, e, generator=gen)
    xd = torch.randn(b, l, 48, generator=gen)
    xd[:, :, 16:32] *= 3.0
    xd[0, 10:20] = 0.0                                           # a stretch of all-zero projections
    Wdt = torch.randn(e, 16, generator=gen) * 0.3
    A = -torch.exp(torch.randn(e, 16, generator=gen) * 2.0)      # |A| from 0.01 to 100
    bias = torch.cat([torch.full((16,), 30.0), torch.full((16,), -25.0), torch.randn(32, generator=gen)])   # huge / tiny / normal steps
    D = torch.randn(e, generator=gen)
    delta = torch.einsum("er,blr->bel", Wdt.double(), xd[:, :, :16].double())
    ref = O.selective_scan(u.transpose(1, 2), delta, A, xd[:, :, 16:32].transpose(1, 2), xd[:, :, 32:].transpose(1, 2), D,
                           z.transpose(1, 2), bias, True, work_dtype=torch.float64).transpose(1, 2)
    (got,) = ops.scan_cl_fwd([dict(u=u.to(DEV), A=A.to(DEV), D=D.to(DEV), delta_bias=bias.to(DEV),
                                   dt_weight=ops.pad_dt_weight(Wdt.to(DEV)), xdbl=xd.to(DEV))], z=z.to(DEV))
    assert torch.isfinite(got).all()
    close(got, ref, 5e-4, 1e-4 * max(1.0, ref.abs().max().item()))


@pytest.mark.parametrize("shape,k,causal", [((2, 48, 250), 31, False), ((3, 256, 1000), 31, False), ((2, 17, 37), 31, True),
                                            ((1, 8, 2500), 15, False), ((2, 144, 5), 31, False)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dwconv1d_forward_backward(ops, shape, k, causal, dtype):
    """cm_dwconv1d_fwd / _bwd (through the autograd node the ConvolutionModule uses) vs torch's depthwise conv1d and its
    autograd in float64 on the CPU: 'same' padding and the causal pad-then-chomp variant, rows longer than one pass
    (2500 steps), rows shorter than the kernel (5 steps), channel counts that are not multiples of anything."""
    b, d, l = shape
    gen = torch.Generator().manual_seed(l + d + k)
    x = torch.randn(b, d, l, generator=gen).to(dtype)
    w = torch.randn(d, 1, k, generator=gen) / k ** 0.5
    bias = torch.randn(d, generator=gen) * 0.1
    dy = torch.randn(b, d, l, generator=gen).to(dtype)
    xr = x.double().requires_grad_(True)
    wr, br = w.double().requires_grad_(True), bias.double().requires_grad_(True)
    pad = k - 1 if causal else k // 2
    ref = torch.nn.functional.conv1d(xr, wr, br, padding=pad, groups=d)
    ref = ref[..., :l] if causal else ref
    ref.backward(dy.double())
    xg = x.to(DEV).requires_grad_(True)
    wg, bg = w.to(DEV).requires_grad_(True), bias.to(DEV).requires_grad_(True)
    got = ops.DepthwiseConv1dFn.apply(xg, wg, bg, pad)
    got.backward(dy.to(DEV))
    ft, gt = ((1e-5, 1e-5), (1e-4, 1e-4)) if dtype == torch.float32 else ((1.6e-2, 1e-2), (2e-2, 2e-2))
    close(got.float(), ref, *ft)
    close(xg.grad.float(), xr.grad, *gt)
    scale = max(1.0, wr.grad.abs().max().item())
    close(wg.grad, wr.grad, gt[0], gt[1] * scale)
    close(bg.grad, br.grad, gt[0], gt[1] * max(1.0, br.grad.abs().max().item()))


@pytest.mark.parametrize("shape,k,causal", [((2, 250, 48), 31, False), ((3, 1000, 256), 31, False), ((2, 37, 17), 31, True),
                                            ((1, 2500, 8), 15, False), ((2, 5, 144), 31, False), ((2, 130, 300), 31, False)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dwconv_channels_last_forward_backward(ops, shape, k, causal, dtype):
    """cm_dwconv_cl_fwd / _bwd (channels-last rows; the autograd node of the module-API ConvolutionModule) vs torch's
    depthwise conv1d + autograd in float64: ragged tiles / runs, dim above and below one workgroup, causal padding; the
    tap gradients are bit-identical between two runs (fixed-order reduction)."""
    b, l, d = shape
    gen = torch.Generator().manual_seed(l + d + k)
    x = torch.randn(b, l, d, generator=gen).to(dtype)
    w = torch.randn(d, 1, k, generator=gen) / k ** 0.5
    bias = torch.randn(d, generator=gen) * 0.1
    dy = torch.randn(b, l, d, generator=gen).to(dtype)
    xr = x.double().requires_grad_(True)
    wr, br = w.double().requires_grad_(True), bias.double().requires_grad_(True)
    pad = k - 1 if causal else k // 2
    ref = torch.nn.functional.conv1d(xr.transpose(1, 2), wr, br, padding=pad, groups=d)
    ref = (ref[..., :l] if causal else ref).transpose(1, 2)
    ref.backward(dy.double())
    xg = x.to(DEV).requires_grad_(True)
    wg, bg = w.to(DEV).requires_grad_(True), bias.to(DEV).requires_grad_(True)
    got = ops.DepthwiseConvClFn.apply(xg, wg, bg, pad)
    got.backward(dy.to(DEV))
    ft, gt = ((1e-5, 1e-5), (1e-4, 1e-4)) if dtype == torch.float32 else ((1.6e-2, 1e-2), (2e-2, 2e-2))
    close(got.float(), ref, *ft)
    close(xg.grad.float(), xr.grad, *gt)
    close(wg.grad, wr.grad, gt[0], gt[1] * max(1.0, wr.grad.abs().max().item()))
    close(bg.grad, br.grad, gt[0], gt[1] * max(1.0, br.grad.abs().max().item()))
    _, dw2, db2 = ops.dwconv_cl_bwd(x.to(DEV), w.to(DEV), dy.to(DEV), True, pad)
    assert torch.equal(dw2.reshape(wg.grad.shape), wg.grad.float()) and torch.equal(db2, bg.grad.float())


@pytest.mark.parametrize("shape", [(3, 37, 256), (1, 1, 144), (2, 1000, 256), (5, 333, 144), (2, 129, 512), (1, 70, 640), (3, 50, 1024),
                                   (4, 4099, 256), (2, 17, 8)])
@pytest.mark.parametrize("mode", ["f32", "bf16_autocast", "bf16_plain"])
def test_layernorm_forward_backward(ops, shape, mode):
    """cm_layernorm_fwd / _bwd through the autograd node the module API uses vs torch's LayerNorm + autograd in float64:
    both lane mappings (16 and 64 lanes per row), dims that leave lanes idle (144, 8, 640), row counts that are not a
    multiple of the rows per wave, bf16 input with fp32 output (autocast) and bf16 in/out; the gamma / beta gradients are
    bit-identical between two runs (fixed-order reduction)."""
    b, l, d = shape
    gen = torch.Generator().manual_seed(l * 7 + d)
    dt_in = torch.float32 if mode == "f32" else torch.bfloat16
    x = (torch.randn(b, l, d, generator=gen) * 2.0 + 0.5).to(dt_in)
    w, bias = torch.randn(d, generator=gen) * 0.5 + 1.0, torch.randn(d, generator=gen) * 0.2
    dt_out = torch.bfloat16 if mode == "bf16_plain" else torch.float32
    dy = torch.randn(b, l, d, generator=gen).to(dt_out)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), bias.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (d,), wr, br, 1e-5)
    ref.backward(dy.double())
    xg, wg, bg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True), bias.to(DEV).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode == "bf16_autocast"):
        got = ops.LayerNormFn.apply(xg, wg, bg, 1e-5)
    assert got.dtype == dt_out and got.shape == x.shape
    got.backward(dy.to(DEV))
    ft = (1e-5, 1e-5) if dt_out == torch.float32 else (1.6e-2, 1e-2)
    gt = (1e-4, 1e-4) if mode == "f32" else (1.6e-2, 1e-2)          # dx is rounded to the input dtype
    close(got.float(), ref, *ft)
    close(xg.grad.float(), xr.grad, *gt)
    scale = lambda t: max(1.0, t.abs().max().item())
    close(wg.grad, wr.grad, 1e-4, 1e-4 * scale(wr.grad))
    close(bg.grad, br.grad, 1e-4, 1e-4 * scale(br.grad))
    y2, x2, stats = ops.layernorm_fwd(x.to(DEV), w.to(DEV), bias.to(DEV), 1e-5, dt_out)
    assert torch.equal(y2, got.detach())
    _, dg2, db2 = ops.layernorm_bwd(dy.to(DEV), x2, stats, w.to(DEV), 1e-5)
    assert torch.equal(dg2, wg.grad) and torch.equal(db2, bg.grad)


@pytest.mark.parametrize("shape,norm", [((2, 50, 40, 64), (40, 64)), ((3, 33, 20, 32), (20, 32)), ((1, 7, 4096), (4096,)), ((2, 9, 1028), (1028,))])
@pytest.mark.parametrize("autocast", [False, True])
def test_layernorm_module_wide_and_multi_axis(ops, shape, norm, autocast):
    """sb_compat.RowsLayerNorm (the module every LayerNorm of the module API is) over several trailing axes and over rows
    wider than 1024 (one workgroup per row) vs torch's LayerNorm in float64: the CNN front end's (frequency, channel)
    norms (40 x 64 = 2560, 20 x 32 = 640), the 4096 limit, a width that leaves threads idle."""
    from mamba_asr_amd import sb_compat as sb
    gen = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(*shape, generator=gen) * 1.5 + 0.3).to(torch.bfloat16 if autocast else torch.float32)
    dy = torch.randn(*shape, generator=gen)
    m = sb.RowsLayerNorm(norm, eps=1e-5)
    with torch.no_grad():
        m.weight.copy_(torch.randn(*norm, generator=gen) * 0.5 + 1.0)
        m.bias.copy_(torch.randn(*norm, generator=gen) * 0.2)
    ref_m = torch.nn.LayerNorm(norm, eps=1e-5).double()
    ref_m.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
    xr = x.double().requires_grad_(True)
    ref = ref_m(xr)
    ref.backward(dy.double())
    m = m.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        got = m(xg)
    assert got.dtype == torch.float32 and got.shape == x.shape
    got.backward(dy.to(DEV))
    close(got, ref, 1e-5, 2e-5)
    close(xg.grad.float(), xr.grad, *((1.6e-2, 1e-2) if autocast else (1e-4, 1e-4)))
    scale = lambda t: max(1.0, t.abs().max().item())
    close(m.weight.grad, ref_m.weight.grad, 1e-4, 1e-4 * scale(ref_m.weight.grad))
    close(m.bias.grad, ref_m.bias.grad, 1e-4, 1e-4 * scale(ref_m.bias.grad))


@pytest.mark.parametrize("shape", [(100, 256, 256), (777, 1024, 256), (130, 256, 1024), (64, 512, 640)])
def test_gemm_bf16_epilogues(ops, shape):
    """cm_gemm_bf16 vs torch fp32 reference on the same bf16-rounded operands; asymmetric data so that a transposed
    fragment map cannot pass."""
    m, n, k = shape
    gen = torch.Generator().manual_seed(m + n)
    a = (torch.randn(m, k, generator=gen) * 0.5).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=gen) * 0.1).to(torch.bfloat16)
    bias = torch.randn(n, generator=gen)
    acc = a.float() @ w.float().t()
    ga, gw, gb = a.to(DEV), w.to(DEV), bias.to(DEV)
    out0 = ops.gemm_bf16(ga, gw, gb, epilogue=0)
    torch.testing.assert_close(out0.float().cpu(), acc + bias, rtol=1.6e-2, atol=2e-2)
    out0n = ops.gemm_bf16(ga, gw, None, epilogue=0)
    torch.testing.assert_close(out0n.float().cpu(), acc, rtol=1.6e-2, atol=2e-2)
    out1 = ops.gemm_bf16(ga, gw, gb, epilogue=1)
    torch.testing.assert_close(out1.float().cpu(), torch.nn.functional.gelu(acc + bias), rtol=1.6e-2, atol=2e-2)
    # strided A (column slice of a wider buffer, as [u_fwd | u_bwd])
    wide = torch.zeros(m, 2 * k, dtype=torch.bfloat16, device=DEV)
    wide[:, k:] = ga
    torch.testing.assert_close(ops.gemm_bf16(wide[:, k:], gw, gb, epilogue=0).float().cpu(), out0.float().cpu(), rtol=0, atol=0)
    if n == 256:
        x = torch.randn(m, n, generator=gen)
        g1, b1, g2, b2 = (torch.randn(n, generator=gen) for _ in range(4))
        r = x + 0.5 * (acc + bias)
        # residual only
        gx = x.clone().to(DEV)
        assert ops.gemm_bf16(ga, gw, gb, epilogue=2, x=gx, alpha=0.5, want_out=False) is None
        torch.testing.assert_close(gx.cpu(), r, rtol=1e-3, atol=5e-3)
        # residual + LN2 -> bf16 out, x keeps the un-normalised residual
        gx = x.clone().to(DEV)
        out = ops.gemm_bf16(ga, gw, gb, epilogue=2, x=gx, alpha=0.5, norm2=(g2.to(DEV), b2.to(DEV), 1e-5))
        torch.testing.assert_close(gx.cpu(), r, rtol=1e-3, atol=5e-3)
        torch.testing.assert_close(out.float().cpu(), torch.nn.functional.layer_norm(r, (n,), g2, b2, 1e-5), rtol=2e-2, atol=3e-2)
        # LN1 into x, LN2 of that into out
        gx = x.clone().to(DEV)
        out = ops.gemm_bf16(ga, gw, gb, epilogue=2, x=gx, alpha=0.5, norm1=(g1.to(DEV), b1.to(DEV), 1e-5),
                            norm2=(g2.to(DEV), b2.to(DEV), 1e-6))
        r1 = torch.nn.functional.layer_norm(r, (n,), g1, b1, 1e-5)
        torch.testing.assert_close(gx.cpu(), r1, rtol=2e-3, atol=1e-2)
        torch.testing.assert_close(out.float().cpu(), torch.nn.functional.layer_norm(r1, (n,), g2, b2, 1e-6), rtol=2e-2, atol=3e-2)


# ---------------------------------------------------------------------------------------------------------------------
# deterministic gradient reductions (round 2): per-workgroup partials + fixed-order second pass instead of fp32 atomics
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_backward_kernels_are_bit_reproducible(dtype, monkeypatch):
    """cm_selective_scan_bwd and cm_causal_conv1d_bwd with their workspaces: every gradient is bit-identical between two
    runs (the atomics path is only close), and equals the atomics path within fp32 round-off."""
    from mamba_asr_amd import ops
    g = torch.Generator(device=DEV).manual_seed(5)
    b, e, l, n = 6, 160, 333, 16
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)
    u, delta, z, dout = (rnd(b, e, l).to(dtype), (rnd(b, e, l) * 0.5).to(dtype), rnd(b, e, l).to(dtype), rnd(b, e, l).to(dtype))
    A = -torch.exp(rnd(e, n) * 0.3)
    B, C = rnd(b, n, l), rnd(b, n, l)
    D, bias = rnd(e), rnd(e) - 1

    def scan_grads():
        _, x, _ = ops.selective_scan_fwd(u, delta, A, B, C, D, z, bias, True, need_out=False)
        return [t for t in ops.selective_scan_bwd(u, delta, A, B, C, D, z, bias, dout, x, True) if t is not None]

    monkeypatch.setattr(ops, "DETERMINISTIC", True)
    r1, r2 = scan_grads(), scan_grads()
    assert all(torch.equal(a_, b_) for a_, b_ in zip(r1, r2))
    monkeypatch.setattr(ops, "DETERMINISTIC", False)
    r3 = scan_grads()
    for a_, c_ in zip(r1, r3):
        torch.testing.assert_close(a_.float(), c_.float(), rtol=2e-4, atol=2e-4 * max(1.0, float(c_.float().abs().max())))
    w, cb = rnd(e, 4) * 0.5, rnd(e) * 0.1
    x = rnd(b, e, l).to(dtype)
    for rev in (False, True):
        monkeypatch.setattr(ops, "DETERMINISTIC", True)
        c1 = ops.causal_conv1d_bwd(x, w, cb, dout, True, reverse=rev)
        c2 = ops.causal_conv1d_bwd(x, w, cb, dout, True, reverse=rev)
        assert all(torch.equal(a_, b_) for a_, b_ in zip(c1, c2))
        monkeypatch.setattr(ops, "DETERMINISTIC", False)
        c3 = ops.causal_conv1d_bwd(x, w, cb, dout, True, reverse=rev)
        for a_, c_ in zip(c1, c3):
            torch.testing.assert_close(a_.float(), c_.float(), rtol=2e-4, atol=2e-4 * max(1.0, float(c_.float().abs().max())))


def test_bimamba_layer_gradients_are_bit_reproducible():
    """The whole BiMamba mixer, forward + backward, twice on the same input: identical bits in every gradient."""
    from mamba_asr_amd.modules.mamba.bimamba import Mamba
    torch.manual_seed(2)
    m = Mamba(256, d_state=16, d_conv=4, expand=2, bimamba_type="v2").to(DEV)
    x = torch.randn(4, 200, 256, device=DEV)
    dy = torch.randn(4, 200, 256, device=DEV)

    def run():
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(xi)
        return torch.autograd.grad(y, [xi] + list(m.parameters()), dy.to(y.dtype))

    g1, g2 = run(), run()
    assert all(torch.equal(a_, b_) for a_, b_ in zip(g1, g2))


@pytest.mark.parametrize("shape", [(2, 9, 7, 1), (3, 40, 20, 64), (1, 5, 6, 3), (2, 11, 8, 8), (2, 6, 5, 2)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("pad", [1, 2])
def test_reflect_pad_tf_forward_and_backward(shape, dtype, pad):
    """cm_reflect_pad_tf against torch's reflection pad on the same channels-last tensor (the padding of the front end's Conv2d
    blocks): forward bit-exact; backward = autograd's fold of the padded gradient (sums of up to four values)."""
    from mamba_asr_amd import ops
    g = torch.Generator().manual_seed(sum(shape) + pad)
    x = torch.randn(*shape, generator=g).to(dtype).to(DEV).requires_grad_(True)
    ref = torch.nn.functional.pad(x.unsqueeze(0), (0, 0, pad, pad, pad, pad), mode="reflect").squeeze(0)
    got = ops.ReflectPadTfFn.apply(x, pad)
    assert got.shape == ref.shape and torch.equal(got, ref)
    dy = torch.randn(ref.shape, generator=g).to(dtype).to(DEV)
    (gref,) = torch.autograd.grad(ref, x, dy)
    (ggot,) = torch.autograd.grad(got, x, dy)
    if dtype == torch.float32:
        torch.testing.assert_close(ggot, gref, rtol=1e-6, atol=1e-6)
    else:
        torch.testing.assert_close(ggot.float(), gref.float(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("rows,m,n", [(64, 128, 128), (1000, 256, 128), (777, 128, 384), (4100, 1024, 256), (32000, 256, 1024)])
@pytest.mark.parametrize("strided", [False, True])
@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_wgrad_bf16_against_fp64(rows, m, n, strided, variant):
    """cm_wgrad_bf16 (split-K over row chunks, both operands transposed on the way out of LDS) against a^T b in fp64 on the same
    bf16 values; operands as column slices of wider tensors (the gradients of the concatenated BiMamba tensors); run twice: same bits."""
    from mamba_asr_amd import ops
    g = torch.Generator().manual_seed(rows + m + n)
    wa, wb = (m + 64, n + 128) if strided else (m, n)
    A = torch.randn(rows, wa, generator=g).bfloat16().to(DEV)
    B = torch.randn(rows, wb, generator=g).bfloat16().to(DEV)
    a, b = (A[:, 64:], B[:, 128:]) if strided else (A, B)
    got = ops.wgrad(a, b, variant=variant)
    ref = (a.double().t() @ b.double())
    err = float((got.double() - ref).abs().max()) / float(ref.abs().max())
    assert got.shape == (m, n) and got.dtype == torch.float32 and err < 2e-6, err
    assert torch.equal(got, ops.wgrad(a, b, variant=variant))


def test_wgrad_falls_back_to_batched_gemms_for_other_shapes():
    from mamba_asr_amd import ops
    g = torch.Generator().manual_seed(5)
    a = torch.randn(600, 48, generator=g).bfloat16().to(DEV)
    b = torch.randn(600, 512, generator=g).bfloat16().to(DEV)
    got = ops.wgrad(a, b, nbatch=6)
    ref = a.double().t() @ b.double()
    assert float((got.double() - ref).abs().max()) / float(ref.abs().max()) < 2e-2      # per-chunk products are rounded to bf16


@pytest.mark.parametrize("shape", [(2, 30, 40, 64), (3, 17, 20, 32), (2, 9, 8, 16)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("drop", [False, True])
def test_layernorm_leaky_dropout2d_epilogue(shape, dtype, drop):
    """ops.LnActDropFn (cm_layernorm_fwd / _bwd with the activation + channel-mask epilogue: the front-end Conv2d block's
    LayerNorm over (freq, channel) -> LeakyReLU(0.01) -> Dropout2d) against the same three steps in torch, fp64, forward and all
    gradients; wide rows (40 x 64 = 2560) and the 64-lane / 16-lane row kernels."""
    from mamba_asr_amd import ops
    b, t, f, c = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(b, t, f, c, generator=g)
    w, bias = 1 + 0.2 * torch.randn(f * c, generator=g), 0.2 * torch.randn(f * c, generator=g)
    mask = (torch.rand(b, c, generator=g) > 0.3).float() / 0.7 if drop else None
    dy = torch.randn(b, t, f, c, generator=g)
    # the reference sees the values the kernel sees (bf16-rounded x, dy): a LayerNorm output next to 0 would otherwise change the
    # LeakyReLU branch between the two
    x, dy = x.to(dtype).float(), dy.to(dtype).float()
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), bias.double().requires_grad_(True)
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.layer_norm(xr.view(b, t, f * c), (f * c,), wr, br, 1e-5), 0.01).view(b, t, f, c)
    if drop:
        ref = ref * mask.double()[:, None, None, :]
    gref = torch.autograd.grad(ref, [xr, wr, br], dy.double())
    xg = x.to(dtype).to(DEV).requires_grad_(True)
    wg, bg = w.to(DEV).requires_grad_(True), bias.to(DEV).requires_grad_(True)
    got = ops.LnActDropFn.apply(xg, wg, bg, 1e-5, 0.01, mask.to(DEV) if drop else None, dtype)
    ggot = torch.autograd.grad(got, [xg, wg, bg], dy.to(dtype).to(DEV))
    if dtype == torch.float32:
        tol = dict(rtol=1e-4, atol=1e-4)
        ref_in = ref
    else:
        tol = dict(rtol=3e-2, atol=3e-2)                          # the bf16 rounding of the outputs
        ref_in = ref
    torch.testing.assert_close(got.double().cpu(), ref_in.detach(), **tol)
    for a_, r_ in zip(ggot, gref):
        scale = max(1.0, float(r_.abs().max()))
        torch.testing.assert_close(a_.double().cpu(), r_, rtol=tol["rtol"], atol=tol["atol"] * scale)
