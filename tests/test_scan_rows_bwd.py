"""cm_scan_cl_bwd (csrc/scan_rows_bwd.hip): the channels-last backward of the row-group scan, both directions in one launch,
against the oracle's analytic gradients (oracle.selective_scan_bwd, pinned to the reference's autograd by G2) and fp64
matmuls for the dt_proj part (reference selective_scan_interface.py:252-279).  The forward that writes the checkpoints and
the pre-gate output is cm_scan_cl_fwd itself (training mode: ckpt / ypre set)."""
import pytest
import torch

from oracle import conmamba_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(a, b, rtol, atol):
    scale = max(1.0, float(b.abs().max()))
    torch.testing.assert_close(a.detach().double().cpu(), b.detach().double().cpu(), rtol=rtol, atol=atol * scale)


def _case(ops, b, l, e, rank, dtype, seed, lanes=0, a_scale=1.0):
    P = 16 if rank <= 16 else 32
    RW = P + 32
    gen = torch.Generator().manual_seed(seed)
    xz = torch.randn(b, l, 2 * e, generator=gen).to(dtype)
    z = xz[:, :, e:]
    dmix = (torch.randn(b, l, e, generator=gen) * 0.5).to(dtype)                    # shared by both directions, as in BiMamba v2
    ucat = torch.randn(b, l, 2 * e, generator=gen).to(dtype)
    xcat = torch.randn(b, l, 2 * RW, generator=gen)
    for i in range(2):
        xcat[:, :, RW * i + rank:RW * i + P] = 0.0
    xcat = xcat.to(dtype)
    dirs, refs = [], []
    ycat = torch.zeros(b, l, 2 * e, dtype=dtype, device=DEV)
    pcat = torch.zeros(b, l, 2 * e, dtype=dtype, device=DEV)
    gz, gd, gu, gx = xz.to(DEV)[:, :, e:], dmix.to(DEV), ucat.to(DEV), xcat.to(DEV)
    for i, rev in enumerate((False, True)):
        u = ucat[:, :, e * i:e * (i + 1)]
        xd = xcat[:, :, RW * i:RW * (i + 1)].double()
        A = -torch.exp(torch.randn(e, 16, generator=gen) * 0.3) * a_scale
        Wdt = torch.randn(e, rank, generator=gen) * 0.3
        D, bias = torch.randn(e, generator=gen), torch.randn(e, generator=gen) - 1
        Wq = Wdt.to(dtype).double()                                                 # bf16 I/O: the product runs on the bf16-rounded weight
        delta = torch.einsum("er,blr->bel", Wq, xd[:, :, :rank])                    # (b, e, l), pre-bias
        Bm, Cm = xd[:, :, P:P + 16].transpose(1, 2), xd[:, :, P + 16:].transpose(1, 2)
        f = (lambda t: t.flip(-1)) if rev else (lambda t: t)
        tr = lambda t: t.double().transpose(1, 2)
        r = O.selective_scan_bwd(f(tr(u)), f(delta), A, f(Bm), f(Cm), D, f(tr(z)), bias, f(tr(dmix)), True)
        r = {k: (f(v) if v is not None and v.dim() == 3 else v) for k, v in r.items()}
        ddelta = r["ddelta"]                                                        # (b, e, l)
        r["ddt"] = torch.einsum("bel,er->blr", ddelta, Wq)                          # :279
        r["dW"] = torch.einsum("bel,blr->er", ddelta, xd[:, :, :rank])              # :278
        refs.append(r)
        dirs.append(dict(u=gu[:, :, e * i:e * (i + 1)], xdbl=gx[:, :, RW * i:RW * (i + 1)], A=A.to(DEV), D=D.to(DEV), delta_bias=bias.to(DEV),
                         dt_weight=ops.pad_dt_weight(Wdt.to(DEV)), reverse=rev, out=ycat[:, :, e * i:e * (i + 1)],
                         ypre=pcat[:, :, e * i:e * (i + 1)], ckpt=torch.empty(ops.scan_ckpt_shape(b, l, e), device=DEV)))
    ops.scan_cl_fwd(dirs, z=gz, delta_softplus=True, time_chunks=1, split=lanes)
    for dd in dirs:
        dd["dout"] = gd
    return dirs, gz, refs, rank, P


@pytest.mark.parametrize("shape", [(2, 37, 64), (1, 130, 200), (3, 64, 72), (2, 300, 512)])
@pytest.mark.parametrize("dtype,rank", [(torch.float32, 16), (torch.float32, 5), (torch.bfloat16, 16), (torch.bfloat16, 9), (torch.bfloat16, 32)])
def test_scan_rows_bwd_two_directions(shape, dtype, rank):
    from mamba_asr_amd import ops
    b, l, e = shape
    dirs, gz, refs, rank, P = _case(ops, b, l, e, rank, dtype, seed=l * 7 + e + rank)
    outs = ops.scan_cl_bwd(dirs, gz)
    torch.cuda.synchronize()
    f32 = dtype == torch.float32
    rt, at = (2e-3, 2e-4) if f32 else (2e-2, 1.2e-2)
    for o, r in zip(outs, refs):
        close(o["du"].float(), r["du"].transpose(1, 2), rt, at)
        close(o["dz"].float(), r["dz"].transpose(1, 2), rt, at)
        close(o["dxdbl"][:, :, P:P + 16].float(), r["dB"].transpose(1, 2), rt, at)
        close(o["dxdbl"][:, :, P + 16:].float(), r["dC"].transpose(1, 2), rt, at)
        close(o["dxdbl"][:, :, :rank].float(), r["ddt"], rt, at)
        assert float(o["dxdbl"][:, :, rank:P].float().abs().max()) == 0.0 if rank < P else True
        # parameter gradients are fp32 sums in both modes; bf16 mode rounds ddelta to bf16 in front of the ddt_weight product
        close(o["dA"], r["dA"], 3e-3 if f32 else 2e-2, 3e-4 if f32 else 5e-3)
        close(o["dD"], r["dD"], 3e-3 if f32 else 2e-2, 3e-4 if f32 else 5e-3)
        close(o["ddelta_bias"], r["ddelta_bias"], 3e-3 if f32 else 2e-2, 3e-4 if f32 else 5e-3)
        close(o["ddt_weight"][:, :rank], r["dW"], 3e-3 if f32 else 2e-2, 3e-4 if f32 else 8e-3)
    # deterministic: a second launch gives the same bits
    outs2 = ops.scan_cl_bwd(dirs, gz)
    for o, o2 in zip(outs, outs2):
        for k in o:
            assert torch.equal(o[k], o2[k]), k


@pytest.mark.parametrize("shape,chunks", [((2, 37, 64), 2), ((1, 130, 200), 3), ((3, 64, 72), 4), ((2, 300, 512), 5), ((1, 1000, 128), 0)])
@pytest.mark.parametrize("dtype,rank", [(torch.float32, 16), (torch.float32, 5), (torch.bfloat16, 16), (torch.bfloat16, 32)])
def test_scan_rows_bwd_time_chunks(shape, chunks, dtype, rank):
    """Launches cut along time (adjoint summaries per chunk, carry fold, full pass per chunk from the carried-in adjoint) against
    the one-pass launch AND the oracle, with a slow decay (|A| x 0.03: the adjoint entering a chunk carries weight over hundreds
    of steps).  chunks = 0: the library's own policy (1 x 1000 x 128 = 4 workgroups: it cuts)."""
    from mamba_asr_amd import ops
    b, l, e = shape
    dirs, gz, refs, rank, P = _case(ops, b, l, e, rank, dtype, seed=l * 11 + e + rank, a_scale=0.03)
    one = ops.scan_cl_bwd(dirs, gz, time_chunks=1)
    one = [{k: v.clone() for k, v in o.items()} for o in one]
    cut = ops.scan_cl_bwd(dirs, gz, time_chunks=chunks)
    torch.cuda.synchronize()
    f32 = dtype == torch.float32
    for o, c, r in zip(one, cut, refs):
        for k in o:
            # same arithmetic per step; only the adjoint's entry value is folded in a different order (P L + E per chunk)
            close(c[k].float(), o[k].float().double(), 1e-4 if f32 else 1.6e-2, 1e-5 if f32 else 8e-3)
        rt, at = (2e-3, 2e-4) if f32 else (2e-2, 1.2e-2)
        close(c["du"].float(), r["du"].transpose(1, 2), rt, at)
        close(c["dxdbl"][:, :, P:P + 16].float(), r["dB"].transpose(1, 2), rt, at)
        close(c["dA"], r["dA"], 3e-3 if f32 else 2e-2, 3e-4 if f32 else 5e-3)
    if chunks == 0:
        from mamba_asr_amd import _native as N
        assert N.lib().cm_scan_cl_bwd_auto_chunks(b, l, e, 2) > 1
    cut2 = ops.scan_cl_bwd(dirs, gz, time_chunks=chunks)
    for o, o2 in zip(cut, cut2):
        for k in o:
            assert torch.equal(o[k], o2[k]), k


@pytest.mark.parametrize("lanes", [4, 8])
def test_scan_rows_fwd_training_outputs(lanes):
    """The training forward's extra outputs: ypre * silu(z) == out (up to the output rounding), checkpoints == the states a
    sequence cut at the half-block boundary carries (h_last of the prefix), for both directions and both lane splits; the
    backward on the 2-states-per-lane forward's checkpoints == on the 4-states-per-lane one's."""
    from mamba_asr_amd import ops
    b, l, e = 2, 75, 64
    dirs, gz, refs, rank, P = _case(ops, b, l, e, 16, torch.float32, seed=5, lanes=lanes)
    outs = ops.scan_cl_bwd(dirs, gz)
    for o, r in zip(outs, refs):
        close(o["du"].float(), r["du"].transpose(1, 2), 2e-3, 2e-4)
        close(o["dA"], r["dA"], 3e-3, 3e-4)
    for i, dd in enumerate(dirs):
        y, yp = dd["out"], dd["ypre"]
        torch.testing.assert_close(y, yp * torch.nn.functional.silu(gz), rtol=1e-5, atol=1e-6)
        ck = dd["ckpt"]
        assert ck.shape == (b, 2 * 5, e, 16)
        rev = bool(dd["reverse"])
        # entry state of half block m == last state of the scan over the steps before it (in scan order)
        for m in (1, 4, 7, 9):
            lo, hi = (8 * m + 8, l) if rev else (0, 8 * m)
            if lo >= hi:
                continue
            hl = torch.zeros(b, e, 16, device=DEV)
            sub = dict(u=dd["u"][:, lo:hi], xdbl=dd["xdbl"][:, lo:hi], A=dd["A"], D=dd["D"], delta_bias=dd["delta_bias"],
                       dt_weight=dd["dt_weight"], reverse=rev, h_last=hl)
            ops.scan_cl_fwd([sub], z=gz[:, lo:hi], delta_softplus=True, time_chunks=1, split=lanes)
            torch.testing.assert_close(ck[:, m], hl, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("shape", [(2, 37, 64), (1, 130, 200), (3, 64, 72), (2, 300, 512)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("two", [True, False])
def test_conv_cl_bwd(shape, dtype, two):
    """cm_conv_cl_bwd vs the oracle's causal-conv backward per direction (oracle.causal_conv1d_bwd, pinned by k_conv.npz);
    the reverse direction through flipped tensors as the reference does it (bimamba.py:237)."""
    from mamba_asr_amd import ops
    b, l, e = shape
    gen = torch.Generator().manual_seed(l + e)
    xz = torch.randn(b, l, 2 * e, generator=gen).to(dtype)
    x = xz[:, :, :e]
    du = torch.randn(b, l, 2 * e, generator=gen).to(dtype)
    dzs = torch.randn(b, l, 2 * e, generator=gen).to(dtype)
    w = [torch.randn(e, 4, generator=gen) * 0.5 for _ in range(2)]
    bs = [torch.randn(e, generator=gen) * 0.2 for _ in range(2)]
    gxz, gdu, gdz = xz.to(DEV), du.to(DEV), dzs.to(DEV)
    dxz = torch.zeros(b, l, 2 * e, dtype=dtype, device=DEV)
    args = dict(du_b=gdu[:, :, e:], weight_b=w[1].to(DEV), bias_b=bs[1].to(DEV), dz_b=gdz[:, :, e:]) if two else {}
    dx, dz, dwf, dbf, dwb, dbb = ops.conv_cl_bwd(gxz[:, :, :e], w[0].to(DEV), bs[0].to(DEV), gdu[:, :, :e], dz_f=gdz[:, :, :e],
                                                 dx=dxz[:, :, :e], dz=dxz[:, :, e:], **args)
    tr = lambda t: t.double().transpose(1, 2)
    rdx, rdw, rdb = O.causal_conv1d_bwd(tr(x), w[0], bs[0], tr(du[:, :, :e]), True)
    want_dx, want_dz = rdx, dzs[:, :, :e].double()
    f32 = dtype == torch.float32
    rt, at = (2e-4, 2e-5) if f32 else (1.6e-2, 1e-2)
    if two:
        bdx, bdw, bdb = O.causal_conv1d_bwd(tr(x).flip(-1), w[1], bs[1], tr(du[:, :, e:]).flip(-1), True)
        want_dx = want_dx + bdx.flip(-1)
        want_dz = want_dz + dzs[:, :, e:].double()
        close(dwb, bdw, 1e-3 if f32 else 1e-2, 1e-4 if f32 else 2e-3)
        close(dbb, bdb, 1e-3 if f32 else 1e-2, 1e-4 if f32 else 2e-3)
    else:
        assert dwb is None and dbb is None
    close(dxz[:, :, :e].float(), want_dx.transpose(1, 2), rt, at)
    close(dxz[:, :, e:].float(), want_dz, rt, at)
    close(dwf, rdw, 1e-3 if f32 else 1e-2, 1e-4 if f32 else 2e-3)
    close(dbf, rdb, 1e-3 if f32 else 1e-2, 1e-4 if f32 else 2e-3)
    again = ops.conv_cl_bwd(gxz[:, :, :e], w[0].to(DEV), bs[0].to(DEV), gdu[:, :, :e], dz_f=gdz[:, :, :e], **args)
    assert torch.equal(again[0], dxz[:, :, :e]) and torch.equal(again[2], dwf) and torch.equal(again[3], dbf)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_ffn_rows_node_vs_module_tree(dtype, monkeypatch):
    """modules/ffn_rows.FfnRowsFn (fused bias + GELU + dropout / bias + dropout + residual kernels, csrc/ffn_train.hip) against the
    same module tree run by torch (CM_FFN_ROWS off) in fp64 on the CPU: output and every gradient, dropout off; then dropout on:
    the masks of forward and backward agree (gradient check by finite consistency), keep rate ~ 1 - p, scale 1 / (1 - p)."""
    import torch.nn as nn
    from mamba_asr_amd.modules import Conmamba as CM
    from mamba_asr_amd.modules import ffn_rows
    torch.manual_seed(3)
    layer = CM.ConmambaEncoderLayer(d_model=256, d_ffn=1024, kernel_size=31, activation=nn.GELU, bias=True, dropout=0.0, causal=False,
                                    mamba_config={"d_state": 16, "expand": 2, "d_conv": 4, "bidirectional": True})
    mod = layer.ffn_module1
    with torch.no_grad():
        for p_ in mod.parameters():
            if p_.dim() == 1:
                p_.add_(0.1 * torch.randn_like(p_))
    x = torch.randn(3, 37, 256)
    dy = torch.randn(3, 37, 256)
    ref_mod = __import__("copy").deepcopy(mod).double()
    xr = x.double().requires_grad_(True)
    want = xr + 0.5 * ref_mod(xr)
    gref = torch.autograd.grad(want, [xr] + list(ref_mod.parameters()), dy.double())
    layer = layer.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
        got = layer._ffn(layer.ffn_module1, xg)
    ggot = torch.autograd.grad(got, [xg] + list(layer.ffn_module1.parameters()), dy.to(DEV))
    rt, at = (2e-4, 2e-5) if dtype == torch.float32 else (2e-2, 1.5e-2)
    close(got, want.detach(), rt, at)
    for a_, b_ in zip(ggot, gref):
        close(a_, b_, rt * 5, at)
    # dropout on: y = x + 0.5 * drop(f): same mask in forward and backward -> d(out)/d(b2) column sums equal 0.5 * sum(dy * mask / (1 - p))
    layer.train()
    for m_ in layer.ffn_module1.modules():
        if isinstance(m_, nn.Dropout):
            m_.p = 0.25
    xg2 = x.to(DEV).requires_grad_(True)
    out = layer._ffn(layer.ffn_module1, xg2)
    b2 = layer.ffn_module1[1].ffn[3].bias
    (gb2,) = torch.autograd.grad(out, [b2], torch.ones_like(out))
    keep = gb2 / (0.5 / 0.75)                                           # kept rows per column
    rate = float(keep.sum() / (3 * 37 * 256))
    assert abs(rate - 0.75) < 0.02, rate
    assert torch.allclose(keep, keep.round(), atol=1e-3)


def _drop_masks(ops, rows, dim, p, seed):
    """The keep decisions of csrc/cm_dropout.h for a (rows, dim) tensor, through the element-wise kernel's stored mask."""
    z = torch.zeros(rows, dim, device=DEV, dtype=torch.bfloat16)
    _, m = ops.bias_act_dropout_fwd(z, None, act=0, p=p, seed=seed, store_mask=True)
    return m.bool()


@pytest.mark.parametrize("rows,hidden", [(300, 1024), (64, 256), (1000, 2048)])
@pytest.mark.parametrize("p1,p2", [(0.0, 0.0), (0.1, 0.1), (0.3, 0.0)])
def test_ffn_fused_training_forward(rows, hidden, p1, p2):
    """cm_ffn_fused's training variant: x_out, the stored pre-activation / normalised input / LayerNorm statistics, and both
    dropouts (decisions = cm_dropout.h's function of (seed, index), read back through the element-wise kernel's mask) against
    the same arithmetic in torch: LN in fp32 -> bf16, GEMM 1 in fp32 on bf16 inputs + bias -> bf16, GELU (erf) x mask x scale ->
    bf16, GEMM 2 + bias, mask x scale, x + alpha x that."""
    from mamba_asr_amd import ops
    g = torch.Generator().manual_seed(rows + hidden)
    D = 256
    x = torch.randn(rows, D, generator=g).to(DEV)
    lnw, lnb = (1 + 0.1 * torch.randn(D, generator=g)).to(DEV), (0.1 * torch.randn(D, generator=g)).to(DEV)
    w1 = (torch.randn(hidden, D, generator=g) / 16).to(DEV).bfloat16()
    w2 = (torch.randn(D, hidden, generator=g) / (hidden ** 0.5)).to(DEV).bfloat16()
    b1, b2 = (0.1 * torch.randn(hidden, generator=g)).to(DEV), (0.1 * torch.randn(D, generator=g)).to(DEV)
    s1, s2 = 1234567 + rows, 7654321 + hidden
    out = torch.empty_like(x)
    _, (pre, xn, stats) = ops.ffn_fused(x, (lnw, lnb, 1e-5), w1, b1, w2, b2, alpha=0.5, x_out=out, train=(p1, p2, s1, s2))
    torch.cuda.synchronize()
    mean, var = x.mean(1), x.var(1, unbiased=False)
    close(stats[0], mean, 1e-5, 1e-5)
    close(stats[1], (var + 1e-5).rsqrt(), 1e-4, 1e-5)
    xn_ref = torch.nn.functional.layer_norm(x, (D,), lnw, lnb, 1e-5)
    close(xn.float(), xn_ref, 1e-2, 1e-2)
    pre_ref = xn.float() @ w1.float().t() + b1                                # from the kernel's own bf16 xn: isolates GEMM 1
    close(pre.float(), pre_ref, 1e-2, 1e-2)
    sc = lambda p: 1.0 / (1.0 - round(p * 65536) / 65536.0)
    m1 = _drop_masks(ops, rows, hidden, p1, s1) if p1 > 0 else torch.ones(rows, hidden, dtype=torch.bool, device=DEV)
    m2 = _drop_masks(ops, rows, D, p2, s2) if p2 > 0 else torch.ones(rows, D, dtype=torch.bool, device=DEV)
    if p1 > 0:
        assert abs(float(m1.float().mean()) - (1 - p1)) < 0.01
    act = (torch.nn.functional.gelu(pre.float()) * m1 * sc(p1)).bfloat16()
    y2 = (act.float() @ w2.float().t() + b2) * m2 * sc(p2)
    close(out, x + 0.5 * y2, 1e-2, 1e-2)
    if p2 > 0:
        # a dropped element of the second dropout leaves the residual stream untouched: the decisions are the mask's, exactly
        assert torch.equal((out == x), ~m2 | (y2 == 0))
    # what the backward recomputes as "the activation the second GEMM saw" is the forward's: same bits through the same function
    da, db, act_k = ops.bias_act_dropout_bwd(torch.ones(rows, hidden, device=DEV, dtype=torch.bfloat16), None, p1, a=pre, act=1,
                                             seed=s1 if p1 > 0 else None, want_act=True)
    close(act_k.float(), act.float(), 1e-2, 4e-3)
    assert torch.equal(act_k == 0, (~m1) | (act_k == 0))
    assert torch.equal(da == 0, (~m1) | (da == 0))


@pytest.mark.parametrize("rows,hidden", [(300, 1024), (64, 256), (1000, 2048)])
@pytest.mark.parametrize("p1,p2", [(0.0, 0.0), (0.1, 0.2)])
def test_ffn_bwd_fused_against_the_kernel_per_stage_chain(rows, hidden, p1, p2):
    """cm_ffn_bwd_fused (da2 -> dg -> da1 / recomputed activation -> dh, both bias gradients, one kernel) against the same chain as
    separate launches (cm_bias_act_dropout_bwd x 2 around two library GEMMs) with the same dropout seeds: da2 and the activation bit
    for bit (same arithmetic), everything behind a GEMM within bf16 rounding of the different summation orders."""
    from mamba_asr_amd import ops
    g = torch.Generator().manual_seed(rows * 3 + hidden)
    D = 256
    dout = torch.randn(rows, D, generator=g).to(DEV)
    pre = torch.randn(rows, hidden, generator=g).bfloat16().to(DEV)
    w1 = (torch.randn(hidden, D, generator=g) / 16).to(DEV)
    w2 = (torch.randn(D, hidden, generator=g) / (hidden ** 0.5)).to(DEV)
    s1, s2, alpha = 99 + rows, 77 + hidden, 0.5
    w1c, w2c = w1.bfloat16(), w2.bfloat16()
    da2, da1, act, dh, db1, db2 = ops.ffn_bwd_fused(dout, ops.PackedWeight(w2c.t().contiguous()), ops.PackedWeight(w1c.t().contiguous()), pre, alpha, p1, p2, s1, s2)
    torch.cuda.synchronize()
    ra2, rb2 = ops.bias_act_dropout_bwd(dout, None, p2, act=0, alpha=alpha, out_dtype=torch.bfloat16, seed=s2 if p2 > 0 else None)
    rdg = torch.mm(ra2, w2c)
    ra1, rb1, ract = ops.bias_act_dropout_bwd(rdg, None, p1, a=pre, act=1, seed=s1 if p1 > 0 else None, want_act=True)
    rdh = torch.mm(ra1, w1c)
    assert torch.equal(da2, ra2) and torch.equal(act, ract)
    close(db2, rb2, 1e-5, 1e-5)
    close(da1.float(), ra1.float(), 2e-2, 1e-2)
    assert torch.equal(da1 == 0, ra1 == 0) or float(((da1 == 0) != (ra1 == 0)).float().mean()) < 1e-3      # same dropout decisions
    close(dh.float(), rdh.float(), 2e-2, 1.5e-2)
    close(db1, rb1, 1e-2, 4e-3)
    # fp64 reference of the chain on the same masks (read back through act / da1 zeros is not exact: use the kernel chain's masks)
    m1 = _drop_masks(ops, rows, hidden, p1, s1).double() / (1 - round(p1 * 65536) / 65536.0) if p1 > 0 else 1.0
    x = pre.double()
    gp = 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5
    ref_da1 = (da2.double() @ w2c.double()) * m1 * gp
    close(da1.float(), ref_da1, 3e-2, 1e-2)
    close(dh.float(), ref_da1.bfloat16().double() @ w1c.double() if False else ref_da1 @ w1c.double(), 3e-2, 1.5e-2)


@pytest.mark.parametrize("p", [0.0, 0.15])
def test_ffn_rows_fused_training_node_gradients(p, monkeypatch):
    """FfnRowsFn on the fused training forward (bf16 autocast, d_model 256) with dropout live: output and every gradient against
    torch autograd on the same function with the SAME masks (read back from the dropout stream with the seeds the node drew),
    in fp64."""
    import torch.nn as nn
    from mamba_asr_amd import ops
    from mamba_asr_amd.modules import Conmamba as CM
    from mamba_asr_amd.modules import ffn_rows
    assert ffn_rows.FUSED_TRAIN
    torch.manual_seed(11)
    layer = CM.ConmambaEncoderLayer(d_model=256, d_ffn=1024, kernel_size=31, activation=nn.GELU, bias=True, dropout=p, causal=False,
                                    mamba_config={"d_state": 16, "expand": 2, "d_conv": 4, "bidirectional": True}).to(DEV)
    layer.train()
    mod = layer.ffn_module2
    with torch.no_grad():
        for p_ in mod.parameters():
            if p_.dim() == 1:
                p_.add_(0.1 * torch.randn_like(p_))
    B, T, D, F_ = 3, 50, 256, 1024
    x = torch.randn(B, T, D, device=DEV)
    dy = torch.randn(B, T, D, device=DEV)
    seeds = iter([424242, 171717])
    drawn = []
    monkeypatch.setattr(ops, "draw_seed", lambda: (drawn.append(next(seeds)), drawn[-1])[1])
    xg = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        got = layer._ffn(mod, xg)
    params = list(mod.parameters())
    ggot = torch.autograd.grad(got, [xg] + params, dy)
    monkeypatch.undo()
    assert len(drawn) == (2 if p > 0 else 0)
    ln = mod[0]
    sc = 1.0 / (1.0 - round(p * 65536) / 65536.0)
    m1 = _drop_masks(ops, B * T, F_, p, drawn[0]).view(B, T, F_).double() * sc if p > 0 else 1.0
    m2 = _drop_masks(ops, B * T, D, p, drawn[1]).view(B, T, D).double() * sc if p > 0 else 1.0
    ps = [q.detach().double().requires_grad_(True) for q in params]
    xr = x.double().requires_grad_(True)
    # parameters come as (ln.weight, ln.bias, lin1.weight, lin1.bias, lin2.weight, lin2.bias)
    lnw, lnb, w1, b1, w2, b2 = ps
    h = torch.nn.functional.layer_norm(xr, (D,), lnw, lnb, ln.eps)
    want = xr + 0.5 * ((torch.nn.functional.gelu(h @ w1.t() + b1) * m1) @ w2.t() + b2) * m2
    gref = torch.autograd.grad(want, [xr] + ps, dy.double())
    close(got, want.detach(), 2e-2, 1.5e-2)
    for a_, b_ in zip(ggot, gref):
        close(a_, b_, 5e-2, 1.5e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("causal", [False, True])
def test_convmod_rows_node_vs_module_tree(dtype, causal):
    """modules/convmod_rows.ConvModuleRowsFn against the module tree (reference modules/Conmamba.py:439-449 semantics, pinned by
    G4) run by torch in fp64 on the CPU: x + convolution_module(x) and every gradient."""
    import copy
    import torch.nn as nn
    from mamba_asr_amd.modules import Conmamba as CM
    from mamba_asr_amd.modules import convmod_rows
    torch.manual_seed(4)
    cm = CM.ConvolutionModule(256, 31, True, nn.GELU, 0.0, causal=causal)
    with torch.no_grad():
        for p_ in cm.parameters():
            if p_.dim() == 1:
                p_.add_(0.1 * torch.randn_like(p_))
    x, dy = torch.randn(3, 50, 256), torch.randn(3, 50, 256)
    ref = copy.deepcopy(cm).double()
    xr = x.double().requires_grad_(True)
    want = xr + ref(xr)
    gref = torch.autograd.grad(want, [xr] + list(ref.parameters()), dy.double())
    cm = cm.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    assert convmod_rows.supported(cm, xg)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
        got = convmod_rows.convmod_rows(cm, xg)
    ggot = torch.autograd.grad(got, [xg] + list(cm.parameters()), dy.to(DEV))
    rt, at = (2e-4, 2e-5) if dtype == torch.float32 else (2e-2, 1.5e-2)
    close(got, want.detach(), rt, at)
    for (k, _), a_, b_ in zip([("x", None)] + list(cm.named_parameters()), ggot, gref):
        close(a_, b_, rt * 5, at * 2)
