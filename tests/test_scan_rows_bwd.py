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


def _case(ops, b, l, e, rank, dtype, seed):
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
        A = -torch.exp(torch.randn(e, 16, generator=gen) * 0.3)
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
    ops.scan_cl_fwd(dirs, z=gz, delta_softplus=True, time_chunks=1)
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


def test_scan_rows_fwd_training_outputs():
    """The training forward's extra outputs: ypre * silu(z) == out (up to the output rounding), checkpoints == the states a
    sequence cut at the half-block boundary carries (h_last of the prefix), for both directions."""
    from mamba_asr_amd import ops
    b, l, e = 2, 75, 64
    dirs, gz, refs, rank, P = _case(ops, b, l, e, 16, torch.float32, seed=5)
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
            ops.scan_cl_fwd([sub], z=gz[:, lo:hi], delta_softplus=True, time_chunks=1)
            torch.testing.assert_close(ck[:, m], hl, rtol=1e-5, atol=1e-6)
