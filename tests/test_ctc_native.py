"""cm_ctc_loss (csrc/ctc.hip) vs torch.nn.functional.ctc_loss in fp64 on the CPU (what speechbrain's ctc_loss wraps, reference
train_CTC.py:405): per-utterance negative log-likelihoods and the gradient w.r.t. the log-probabilities, ragged input / target
lengths, repeated labels, an utterance with no valid alignment (zero_infinity), empty targets; bit-reproducible."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ref(lp, tg, il, tl, blank=0):
    lp64 = lp.double().requires_grad_(True)
    nll = F.ctc_loss(lp64.transpose(0, 1), tg, il, tl, blank, reduction="none", zero_infinity=True)
    (g,) = torch.autograd.grad(nll.sum(), lp64)
    return nll.detach(), g


@pytest.mark.parametrize("b,t,v,s,seed", [(4, 50, 31, 12, 0), (3, 200, 31, 60, 1), (2, 1000, 31, 500, 2), (5, 64, 8, 20, 3), (2, 30, 5000, 9, 4), (2, 600, 5000, 400, 5),
                                          (3, 900, 1000, 511, 6)])
def test_ctc_loss_and_gradient_vs_torch(b, t, v, s, seed):
    from mamba_asr_amd import ops
    g = torch.Generator().manual_seed(seed)
    lp = torch.log_softmax(torch.randn(b, t, v, generator=g) * 2, -1)
    # few classes: many repeated neighbours; the long-target cases (seed >= 5) draw from 300 classes: repeats at a distance
    tg = torch.randint(1, min(v, 6 if seed < 5 else 300), (b, s), generator=g)
    il = torch.tensor([t - (7 * i) % max(1, t // 3) for i in range(b)], dtype=torch.int32)
    tl = torch.tensor([max(0, s - (5 * i) % (s + 1)) for i in range(b)], dtype=torch.int32)
    if b >= 3:
        il[1], tl[1] = min(t, 5), min(s, 12)                                    # more labels than frames: no alignment
        tl[2] = 0                                                               # empty target
    want_nll, want_g = _ref(lp, tg, il, tl)
    nll, grad = ops.ctc_loss_grad(lp.to(DEV), tg.to(DEV), il.to(DEV), tl.to(DEV))
    torch.testing.assert_close(nll.cpu().double(), want_nll, rtol=2e-5, atol=2e-4)
    # fp32 log-space tables: alpha + beta + nll - lp cancels numbers of size |nll| (up to ~2000 here), so a posterior carries
    # an absolute error of a few ulp(|nll|); torch's own fp32 GPU kernel is compared on the same footing below
    tol = 2e-5 + 1.5e-6 * float(want_nll.abs().max())
    torch.testing.assert_close(grad.cpu().double(), want_g, rtol=1e-4, atol=tol)
    lpg = lp.to(DEV).requires_grad_(True)
    tnll = F.ctc_loss(lpg.transpose(0, 1), tg.to(DEV), il.to(DEV), tl.to(DEV), 0, reduction="none", zero_infinity=True)
    (tgrad,) = torch.autograd.grad(tnll.sum(), lpg)
    err_native = float((grad.cpu().double() - want_g).abs().max())
    err_torch = float((tgrad.cpu().double() - want_g).abs().max())
    print(f"CTC gradient max |error| vs fp64: native {err_native:.2e}, torch fp32 GPU kernel {err_torch:.2e}")
    assert err_native <= max(3.0 * err_torch, tol)
    nll2, grad2 = ops.ctc_loss_grad(lp.to(DEV), tg.to(DEV), il.to(DEV), tl.to(DEV))
    assert torch.equal(nll, nll2) and torch.equal(grad, grad2)


def test_sb_ctc_loss_wrapper_native_vs_torch(monkeypatch):
    """sb_compat.ctc_loss ('batchmean', relative lengths) through the native op == through torch, value and gradient."""
    from mamba_asr_amd import sb_compat as sb
    g = torch.Generator().manual_seed(9)
    lp0 = torch.log_softmax(torch.randn(4, 120, 31, generator=g), -1)
    tg = torch.randint(1, 31, (4, 25), generator=g)
    il, tl = torch.tensor([1.0, 0.8, 0.55, 0.9]), torch.tensor([1.0, 0.6, 0.4, 0.8])
    res = {}
    for native in (True, False):
        monkeypatch.setattr(sb, "USE_NATIVE_CTC", native)
        lp = lp0.to(DEV).requires_grad_(True)
        loss = sb.ctc_loss(lp, tg.to(DEV), il.to(DEV), tl.to(DEV), 0, reduction="batchmean")
        (gr,) = torch.autograd.grad(loss, lp)
        res[native] = (loss.detach(), gr)
    torch.testing.assert_close(res[True][0], res[False][0], rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(res[True][1], res[False][1], rtol=1e-4, atol=1e-5)
