"""The element-wise kernels of the feed-forward module's training step (csrc/ffn_train.hip) at ConMamba-large sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd import ops

dev, dt = "cuda", torch.bfloat16


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


if __name__ == "__main__":
    rows, F_, D = 32000, 1024, 256
    pre = torch.randn(rows, F_, device=dev).to(dt)
    dg = torch.randn(rows, F_, device=dev).to(dt)
    dout = torch.randn(rows, D, device=dev)
    t = timeit(lambda: ops.bias_act_dropout_bwd(dg, None, 0.1, a=pre, act=1, seed=1234, want_act=True))
    print(f"FFN middle backward (dg, pre -> da1, act, db1), {rows} x {F_}: {t:6.1f} us  ({4 * rows * F_ * 2 / t / 1e6:5.2f} TB/s)")
    t = timeit(lambda: ops.bias_act_dropout_bwd(dg, None, 0.1, a=pre, act=1, seed=1234))
    print(f"  without the recomputed activation: {t:6.1f} us  ({3 * rows * F_ * 2 / t / 1e6:5.2f} TB/s)")
    t = timeit(lambda: ops.bias_act_dropout_bwd(dout, None, 0.1, act=0, alpha=0.5, out_dtype=dt, seed=77))
    print(f"FFN output backward (dout fp32 -> da2 bf16, db2), {rows} x {D}: {t:6.1f} us  ({rows * D * 6 / t / 1e6:5.2f} TB/s)")
    x = torch.randn(rows, D, device=dev)
    w1 = (torch.randn(F_, D, device=dev) / 16).to(dt); w2 = (torch.randn(D, F_, device=dev) / 32).to(dt)
    b1, b2 = torch.zeros(F_, device=dev), torch.zeros(D, device=dev)
    ln = (torch.ones(D, device=dev), torch.zeros(D, device=dev), 1e-5)
    p1, p2 = ops.PackedWeight(w1), ops.PackedWeight(w2)
    out = torch.empty_like(x)
    t = timeit(lambda: ops.ffn_fused(x, ln, p1, b1, p2, b2, alpha=0.5, x_out=out, train=(0.1, 0.1, 5, 6)))
    print(f"cm_ffn_fused training forward, {rows} rows: {t:6.1f} us")
    t = timeit(lambda: ops.ffn_fused(x, ln, p1, b1, p2, b2, alpha=0.5, x_out=out, want_h=False))
    print(f"cm_ffn_fused inference forward, {rows} rows: {t:6.1f} us")
    w2t, w1t = ops.PackedWeight(w2.t().contiguous()), ops.PackedWeight(w1.t().contiguous())
    t = timeit(lambda: ops.ffn_bwd_fused(dout, w2t, w1t, pre, 0.5, 0.1, 0.1, 5, 6))
    print(f"cm_ffn_bwd_fused (dout -> da2, da1, act, dh, db1, db2), {rows} rows: {t:6.1f} us")

    def chain():
        a2, b2_ = ops.bias_act_dropout_bwd(dout, None, 0.1, act=0, alpha=0.5, out_dtype=dt, seed=6)
        g_ = torch.mm(a2, w2)
        a1, b1_, ac = ops.bias_act_dropout_bwd(g_, None, 0.1, a=pre, act=1, seed=5, want_act=True)
        return torch.mm(a1, w1)
    print(f"  the same chain as separate launches: {timeit(chain):6.1f} us")
