"""The front end's wide LayerNorm (+ LeakyReLU + channel mask) at block 1's training size: 32 x 2000 rows of 40 x 64 = 2560, bf16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd import ops

dev, dt = "cuda", torch.bfloat16


def timeit(fn, iters=10, warm=2):
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
    for (b, t, f, c) in ((32, 2000, 40, 64), (32, 1000, 20, 32)):
        x = torch.randn(b * t, f * c, device=dev).to(dt)
        w, bias = torch.ones(f * c, device=dev), torch.zeros(f * c, device=dev)
        mask = (torch.rand(b, c, device=dev) > 0.1).float() / 0.9
        dy = torch.randn(b * t, f * c, device=dev).to(dt)
        y, x2, stats = ops.layernorm_fwd(x, w, bias, 1e-5, dt, leaky_slope=0.01, chan_mask=mask, mask_rows=t)
        tf = timeit(lambda: ops.layernorm_fwd(x, w, bias, 1e-5, dt, leaky_slope=0.01, chan_mask=mask, mask_rows=t))
        tb = timeit(lambda: ops.layernorm_bwd(dy, x2, stats, w, 1e-5, bias=bias, leaky_slope=0.01, chan_mask=mask, mask_rows=t))
        n = b * t * f * c * 2
        print(f"{b * t} rows x {f * c}: forward {tf:7.1f} us ({2 * n / tf / 1e6:5.2f} TB/s), backward {tb:7.1f} us ({3 * n / tb / 1e6:5.2f} TB/s)", flush=True)
