"""cm_conv_cl_bwd (both directions' causal conv + SiLU backward, dz = dz_f + dz_b) at ConMamba-large training sizes."""
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
    for b, l, d in ((32, 1000, 512), (16, 1000, 512), (4, 4000, 1024)):
        r = lambda *s: torch.randn(*s, device=dev)
        x, duf, dub, dzf, dzb = (r(b, l, d).to(dt) for _ in range(5))
        wf, wb, bf, bb = r(d, 4) / 2, r(d, 4) / 2, r(d) * 0.1, r(d) * 0.1
        t = timeit(lambda: ops.conv_cl_bwd(x, wf, bf, duf, wb, bb, dub, dzf, dzb))
        byt = 7 * b * l * d * 2
        print(f"{b} x {l} x {d}: {t:7.1f} us  ({byt / t / 1e6:5.2f} TB/s of 7 row tensors)", flush=True)
