"""cm_dwconv_cl_fwd / _bwd (training path's depthwise conv on rows) at ConMamba-large sizes, us per call."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from mamba_asr_amd import ops
from bench_scan_bwd import timeit
dev = "cuda"
for dt in (torch.bfloat16, torch.float32):
    for (b, l, d) in ((32, 1000, 256), (4, 4000, 512), (64, 250, 144)):
        x = torch.randn(b, l, d, device=dev).to(dt); dy = torch.randn(b, l, d, device=dev).to(dt)
        w, bias = torch.randn(d, 31, device=dev) / 6, torch.randn(d, device=dev) * 0.1
        tf = timeit(lambda: ops.dwconv_cl_fwd(x, w, bias))
        tb = timeit(lambda: ops.dwconv_cl_bwd(x, w, dy))
        byt = b * l * d * x.element_size()
        print(f"{dt} {b} x {l} x {d}: fwd {tf:6.1f} us ({2 * byt / tf / 1e6:5.2f} TB/s)  bwd {tb:6.1f} us ({3 * byt / tb / 1e6:5.2f} TB/s)", flush=True)
