#!/usr/bin/env python3
"""cm_selective_scan_bwd alone (B x 512 x 1000, bf16) for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd import ops
b, e, l, n = int(os.environ.get("B", 32)), 512, 1000, 16
dev, dt = "cuda", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
u, delta, z, dout = rnd(b, e, l).to(dt), (rnd(b, e, l) * 0.5).to(dt), rnd(b, e, l).to(dt), rnd(b, e, l).to(dt)
A = -torch.exp(rnd(e, n) * 0.3)
B, C = rnd(b, 1, n, l).to(dt), rnd(b, 1, n, l).to(dt)
D, bias = rnd(e), rnd(e) - 1
_, x, _ = ops.selective_scan_fwd(u, delta, A, B, C, D, z, bias, True, need_out=False)
for _ in range(4):
    ops.selective_scan_bwd(u, delta, A, B, C, D, z, bias, dout, x, True)
torch.cuda.synchronize()
