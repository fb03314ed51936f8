#!/usr/bin/env python3
"""cm_ln_pw_glu vs cm_add_layernorm + library pointwise GEMM (rows = B x 1000, d_model 256), hipGraph-timed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd import ops
from bench_ffn import timeit

dev = "cuda"
for b in (32, 64):
    rows, D = b * 1000, 256
    x = torch.randn(rows, D, device=dev)
    y = (torch.randn(rows, D, device=dev) * 0.5).bfloat16()
    ln = (torch.ones(D, device=dev), torch.zeros(D, device=dev), 1e-5)
    w = (torch.randn(2 * D, D, device=dev) / 16).bfloat16()
    bias = torch.randn(2 * D, device=dev) * 0.1
    wp = ops.PackedWeight(w)
    bb = bias.bfloat16()
    t1 = timeit(lambda: ops.ln_pw_glu(x, y, 0.0, ln, wp, bias))

    def lib():
        _, h = ops.add_layernorm(x, y, 0.0, x_out=x, norm2=ln, out_dtype=torch.bfloat16)
        return torch.addmm(bb, h, w.t())
    t2 = timeit(lib)
    mb = rows * (D * 4 * 2 + D * 2 * 2) / 1e6
    print(f"B={b}: ln_pw_glu {t1:6.1f} us ({mb / t1:5.2f} TB/s on {mb:.0f} MB)   add_ln + GEMM {t2:6.1f} us", flush=True)
