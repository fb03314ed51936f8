#!/usr/bin/env python3
"""Weight-gradient reduction of the training path: sum_b dy_b^T x_b as a batched product over G groups of utterances + a sum
over the groups (sb_compat._LinearRowsFn.backward), G from the batch size down to 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_ffn import timeit
dev = "cuda"
B, T = int(os.environ.get("B", 32)), 1000
for n, k in ((1024, 256), (256, 1024), (256, 256), (1024, 512)):
    dy = torch.randn(B, T, n, device=dev).bfloat16()
    x = torch.randn(B, T, k, device=dev).bfloat16()
    line = f"B={B} N={n} K={k}:"
    for g in (B, 16, 8, 4, 2, 1):
        if B % g:
            continue
        f = lambda: torch.bmm(dy.reshape(g, -1, n).transpose(1, 2), x.reshape(g, -1, k)).sum(0)
        line += f"  G={g}: {timeit(f, iters=10):6.1f} us"
    print(line, flush=True)
