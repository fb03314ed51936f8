#!/usr/bin/env python3
"""cm_ffn_fused alone (rows = B x 1000, d_model 256, hidden 1024) for rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd import ops
dev = "cuda"
rows, hidden = int(os.environ.get("B", 64)) * 1000, 1024
x = torch.randn(rows, 256, device=dev)
w1 = (torch.randn(hidden, 256, device=dev) / 16).bfloat16()
w2 = (torch.randn(256, hidden, device=dev) / 32).bfloat16()
b1, b2 = torch.randn(hidden, device=dev) * 0.1, torch.randn(256, device=dev) * 0.1
ln = (torch.ones(256, device=dev), torch.zeros(256, device=dev), 1e-5)
w1p, w2p = ops.PackedWeight(w1), ops.PackedWeight(w2)
for _ in range(5):
    ops.ffn_fused(x, ln, w1p, b1, w2p, b2, alpha=0.5, norm2=ln)
torch.cuda.synchronize()
