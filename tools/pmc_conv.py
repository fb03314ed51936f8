#!/usr/bin/env python3
"""cm_conv_xproj alone at ConMamba-large shapes for rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd import ops
dev = "cuda"
e, l, b = 512, 1000, int(os.environ.get("B", 64))
xz = torch.randn(b, l, 2 * e, device=dev).bfloat16()
x = xz[:, :, :e]
wf, wb = torch.randn(e, 4, device=dev) * 0.5, torch.randn(e, 4, device=dev) * 0.5
bf, bb = torch.randn(e, device=dev) * 0.1, torch.randn(e, device=dev) * 0.1
pk = [ops.PackedWeight((torch.randn(48, e, device=dev) * 0.1).bfloat16()) for _ in range(2)]
ucat = torch.empty(b, l, 2 * e, device=dev, dtype=torch.bfloat16)
for _ in range(5):
    ops.conv_xproj(x, wf, bf, wb, bb, pk[0], pk[1], out_f=ucat[:, :, :e], out_b=ucat[:, :, e:])
    ops.conv_cl_fwd(x, wf, bf, wb, bb, True, out_f=ucat[:, :, :e], out_b=ucat[:, :, e:])
torch.cuda.synchronize()
