#!/usr/bin/env python3
"""Which kernels does each library GEMM of the fused layer launch?  Run under rocprofv3 --kernel-trace; markers are
tiny fill kernels with distinctive sizes."""
import torch
dev = "cuda"
rows, D, E = 64000, 256, 512
bf = torch.bfloat16
h = torch.randn(rows, D, device=dev).to(bf)
in_w = torch.randn(2 * E, D, device=dev).to(bf)
ycat = torch.randn(rows, 2 * E, device=dev).to(bf)
out_w = torch.randn(D, 2 * E, device=dev).to(bf)
pw_w, pw_b = torch.randn(2 * D, D, device=dev).to(bf), torch.randn(2 * D, device=dev).to(bf)
lin_w, lin_b = torch.randn(D, D, device=dev).to(bf), torch.randn(D, device=dev).to(bf)
src = torch.randn(rows, 640, device=dev).to(bf)
src_w, src_b = torch.randn(D, 640, device=dev).to(bf), torch.randn(D, device=dev).to(bf)
mark = torch.zeros(7, device=dev)
def m(): mark.fill_(1.0)
for _ in range(2):
    m(); a = h @ in_w.t()
    m(); b = ycat @ out_w.t()
    m(); c = torch.addmm(pw_b, h, pw_w.t())
    m(); d = torch.addmm(lin_b, h, lin_w.t())
    m(); e = torch.addmm(src_b, src, src_w.t())
    m()
torch.cuda.synchronize()
