import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd import ops
dev, rows, D, F = "cuda", 64000, 256, 1024
x = torch.randn(rows, D, device=dev)
n1 = (torch.ones(D, device=dev), torch.zeros(D, device=dev), 1e-5)
w1 = ops.PackedWeight((torch.randn(F, D, device=dev) / 16).bfloat16())
w2 = ops.PackedWeight((torch.randn(D, F, device=dev) / 32).bfloat16())
b1, b2 = torch.randn(F, device=dev) * 0.1, torch.randn(D, device=dev) * 0.1
win = (torch.randn(1024, D, device=dev) / 16).bfloat16()
wp = ops.PackedWeight(win)
out = torch.empty(rows, 1024, dtype=torch.bfloat16, device=dev)
def sep():
    _, h = ops.ffn_fused(x, n1, w1, b1, w2, b2, alpha=0.5, norm2=n1)
    torch.mm(h, win.t(), out=out)
def fus():
    ops.ffn_fused(x, n1, w1, b1, w2, b2, alpha=0.5, norm2=n1, proj_w=wp, proj_out=out)
for name, fn in (("ffn1 + library in_proj", sep), ("ffn1 with projection", fus)) * 2:
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) * 50:.1f} us")
