"""cm_ffn_fused at 64k rows: kernel variants selected with cm_debug_set (0 = the kernel as shipped, 25 = residual rows read
straight into the accumulator layout, 26 = stream rows stored from it; 1-5 the round-1 ablations)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd import ops, _native

dev, rows, D, F = "cuda", int(os.environ.get("ROWS", 64000)), 256, 1024
x = torch.randn(rows, D, device=dev)
add = (torch.randn(rows, D, device=dev) * 0.5).bfloat16()
n1 = (torch.ones(D, device=dev), torch.zeros(D, device=dev), 1e-5)
w1 = ops.PackedWeight((torch.randn(F, D, device=dev) / 16).bfloat16())
w2 = ops.PackedWeight((torch.randn(D, F, device=dev) / 32).bfloat16())
b1, b2 = torch.randn(F, device=dev) * 0.1, torch.randn(D, device=dev) * 0.1
lib = _native.lib()
ffn2 = lambda: ops.ffn_fused(x, n1, w1, b1, w2, b2, alpha=0.5, addend=add, norm1=n1, want_h=False)      # second FFN of a layer
ffn1 = lambda: ops.ffn_fused(x, n1, w1, b1, w2, b2, alpha=0.5, norm2=n1)                                 # first FFN (h out)
for variant in [int(v) for v in os.environ.get("VARIANTS", "0,25,26,0,25,26").split(",")]:
    lib.cm_debug_set(variant)
    out = []
    for fn in (ffn1, ffn2):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 50)
    print("variant %2d: ffn1 %.1f us  ffn2 %.1f us" % (variant, out[0], out[1]), flush=True)
lib.cm_debug_set(0)
