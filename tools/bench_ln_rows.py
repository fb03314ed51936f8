#!/usr/bin/env python3
"""cm_layernorm_bwd on the training step's row shapes ((batch x time, d_model) fp32 stream, bf16 dy, fused residual gradient):
average launch time with the tensors cycled through more memory than the 256 MB memory-side cache, and the bytes moved.
With the ablation library (CM_LIB_PATH=.../libconmamba_hip_ablate.so) `--debug 51` runs the kernel on 512 workgroups (two rounds) and
`--debug 50` also without the cross-iteration prefetch."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mamba_asr_amd import ops, _native as N

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=32000)
ap.add_argument("--dim", type=int, default=256)
ap.add_argument("--debug", type=int, nargs="*", default=[0])
ap.add_argument("--iters", type=int, default=40)
a = ap.parse_args()
dev = "cuda"
nset = max(2, int(600e6 / (a.rows * a.dim * 14)) + 1)
sets = []
for i in range(nset):
    x = torch.randn(a.rows, a.dim, device=dev)
    dy = torch.randn(a.rows, a.dim, device=dev).bfloat16()
    dres = torch.randn(a.rows, a.dim, device=dev)
    w, b = torch.randn(a.dim, device=dev), torch.randn(a.dim, device=dev)
    y, x2, stats = ops.layernorm_fwd(x, w, b, 1e-5, torch.bfloat16)
    sets.append((dy, x2, stats, w, dres))
for dbg in a.debug:
    if dbg and hasattr(N.lib(), "cm_debug_set"):
        N.lib().cm_debug_set(dbg)
    elif dbg:
        print(f"debug {dbg}: the loaded library has no cm_debug_set (use the ablation build)"); continue
    for cfg, kw in (("dy bf16, x fp32, + dres", True), ("dy bf16, x fp32", False)):
        for s in sets:
            ops.layernorm_bwd(s[0], s[1], s[2], s[3], 1e-5, dres=s[4] if kw else None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(a.iters):
            s = sets[i % nset]
            ops.layernorm_bwd(s[0], s[1], s[2], s[3], 1e-5, dres=s[4] if kw else None)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        nbytes = a.rows * a.dim * (2 + 4 + 4 + (4 if kw else 0))
        print(f"debug {dbg:2d}  {cfg:26s} {a.rows} x {a.dim}: {us:7.1f} us per call (kernel + reduce + allocations)  {nbytes / us / 1e6:6.2f} TB/s")
    if dbg:
        N.lib().cm_debug_set(0)
