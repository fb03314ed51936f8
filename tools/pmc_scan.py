#!/usr/bin/env python3
"""Runs only the dominant kernel (cm_scan_cl_fwd, ConMamba-large shapes, both directions, in-kernel dt_proj) a few
times, for `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (profiles/r01/pmc_scan_*.csv)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd import ops

b, l, e, n, r = int(os.environ.get("B", 16)), 1000, 512, 16, 16
dev, dt = "cuda", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
xz = torch.randn(b, l, 2 * e, device=dev, generator=g).to(dt)
ucat = torch.randn(b, l, 2 * e, device=dev, generator=g).to(dt)
ycat = torch.empty_like(ucat)
dirs = []
for i in range(2):
    feat = ops.alloc_bc(r + 2 * n, b, l, dev)
    feat.normal_(generator=g)
    dirs.append(dict(u=ucat[:, :, i * e:(i + 1) * e], A=-torch.rand(e, n, device=dev, generator=g) - 0.1, B=feat[r:r + n],
                     C=feat[r + n:], dt_low=feat[:r], dt_weight=torch.randn(e, r, device=dev, generator=g) * 0.2,
                     D=torch.ones(e, device=dev), delta_bias=torch.zeros(e, device=dev) - 2, out=ycat[:, :, i * e:(i + 1) * e],
                     reverse=bool(i)))
for _ in range(5):
    ops.scan_cl_fwd(dirs, z=xz[:, :, e:])
torch.cuda.synchronize()
print("alg bytes per launch (4E+2N)*s*B*T*2 =", (4 * e + 2 * n) * 2 * b * l * 2)
