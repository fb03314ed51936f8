#!/usr/bin/env python3
"""Runs only the dominant kernel (cm_scan_cl_fwd in xdbl mode = scan_rows_fwd_kernel, ConMamba-large shapes, both
directions, in-kernel dt_proj) a few times, for `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes
(profiles/r01/pmc_scan_rows_*.csv).  B=<utterances> selects the batch; OLD=1 runs the state-split kernel instead."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd import ops

b, l, e, n, r = int(os.environ.get("B", 64)), 1000, 512, 16, 16
old = os.environ.get("OLD", "0") == "1"
dev, dt = "cuda", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
xz = torch.randn(b, l, 2 * e, device=dev, generator=g).to(dt)
ucat = torch.randn(b, l, 2 * e, device=dev, generator=g).to(dt)
xdbl = torch.randn(b, l, 96, device=dev, generator=g).to(dt)
ycat = torch.empty_like(ucat)
dirs = []
for i in range(2):
    common = dict(u=ucat[:, :, i * e:(i + 1) * e], A=-torch.rand(e, n, device=dev, generator=g) - 0.1,
                  dt_weight=torch.randn(e, r, device=dev, generator=g) * 0.2, D=torch.ones(e, device=dev),
                  delta_bias=torch.zeros(e, device=dev) - 2, out=ycat[:, :, i * e:(i + 1) * e], reverse=bool(i))
    if old:
        feat = ops.alloc_bc(r + 2 * n, b, l, dev)
        feat.copy_(xdbl[:, :, 48 * i:48 * (i + 1)].float().permute(2, 0, 1))
        dirs.append(dict(common, B=feat[r:r + n], C=feat[r + n:], dt_low=feat[:r]))
    else:
        dirs.append(dict(common, xdbl=xdbl[:, :, 48 * i:48 * (i + 1)]))
for _ in range(5):
    ops.scan_cl_fwd(dirs, z=xz[:, :, e:])
torch.cuda.synchronize()
print("alg bytes per launch (4E+2N)*s*B*T*2 =", (4 * e + 2 * n) * 2 * b * l * 2)
