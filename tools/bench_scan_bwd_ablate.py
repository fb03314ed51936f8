"""cm_scan_cl_bwd at 32 x 1000 x 512 bf16 through whichever library CM_LIB_PATH names: used with timing-only builds of
csrc/scan_rows_bwd.hip compiled with -DCM_BWD_ABL=n (1: no dB / dC reduction, 2: no d dt / ddt_weight MFMAs, 3: exponentials of
the recompute sweep replaced by a multiply-add, 4: no cross-wave sum / second barrier) -- results of those builds are wrong by
construction; profiles/r03/scan_bwd_ablations.log holds the numbers."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from mamba_asr_amd import ops
from bench_scan_bwd import timeit
dev, dt = "cuda", torch.bfloat16
b, l, e, P = 32, 1000, 512, 16
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
RW = 48
xz, ucat, xcat, dmix = rnd(b, l, 2 * e).to(dt), rnd(b, l, 2 * e).to(dt), (rnd(b, l, 2 * RW) * 0.5).to(dt), rnd(b, l, e).to(dt)
ycat, pcat = torch.empty(b, l, 2 * e, device=dev, dtype=dt), torch.empty(b, l, 2 * e, device=dev, dtype=dt)
dirs = [dict(u=ucat[:, :, e * i:e * (i + 1)], xdbl=xcat[:, :, RW * i:RW * (i + 1)], A=-torch.exp(rnd(e, 16) * 0.06), D=torch.ones(e, device=dev), delta_bias=rnd(e) - 4,
             dt_weight=ops.pad_dt_weight(rnd(e, P) * 0.25), reverse=bool(i), out=ycat[:, :, e * i:e * (i + 1)], ypre=pcat[:, :, e * i:e * (i + 1)],
             ckpt=torch.empty(ops.scan_ckpt_shape(b, l, e), device=dev)) for i in range(2)]
z = xz[:, :, e:]
ops.scan_cl_fwd(dirs, z=z, time_chunks=1)
for d in dirs: d["dout"] = dmix
print(os.environ.get("CM_LIB_PATH", "product"), f"{timeit(lambda: ops.scan_cl_bwd(dirs, z), iters=10):.1f} us")
