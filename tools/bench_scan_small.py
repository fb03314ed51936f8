"""Row-group scan forward at launches that leave SIMDs idle: 4 states per lane (scan_rows_fwd.hip), 2 states per lane
(scan_rows_fwd2.hip), and the chunked launch, us per launch (both directions, bf16, E = 512)."""
import os, sys
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from mamba_asr_amd import ops
from bench_scan_bwd import timeit
dev, dt = "cuda", torch.bfloat16
for (b, l, e) in ((16, 1000, 512), (8, 1000, 512), (24, 1000, 512), (32, 1000, 512), (64, 250, 288), (4, 4000, 512)):
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    xz, ucat, xcat = rnd(b, l, 2 * e).to(dt), rnd(b, l, 2 * e).to(dt), (rnd(b, l, 96) * 0.5).to(dt)
    ycat = torch.empty(b, l, 2 * e, device=dev, dtype=dt)
    dirs = [dict(u=ucat[:, :, e * i:e * (i + 1)], xdbl=xcat[:, :, 48 * i:48 * (i + 1)], A=-torch.exp(rnd(e, 16) * 0.06), D=torch.ones(e, device=dev),
                 delta_bias=rnd(e) - 4, dt_weight=ops.pad_dt_weight(rnd(e, 16) * 0.25), reverse=bool(i), out=ycat[:, :, e * i:e * (i + 1)]) for i in range(2)]
    z = xz[:, :, e:]
    res = {}
    for name, kw in (("4/lane", dict(split=4, time_chunks=1)), ("2/lane", dict(split=8, time_chunks=1)), ("auto", dict())):
        res[name] = timeit(lambda: ops.scan_cl_fwd(dirs, z=z, **kw))
    alg = 2 * b * l * (4 * e + 32) * 2
    print(f"{b:3d} x {l} x {e}: " + "  ".join(f"{k} {v:6.1f} us ({alg / v / 1e6 / 8 * 100:4.1f} %)" for k, v in res.items()), flush=True)
