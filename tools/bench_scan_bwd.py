"""cm_scan_cl_bwd (row-group backward, both directions per launch) and the training forward at ConMamba-large sizes:
us per launch, against the (B,E,T)-layout operator kernels (cm_selective_scan_bwd, one direction per launch)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from mamba_asr_amd import ops, _native as N

dev = "cuda"
dt = torch.bfloat16


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


if __name__ == "__main__":
  for (b, l, e, P) in ((32, 1000, 512, 16), (16, 1000, 512, 16), (64, 1000, 512, 16), (4, 4000, 1024, 32)):
      g = torch.Generator(device=dev).manual_seed(0)
      rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
      RW = P + 32
      xz = rnd(b, l, 2 * e).to(dt)
      ucat = rnd(b, l, 2 * e).to(dt)
      xcat = (rnd(b, l, 2 * RW) * 0.5).to(dt)
      dmix = rnd(b, l, e).to(dt)
      ycat = torch.empty(b, l, 2 * e, device=dev, dtype=dt)
      pcat = torch.empty(b, l, 2 * e, device=dev, dtype=dt)
      dirs = []
      for i in range(2):
          dirs.append(dict(u=ucat[:, :, e * i:e * (i + 1)], xdbl=xcat[:, :, RW * i:RW * (i + 1)], A=-torch.exp(rnd(e, 16) * 0.06),
                           D=torch.ones(e, device=dev), delta_bias=rnd(e) - 4, dt_weight=ops.pad_dt_weight(rnd(e, P) * 0.25), reverse=bool(i),
                           out=ycat[:, :, e * i:e * (i + 1)]))
      z = xz[:, :, e:]
      t_inf = timeit(lambda: ops.scan_cl_fwd(dirs, z=z, time_chunks=1))
      for i in range(2):
          dirs[i]["ypre"] = pcat[:, :, e * i:e * (i + 1)]
          dirs[i]["ckpt"] = torch.empty(ops.scan_ckpt_shape(b, l, e), device=dev)
      t_trn = timeit(lambda: ops.scan_cl_fwd(dirs, z=z, time_chunks=1))
      ducat = torch.empty(b, l, 2 * e, device=dev, dtype=dt)
      dzcat = torch.empty(b, l, 2 * e, device=dev, dtype=dt)
      dxcat = torch.empty(b, l, 2 * RW, device=dev, dtype=dt)
      for i in range(2):
          dirs[i].update(dout=dmix, du=ducat[:, :, e * i:e * (i + 1)], dz=dzcat[:, :, e * i:e * (i + 1)], dxdbl=dxcat[:, :, RW * i:RW * (i + 1)])
      t_bwd = timeit(lambda: ops.scan_cl_bwd(dirs, z), iters=10)
      t_one = timeit(lambda: ops.scan_cl_bwd(dirs, z, time_chunks=1), iters=10)
      nck = N.lib().cm_scan_cl_bwd_auto_chunks(b, l, e, 2)
      s = 2
      alg = 2 * b * l * (9 * e + 4 * 16) * s                     # SURVEY §8d's backward bytes per step and direction, both directions
      print(f"{b} x {l} x {e} (P {P}): forward {t_inf:7.1f} us, training forward (+ckpt, ypre) {t_trn:7.1f} us, backward (both directions, "
            f"reduce included, {nck} time chunk(s); uncut {t_one:7.1f} us) {t_bwd:7.1f} us = {t_bwd / 2:6.1f} us per direction; {alg / t_bwd / 1e6:6.2f} TB/s = {alg / t_bwd / 1e6 / 8 * 100:4.1f} % of 8 TB/s")
      if P == 16:
          # the operator-API kernels on (B, E, T)
          u = rnd(b, e, l).to(dt); delta = (rnd(b, e, l) * 0.5).to(dt); zz = rnd(b, e, l).to(dt)
          A = -torch.exp(rnd(e, 16) * 0.06); B = rnd(b, 1, 16, l).to(dt); C = rnd(b, 1, 16, l).to(dt)
          D = torch.ones(e, device=dev); bias = rnd(e) - 4
          _, x, _ = ops.selective_scan_fwd(u, delta, A, B, C, D, zz, bias, True, need_out=False)
          dout = rnd(b, e, l).to(dt)
          t_old = timeit(lambda: ops.selective_scan_bwd(u, delta, A, B, C, D, zz, bias, dout, x, True), iters=10)
          print(f"    operator-API cm_selective_scan_bwd, one direction: {t_old:7.1f} us")
