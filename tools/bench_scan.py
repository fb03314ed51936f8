#!/usr/bin/env python3
"""Scan kernels alone at ConMamba-large shapes (E=512, N=16, R=16, T=1000, both directions, bf16):
state-split kernel (scan_cl_fwd.hip, B/C/dt as fp32 (16, batch, T) rows) vs row-group kernel (scan_rows_fwd.hip, x_dbl rows).
Prints us per launch and the fraction of the HBM roofline on the algorithmic bytes (4E+2N)*s per step per direction."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd import ops


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    dev = "cuda"
    e, l = 512, int(os.environ.get("T", "1000"))
    for b in [int(x) for x in os.environ.get("BATCHES", "11,16,32,64").split(",")]:
        gen = torch.Generator(device=dev).manual_seed(1)
        xz = torch.randn(b, l, 2 * e, device=dev, generator=gen).bfloat16()
        ucat = torch.randn(b, l, 2 * e, device=dev, generator=gen).bfloat16()
        ycat = torch.empty_like(ucat)
        xdbl = (torch.randn(b, l, 96, device=dev, generator=gen) * 0.5).bfloat16()
        old, new = [], []
        for i in range(2):
            A = -torch.exp(torch.randn(e, 16, device=dev, generator=gen) * 0.06)
            Wdt = torch.randn(e, 16, device=dev, generator=gen) * 0.25
            D, bias = torch.ones(e, device=dev), torch.randn(e, device=dev, generator=gen) - 4
            feat = ops.alloc_bc(48, b, l, dev)
            feat.copy_(xdbl[:, :, 48 * i:48 * (i + 1)].float().permute(2, 0, 1))
            common = dict(u=ucat[:, :, i * e:(i + 1) * e], A=A, D=D, delta_bias=bias, out=ycat[:, :, i * e:(i + 1) * e], reverse=bool(i))
            old.append(dict(common, B=feat[16:32], C=feat[32:48], dt_low=feat[:16], dt_weight=Wdt))
            new.append(dict(common, xdbl=xdbl[:, :, 48 * i:48 * (i + 1)], dt_weight=Wdt))
        z = xz[:, :, e:]
        alg = b * l * 2 * (4 * e + 32) * 2
        ops.scan_cl_fwd(old, z=z)
        ref = ycat.clone()
        ops.scan_cl_fwd(new, z=z)
        err = (ycat.float() - ref.float()).abs().max().item()
        from mamba_asr_amd import _native
        abls = [int(x) for x in os.environ.get("ABL", "").split(",") if x]
        cases = [("state-split", old, 0), ("row-group", new, 0)] + [(f"row-group abl{a}", new, a) for a in abls]
        for name, dirs, abl in cases:
            _native.lib().cm_debug_set(abl)
            ms = timeit(lambda: ops.scan_cl_fwd(dirs, z=z))
            _native.lib().cm_debug_set(0)
            print(f"B={b:3d} T={l} {name:16s} {ms * 1e3:8.1f} us  {alg / ms / 1e6:7.1f} GB/s  frac={alg / ms / 1e6 / 8000:.3f}  (max|new-old|={err:.3g})", flush=True)


if __name__ == "__main__":
    main()
