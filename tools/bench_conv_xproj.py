#!/usr/bin/env python3
"""cm_conv_xproj vs cm_conv_cl_fwd + library x_proj GEMM at ConMamba-large shapes (E=512, T=1000, bf16)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd import ops
from bench_ffn import timeit

dev = "cuda"
e, l = 512, 1000
for b in (16, 32, 64):
    xz = torch.randn(b, l, 2 * e, device=dev).bfloat16()
    x = xz[:, :, :e]
    wf, wb = torch.randn(e, 4, device=dev) * 0.5, torch.randn(e, 4, device=dev) * 0.5
    bf, bb = torch.randn(e, device=dev) * 0.1, torch.randn(e, device=dev) * 0.1
    wx = [(torch.randn(48, e, device=dev) * 0.1).bfloat16() for _ in range(2)]
    wbd = torch.zeros(96, 2 * e, device=dev, dtype=torch.bfloat16)
    wbd[:48, :e], wbd[48:, e:] = wx[0], wx[1]
    pk = [ops.PackedWeight(w) for w in wx]
    ucat = torch.empty(b, l, 2 * e, device=dev, dtype=torch.bfloat16)
    xdbl = torch.empty(b, l, 96, device=dev, dtype=torch.bfloat16)
    t1 = timeit(lambda: ops.conv_xproj(x, wf, bf, wb, bb, pk[0], pk[1], out_f=ucat[:, :, :e], out_b=ucat[:, :, e:], xdbl=xdbl))
    # the same call cycling through buffer sets whose total exceeds the 256 MB memory-side cache: what the kernel sees in the
    # encoder, where its input was written and its output is read by other kernels with ~1 GB of traffic in between
    from mamba_asr_amd import _native
    nset = max(2, int(1.2e9 / (b * l * (2 * e * 2 * 2 + 192))))
    sets = [(torch.randn(b, l, 2 * e, device=dev).bfloat16(), torch.empty(b, l, 2 * e, device=dev, dtype=torch.bfloat16),
             torch.empty(b, l, 96, device=dev, dtype=torch.bfloat16)) for _ in range(nset)]
    state = {"i": 0}

    def cold():
        xs, us, xd = sets[state["i"] % nset]
        state["i"] += 1
        ops.conv_xproj(xs[:, :, :e], wf, bf, wb, bb, pk[0], pk[1], out_f=us[:, :, :e], out_b=us[:, :, e:], xdbl=xd)
    res = []
    for dbg in (0, 16, 0, 16):
        _native.lib().cm_debug_set(dbg)
        res.append(timeit(cold, iters=4 * nset))
    _native.lib().cm_debug_set(0)
    print(f"B={b}: cold buffers ({nset} sets): default tile {res[0]:6.1f} / {res[2]:6.1f} us, 16-step tile {res[1]:6.1f} / {res[3]:6.1f} us", flush=True)
    del sets
    t2 = timeit(lambda: ops.conv_cl_fwd(x, wf, bf, wb, bb, True, out_f=ucat[:, :, :e], out_b=ucat[:, :, e:]))
    t3 = timeit(lambda: torch.matmul(ucat.view(-1, 2 * e), wbd.t()))
    mb = b * l * (e * 2 + 2 * e * 2 + 192) / 1e6
    print(f"B={b}: conv_xproj {t1:6.1f} us ({mb / t1:5.2f} TB/s on {mb:.0f} MB)   conv_cl {t2:6.1f} us + x_proj GEMM {t3:6.1f} us", flush=True)
