#!/usr/bin/env python3
"""Kernel-level timing of the native ops at the BASELINE config-3 shapes (ConMamba-large:
E=512, N=16, B=16, T=1000).  Prints ms, algorithmic GB/s ((4E+2N)*s bytes per scan step)
and the fraction of the 8 TB/s HBM peak, per lane split."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd import ops, _native


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
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--len", type=int, default=1000)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    dt = {"bf16": torch.bfloat16, "f32": torch.float32}[a.dtype]
    dev = "cuda"
    b, e, l, n = a.batch, a.dim, a.len, 16
    g = torch.Generator(device=dev).manual_seed(0)
    u = torch.randn(b, e, l, device=dev, generator=g).to(dt)
    delta = (torch.randn(b, e, l, device=dev, generator=g) * 0.5).to(dt)
    z = torch.randn(b, e, l, device=dev, generator=g).to(dt)
    A = -torch.exp(torch.randn(e, n, device=dev, generator=g) * 0.3)
    B = torch.randn(b, 1, n, l, device=dev, generator=g).to(dt)
    C = torch.randn(b, 1, n, l, device=dev, generator=g).to(dt)
    D = torch.randn(e, device=dev, generator=g)
    bias = torch.randn(e, device=dev, generator=g) - 1
    s = u.element_size()
    bytes_fwd = b * l * (4 * e + 2 * n) * s
    lib = _native.lib()
    for split in (0, 1, 2, 4, 8, 16):
        ms = timeit(lambda: ops.selective_scan_fwd(u, delta, A, B, C, D, z, bias, True, need_out=False, need_x=False, split=split))
        ms_r = timeit(lambda: ops.selective_scan_fwd(u, delta, A, B, C, D, z, bias, True, reverse=True, need_out=False, need_x=False, split=split))
        print(f"scan_fwd split={split:2d}: fwd {ms*1e3:8.1f} us  rev {ms_r*1e3:8.1f} us  {bytes_fwd/ms/1e6:8.1f} GB/s "
              f"= {bytes_fwd/ms/1e6/8000*100:5.1f}% of 8 TB/s")
    # both directions concurrently on two streams
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    cur = [0]
    def both():
        s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s1):
            ops.selective_scan_fwd(u, delta, A, B, C, D, z, bias, True, need_out=False, need_x=False, split=cur[0])
        with torch.cuda.stream(s2):
            ops.selective_scan_fwd(u, delta, A, B, C, D, z, bias, True, reverse=True, need_out=False, need_x=False, split=cur[0])
        torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    for split in (0, 4, 8, 16):
        cur[0] = split
        ms = timeit(both)
        print(f"scan_fwd both directions on 2 streams, split={split}: {ms*1e3:8.1f} us  {2*bytes_fwd/ms/1e6:8.1f} GB/s")
    # channels-last, both directions in one launch
    ucl, dcl = u.transpose(1, 2).contiguous(), delta.transpose(1, 2).contiguous()
    xz = torch.randn(b, l, 2 * e, device=dev, generator=g).to(dt)
    Bcl, Ccl = ops.alloc_bc(16, b, l, dev), ops.alloc_bc(16, b, l, dev)
    Bcl.normal_(generator=g), Ccl.normal_(generator=g)
    ycat = torch.empty(b, l, 2 * e, device=dev, dtype=dt)
    dirs = [dict(u=ucl, delta=dcl, A=A, B=Bcl, C=Ccl, D=D, delta_bias=bias, out=ycat[:, :, i * e:(i + 1) * e], reverse=bool(i))
            for i in range(2)]
    for split in (4, 8, 16):
        ms = timeit(lambda: ops.scan_cl_fwd(dirs, z=xz[:, :, e:], split=split))
        print(f"scan_cl_fwd both directions one launch, lanes/channel={split:2d}: {ms*1e3:8.1f} us  {2*bytes_fwd/ms/1e6:8.1f} GB/s "
              f"= {2*bytes_fwd/ms/1e6/8000*100:5.1f}% of 8 TB/s")
    _, x, _ = ops.selective_scan_fwd(u, delta, A, B, C, D, z, bias, True, need_out=False)
    dout = torch.randn(b, e, l, device=dev, generator=g).to(dt)
    for split in (4, 8, 16):
        ms = timeit(lambda: ops.selective_scan_bwd(u, delta, A, B, C, D, z, bias, dout, x, True, split=split), iters=5)
        print(f"scan_bwd split={split:2d}: {ms*1e3:8.1f} us")
    if dt == torch.bfloat16:
        rows = b * l
        for (n_, k_, epi) in ((1024, 256, 1), (256, 1024, 2), (1024, 256, 0), (512, 256, 0), (256, 256, 2)):
            a_ = (torch.randn(rows, k_, device=dev, generator=g) * 0.5).to(dt)
            w_ = (torch.randn(n_, k_, device=dev, generator=g) * 0.05).to(dt)
            bias_ = torch.randn(n_, device=dev, generator=g)
            xres = torch.randn(rows, 256, device=dev, generator=g)
            ln = (torch.ones(256, device=dev), torch.zeros(256, device=dev), 1e-5)
            if epi == 2:
                ms = timeit(lambda: ops.gemm_bf16(a_, w_, bias_, epilogue=2, x=xres, alpha=0.5, norm2=ln))
            else:
                ms = timeit(lambda: ops.gemm_bf16(a_, w_, bias_, epilogue=epi))
            ms_l = timeit(lambda: torch.addmm(bias_.to(dt), a_, w_.t()))
            fl = 2.0 * rows * n_ * k_
            print(f"gemm M={rows} N={n_} K={k_} epi={epi}: native {ms*1e3:7.1f} us ({fl/ms/1e9:6.0f} TF/s)   library addmm {ms_l*1e3:7.1f} us ({fl/ms_l/1e9:6.0f} TF/s)")
    w = torch.randn(e, 4, device=dev, generator=g)
    cb = torch.randn(e, device=dev, generator=g)
    ms = timeit(lambda: ops.causal_conv1d_fwd(u, w, cb, True))
    print(f"conv_fwd: {ms*1e3:8.1f} us  {2*b*e*l*s/ms/1e6:8.1f} GB/s")
    ms = timeit(lambda: ops.causal_conv1d_bwd(u, w, cb, dout, True))
    print(f"conv_bwd: {ms*1e3:8.1f} us  {3*b*e*l*s/ms/1e6:8.1f} GB/s")


if __name__ == "__main__":
    main()
