"""Weight gradients as batched GEMMs over row chunks + cm_sum_leading: how many chunks?  dW (M, N) = sum_k A[k, m] B[k, n] with
A (rows, M), B (rows, N) bf16, rows = 32000 (ConMamba-large, 32 x 40 s)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd import ops

dev, dt = "cuda", torch.bfloat16


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
    rows = 32000
    for M, N in ((1024, 256), (256, 1024), (256, 256), (512, 256), (1024, 1024)):
        A = torch.randn(rows, M, device=dev).to(dt)
        B = torch.randn(rows, N, device=dev).to(dt)
        line = f"dW ({M:4d}, {N:4d}): "
        for nb in (32, 64):
            if rows % nb:
                continue
            k = rows // nb
            f = lambda: ops.sum_leading(torch.bmm(A.view(nb, k, M).transpose(1, 2), B.view(nb, k, N)))
            line += f" {nb:3d} chunks {timeit(f):6.1f} us |"
        line += f" one GEMM {timeit(lambda: torch.mm(A.t(), B)):6.1f} us | cm_wgrad_bf16 4-stage LDS-DMA ring {timeit(lambda: ops.wgrad(A, B, variant=2)):6.1f} us, 2-buffer LDS-DMA {timeit(lambda: ops.wgrad(A, B, variant=3)):6.1f} us, through registers {timeit(lambda: ops.wgrad(A, B, variant=1)):6.1f} us"
        print(line, flush=True)
