#!/usr/bin/env python3
"""cm_ffn_fused vs the library path (LayerNorm kernel + 2 GEMMs + residual/LayerNorm kernel) at the headline shapes:
rows = 32 utterances x 1000 encoder steps, d_model 256, hidden 1024.  Prints us per call and TFLOP/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd import ops


def timeit(fn, iters=20, warm=3, reps=5):
    """GPU time per call: `iters` calls captured in one hipGraph (no host launch overhead), replayed `reps` times."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(iters):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * reps) * 1e3


def main():
    dev = "cuda"
    variants = [int(v) for v in os.environ.get("VAR", "").split(",") if v]      # timing-only kernel variants (ffn_fused.hip)
    if variants:
        from mamba_asr_amd import _native
        rows, hidden = 64000, 1024
        x = torch.randn(rows, 256, device=dev)
        w1p = ops.PackedWeight((torch.randn(hidden, 256, device=dev) / 16).bfloat16())
        w2p = ops.PackedWeight((torch.randn(256, hidden, device=dev) / 32).bfloat16())
        b1, b2 = torch.randn(hidden, device=dev) * 0.1, torch.randn(256, device=dev) * 0.1
        ln = (torch.ones(256, device=dev), torch.zeros(256, device=dev), 1e-5)
        for _ in range(300):                                   # clocks settle: the first case measured was 5-7 % slow without this
            ops.ffn_fused(x, ln, w1p, b1, w2p, b2, alpha=0.5, norm2=ln)
        for v in [0] + variants + [0]:
            _native.lib().cm_debug_set(v)
            t = timeit(lambda: ops.ffn_fused(x, ln, w1p, b1, w2p, b2, alpha=0.5, norm2=ln))
            print(f"rows {rows}: variant {v}: {t:7.1f} us", flush=True)
        _native.lib().cm_debug_set(0)
        return
    for rows in (10667, 16000, 32000, 64000):
        hidden = 1024
        x = torch.randn(rows, 256, device=dev)
        w1 = (torch.randn(hidden, 256, device=dev) / 16).bfloat16()
        w2 = (torch.randn(256, hidden, device=dev) / 32).bfloat16()
        b1, b2 = torch.randn(hidden, device=dev) * 0.1, torch.randn(256, device=dev) * 0.1
        ln = (torch.ones(256, device=dev), torch.zeros(256, device=dev), 1e-5)
        add = torch.randn(rows, 256, device=dev).bfloat16()
        flops = 4.0 * rows * 256 * hidden
        w1p, w2p = ops.PackedWeight(w1), ops.PackedWeight(w2)
        t = timeit(lambda: ops.ffn_fused(x, ln, w1p, b1, w2p, b2, alpha=0.5, norm2=ln))
        t2 = timeit(lambda: ops.ffn_fused(x, ln, w1p, b1, w2p, b2, alpha=0.5, addend=add, norm1=ln, want_h=False))
        b1h, b2h = b1.bfloat16(), b2.bfloat16()

        def lib():
            _, h = ops.add_layernorm(x, None, norm2=ln, out_dtype=torch.bfloat16)
            u = torch._addmm_activation(b1h, h, w1.t(), use_gelu=True)
            y = torch.addmm(b2h, u, w2.t())
            ops.add_layernorm(x, y, 0.5, x_out=x, norm2=ln, out_dtype=torch.bfloat16)
        t3 = timeit(lib)
        print(f"rows {rows:6d}: ffn_fused {t:7.1f} us ({flops / t / 1e6:6.1f} TFLOP/s)  with addend+norm1 {t2:7.1f} us   "
              f"library path {t3:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
