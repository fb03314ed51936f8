"""cm_ffn_fused on v_mfma_f32_16x16x32_bf16 (layout 16) vs v_mfma_f32_32x32x16_bf16 (layout 32): the three shapes of a ConMamba-large
layer's forward (FFN 1 + norm1 + in_proj, FFN 2 with addend + norm2) at 64 k and 32 k rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd import ops

dev = "cuda"


def timeit(fn, iters=30, warm=5):
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
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s, scale=1.0: torch.randn(*s, device=dev, generator=g) * scale
    for rows in (64000, 32000, 16000, 8000):
        x = rn(rows, 256)
        add = rn(rows, 256).bfloat16()
        w1, w2, wp = rn(1024, 256, scale=1 / 16).bfloat16(), rn(256, 1024, scale=1 / 32).bfloat16(), rn(1024, 256, scale=1 / 16).bfloat16()
        b1, b2 = rn(1024, scale=0.1), rn(256, scale=0.1)
        ln = lambda: (1 + 0.1 * rn(256), 0.1 * rn(256), 1e-5)
        pre, n1, n2 = ln(), ln(), ln()
        line = f"{rows} rows:"
        for lay, tok in ((16, None), (32, 64), (32, 32)):
            p1, p2, pp = ops.PackedWeight(w1, lay), ops.PackedWeight(w2, lay), ops.PackedWeight(wp, lay)
            out = torch.empty(rows, 1024, device=dev, dtype=torch.bfloat16)
            t_a = timeit(lambda: ops.ffn_fused(x, pre, p1, b1, p2, b2, alpha=0.5, norm2=n2, proj_w=pp, proj_out=out, tokens=tok))
            t_b = timeit(lambda: ops.ffn_fused(x, pre, p1, b1, p2, b2, alpha=0.5, addend=add, norm1=n1, want_h=False, tokens=tok))
            line += f"  layout {lay}{' / 32 tokens' if tok == 32 else ''}: FFN 1 + in_proj {t_a:6.1f} us, FFN 2 {t_b:6.1f} us |"
        print(line, flush=True)
