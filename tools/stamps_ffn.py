"""Per-phase s_memtime stamps of one mid-grid workgroup of cm_ffn_fused (cm_debug_set(18), wave 0) at 64k rows."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd import ops, _native

dev, rows, D, F = "cuda", 64000, 256, 1024
x = torch.randn(rows, D, device=dev)
add = (torch.randn(rows, D, device=dev) * 0.5).bfloat16()
ln = lambda: (torch.ones(D, device=dev), torch.zeros(D, device=dev), 1e-5)
w1 = ops.PackedWeight((torch.randn(F, D, device=dev) / 16).bfloat16())
w2 = ops.PackedWeight((torch.randn(D, F, device=dev) / 32).bfloat16())
b1, b2 = torch.randn(F, device=dev) * 0.1, torch.randn(D, device=dev) * 0.1
lib = _native.lib()
h = ctypes.CDLL(_native.LIB_PATH)
run = lambda: ops.ffn_fused(x, ln(), w1, b1, w2, b2, alpha=0.5, addend=add, norm1=ln(), want_h=False)
for _ in range(3):
    run()
names = ["entry", "rows arrived", "LN issued", "barrier"]
for c in range(4):
    names += [f"slab{c} GEMM1", f"slab{c} GELU", f"slab{c} barrier"]
names += ["last GEMM2", "all landed"]
lib.cm_debug_set(18)
for rep in range(2):
    run()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 24)()
    h.cm_debug_read_stamps_ffn(buf)
    n = int(buf[20])
    t = [buf[i] - buf[0] for i in range(n)]
    print(" | ".join(f"{names[i] if i < len(names) else i}: {t[i]} (+{t[i] - t[i - 1] if i else 0})" for i in range(n)))
lib.cm_debug_set(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
print("kernel %.1f us" % (e0.elapsed_time(e1) * 100))
