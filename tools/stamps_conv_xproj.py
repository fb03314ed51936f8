"""Per-phase s_memtime stamps of one mid-grid workgroup of cm_conv_xproj (cm_debug_set(18)) at 64 x 1000 x 512."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd import ops, _native

dev, e, l, b = "cuda", 512, 1000, 64
xz = torch.randn(b, l, 2 * e, device=dev).bfloat16()
wf, wb = torch.randn(e, 4, device=dev) * 0.5, torch.randn(e, 4, device=dev) * 0.5
bf, bb = torch.randn(e, device=dev) * 0.1, torch.randn(e, device=dev) * 0.1
pk = [ops.PackedWeight((torch.randn(48, e, device=dev) * 0.1).bfloat16()) for _ in range(2)]
ucat = torch.empty(b, l, 2 * e, device=dev, dtype=torch.bfloat16)
xdbl = torch.empty(b, l, 96, device=dev, dtype=torch.bfloat16)
lib = _native.lib()
run = lambda: ops.conv_xproj(xz[:, :, :e], wf, bf, wb, bb, pk[0], pk[1], out_f=ucat[:, :, :e], out_b=ucat[:, :, e:], xdbl=xdbl)
for _ in range(3):
    run()
lib.cm_debug_set(18)
for rep in range(3):
    run()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    lib._handle if False else None
    h = ctypes.CDLL(_native.LIB_PATH)
    h.cm_debug_read_stamps_cx(buf)
    w0 = [buf[i] - buf[0] for i in range(6)]
    w3 = [buf[8 + i] - buf[8] for i in range(4)]
    print("wave 0: rows arrived %d, phase 1 issued %d, barrier passed %d, MFMA loop issued %d, all landed %d   (s_memtime ticks from kernel entry)" % tuple(w0[1:]))
    print("wave 3: rows arrived %d, phase 1 issued %d, barrier passed %d" % tuple(w3[1:]))
lib.cm_debug_set(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
print("kernel %.1f us" % (e0.elapsed_time(e1) * 100))
