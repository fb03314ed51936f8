import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from test_hip_parity_r3 import _train_case, DEV
torch.backends.cudnn.deterministic = (os.environ.get("DET","0")=="1")
cfg, model, wavs, lens, tokens, tok_lens = _train_case(batch=2, frames=200)
runs=[]; g=None
for _ in range(3):
    for p in model.parameters(): p.grad=None
    logp = model.forward_ctc(wavs, lens)
    if g is None: g = torch.randn(logp.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))*1e-2
    logp.backward(g)
    runs.append({k: p.grad.clone() for k,p in model.named_parameters()})
for k in runs[0]:
    d01 = (runs[0][k]-runs[1][k]).abs().max().item(); d12=(runs[1][k]-runs[2][k]).abs().max().item()
    if d01>0 or d12>0: print(k, d01, d12)
print("done")
import time
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(5):
    for p in model.parameters(): p.grad=None
    logp = model.forward_ctc(wavs, lens); logp.backward(g)
torch.cuda.synchronize(); print("ms/iter", (time.perf_counter()-t0)/5*1e3)
