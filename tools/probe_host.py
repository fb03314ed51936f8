import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd.asr import CONFIGS, ConMambaASR, samples_for_frames, synthetic_wavs
dev = torch.device("cuda")
cfg = CONFIGS["conmamba_large_ctc"]
model = ConMambaASR(cfg).to(dev).eval()
wavs, lens = synthetic_wavs(16, samples_for_frames(4000), cfg.seed, dev)
def step():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        return model.encode(wavs, lens)
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t_issue = (time.perf_counter() - t0) / 10
torch.cuda.synchronize()
t_total = (time.perf_counter() - t0) / 10
print(f"host issue {t_issue*1e3:.2f} ms/step, issue+drain {t_total*1e3:.2f} ms/step")
# CUDA/HIP graph
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(s)
try:
    with torch.cuda.graph(g):
        out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    print(f"graph replay {(time.perf_counter()-t0)/10*1e3:.2f} ms/step")
    ref = step()
    print("graph vs eager max diff", (out - ref).abs().max().item())
except Exception as e:
    print("graph capture failed:", repr(e)[:500])
