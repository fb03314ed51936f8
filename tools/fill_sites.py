#!/usr/bin/env python3
"""Where the small glue kernels of a training step are launched from: one ConMamba-large micro-batch (8 x 10 s) under
torch.profiler with Python stacks; for every aten op matching PATTERN (default: fill / zero / copy), the count per innermost
repository frame."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

pat = sys.argv[1].split(",") if len(sys.argv) > 1 else ["aten::fill_", "aten::zero_", "aten::copy_", "aten::add_", "aten::mul"]
from mamba_asr_amd.asr import CONFIGS, ConMambaASR
cfg = CONFIGS["conmamba_large_ctc"]
dev = "cuda"
torch.manual_seed(0)
model = ConMambaASR(cfg).to(dev).train()
B, S = 8, 160000
wavs = (0.1 * torch.randn(B, S, device=dev)).clamp(-1, 1)
lens = torch.ones(B, device=dev)
tokens = torch.randint(3, cfg.output_neurons, (B, 100), device=dev)


def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logp = model.forward_ctc(wavs, lens)
    from mamba_asr_amd import sb_compat
    loss = sb_compat.ctc_loss(logp, tokens, lens, lens, blank_index=0)
    loss.backward()
    for p in model.parameters():
        p.grad = None


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
torch.cuda.synchronize()
root = os.getcwd()
sites = collections.Counter()
for ev in prof.events():
    if ev.name in pat:
        fr = [s for s in (ev.stack or []) if "mamba_asr_amd" in s]
        sites[(ev.name, fr[0].strip() if fr else (ev.stack[0].strip() if ev.stack else "<autograd engine>"))] += 1
for (n, s), c in sites.most_common(60):
    print(f"{c:5d}  {n:14s} {s[-110:]}")
