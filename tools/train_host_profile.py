#!/usr/bin/env python3
"""cProfile of the host side of training steps (ConMamba-large, 32 x 40 s, bf16): where the ~45 ms of enqueue time per step go."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd.asr import CONFIGS, ConMambaASR
from mamba_asr_amd import sb_compat

cfg = CONFIGS["conmamba_large_ctc"]
B, S = 32, 640000
dev = "cuda"
torch.manual_seed(0)
model = ConMambaASR(cfg).to(dev).train()
wavs = (0.1 * torch.randn(B, S, device=dev)).clamp(-1, 1)
lens = torch.ones(B, device=dev)
tokens = torch.randint(3, cfg.output_neurons, (B, 500), device=dev)


def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logp = model.forward_ctc(wavs, lens)
    loss = sb_compat.ctc_loss(logp, tokens, lens, lens, blank_index=0)
    loss.backward()
    for p in model.parameters():
        p.grad = None


for _ in range(3):
    step()
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)                 # the backward's Python on this thread: visible to cProfile
pr = cProfile.Profile()
pr.enable()
for _ in range(4):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 45)
