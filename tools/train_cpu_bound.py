#!/usr/bin/env python3
"""Is the training step CPU-launch-bound?  ConMamba-large, 32 x 40 s, bf16 autocast: host time to ENQUEUE forward + loss + backward
(no synchronisation) against the GPU time of the same work, and the launch count (LAUNCH_LOG off)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd.asr import CONFIGS, ConMambaASR
from mamba_asr_amd import sb_compat

cfg = CONFIGS[os.environ.get("CFG", "conmamba_large_ctc")]
B, S = int(os.environ.get("B", 32)), int(os.environ.get("SAMPLES", 640000))
dev = "cuda"
torch.manual_seed(0)
model = ConMambaASR(cfg).to(dev).train()
wavs = (0.1 * torch.randn(B, S, device=dev)).clamp(-1, 1)
lens = torch.ones(B, device=dev)
tokens = torch.randint(3, cfg.output_neurons, (B, S // 160 // 8), device=dev)


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
n = 8
t0 = time.perf_counter()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    step()
t_enq = time.perf_counter() - t0
e1.record()
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"{n} steps: host enqueue {t_enq / n * 1e3:.1f} ms per step, wall {t_all / n * 1e3:.1f} ms per step, GPU interval {e0.elapsed_time(e1) / n:.1f} ms per step")
print("-> CPU-bound" if t_enq > 0.9 * t_all else "-> GPU-bound (the host runs ahead)")
