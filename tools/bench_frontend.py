#!/usr/bin/env python3
"""GPU time of the front-end kernels at the headline shapes (32 utterances x 4000 frames), hipGraph-timed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd import ops, _native
from tools.bench_ffn import timeit


def main():
    dev = "cuda"
    B, T, F, M = 32, 4001, 257, 80
    buf = torch.randn(B, T, F, dtype=torch.complex64, device=dev)
    spec = buf.transpose(1, 2)                                  # what torch.stft hands out: (B, F, T) view of (B, T, F)
    fb = torch.zeros(F, M, device=dev)
    edges = torch.linspace(0, F - 1, M + 2)
    for m in range(M):
        lo, c, hi = edges[m].item(), edges[m + 1].item(), edges[m + 2].item()
        f = torch.arange(F, dtype=torch.float32)
        fb[:, m] = torch.clamp(torch.minimum((f - lo) / (c - lo), (hi - f) / (hi - c)), min=0).to(dev)
    mean, std = torch.zeros(M, device=dev), torch.ones(M, device=dev)
    abls = [int(a) for a in os.environ.get("ABLS", "0").split(",")]
    for abl in abls:
        _native.lib().cm_debug_set(abl)
        t = timeit(lambda: ops.fbank_from_stft(spec, fb, mean=mean, std=std), iters=5)
        print(f"fbank (mel_db + finish) abl={abl}: {t:8.1f} us", flush=True)
    _native.lib().cm_debug_set(0)
    feats = torch.randn(B, T, M, device=dev)
    w1, b1 = torch.randn(64, 1, 3, 3, device=dev) / 3, torch.randn(64, device=dev) * 0.1
    g1, be1 = torch.ones(40, 64, device=dev), torch.zeros(40, 64, device=dev)
    t = timeit(lambda: ops.cnn_block1(feats, w1, b1, g1, be1, 1e-5, 0.01, out_dtype=torch.bfloat16, pad_out=1), iters=5)
    print(f"cnn_block1: {t:8.1f} us", flush=True)
    y1 = ops.cnn_block1(feats, w1, b1, g1, be1, 1e-5, 0.01, out_dtype=torch.bfloat16, pad_out=1)
    w2 = (torch.randn(32, 3, 3, 64, device=dev) / 24).bfloat16()
    b2 = torch.randn(32, device=dev) * 0.1
    g2, be2 = torch.ones(20, 32, device=dev), torch.zeros(20, 32, device=dev)
    t = timeit(lambda: ops.cnn_block2(y1, w2, b2, g2, be2, 1e-5, 0.01), iters=5)
    print(f"cnn_block2: {t:8.1f} us   (in {y1.numel()*2/1e6:.0f} MB)", flush=True)


if __name__ == "__main__":
    main()
