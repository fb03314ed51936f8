#!/usr/bin/env python3
"""cm_fbank_wav (+ cm_fbank_finish) vs torch.stft + cm_fbank_mel_db at 64 utterances x 40 s (hipGraph-timed)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd import sb_compat
from mamba_asr_amd.sb_compat import Fbank
from bench_ffn import timeit

dev = "cuda"
for b in (32, 64):
    wav = (0.1 * torch.randn(b, 640000, device=dev)).clamp(-1, 1)
    fb = Fbank(sample_rate=16000, n_fft=512, n_mels=80, win_length=25).to(dev)
    mean, std = torch.zeros(80, device=dev), torch.ones(80, device=dev)
    sb_compat.USE_FBANK_WAV = True
    t1 = timeit(lambda: fb(wav, norm=(mean, std)), iters=5)
    sb_compat.USE_FBANK_WAV = False
    t2 = timeit(lambda: fb(wav, norm=(mean, std)), iters=5)
    print(f"B={b}: cm_fbank_wav path {t1:8.1f} us   torch.stft + cm_fbank_mel_db path {t2:8.1f} us", flush=True)
