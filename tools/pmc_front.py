#!/usr/bin/env python3
"""Runs only the front end (cm_fbank_wav + cm_fbank_finish + cm_cnn_front, 64 x 40 s) a few times, for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd import fused, ops
from mamba_asr_amd.asr import CONFIGS, ConMambaASR, samples_for_frames, synthetic_wavs

b = int(os.environ.get("B", 64))
cfg = CONFIGS["conmamba_large_ctc"]
model = ConMambaASR(cfg).to("cuda").eval()
wavs, lens = synthetic_wavs(b, samples_for_frames(4000), 0, "cuda")
with torch.no_grad():
    model.calibrate(wavs, lens)
    c = fused._frontend_cache(model, torch.bfloat16)
    b0 = model.CNN.blocks[0]
    for _ in range(int(os.environ.get("REPS", 4))):
        feats = model.compute_features(wavs, norm=(model.normalize.glob_mean, model.normalize.glob_std))
        src = ops.cnn_front(feats, b0.conv.weight, b0.conv.bias, b0.norm.norm.weight, b0.norm.norm.bias, b0.norm.norm.eps,
                            c["w2_ohwi"], c["b2f"], c["ln2"][0], c["ln2"][1], c["ln2"][2], 0.01)
torch.cuda.synchronize()
print("front end ran", tuple(src.shape))
