#!/usr/bin/env python3
"""CTC-loss delta of the GPU path vs the CPU oracle on ConMamba-large (2 utterances x 10 s): the north-star parity
figure (target |delta| <= 1e-3 in fp32; bf16 delta reported)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd.asr import CONFIGS, ConMambaASR, synthetic_wavs, samples_for_frames
from oracle import conmamba_oracle as O

cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "conmamba_large_ctc"]
dev = "cuda"
model = ConMambaASR(cfg).to(dev).eval()
wavs, lens = synthetic_wavs(2, samples_for_frames(1000), cfg.seed, dev)
gen = torch.Generator().manual_seed(1)
tokens = torch.randint(3, 31, (2, 140), generator=gen)
tl = torch.tensor([1.0, 0.9])
with torch.no_grad():
    model.calibrate(wavs, lens)
    l32 = model.ctc_objective(model.forward_ctc(wavs, lens), tokens.to(dev), lens, tl.to(dev))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        pbf = model.forward_ctc(wavs, lens)
    lbf = model.ctc_objective(pbf.float(), tokens.to(dev), lens, tl.to(dev))
O.set_threads(O.host_threads(16))
try:
    O.load_c_oracle(); scan = O.selective_scan_c
except OSError:
    scan = O.selective_scan
p = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
enc = O.asr_encode(p, wavs.cpu(), lens.cpu(), cfg.num_encoder_layers, p["normalize.glob_mean"], p["normalize.glob_std"],
                   scan=scan, n_fft=cfg.n_fft, win_ms=cfg.win_length)
logp = torch.log_softmax(torch.nn.functional.linear(enc, p["ctc_lin.w.weight"], p["ctc_lin.w.bias"]), -1)
ref = O.ctc_loss_batchmean(logp, tokens, lens.cpu(), tl)
print(f"{cfg.name}: CTC loss oracle {float(ref):.6f} | GPU fp32 {float(l32):.6f} |delta| {abs(float(l32)-float(ref)):.3e} | "
      f"GPU bf16 {float(lbf):.6f} |delta| {abs(float(lbf)-float(ref)):.3e}")
