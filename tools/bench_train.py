#!/usr/bin/env python3
"""Training-step throughput of the CTC recipe on one GPU (secondary figure; the headline is the encoder forward):
ConMamba-large, synthetic batch, bf16 autocast, SpecAugment on, AdamW + Noam, through the Brain-style loop
(mamba_asr_amd.brain) and the autograd operator API (HIP scan/conv forward+backward kernels)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mamba_asr_amd.asr import CONFIGS, ConMambaASR, synthetic_wavs, samples_for_frames
from mamba_asr_amd.brain import Brain, Stage
from mamba_asr_amd import sb_compat as sb

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--frames", type=int, default=2000)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--config", default="conmamba_large_ctc")
ap.add_argument("--op-profile", action="store_true", help="one more step under torch.profiler: top ops by GPU time with input shapes")
a = ap.parse_args()
dev = torch.device("cuda")
cfg = CONFIGS[a.config]
model = ConMambaASR(cfg).to(dev)
aug = sb.Augmenter(augmentations=[sb.SpectrogramDrop(6, 12, 1, 5, "mean", 1), sb.SpectrogramDrop(10, 20, 1, 3, "mean", 2)])


class ASR(Brain):
    def compute_forward(self, batch, stage):
        wavs, lens, tokens, tlens = batch
        return self.modules["asr"].forward_ctc(wavs, lens, epoch=0, augment=aug if stage == Stage.TRAIN else None)

    def compute_objectives(self, p_ctc, batch, stage):
        wavs, lens, tokens, tlens = batch
        return self.modules["asr"].ctc_objective(p_ctc, tokens, lens, tlens)

    def on_fit_batch_end(self, batch, outputs, loss, should_step):
        if should_step:
            self.hparams.noam(self.optimizer)


noam = sb.NoamScheduler(1e-3, 50)
brain = ASR({"asr": model}, opt_class=lambda ps: torch.optim.AdamW(ps, lr=1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=5e-4),
            hparams={"precision": "bf16", "grad_accumulation_factor": 1, "max_grad_norm": 5.0, "noam": noam},
            run_opts={"device": "cuda"})
brain.on_fit_start()
brain.modules.train()
wavs, lens = synthetic_wavs(a.batch, samples_for_frames(a.frames), cfg.seed, dev)
g = torch.Generator().manual_seed(0)
tokens = torch.randint(3, 31, (a.batch, a.frames // 8), generator=g).to(dev)
tlens = torch.ones(a.batch, device=dev)
batch = (wavs, lens, tokens, tlens)
losses = [float(brain.fit_batch(batch)) for _ in range(2)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    losses.append(float(brain.fit_batch(batch)))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(f"{cfg.name} train step: batch {a.batch} x {a.frames} frames, {dt*1e3:.1f} ms/step, {a.batch*a.frames/dt/1e3:.1f} k audio-frames/s, "
      f"peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB, loss {losses[0]:.2f} -> {losses[-1]:.2f}, optimizer steps {brain.optimizer_step}")
assert all(l == l for l in losses) and losses[-1] < losses[0], "loss did not decrease"

if a.op_profile:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        brain.fit_batch(batch)
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=45,
                                                             max_shapes_column_width=70))
    print("copies by input shape (GPU time, calls, shapes):")
    for ev in sorted(prof.key_averages(group_by_input_shape=True), key=lambda e: -e.device_time_total):
        if any(k in ev.key for k in ("copy_", "contiguous", "clone", "aten::to", "_to_copy")) and ev.device_time_total > 200:
            print(f"  {ev.key:24s} {ev.device_time_total / 1e3:8.2f} ms {ev.count:5d}  {ev.input_shapes}")
