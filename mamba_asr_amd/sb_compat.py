"""Minimal stand-ins for the speechbrain==1.0.0 classes the ConMamba recipes instantiate
(reference hparams/CTC/conmamba_large.yaml:187-326, modules/Conmamba.py:112-121,
modules/TransformerASR.py:726-734).  speechbrain is not part of the reference tree and is not
installable here, so these restate its documented semantics with the same
constructor arguments and state_dict key names; if speechbrain is importable the real classes can be
used instead — nothing below is needed then.  Parity status: "unpinned" (no reference test covers them)."""
from __future__ import annotations

import math
import os
from typing import Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


class Swish(nn.Module):
    """speechbrain.nnet.activations.Swish(beta=1)."""

    def __init__(self, beta: float = 1.0):
        super().__init__()
        self.beta = beta

    def forward(self, x):
        return x * torch.sigmoid(self.beta * x)


# LayerNorm of row tensors on cm_layernorm_fwd / _bwd (csrc/layernorm_train.hip); CM_NATIVE_LN=0 keeps torch's kernels
USE_NATIVE_LN = os.environ.get("CM_NATIVE_LN", "1") == "1"


# the front-end Conv2d block's LayerNorm -> LeakyReLU -> Dropout2d as one kernel each way (ops.LnActDropFn); CM_CNN_TAIL=0 = three
FUSED_CNN_TAIL = os.environ.get("CM_CNN_TAIL", "1") == "1"


# CM_LN_LOW_OUT=0: LayerNorms in front of a projection return fp32 under autocast (torch's behaviour) instead of the
# projection's operand dtype
LN_LOW_OUT = os.environ.get("CM_LN_LOW_OUT", "1") == "1"


class RowsLayerNorm(nn.LayerNorm):
    """nn.LayerNorm (same parameters and state_dict keys) whose GPU forward/backward run on the native kernels (rows of up
    to 4096 normalised elements, contiguous); anything else and CPU tensors take torch's path.

    ``low_out``: set by a parent module whose ONLY consumer of this norm's output is a Linear / projection.  Under autocast
    torch returns fp32 from layer_norm and the Linear then rounds it to the autocast dtype with a cast kernel (and a cast
    back in backward); with ``low_out`` the native kernel stores that same rounding itself (round to nearest even of the
    same fp32 value: bit-identical operands for the GEMM), and the backward kernel takes the low-precision gradient."""

    low_out = False

    def forward(self, x):
        dim = math.prod(self.normalized_shape)
        if (USE_NATIVE_LN and x.is_cuda and self.weight is not None and self.bias is not None and dim % 4 == 0 and dim <= 4096
                and x.dtype in (torch.float32, torch.bfloat16) and x.numel() > 0
                and (len(self.normalized_shape) == 1 or x.is_contiguous())):
            from . import ops
            if len(self.normalized_shape) == 1:
                low = (torch.get_autocast_dtype("cuda") if (self.low_out and LN_LOW_OUT and torch.is_autocast_enabled("cuda")) else None)
                return ops.LayerNormFn.apply(x, self.weight, self.bias, self.eps, low)
            # several trailing axes (the CNN front end's (frequency, channel) norm): rows of their product
            lead = x.shape[:x.dim() - len(self.normalized_shape)]
            y = ops.LayerNormFn.apply(x.reshape(*lead, dim), self.weight.reshape(dim), self.bias.reshape(dim), self.eps)
            return y.view(x.shape)
        return super().forward(x)


class LayerNorm(nn.Module):
    """speechbrain.nnet.normalization.LayerNorm: nn.LayerNorm under ``.norm`` (keys norm.weight/bias)."""

    def __init__(self, input_size=None, input_shape=None, eps=1e-05, elementwise_affine=True):
        super().__init__()
        if input_shape is not None:
            input_size = input_shape[2:]
        self.eps = eps
        self.norm = RowsLayerNorm(input_size, eps=eps, elementwise_affine=elementwise_affine)

    def forward(self, x):
        return self.norm(x)


class _LinearRowsFn(torch.autograd.Function):
    """y = x W^T + b with the WEIGHT gradient formed per utterance and summed: as one GEMM with K = batch*time and a
    256 x 1024 output the library runs 64 workgroups on 256 CUs (no split-K: 165 us, 100 TFLOP/s at 32 x 1000 rows);
    a batched product over the batch axis fills the chip, then one small reduction."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias):
        if torch.is_autocast_enabled("cuda") and x.dtype == torch.float32:
            x = x.to(torch.get_autocast_dtype("cuda"))     # the cast autocast would make inside F.linear, made once: the
        ctx.save_for_backward(x, weight)                    # backward reuses it (and the fp32 input need not stay alive)
        ctx.has_bias = bias is not None
        ctx.weight_owner = weight                           # the Parameter OBJECT: saved_tensors hands back a new wrapper, on which
        return F.linear(x, weight, bias)                    # ops.cast_cached's per-object cache would never hit

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            from . import ops
            # the cache lives on a long-lived object: the parameter itself, or -- for a reshaped view of one (the pointwise Conv1d's
            # weight.squeeze(-1), a new tensor object per call) -- the parameter it is a view of
            own = ctx.weight_owner
            base = own._base
            if base is not None and base.numel() == own.numel() and base.is_contiguous() and own.is_contiguous():
                w_c = ops.cast_cached(base, dy.dtype).view(own.shape)
            else:
                w_c = ops.cast_cached(own, dy.dtype)
            dx = torch.matmul(dy, w_c)
        if ctx.needs_input_grad[1]:
            if x.dim() == 3 and x.shape[0] > 1:
                from . import ops
                dw = ops.sum_leading(torch.bmm(dy.transpose(1, 2), x.to(dy.dtype)), weight.dtype if weight.dtype in (torch.float32, torch.bfloat16) else torch.float32)
            else:
                dw = dy.reshape(-1, dy.shape[-1]).t() @ x.reshape(-1, x.shape[-1]).to(dy.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.reshape(-1, dy.shape[-1]).sum(0)
        return dx, dw, db


def linear_rows(x, weight, bias=None):
    """F.linear whose weight gradient is a batched (per-utterance) product on the GPU; plain F.linear elsewhere."""
    if x.is_cuda and torch.is_grad_enabled() and (weight.requires_grad or x.requires_grad):
        return _LinearRowsFn.apply(x, weight, bias)
    return F.linear(x, weight, bias)


class RowsLinear(nn.Linear):
    """nn.Linear (same parameters and state_dict keys) on linear_rows."""

    def forward(self, x):
        return linear_rows(x, self.weight, self.bias)


class Linear(nn.Module):
    """speechbrain.nnet.linear.Linear: nn.Linear under ``.w``."""

    def __init__(self, n_neurons, input_shape=None, input_size=None, bias=True, combine_dims=False):
        super().__init__()
        if input_size is None:
            input_size = input_shape[-1]
            if len(input_shape) == 4 and combine_dims:
                input_size = input_shape[2] * input_shape[3]
        self.combine_dims = combine_dims
        self.w = nn.Linear(input_size, n_neurons, bias=bias)

    def forward(self, x):
        if x.ndim == 4 and self.combine_dims:
            x = x.reshape(x.shape[0], x.shape[1], x.shape[2] * x.shape[3])
        return self.w(x)


class ModuleList(nn.Module):
    """speechbrain.nnet.containers.ModuleList(*layers): sequential application, keys layers.N.*"""

    def __init__(self, *layers):
        super().__init__()
        self.layers = nn.ModuleList(layers)

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        return x


class PositionalwiseFeedForward(nn.Module):
    """speechbrain.nnet.attention.PositionalwiseFeedForward: Linear -> act -> Dropout -> Linear (keys ffn.0, ffn.3)."""

    def __init__(self, d_ffn, input_shape=None, input_size=None, dropout=0.0, activation=nn.ReLU):
        super().__init__()
        if input_size is None:
            input_size = input_shape[-1]
        self.ffn = nn.Sequential(RowsLinear(input_size, d_ffn), activation(), nn.Dropout(dropout),
                                 RowsLinear(d_ffn, input_size))

    def forward(self, x):
        return self.ffn(x)


# -------------------------------------------------------------------------------------------
# frontend: Fbank, InputNormalization, SpectrogramDrop / Augmenter, ConvolutionFrontEnd
# -------------------------------------------------------------------------------------------
# cm_fbank_wav (in-LDS FFT, waveform -> log-mel in one kernel) when n_fft == 512; CM_FBANK_WAV=0 = torch.stft + cm_fbank_mel_db
USE_FBANK_WAV = os.environ.get("CM_FBANK_WAV", "1") == "1"


def mel_filterbank(n_mels=80, n_fft=512, sample_rate=16000, f_min=0.0, f_max=None) -> torch.Tensor:
    """Triangular mel filters of speechbrain's Filterbank: (n_fft//2+1, n_mels)."""
    f_max = sample_rate / 2 if f_max is None else f_max
    to_mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)
    mel = torch.linspace(to_mel(f_min), to_mel(f_max), n_mels + 2)
    hz = 700.0 * (10.0 ** (mel / 2595.0) - 1.0)
    band = (hz[1:] - hz[:-1])[:-1]
    centre = hz[1:-1]
    freqs = torch.linspace(0, sample_rate // 2, n_fft // 2 + 1)
    slope = (freqs[None, :] - centre[:, None]) / band[:, None]
    return torch.clamp(torch.minimum(slope + 1.0, -slope + 1.0), min=0.0).t().contiguous()


class Fbank(nn.Module):
    """speechbrain.lobes.features.Fbank(sample_rate, n_fft, n_mels, win_length[ms], hop_length[ms]=10):
    STFT (hamming, centre, constant pad) -> power spectrum -> mel -> 10*log10 (amin 1e-10, top_db 80)."""

    def __init__(self, sample_rate=16000, n_fft=400, n_mels=40, win_length=25, hop_length=10, f_min=0, f_max=None,
                 deltas=False, context=False, requires_grad=False, top_db=80.0, amin=1e-10):
        super().__init__()
        assert not deltas and not context, "deltas/context are not used by the ConMamba recipes"
        self.sample_rate, self.n_fft, self.n_mels = sample_rate, n_fft, n_mels
        self.win = int(round(sample_rate / 1000.0 * win_length))
        self.hop = int(round(sample_rate / 1000.0 * hop_length))
        self.top_db, self.amin = top_db, amin
        self.register_buffer("window", torch.hamming_window(self.win), persistent=False)
        self.register_buffer("fbank", mel_filterbank(n_mels, n_fft, sample_rate, f_min, f_max), persistent=False)

    @torch.no_grad()
    def forward(self, wav, norm=None):
        """``norm`` = (mean, std) folds the global normalisation into the native back end (GPU only)."""
        with torch.autocast(device_type=wav.device.type, enabled=False):          # speechbrain forces fp32 here
            if wav.is_cuda and USE_FBANK_WAV:
                from . import ops
                if ops.fbank_wav_supported(self.n_fft, self.hop, self.n_mels):     # STFT + mel in one native kernel
                    mean, std = norm if norm is not None else (None, None)
                    return ops.fbank_from_wav(wav, self.window, self.n_fft, self.hop, self.fbank, self.amin, self.top_db, mean, std)
            spec = torch.stft(wav.float(), self.n_fft, self.hop, self.win, self.window, center=True,
                              pad_mode="constant", normalized=False, onesided=True, return_complex=True)
            if wav.is_cuda:                                                        # native back end (cm_fbank_*)
                from . import ops
                mean, std = norm if norm is not None else (None, None)
                return ops.fbank_from_stft(spec, self.fbank, self.amin, self.top_db, mean, std)
            power = (spec.real ** 2 + spec.imag ** 2).transpose(1, 2)
            mel = power @ self.fbank
            db = 10.0 * torch.log10(torch.clamp(mel, min=self.amin))
            floor = db.amax(dim=(-2, -1), keepdim=True) - self.top_db
            return torch.maximum(db, floor)


class InputNormalization(nn.Module):
    """speechbrain.processing.features.InputNormalization(norm_type='global', update_until_epoch):
    running average (weight 1/(count+1)) of per-utterance mean/std over valid frames.

    As in speechbrain the statistics move only in training mode (first batch, then while epoch < update_until_epoch);
    in eval they are frozen, and before any update they are mean 0 / std 1.  ``count`` travels with glob_mean /
    glob_std in the state_dict (speechbrain checkpoints it alongside them), and the buffers take the checkpoint's
    shape on load, so a trained model restores into a fresh one.  ``update_statistics`` is the explicit way to fill them
    from a batch outside training (benchmarks and tests calibrate on their first batch with it)."""

    def __init__(self, mean_norm=True, std_norm=True, norm_type="global", avg_factor=None, requires_grad=False,
                 update_until_epoch=3):
        super().__init__()
        assert norm_type == "global"
        self.mean_norm, self.std_norm, self.avg_factor = mean_norm, std_norm, avg_factor
        self.update_until_epoch = update_until_epoch
        self.register_buffer("glob_mean", torch.zeros(1))
        self.register_buffer("glob_std", torch.ones(1))
        self.register_buffer("count_buf", torch.zeros((), dtype=torch.long))
        self.count = 0
        self.eps = 1e-10

    # ``count`` stays a host integer (no device sync per batch); the buffer mirrors it for the state_dict
    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self.count_buf.fill_(self.count)
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for name in ("glob_mean", "glob_std"):                   # (1,) placeholders become (n_mels,) after the first batch
            t = state_dict.get(prefix + name)
            if t is not None and t.shape != getattr(self, name).shape:
                setattr(self, name, torch.empty_like(t, device=getattr(self, name).device))
        if prefix + "count_buf" not in state_dict and prefix + "glob_mean" in state_dict:
            # checkpoint written before count was saved: statistics present -> treat as calibrated
            state_dict = dict(state_dict)
            state_dict[prefix + "count_buf"] = torch.tensor(int(state_dict[prefix + "glob_mean"].numel() > 1))
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)
        self.count = int(self.count_buf)

    @torch.no_grad()
    def update_statistics(self, x, lengths):
        """One running-average update from batch ``x`` (batch, frames, feats) with relative ``lengths``."""
        n = torch.round(lengths * x.shape[1]).long().clamp(min=1)
        mask = (torch.arange(x.shape[1], device=x.device)[None, :] < n[:, None]).to(x.dtype)[..., None]
        cnt = n.to(x.dtype)[:, None]
        mean = (x * mask).sum(1) / cnt
        var = (((x - mean[:, None]) * mask) ** 2).sum(1) / (cnt - 1).clamp(min=1)
        cur_mean, cur_std = mean.mean(0), var.sqrt().clamp(min=self.eps).mean(0)
        w = 1.0 / (self.count + 1) if self.avg_factor is None else self.avg_factor
        if self.count == 0:
            self.glob_mean, self.glob_std = cur_mean.detach(), cur_std.detach()
        else:
            self.glob_mean = ((1 - w) * self.glob_mean + w * cur_mean).detach()
            self.glob_std = ((1 - w) * self.glob_std + w * cur_std).detach()
        self.count += 1

    def forward(self, x, lengths, spk_ids=None, epoch=0):
        if self.training and (self.count == 0 or epoch < self.update_until_epoch):
            self.update_statistics(x, lengths)
        return (x - self.glob_mean) / self.glob_std


class SpectrogramDrop(nn.Module):
    """speechbrain.augment.freq_domain.SpectrogramDrop(dim in {1,2}, replace='mean'): n_masks ~ U{low..high}
    per batch, mask length ~ U{len_low..len_high-1}, start ~ U{0..max(1, D-len)-1}; masked cells take the
    tensor's global mean."""

    def __init__(self, drop_length_low=5, drop_length_high=15, drop_count_low=1, drop_count_high=3, replace="zeros",
                 dim=1):
        super().__init__()
        self.lo, self.hi, self.clo, self.chi, self.replace, self.dim = (drop_length_low, drop_length_high,
                                                                         drop_count_low, drop_count_high, replace, dim)

    def forward(self, spec):
        b, size = spec.shape[0], spec.shape[self.dim]
        n_masks = int(torch.randint(self.clo, self.chi + 1, (1,)))
        length = torch.randint(self.lo, self.hi, (b, n_masks), device=spec.device)
        start = torch.randint(0, max(1, size - int(length.max())), (b, n_masks), device=spec.device)
        ar = torch.arange(size, device=spec.device).view(1, 1, -1)
        mask = ((ar >= start[..., None]) & (ar < (start + length)[..., None])).any(1)       # (b, size)
        mask = mask[:, :, None] if self.dim == 1 else mask[:, None, :]
        val = spec.mean() if self.replace == "mean" else torch.zeros((), device=spec.device)
        if spec.is_cuda and spec.dtype == torch.float32:                           # native masking kernel (cm_spec_drop)
            from . import ops
            return ops.spec_drop_(spec.contiguous().clone(), start, length, self.dim, val)
        return torch.where(mask, val.to(spec.dtype), spec)


class Warping(nn.Module):
    """speechbrain.augment.freq_domain.Warping(warp_window=5, warp_mode='bicubic', dim=1) -- the S2S recipes' time warp
    (reference hparams/S2S/conmambamamba_large.yaml:471-491): one random centre c in [window, T - window) and target
    w in (c - window, c + window] per batch; the part before c is resized to w steps and the part from c on to T - w steps
    (interpolation with align_corners=True along the warped axis only; the other axis keeps its size), so the spectrogram
    keeps its shape.  Tensors shorter than 2 * window + 1 pass through.  (speechbrain absent: restated, parity unpinned;
    tests/test_warping.py pins shape, the untouched cases and the closed form on a ramp.)"""

    def __init__(self, warp_window=5, warp_mode="bicubic", dim=1):
        super().__init__()
        self.warp_window, self.warp_mode, self.dim = warp_window, warp_mode, dim

    def forward(self, spectrogram):
        x = spectrogram.transpose(1, 2) if self.dim == 2 else spectrogram
        squeeze = x.dim() == 3
        if squeeze:
            x = x.unsqueeze(1)                                   # (batch, 1, T, F): 2-d interpolation wants 4-d
        T, window = x.shape[2], self.warp_window
        if T - window <= window:
            return spectrogram
        c = int(torch.randint(window, T - window, (1,)))
        w = int(torch.randint(c - window, c + window, (1,))) + 1
        out = self.warp(x, c, w)
        if squeeze:
            out = out.squeeze(1)
        return out.transpose(1, 2) if self.dim == 2 else out

    def warp(self, x, c: int, w: int):
        """x (batch, ch, T, F): [0, c) -> w steps, [c, T) -> T - w steps."""
        T, Fq = x.shape[2], x.shape[3]
        kw = dict(mode=self.warp_mode, align_corners=True) if self.warp_mode in ("bilinear", "bicubic") else dict(mode=self.warp_mode)
        left = F.interpolate(x[:, :, :c], (w, Fq), **kw)
        right = F.interpolate(x[:, :, c:], (T - w, Fq), **kw)
        return torch.cat([left, right], dim=2)


class Augmenter(nn.Module):
    """speechbrain.augment.augmenter.Augmenter with the recipe settings (sequential, prob 1, no concat)."""

    def __init__(self, parallel_augment=False, concat_original=False, min_augmentations=None, max_augmentations=None,
                 shuffle_augmentations=False, repeat_augment=1, augment_prob=1.0, augmentations=()):
        super().__init__()
        assert not parallel_augment and not concat_original and repeat_augment == 1
        self.augmentations = nn.ModuleList(augmentations)
        self.augment_prob = augment_prob

    def forward(self, x, lengths):
        if float(torch.rand(1)) <= self.augment_prob:
            for aug in self.augmentations:
                x = aug(x)
        return x, lengths

    def replicate_labels(self, labels):
        return labels


class Resample(nn.Module):
    """speechbrain.processing.speech_augmentation.Resample (Kaldi's LinearResample): windowed-sinc polyphase resampling of
    (batch, time) waveforms.  cutoff = 0.99 * 0.5 * min(orig, new); Hann-windowed sinc with `lowpass_filter_width` zero
    crossings on each side; one FIR filter per output phase, applied as a strided conv1d; the output has
    floor(len * new / orig) samples.  (speechbrain is absent here: restated from the published algorithm, parity unpinned;
    tests pin its properties — identity at equal rates, tone frequency / amplitude / length after resampling.)"""

    def __init__(self, orig_freq=16000, new_freq=16000, lowpass_filter_width=4):
        super().__init__()
        self.orig_freq, self.new_freq, self.width = int(orig_freq), int(new_freq), lowpass_filter_width
        base = math.gcd(self.orig_freq, self.new_freq)
        self.in_unit, self.out_unit = self.orig_freq // base, self.new_freq // base
        if self.orig_freq == self.new_freq:
            return
        cutoff = 0.99 * 0.5 * min(self.orig_freq, self.new_freq)
        window = lowpass_filter_width / (2.0 * cutoff)                       # half width of the filter in seconds
        out_t = torch.arange(self.out_unit, dtype=torch.float64) / self.new_freq
        first = torch.ceil((out_t - window) * self.orig_freq).long()         # first input sample each phase uses
        taps = int(torch.max(torch.floor((out_t + window) * self.orig_freq).long() - first)) + 1
        j = torch.arange(taps, dtype=torch.float64)[None, :]
        dt = (first[:, None] + j) / self.orig_freq - out_t[:, None]
        w = torch.where(dt.abs() < window, 0.5 * (1.0 + torch.cos(2 * math.pi * cutoff / lowpass_filter_width * dt)), torch.zeros_like(dt))
        sinc = torch.where(dt == 0, torch.full_like(dt, 2 * cutoff), torch.sin(2 * math.pi * cutoff * dt) / (math.pi * dt))
        self.register_buffer("weights", (w * sinc / self.orig_freq).float()[:, None, :], persistent=False)    # (phases, 1, taps)
        self.first = [int(v) for v in first]

    @torch.no_grad()
    def forward(self, wav):
        if self.orig_freq == self.new_freq:
            return wav
        squeeze = wav.dim() == 1
        x = wav[None] if squeeze else wav
        n = x.shape[-1]
        n_out = n * self.new_freq // self.orig_freq
        lo = min(self.first)                                               # most negative start (<= 0)
        units = (n_out + self.out_unit - 1) // self.out_unit
        taps = self.weights.shape[-1]
        need = (units - 1) * self.in_unit + max(self.first) + taps
        xp = F.pad(x.float(), (-lo, max(0, need - n)))
        outs = []
        for ph in range(self.out_unit):                                    # phase ph reads x[u * in_unit + first[ph] + j]
            seg = xp[:, self.first[ph] - lo:]
            outs.append(F.conv1d(seg[:, None, :], self.weights[ph:ph + 1].to(xp.device), stride=self.in_unit)[:, 0, :units])
        y = torch.stack(outs, dim=-1).reshape(x.shape[0], -1)[:, :n_out].to(wav.dtype)
        return y[0] if squeeze else y


class SpeedPerturb(nn.Module):
    """speechbrain.augment.time_domain.SpeedPerturb (reference hparams/CTC/conmamba_large.yaml:260-264, applied per
    utterance in the data pipeline, train_CTC.py:932-934): one of `speeds` (percent) drawn per call; the waveform is
    resampled from orig_freq to orig_freq * speed // 100 and then treated as orig_freq audio."""

    def __init__(self, orig_freq, speeds=(90, 100, 110), device="cpu"):
        super().__init__()
        self.orig_freq, self.speeds = orig_freq, list(speeds)
        self.resamplers = nn.ModuleList(Resample(orig_freq, orig_freq * sp // 100) for sp in self.speeds)
        self.samp_index = 0

    def forward(self, waveform):
        self.samp_index = int(torch.randint(len(self.speeds), (1,)))
        return self.resamplers[self.samp_index](waveform)


class _ConvLayer(nn.Module):
    def __init__(self, cin, cout, freq, kernel, stride, dropout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel, stride=stride)
        self.norm = LayerNorm(input_size=(freq, cout))
        self.act = nn.LeakyReLU(0.01)
        self.drop = nn.Dropout2d(dropout)
        self.kernel = kernel

    def forward(self, x):                         # x: (b, t, f, c) channels-last like speechbrain
        p = self.kernel // 2
        # logical NCHW view over the NHWC bytes (torch channels_last): conv, LayerNorm over (f, c) and the
        # activation all run without a layout copy
        # reflect padding of the time and frequency axes ON the channels-last tensor (as a 3-d reflection pad of
        # (t, f, c) with no padding on c): torch's 2-d reflection pad wants an NCHW-contiguous input, which cost a transposing
        # copy of the activations into NCHW and another one back, forward and backward
        from . import ops
        if x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.is_contiguous() and 2 * p < min(x.shape[1], x.shape[2]) and p >= 1:
            y = ops.ReflectPadTfFn.apply(x, p).permute(0, 3, 1, 2)             # cm_reflect_pad_tf, forward and backward
        else:
            y = F.pad(x.unsqueeze(0), (0, 0, p, p, p, p), mode="reflect").squeeze(0).permute(0, 3, 1, 2)
        y = self.conv(y).permute(0, 2, 3, 1)                       # (b, t', f', c) view, contiguous for channels_last
        ln = self.norm.norm
        dim = y.shape[2] * y.shape[3]
        if (USE_NATIVE_LN and FUSED_CNN_TAIL and y.is_cuda and y.is_contiguous() and y.dtype in (torch.float32, torch.bfloat16) and dim % 4 == 0
                and dim <= 4096 and y.shape[3] % 4 == 0 and ln.weight is not None and ln.bias is not None and tuple(ln.normalized_shape) == tuple(y.shape[2:])):
            # LayerNorm -> LeakyReLU -> Dropout2d as one kernel each way; the result is rounded ONCE to what the consumer (the next
            # block's Conv2d / the Linear behind the front end) would round it to under autocast
            mask = None
            if self.training and self.drop.p > 0:
                keep = 1.0 - self.drop.p
                mask = torch.empty((y.shape[0], y.shape[3]), dtype=torch.float32, device=y.device).bernoulli_(keep).div_(keep)
            out_dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else y.dtype
            return ops.LnActDropFn.apply(y, ln.weight.reshape(dim), ln.bias.reshape(dim), ln.eps, self.act.negative_slope, mask, out_dtype)
        y = self.act(self.norm(y))
        if self.training and self.drop.p > 0:
            # Dropout2d = one Bernoulli draw per (sample, channel), scaled by 1/keep.  Applied as a broadcast product on
            # the channels-last tensor: nn.Dropout2d on the permuted view returned an NCHW-contiguous result, i.e. two
            # transposing copies of the (b, t, f, c) activations per block forward and two more backward (8 ms per step).
            keep = 1.0 - self.drop.p
            mask = torch.empty((y.shape[0], 1, 1, y.shape[3]), dtype=y.dtype, device=y.device).bernoulli_(keep).div_(keep)
            y = y * mask
        return y


class ConvolutionFrontEnd(nn.Module):
    """speechbrain.lobes.models.convolution.ConvolutionFrontEnd(input_shape=(b,t,f), num_blocks,
    num_layers_per_block=1, out_channels, kernel_sizes, strides, residuals): per block Conv2d (stride in time
    and frequency, 'same' reflect padding) -> LayerNorm over (freq, channel) -> LeakyReLU -> Dropout2d."""

    def __init__(self, input_shape, num_blocks=3, num_layers_per_block=5, out_channels=(128, 256, 512),
                 kernel_sizes=(3, 3, 3), strides=(1, 2, 2), dilations=(1, 1, 1), residuals=(True, True, True),
                 dropout=0.1, **unused):
        super().__init__()
        assert num_layers_per_block == 1 and not any(residuals[:num_blocks])
        freq, cin = input_shape[-1], 1
        blocks = []
        for i in range(num_blocks):
            freq = (freq + strides[i] - 1) // strides[i]
            blocks.append(_ConvLayer(cin, out_channels[i], freq, kernel_sizes[i], strides[i], dropout))
            cin = out_channels[i]
        self.blocks = nn.ModuleList(blocks)

    def forward(self, x):
        x = x[:, :, :, None]
        for blk in self.blocks:
            x = blk(x)
        return x


# CTC loss + gradient on cm_ctc_loss (csrc/ctc.hip) for GPU tensors; CM_NATIVE_CTC=0 = torch.nn.functional.ctc_loss
USE_NATIVE_CTC = os.environ.get("CM_NATIVE_CTC", "1") == "1"


def ctc_loss(log_probs, targets, input_lens, target_lens, blank_index, reduction="mean"):
    """speechbrain.nnet.losses.ctc_loss: relative lengths -> F.ctc_loss(sum, zero_infinity); 'batchmean' = / batch."""
    t = log_probs.shape[1]
    il = torch.round(input_lens * t).int()
    tl = torch.round(target_lens * targets.shape[1]).int()
    from . import ops
    if USE_NATIVE_CTC and ops.ctc_supported(log_probs, targets):
        # cm_ctc_loss: alpha and beta concurrently, deterministic gradient (torch: three launches, atomics in the backward)
        loss = ops.CtcLossFn.apply(log_probs, targets, il, tl, blank_index)
    else:
        loss = F.ctc_loss(log_probs.transpose(0, 1).float(), targets, il, tl, blank_index, reduction="sum", zero_infinity=True)
    if reduction == "batchmean":
        return loss / targets.shape[0]
    if reduction == "mean":
        return loss / tl.sum()
    return loss


def kldiv_loss(log_probs, targets, length=None, label_smoothing=0.0, pad_idx=0, reduction="mean"):
    """speechbrain.nnet.losses.kldiv_loss as the S2S recipes call it (reference hparams/S2S/conmambamamba_large.yaml
    `seq_cost` with label_smoothing 0.1, reduction batchmean; train_S2S.py:518-529): label-smoothed KL divergence
    between log-probabilities (batch, steps, classes) and integer targets (batch, steps); target distribution =
    1 - label_smoothing on the target class, label_smoothing / (classes - 1) elsewhere; padded steps (target ==
    pad_idx) contribute nothing; 'batchmean' = sum / batch.  (speechbrain is absent: restated semantics, parity unpinned.)"""
    if log_probs.dim() == 2:
        log_probs = log_probs.unsqueeze(1)
    bz, steps, n_class = log_probs.shape
    flat = log_probs.reshape(-1, n_class).float()
    tgt = targets.reshape(-1).long().detach()
    ignore = tgt == pad_idx
    if label_smoothing > 0:
        with torch.no_grad():
            true_dist = torch.full_like(flat, label_smoothing / (n_class - 1))
            true_dist.scatter_(1, tgt.masked_fill(ignore, 0).unsqueeze(1), 1.0 - label_smoothing)
        loss = F.kl_div(flat, true_dist, reduction="none")
    else:
        loss = F.nll_loss(flat, tgt.masked_fill(ignore, 0), reduction="none").unsqueeze(1)
    loss = loss.masked_fill(ignore.unsqueeze(1), 0.0)
    if reduction == "batchmean":
        return loss.sum() / bz
    if reduction == "sum":
        return loss.sum()
    if reduction == "batch":
        per = loss.view(bz, -1).sum(1)
        return per / (length * steps if length is not None else steps)
    return loss.sum() / max(int((~ignore).sum()), 1)


class NoamScheduler:
    """speechbrain.nnet.schedulers.NoamScheduler: lr = lr0 * sqrt(warm) * min(step^-0.5, step * warm^-1.5)."""

    def __init__(self, lr_initial, n_warmup_steps, model_size=None):
        self.lr_initial, self.n_warmup_steps, self.n_steps = lr_initial, n_warmup_steps, 0
        self.normalize = n_warmup_steps ** 0.5 if model_size is None else model_size ** -0.5
        self.current_lr = lr_initial

    def __call__(self, opt):
        self.n_steps += 1
        lr = self.lr_initial * self.normalize * min(self.n_steps ** -0.5, self.n_steps * self.n_warmup_steps ** -1.5)
        for g in opt.param_groups:
            g["lr"] = lr
        old, self.current_lr = self.current_lr, lr
        return old, lr

    def state_dict(self):
        return {"n_steps": self.n_steps}

    def load_state_dict(self, sd):
        self.n_steps = sd["n_steps"]
