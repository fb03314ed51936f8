"""Recipe-level assembly of the CTC / S2S ConMamba models, mirroring what the reference's YAML files
instantiate (hparams/CTC/conmamba_large.yaml:146-225, 322-326; hparams/S2S/conmamba_small.yaml:229-259;
hparams/S2S/conmambamamba_large.yaml:251-257) and what train_CTC.py's compute_forward / compute_objectives
do with them (train_CTC.py:281-312, 392-422)."""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import sb_compat as sb
from .modules.TransformerASR import TransformerASR


@dataclass
class ASRConfig:
    name: str
    d_model: int = 256
    d_ffn: int = 1024
    num_encoder_layers: int = 18
    num_decoder_layers: int = 0
    output_neurons: int = 31
    n_fft: int = 512
    win_length: int = 25            # ms
    n_mels: int = 80
    sample_rate: int = 16000
    transformer_dropout: float = 0.1
    d_state: int = 16
    expand: int = 2
    d_conv: int = 4
    bidirectional: bool = True
    seed: int = 3402
    blank_index: int = 0


CONFIGS = {
    # hparams/CTC/conmamba_large.yaml:58, 102-105, 153-183
    "conmamba_large_ctc": ASRConfig("conmamba_large_ctc"),
    # derived as SURVEY.md §8d row 1: CTC recipe + encoder dims of hparams/S2S/conmamba_small.yaml:129,188-189,229-233
    "conmamba_small_ctc": ASRConfig("conmamba_small_ctc", d_model=144, d_ffn=1024, num_encoder_layers=12, n_fft=400,
                                    seed=7775),
    # hparams/S2S/conmambamamba_large.yaml:251-257 (encoder + Mamba decoder, 5000-piece vocabulary)
    "conmambamamba_large_s2s": ASRConfig("conmambamamba_large_s2s", d_model=512, d_ffn=2048, num_encoder_layers=12,
                                         num_decoder_layers=6, output_neurons=5000, win_length=32),
}


class ConMambaASR(nn.Module):
    """modules dict of the recipe (CNN, Transformer, ctc_lin[, seq_lin], normalize) + feature extractor."""

    def __init__(self, cfg: ASRConfig):
        super().__init__()
        self.cfg = cfg
        torch.manual_seed(cfg.seed)                                        # `__set_seed` in the YAML
        self.compute_features = sb.Fbank(sample_rate=cfg.sample_rate, n_fft=cfg.n_fft, n_mels=cfg.n_mels,
                                         win_length=cfg.win_length)
        self.normalize = sb.InputNormalization(norm_type="global", update_until_epoch=4)
        self.CNN = sb.ConvolutionFrontEnd(input_shape=(8, 10, cfg.n_mels), num_blocks=2, num_layers_per_block=1,
                                          out_channels=(64, 32), kernel_sizes=(3, 3), strides=(2, 2),
                                          residuals=(False, False))
        mamba_config = {"d_state": cfg.d_state, "expand": cfg.expand, "d_conv": cfg.d_conv,
                        "bidirectional": cfg.bidirectional}
        self.Transformer = TransformerASR(
            input_size=(cfg.n_mels // 4) * 32, tgt_vocab=cfg.output_neurons, d_model=cfg.d_model, nhead=4,
            num_encoder_layers=cfg.num_encoder_layers, num_decoder_layers=cfg.num_decoder_layers, d_ffn=cfg.d_ffn,
            dropout=cfg.transformer_dropout, activation=nn.GELU, encoder_module="conmamba",
            decoder_module="mamba", attention_type="RelPosMHAXL", normalize_before=True, causal=False,
            mamba_config=mamba_config)
        self.ctc_lin = sb.Linear(input_size=cfg.d_model, n_neurons=cfg.output_neurons)
        if cfg.num_decoder_layers > 0:
            self.seq_lin = sb.Linear(input_size=cfg.d_model, n_neurons=cfg.output_neurons)
        from . import ops
        self.register_load_state_dict_post_hook(lambda module, incompatible: ops.invalidate_caches(module))

    # -- train_CTC.py:285-298 -------------------------------------------------------------
    def features(self, wavs, wav_lens, epoch=0, augment=None):
        feats = self.compute_features(wavs)                                # (B, T, 80), fp32
        feats = self.normalize(feats, wav_lens, epoch=epoch)
        if augment is not None and self.training:
            feats, _ = augment(feats, wav_lens)
        return feats

    @torch.no_grad()
    def calibrate(self, wavs, wav_lens):
        """Fill the global normalisation statistics from one batch outside training (the reference gets them from its
        first training batches, train_CTC.py:287; benchmarks / parity runs on random-init models call this once)."""
        self.normalize.update_statistics(self.compute_features(wavs), wav_lens)

    def encode(self, wavs, wav_lens, epoch=0, augment=None, feats=None):
        """wav (B, samples) -> encoder output (B, ceil(T/4), d_model): the path the headline metric times.
        ``feats``: features() of the batch computed by the caller (a graphed training step keeps Fbank, the running
        normalisation statistics and the host-drawn augmentation outside its hipGraph, brain.Brain.graph_prologue)."""
        if feats is not None:
            return self.Transformer.encode(self.CNN(feats), wav_lens)
        if (not torch.is_grad_enabled()) and (not self.training) and wavs.is_cuda and augment is None \
                and self.normalize.count > 0:
            from . import fused
            if all(fused.supports(layer) for layer in self.Transformer.encoder.layers):
                return fused.asr_encode(self, wavs, wav_lens)                    # native inference path
        src = self.CNN(self.features(wavs, wav_lens, epoch, augment))
        return self.Transformer.encode(src, wav_lens)

    def forward_ctc(self, wavs, wav_lens, epoch=0, augment=None, feats=None):
        """-> log-probabilities (B, T', vocab), train_CTC.py:296-302."""
        enc = self.encode(wavs, wav_lens, epoch, augment, feats=feats)
        return torch.log_softmax(self.ctc_lin(enc), dim=-1)

    def forward_s2s(self, wavs, wav_lens, tokens_bos, epoch=0, augment=None, pad_idx=0, feats=None):
        """train_S2S.py:285-320: features -> CNN -> Transformer(src, <bos> tokens) -> (p_ctc over encoder steps,
        p_seq over decoder steps), both log-probabilities."""
        assert self.cfg.num_decoder_layers > 0, "forward_s2s needs a decoder (S2S configuration)"
        src = self.CNN(self.features(wavs, wav_lens, epoch, augment) if feats is None else feats)
        enc_out, pred = self.Transformer(src, tokens_bos, wav_lens, pad_idx=pad_idx)
        return torch.log_softmax(self.ctc_lin(enc_out), dim=-1), torch.log_softmax(self.seq_lin(pred), dim=-1)

    def s2s_objective(self, p_ctc, p_seq, tokens, tokens_lens, tokens_eos, tokens_eos_lens, wav_lens, ctc_weight=0.3,
                      label_smoothing=0.1, pad_idx=0):
        """train_S2S.py:518-529: ctc_weight * CTC(p_ctc, tokens) + (1 - ctc_weight) * label-smoothed KL(p_seq, tokens_eos)."""
        loss_seq = sb.kldiv_loss(p_seq, tokens_eos, length=tokens_eos_lens, label_smoothing=label_smoothing, pad_idx=pad_idx,
                                 reduction="batchmean")
        loss_ctc = sb.ctc_loss(p_ctc, tokens, wav_lens, tokens_lens, self.cfg.blank_index, reduction="batchmean")
        return ctc_weight * loss_ctc + (1.0 - ctc_weight) * loss_seq

    def ctc_objective(self, p_ctc, tokens, wav_lens, tokens_lens):
        """train_CTC.py:405 with loss_reduction batchmean."""
        return sb.ctc_loss(p_ctc, tokens, wav_lens, tokens_lens, self.cfg.blank_index, reduction="batchmean")


def synthetic_wavs(batch: int, n_samples: int, seed: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """16 kHz waveforms 0.1*N(0,1) clipped to [-1, 1], all utterances full length."""
    g = torch.Generator().manual_seed(seed)
    wav = (0.1 * torch.randn(batch, n_samples, generator=g)).clamp_(-1.0, 1.0)
    return wav.to(device), torch.ones(batch, device=device)


def samples_for_frames(n_frames: int, hop: int = 160) -> int:
    """Fbank with centre padding yields 1 + floor(samples / hop) frames."""
    return (n_frames - 1) * hop
