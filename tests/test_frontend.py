"""Front end.  speechbrain is absent from the reference tree and this image, so these tests
pin the BUILD's restatement of its semantics: Fbank against an independent numpy/scipy STFT + mel computation (G6),
the native GPU back end and SpecAugment kernel against the torch restatement."""
import numpy as np
import pytest
import torch

from oracle import conmamba_oracle as O


def _numpy_fbank(wav, n_fft=512, win=400, hop=160, n_mels=80, sr=16000):
    """independent reference: explicit framing + numpy rfft + hand-built triangular mel filters"""
    w = np.hamming(win + 1)[:-1] if False else 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(win) / win)   # periodic Hamming
    wpad = np.zeros(n_fft); off = (n_fft - win) // 2; wpad[off:off + win] = w
    x = np.pad(wav, (n_fft // 2, n_fft // 2))
    n_frames = 1 + (len(x) - n_fft) // hop
    frames = np.stack([x[i * hop:i * hop + n_fft] * wpad for i in range(n_frames)])
    power = np.abs(np.fft.rfft(frames, n=n_fft, axis=1)) ** 2
    mel = lambda f: 2595.0 * np.log10(1 + f / 700.0)
    pts = np.linspace(mel(0), mel(sr / 2), n_mels + 2)
    hz = 700.0 * (10 ** (pts / 2595.0) - 1)
    freqs = np.linspace(0, sr // 2, n_fft // 2 + 1)
    fb = np.zeros((n_fft // 2 + 1, n_mels))
    for m in range(n_mels):
        band = hz[m + 1] - hz[m]
        slope = (freqs - hz[m + 1]) / band
        fb[:, m] = np.maximum(0, np.minimum(slope + 1, -slope + 1))
    db = 10 * np.log10(np.maximum(power @ fb, 1e-10))
    return np.maximum(db, db.max() - 80.0)


def test_fbank_restatement_vs_independent_numpy():
    from mamba_asr_amd.sb_compat import Fbank
    gen = torch.Generator().manual_seed(1)
    wav = (0.1 * torch.randn(2, 16000, generator=gen)).clamp(-1, 1)
    fb = Fbank(sample_rate=16000, n_fft=512, n_mels=80, win_length=25)
    got = fb(wav)
    assert got.shape == (2, 101, 80)
    for i in range(2):
        ref = _numpy_fbank(wav[i].double().numpy())
        np.testing.assert_allclose(got[i].numpy(), ref, rtol=2e-3, atol=2e-2)
    torch.testing.assert_close(O.fbank(wav), got, rtol=1e-5, atol=1e-4)


def test_input_normalization_and_noam_semantics():
    from mamba_asr_amd.sb_compat import InputNormalization
    x = torch.randn(3, 50, 8) * 3 + 1
    lens = torch.tensor([1.0, 0.5, 0.8])
    norm = InputNormalization(norm_type="global", update_until_epoch=4).train()
    y = norm(x, lens, epoch=0)
    mean, std = O.global_norm_stats(x, lens)
    torch.testing.assert_close(y, (x - mean) / std, rtol=1e-4, atol=1e-4)
    norm.eval()
    y2 = norm(x * 2, lens, epoch=9)                       # frozen statistics
    torch.testing.assert_close(y2, (x * 2 - mean) / std, rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
def test_native_fbank_backend_matches_torch_restatement():
    from mamba_asr_amd.sb_compat import Fbank
    gen = torch.Generator().manual_seed(2)
    wav = (0.1 * torch.randn(3, 24000, generator=gen)).clamp(-1, 1)
    fb = Fbank(sample_rate=16000, n_fft=512, n_mels=80, win_length=25)
    ref = fb(wav)                                          # CPU: torch restatement
    got = fb.to("cuda")(wav.cuda())                        # GPU: rocFFT + cm_fbank_mel_db + cm_fbank_finish
    torch.testing.assert_close(got.cpu(), ref, rtol=1e-3, atol=2e-2)
    mean, std = torch.randn(80), torch.rand(80) + 0.5
    got_n = fb(wav.cuda(), norm=(mean.cuda(), std.cuda()))
    torch.testing.assert_close(got_n.cpu(), (ref - mean) / std, rtol=1e-3, atol=5e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("samples,win_ms", [(16000, 25), (24317, 25), (4000, 32), (640, 25)])
def test_fbank_wav_kernel_vs_float64_reference_and_vendor_fft(samples, win_ms, monkeypatch):
    """cm_fbank_wav (in-LDS radix-4 real FFT + mel + dB) against (a) the independent float64 numpy STFT + mel of
    _numpy_fbank and (b) the torch.stft (vendor FFT) + cm_fbank_mel_db path it replaces; ragged lengths (samples not a
    multiple of the hop, fewer frames than a tile), both window lengths the recipes use."""
    from mamba_asr_amd import sb_compat
    from mamba_asr_amd.sb_compat import Fbank
    gen = torch.Generator().manual_seed(samples)
    wav = (0.1 * torch.randn(3, samples, generator=gen)).clamp(-1, 1)
    wav[1, samples // 2:] = 0.0                              # a silent tail exercises the amin floor / top_db clamp
    fb = Fbank(sample_rate=16000, n_fft=512, n_mels=80, win_length=win_ms).to("cuda")
    got = fb(wav.cuda())
    assert got.shape == (3, 1 + samples // 160, 80)
    win = int(round(16 * win_ms))
    for i in range(3):
        ref = _numpy_fbank(wav[i].double().numpy(), win=win)
        np.testing.assert_allclose(got[i].cpu().numpy(), ref, rtol=1e-4, atol=2e-3)
    monkeypatch.setattr(sb_compat, "USE_FBANK_WAV", False)
    old = fb(wav.cuda())
    torch.testing.assert_close(got, old, rtol=1e-4, atol=2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [1, 2])
def test_spec_drop_kernel(dim):
    from mamba_asr_amd import ops
    gen = torch.Generator().manual_seed(3)
    feats = torch.randn(4, 60, 80, generator=gen)
    size = feats.shape[dim]
    start = torch.randint(0, size - 12, (4, 3), generator=gen)
    length = torch.randint(6, 12, (4, 3), generator=gen)
    fill = feats.mean()
    got = ops.spec_drop_(feats.clone().cuda(), start.cuda(), length.cuda(), dim, fill.cuda()).cpu()
    ar = torch.arange(size).view(1, 1, -1)
    mask = ((ar >= start[..., None]) & (ar < (start + length)[..., None])).any(1)
    mask = mask[:, :, None] if dim == 1 else mask[:, None, :]
    ref = torch.where(mask, fill, feats)
    torch.testing.assert_close(got, ref)
    assert mask.any() and not mask.all()


@pytest.mark.gpu
def test_spectrogram_drop_module_gpu_statistics():
    from mamba_asr_amd.sb_compat import SpectrogramDrop
    torch.manual_seed(0)
    feats = torch.randn(8, 200, 80, device="cuda")
    aug = SpectrogramDrop(drop_length_low=6, drop_length_high=12, drop_count_low=1, drop_count_high=5, replace="mean", dim=1)
    out = aug(feats)
    changed = (out != feats).any(dim=2)                     # (8, 200) time steps touched
    per_utt = changed.sum(1)
    assert int(per_utt.min()) >= 6 and int(per_utt.max()) <= 5 * 11
    assert torch.allclose(out[changed], feats.mean().expand_as(out[changed]))
