"""sb_compat.Resample / SpeedPerturb (the recipe's speed perturbation, reference
hparams/CTC/conmamba_large.yaml:260-264, train_CTC.py:932-934).  speechbrain is absent: the restated windowed-sinc
polyphase resampler is pinned by its defining formula evaluated directly and by signal properties (parity unpinned
w.r.t. speechbrain's own implementation)."""
import math

import pytest
import torch

from mamba_asr_amd import sb_compat as sb


def direct_resample(x, orig, new, width=4):
    """y[m] = sum_n x[n] h(n / orig - m / new), h = Hann-windowed sinc with cutoff 0.99 * min(orig, new) / 2 — O(N * taps)."""
    cutoff = 0.99 * 0.5 * min(orig, new)
    window = width / (2.0 * cutoff)
    n_out = x.numel() * new // orig
    y = torch.zeros(n_out, dtype=torch.float64)
    for m in range(n_out):
        t = m / new
        lo, hi = math.ceil((t - window) * orig), math.floor((t + window) * orig)
        for n in range(max(lo, 0), min(hi, x.numel() - 1) + 1):
            dt = n / orig - t
            if abs(dt) >= window:
                continue
            w = 0.5 * (1 + math.cos(2 * math.pi * cutoff / width * dt))
            s = 2 * cutoff if dt == 0 else math.sin(2 * math.pi * cutoff * dt) / (math.pi * dt)
            y[m] += float(x[n]) * w * s / orig
    return y


@pytest.mark.parametrize("speed", [95, 105, 90])
def test_resample_matches_the_defining_sum(speed):
    gen = torch.Generator().manual_seed(speed)
    x = torch.randn(403, generator=gen)
    r = sb.Resample(16000, 16000 * speed // 100)
    got = r(x)
    want = direct_resample(x.double(), 16000, 16000 * speed // 100)
    assert got.shape == want.shape == (403 * speed // 100,)
    torch.testing.assert_close(got.double(), want, rtol=1e-5, atol=1e-6)
    # batched input, ragged tail: same numbers
    torch.testing.assert_close(r(torch.stack([x, x.flip(0)]))[0], got)


def test_speed_perturb_properties():
    sr = 16000
    t = torch.arange(2 * sr) / sr
    tone = 0.5 * torch.sin(2 * math.pi * 1000 * t)[None]
    assert sb.Resample(sr, sr)(tone) is tone                      # equal rates: identity
    for speed in (95, 105):
        y = sb.Resample(sr, sr * speed // 100)(tone)
        n = y.shape[-1]
        assert n == 2 * sr * speed // 100
        spec = torch.fft.rfft(y[0] * torch.hann_window(n)).abs()
        assert abs(spec.argmax().item() * sr / n - 1000 * 100 / speed) < 1.0          # pitch moves with the speed
        assert abs(y[0, n // 4: 3 * n // 4].abs().max().item() - 0.5) < 5e-3         # pass band gain 1
    torch.manual_seed(0)
    sp = sb.SpeedPerturb(sr, speeds=[95, 100, 105])
    lens = {sp(tone).shape[-1] for _ in range(40)}
    assert lens == {30400, 32000, 33600}                          # every speed is drawn, lengths follow
    # unit gain inside the band (with 4 zero crossings the transition band of this Kaldi-style filter is ~2 kHz wide, so
    # nothing sharper is asserted near the band edge)
    gain = lambda f: sb.Resample(sr, sr * 95 // 100)(torch.sin(2 * math.pi * f * t)[None])[0, 2000:-2000].abs().max().item()
    assert abs(gain(300) - 1.0) < 0.01 and abs(gain(4000) - 1.0) < 0.01
