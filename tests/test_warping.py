"""Time warp of the S2S recipes (reference hparams/S2S/conmambamamba_large.yaml:471-491 -> speechbrain Warping;
speechbrain is absent: restated, parity unpinned).  Pinned by its closed form: linear interpolation with
align_corners=True reproduces affine functions exactly, so a ramp in time becomes piecewise affine with a knee at the
new centre (torch's bicubic kernel, a = -0.75, stays within 1/16 of it and is exact at the segment ends); shape is
preserved; short inputs pass through; the YAML tag resolves to the class."""
import torch

from mamba_asr_amd import sb_compat as sb


def test_warp_closed_form_on_a_ramp():
    T, Fq, c, w = 40, 6, 17, 20
    ramp = torch.arange(T, dtype=torch.float32)[None, None, :, None].expand(2, 1, T, Fq).contiguous()
    left = torch.linspace(0, c - 1, w)                          # [0, c-1] resampled on w points
    right = torch.linspace(c, T - 1, T - w)                     # [c, T-1] resampled on T - w points
    want = torch.cat([left, right])[None, None, :, None].expand(2, 1, T, Fq)
    out = sb.Warping(5, "bilinear").warp(ramp, c, w)
    assert out.shape == ramp.shape
    torch.testing.assert_close(out, want, rtol=1e-5, atol=1e-4)
    cub = sb.Warping(5, "bicubic").warp(ramp, c, w)
    assert (cub - want).abs().max() <= 1.0 / 16 + 1e-4
    for t in (0, w - 1, w, T - 1):                              # segment ends are sample points: exact
        torch.testing.assert_close(cub[:, :, t], want[:, :, t], rtol=1e-5, atol=1e-4)
    assert (cub[0, 0, 1:, 0] > cub[0, 0, :-1, 0]).all()         # still increasing in time


def test_warp_forward_shapes_and_passthrough():
    torch.manual_seed(0)
    x = torch.randn(3, 50, 80)
    wp = sb.Warping(warp_window=5, warp_mode="bicubic", dim=1)
    y = wp(x)
    assert y.shape == x.shape and torch.isfinite(y).all()
    assert torch.equal(y[:, 0], x[:, 0]) and torch.allclose(y[:, -1], x[:, -1], atol=1e-5)    # end points stay (align_corners)
    short = torch.randn(2, 10, 80)
    assert wp(short) is short                                  # T - window <= window: untouched
    yf = sb.Warping(5, "bicubic", dim=2)(x)                     # frequency warping keeps the shape too
    assert yf.shape == x.shape
    # a constant spectrogram is a fixed point
    const = torch.full((2, 30, 8), 3.5)
    torch.testing.assert_close(wp(const), const)


def test_recipe_tag_resolves():
    from mamba_asr_amd import hparams
    assert hparams._SB_MAP["speechbrain.augment.freq_domain.Warping"].endswith("sb_compat.Warping")
