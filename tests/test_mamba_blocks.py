"""modules.mamba.mamba_blocks: the reference's generic Mamba stack (mamba_blocks.py:22-49, 111-251) keeps its
operator-API names on this package; no ASR recipe uses it."""
import pytest
import torch
import torch.nn as nn


def test_names_importable_and_stack_structure():
    from mamba_asr_amd.modules.mamba import mamba_blocks as mb
    from mamba_asr_amd.modules.mamba.bimamba import Mamba as BiMamba, UniMamba
    stack = mb.MambaBlocksSequential(n_mamba=3, bidirectional=True, d_model=64, fused_add_norm=False)
    assert len(stack.layers) == 3 and all(isinstance(b.mixer, BiMamba) for b in stack.layers)
    assert isinstance(stack.norm_f, nn.LayerNorm) and [b.layer_idx for b in stack.layers] == [0, 1, 2]
    keys = set(stack.state_dict())
    assert {"layers.0.mixer.in_proj.weight", "layers.0.mixer.A_b_log", "layers.2.norm.weight", "norm_f.bias"} <= keys
    uni = mb.MambaBlocksSequential(n_mamba=1, bidirectional=False, d_model=32, rms_norm=True)
    assert isinstance(uni.layers[0].mixer, UniMamba) and isinstance(uni.norm_f, mb.RMSNorm)
    blk = mb.create_block(48, ssm_cls=UniMamba, ssm_cfg={"d_state": 16}, layer_idx=7, fused_add_norm=False)
    assert blk.layer_idx == 7 and blk.mixer.d_model == 48
    with pytest.raises(NotImplementedError):
        mb.MambaBlocksSequential(n_mamba=1, bidirectional=True, d_model=32, use_simple_block=True)


@pytest.mark.gpu
@pytest.mark.parametrize("bidirectional", [True, False])
def test_stack_forward_equals_manual_composition(bidirectional):
    from mamba_asr_amd.modules.mamba import mamba_blocks as mb
    torch.manual_seed(0)
    stack = mb.MambaBlocksSequential(n_mamba=2, bidirectional=bidirectional, d_model=64, fused_add_norm=False).cuda().eval()
    x = torch.randn(2, 50, 64, device="cuda")
    with torch.no_grad():
        got = stack(x)
        res, h = None, x
        for blk in stack.layers:                               # Add -> LN -> mixer, then the final Add -> LN
            res = h if res is None else h + res
            h = blk.mixer(blk.norm(res))
        want = stack.norm_f(h + res)
    assert got.shape == x.shape and torch.isfinite(got).all()
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
