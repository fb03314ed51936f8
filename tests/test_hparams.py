"""The recipe/config surface: the HyperPyYAML-subset loader builds live ConMamba modules from a recipe file written
in the reference's style (and, when the reference checkout is present, from its own CTC YAML unmodified)."""
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_small_ctc_recipe_builds_modules():
    from mamba_asr_amd.hparams import load_hparams, Opaque
    from mamba_asr_amd.modules.TransformerASR import TransformerASR
    with open(os.path.join(ROOT, "hparams", "CTC", "conmamba_small.yaml")) as f:
        hp = load_hparams(f, overrides={"data_folder": "/nowhere"})
    assert hp["output_folder"] == "results/CTC_char/conmamba_S_CTC/7775"
    assert isinstance(hp["Transformer"], TransformerASR)
    assert hp["modules"]["Transformer"] is hp["Transformer"]                 # !ref shares the object
    assert len(hp["Transformer"].encoder.layers) == 12
    assert hp["mamba_config"] == {"d_state": 16, "expand": 2, "d_conv": 4, "bidirectional": True}
    assert hp["CNN"].blocks[1].conv.out_channels == 32
    assert isinstance(hp["checkpointer"], Opaque)
    opt = hp["model_opt_class"](hp["model"].parameters())
    assert opt.defaults["betas"] == (0.9, 0.98) and opt.defaults["eps"] == 1e-9
    old, new = hp["noam_annealing"](opt)
    assert new == pytest.approx(1e-3 * 7500 ** 0.5 * 7500 ** -1.5)
    n = sum(p.numel() for p in hp["model"].parameters())
    assert 9_000_000 < n < 11_000_000                                            # README: ~10 M
    with pytest.raises(ValueError):
        with open(os.path.join(ROOT, "hparams", "CTC", "conmamba_small.yaml")) as f:
            load_hparams(f)                                                      # data_folder: !PLACEHOLDER


def test_ref_arithmetic_and_tuples():
    from mamba_asr_amd.hparams import load_hparams
    hp = load_hparams("a: 4\nb: !ref 30000 // <a>\nc: !ref <a> * 2 + 1\nshape: (8, 10, 80)\nd: !ref x/<a>/y\n")
    assert hp["b"] == 7500 and hp["c"] == 9 and hp["shape"] == (8, 10, 80) and hp["d"] == "x/4/y"


@pytest.mark.skipif(not os.path.exists("/root/reference/hparams/CTC/conmamba_large.yaml"),
                    reason="reference checkout not present (GPU box)")
def test_reference_large_ctc_yaml_loads_unmodified():
    from mamba_asr_amd.hparams import load_hparams
    with open("/root/reference/hparams/CTC/conmamba_large.yaml") as f:
        hp = load_hparams(f, overrides={"data_folder": "/nowhere"})
    tr = hp["Transformer"]
    assert sum(p.numel() for p in tr.parameters()) == 31_522_048                # 31.52 M
    assert hp["grad_accumulation_factor"] == 4 and hp["precision"] == "bf16"
    from mamba_asr_amd import sb_compat as sb
    assert isinstance(hp["speed_perturb"], sb.SpeedPerturb) and hp["speed_perturb"].speeds == [95, 100, 105]
