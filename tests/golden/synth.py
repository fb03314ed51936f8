"""Seeded synthetic parameters / inputs for the LARGE golden cases (test infrastructure).

A 2-layer ConMamba-large encoder has 3.5 M parameters (14 MB fp32): too large to commit as a fixture.  Instead
the golden generator (make_golden_r2.py, build container only) and the tests both derive the parameter values
from this function — numpy PCG64, one independent stream per state_dict key (crc32 of the key), so the values
do not depend on key order — load them into the reference's module and into this package's module
respectively, and only the reference's OUTPUT tensors are committed.

Value distributions follow what the recipes produce: every >=2-D tensor ~ Xavier-normal
(reference modules/TransformerASR.py:1051-1054 re-initialises them all, A_log / conv weights included),
LayerNorm weights ~ 1 + 0.1 N, biases ~ 0.1 N (a fresh model has 1 / 0: perturbed so that a mixed-up weight/bias
shows), dt_proj.bias = inverse-softplus of log-uniform[1e-3, 0.1] (reference modules/mamba/bimamba.py:111-118),
D ~ 1 + 0.1 N.
"""
import zlib

import numpy as np
import torch


def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(key.encode())]))


def synth_tensor(key: str, shape, seed: int) -> torch.Tensor:
    g = _rng(seed, key)
    shape = tuple(int(s) for s in shape)
    leaf = key.rsplit(".", 1)[-1]
    if len(shape) >= 2:
        recept = int(np.prod(shape[2:])) if len(shape) > 2 else 1
        fan_out, fan_in = shape[0] * recept, shape[1] * recept
        v = g.standard_normal(shape) * np.sqrt(2.0 / (fan_in + fan_out))
    elif "dt_proj" in key and leaf == "bias":
        dt = np.exp(g.random(shape) * (np.log(0.1) - np.log(1e-3)) + np.log(1e-3)).clip(min=1e-4)
        v = dt + np.log(-np.expm1(-dt))
    elif leaf in ("D", "D_b") or leaf == "weight":           # 1-D weights are LayerNorm gains
        v = 1.0 + 0.1 * g.standard_normal(shape)
    else:
        v = 0.1 * g.standard_normal(shape)
    return torch.from_numpy(v.astype(np.float32))


def synth_state(shapes: dict, seed: int) -> dict:
    """{key: shape} -> {key: fp32 tensor}."""
    return {k: synth_tensor(k, s, seed) for k, s in shapes.items()}


def synth_like(module: torch.nn.Module, seed: int) -> dict:
    return synth_state({k: tuple(v.shape) for k, v in module.state_dict().items()}, seed)


def synth_input(key: str, shape, seed: int, scale: float = 1.0) -> torch.Tensor:
    v = _rng(seed, "input:" + key).standard_normal(tuple(shape)) * scale
    return torch.from_numpy(v.astype(np.float32))
