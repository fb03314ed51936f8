"""Loader for the HyperPyYAML subset the reference's recipes use (hparams/{CTC,S2S}/*.yaml, loaded at
train_CTC.py:1056-1058 with `load_hyperpyyaml(fin, overrides)`): tags !ref (with <key> substitution and
arithmetic), !new:, !name:, !apply:, !PLACEHOLDER, tuple literals.  hyperpyyaml itself is not installable here.

Class paths are imported as written, except that
  * `modules.*`            resolves to mamba_asr_amd.modules.* (the HIP-backed ConMamba modules),
  * `speechbrain.*` paths with a counterpart in mamba_asr_amd.sb_compat resolve to it,
  * other speechbrain / wandb paths (checkpointer, loggers, metric stats, samplers: control plane, out of scope)
    become inert `Opaque` records so that an unmodified recipe file still loads.
"""
from __future__ import annotations

import ast
import functools
import importlib
import re
from typing import Any, Dict, Optional

import yaml

_SB_MAP = {
    "speechbrain.lobes.models.convolution.ConvolutionFrontEnd": "mamba_asr_amd.sb_compat.ConvolutionFrontEnd",
    "speechbrain.nnet.linear.Linear": "mamba_asr_amd.sb_compat.Linear",
    "speechbrain.processing.features.InputNormalization": "mamba_asr_amd.sb_compat.InputNormalization",
    "speechbrain.lobes.features.Fbank": "mamba_asr_amd.sb_compat.Fbank",
    "speechbrain.augment.freq_domain.SpectrogramDrop": "mamba_asr_amd.sb_compat.SpectrogramDrop",
    "speechbrain.augment.freq_domain.Warping": "mamba_asr_amd.sb_compat.Warping",
    "speechbrain.augment.augmenter.Augmenter": "mamba_asr_amd.sb_compat.Augmenter",
    "speechbrain.augment.time_domain.SpeedPerturb": "mamba_asr_amd.sb_compat.SpeedPerturb",
    "speechbrain.nnet.losses.kldiv_loss": "mamba_asr_amd.sb_compat.kldiv_loss",
    "speechbrain.nnet.losses.ctc_loss": "mamba_asr_amd.sb_compat.ctc_loss",
    "speechbrain.nnet.schedulers.NoamScheduler": "mamba_asr_amd.sb_compat.NoamScheduler",
    "speechbrain.nnet.activations.Swish": "mamba_asr_amd.sb_compat.Swish",
}


class Opaque:
    """Record of an out-of-scope object (checkpointer, logger, ...): keeps its path and arguments, does nothing."""

    def __init__(self, path, args, kwargs):
        self.path, self.args, self.kwargs = path, args, kwargs

    def __call__(self, *a, **k):
        return Opaque(self.path, self.args + a, {**self.kwargs, **k})

    def __repr__(self):
        return f"Opaque({self.path})"


class _Tagged:
    def __init__(self, kind, path, value):
        self.kind, self.path, self.value = kind, path, value


class _Loader(yaml.SafeLoader):
    pass


def _multi(kind):
    def ctor(loader, suffix, node):
        if isinstance(node, yaml.MappingNode):
            val = loader.construct_mapping(node, deep=True)
        elif isinstance(node, yaml.SequenceNode):
            val = loader.construct_sequence(node, deep=True)
        else:
            v = loader.construct_scalar(node)
            val = None if v in ("", None) else v
        return _Tagged(kind, suffix, val)
    return ctor


for _k in ("new", "name", "apply"):
    _Loader.add_multi_constructor(f"!{_k}:", _multi(_k))
_Loader.add_constructor("!ref", lambda l, n: _Tagged("ref", None, l.construct_scalar(n)))
_Loader.add_constructor("!PLACEHOLDER", lambda l, n: _Tagged("placeholder", None, None))
_Loader.add_constructor("!copy", lambda l, n: _Tagged("ref", None, l.construct_scalar(n)))


def _import(path: str):
    if path.startswith("modules."):
        path = "mamba_asr_amd." + path
    path = _SB_MAP.get(path, path)
    if path.startswith(("speechbrain.", "wandb.", "sentencepiece.")):
        return None
    mod, _, attr = path.rpartition(".")
    obj = importlib.import_module(mod)
    return getattr(obj, attr)


_REF = re.compile(r"<([A-Za-z0-9_\[\]\.]+)>")


class _Resolver:
    def __init__(self, raw: Dict[str, Any]):
        self.raw, self.done, self.busy = raw, {}, set()

    def key(self, k: str):
        if k in self.done:
            return self.done[k]
        if k not in self.raw:
            raise KeyError(f"!ref <{k}>: no such key")
        if k in self.busy:
            raise ValueError(f"circular !ref through <{k}>")
        self.busy.add(k)
        v = self.value(self.raw[k])
        self.busy.discard(k)
        self.done[k] = v
        return v

    def value(self, v):
        if isinstance(v, _Tagged):
            return self.tagged(v)
        if isinstance(v, dict):
            return {k: self.value(x) for k, x in v.items()}
        if isinstance(v, list):
            return [self.value(x) for x in v]
        if isinstance(v, str):
            t = v.strip()
            if t.startswith("(") and t.endswith(")"):
                try:
                    return ast.literal_eval(t)                  # tuple literals such as (8, 10, 80), (False, False)
                except (ValueError, SyntaxError):
                    pass
        return v

    def tagged(self, t: _Tagged):
        if t.kind == "placeholder":
            raise ValueError("!PLACEHOLDER was not overridden")
        if t.kind == "ref":
            expr = str(t.value).strip()
            m = _REF.fullmatch(expr)
            if m:
                return self.key(m.group(1))
            vals = {}

            def sub(mm):
                val = self.key(mm.group(1))
                vals[mm.group(1)] = val
                return str(val)
            text = _REF.sub(sub, expr)
            if vals and all(isinstance(x, (int, float)) for x in vals.values()) and re.fullmatch(r"[-+*/%() .0-9e]+", text):
                return eval(compile(ast.parse(text, mode="eval"), "<ref>", "eval"), {"__builtins__": {}})   # arithmetic
            return text                                         # string interpolation
        val = self.value(t.value)
        args, kwargs = [], {}
        if isinstance(val, dict):
            kwargs = val
        elif isinstance(val, list):
            args = val
        elif val is not None:
            args = [val]
        target = _import(t.path)
        if target is None:
            return Opaque(t.path, tuple(args), kwargs)
        if t.kind == "new":
            return target(*args, **kwargs)
        if t.kind == "name":
            return functools.partial(target, *args, **kwargs) if (args or kwargs) else target
        return target(*args, **kwargs)                          # apply


def load_hparams(stream, overrides: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
    """YAML text / file object -> dict of live objects, like hyperpyyaml.load_hyperpyyaml(fin, overrides)."""
    raw = yaml.load(stream, Loader=_Loader)
    if overrides:
        if isinstance(overrides, str):
            overrides = yaml.safe_load(overrides) or {}
        raw.update(overrides)
    res = _Resolver(raw)
    return {k: res.key(k) for k in raw}
