#!/usr/bin/env python3
"""Debug aid: one training step with Tensor.contiguous / Tensor.reshape wrapped, listing every call that COPIES a large
non-contiguous tensor (shape, strides, dtype, call site).  Used to find layout glue in the autograd path."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

seen = collections.Counter()
_contig, _reshape = torch.Tensor.contiguous, torch.Tensor.reshape


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "mamba" in fr.filename and "find_copies" not in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno}"
    return "?"


def contiguous(self, *a, **k):
    if self.numel() > 1_000_000 and not self.is_contiguous():
        seen[("contiguous", tuple(self.shape), tuple(self.stride()), str(self.dtype), site())] += 1
    return _contig(self, *a, **k)


def reshape(self, *shape):
    out = _reshape(self, *shape)
    if self.numel() > 1_000_000 and not self.is_contiguous() and out.data_ptr() != self.data_ptr():
        seen[("reshape", tuple(self.shape), tuple(self.stride()), str(self.dtype), site())] += 1
    return out


torch.Tensor.contiguous, torch.Tensor.reshape = contiguous, reshape
sys.argv = [sys.argv[0], "--batch", "32", "--frames", "4000", "--steps", "1"]
import runpy
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_train.py"), run_name="__main__")
print("copying calls over 3 steps (kind, shape, strides, dtype, site): count")
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(" ", k, v)
