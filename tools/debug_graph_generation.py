#!/usr/bin/env python3
"""CM_CACHE_TRACE=1: which cache lookups allocate new storage during a graphed training run of the tiny model (accumulation 2, two shapes)."""
import os, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
os.environ["CM_CACHE_TRACE"] = "1"
import torch
from test_graph_train import _tiny_brain, _batches
batches = _batches()
brain = _tiny_brain(True, dropout=0.0, accum=2)
for n, i in enumerate([0, 0, 0, 0, 1, 1, 0, 1]):
    print("micro-batch", n, "shape", i, flush=True)
    brain.fit_batch(batches[i])
    print("   graphs:", {k[0][0]: sorted(v) for k, v in brain._graphs.items()}, flush=True)
