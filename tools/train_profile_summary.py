#!/usr/bin/env python3
"""Per-micro-batch kernel time of a training step by group, from a rocprofv3 --kernel-trace --stats kernel_stats.csv of
`bench.py --mode train` (MIOpen's one-off solver-search kernels excluded)."""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = [int(r["Calls"]) for r in rows if "scan_rows_fwd_kernel" in r["Name"]][0] / int(sys.argv[2] if len(sys.argv) > 2 else 18)
bad = ("naive_conv", "ck16", "ck::", "kernel_batched_gemm_xdlops_bwd_weight")
grp = {"scan fwd": ("scan_rows_fwd",), "scan bwd": ("scan_rows_bwd",), "wgrad + folds": ("wgrad_", "sum_leading"), "library GEMMs": ("Cijk",),
       "ffn fused fwd": ("ffn_fused",), "ffn fused bwd": ("ffn_bwd",), "module element-wise": ("bias_act", "bias_glu", "colsum"), "LayerNorm": ("ln_",),
       "conv kernels": ("conv_cl", "conv_xproj", "dwconv"), "CNN front end": ("igemm", "reflect", "leaky", "SubTensor"), "CTC": ("ctc_",),
       "torch element-wise / reduce / copy": ("at::native", "copyBuffer", "fillBuffer")}
acc = {k: 0.0 for k in grp}
other = 0.0
for r in rows:
    n = r["Name"]
    if any(b in n for b in bad):
        continue
    for k, pats in grp.items():
        if any(p in n for p in pats):
            acc[k] += float(r["TotalDurationNs"])
            break
    else:
        other += float(r["TotalDurationNs"])
tot = sum(acc.values()) + other
print(f"{steps:.0f} micro-batches, {tot / 1e6 / steps:.2f} ms of kernels per micro-batch")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:36s} {v / 1e6 / steps:6.2f} ms  {100 * v / tot:5.1f} %")
print(f"  {'other':36s} {other / 1e6 / steps:6.2f} ms")
if len(sys.argv) > 3:
    for r in rows[:int(sys.argv[3])]:
        if not any(b in r["Name"] for b in bad):
            print(f"{r['Name'][:96]:96s} {int(r['Calls']) / steps:7.1f}/step {float(r['TotalDurationNs']) / 1e6 / steps:6.2f} ms {float(r['AverageNs']) / 1e3:8.1f} us")
