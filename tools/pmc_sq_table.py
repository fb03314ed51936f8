#!/usr/bin/env python3
"""Per-kernel SQ / LDS table from the two passes of tools/pmc_layer.sh: duration, VALU / MFMA / LDS busy fractions of the
kernel's duration, share of LDS cycles lost to bank conflicts, instruction counts per wave.
usage: pmc_sq_table.py <dir pass a> <dir pass b> [rows]"""
import collections, csv, glob, os, sys


def load(d):
    f = sorted(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    t = sorted(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(t)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return acc, dur


a, dur = load(sys.argv[1])
b, _ = load(sys.argv[2])
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 14
print(f"{'kernel':52s} {'n':>4s} {'us':>7s} {'GHz':>5s} {'VALU%':>6s} {'MFMA%':>6s} {'LDS%':>6s} {'confl%':>6s} {'wait%':>6s} {'valu/w':>7s} {'lds/w':>6s} {'vmr/w':>6s} {'vmw/w':>6s} {'waves':>7s}")
for k in sorted(dur, key=lambda k: -sum(dur[k]))[:rows]:
    n = len(dur[k])
    us = sum(dur[k]) / n
    av = lambda t, c: (sum(t[k][c]) / len(t[k][c])) if t[k].get(c) else float("nan")
    cyc = av(a, "GRBM_GUI_ACTIVE") / 8.0
    waves = av(b, "SQ_WAVES")
    print(f"{k.replace('(anonymous namespace)::', '').replace('void ', '')[:52]:52s} {n:4d} {us:7.1f} {cyc / us / 1e3:5.2f} "
          f"{av(a, 'SQ_ACTIVE_INST_VALU') * 4 / (cyc * 1024) * 100:6.1f} {av(a, 'SQ_VALU_MFMA_BUSY_CYCLES') / (cyc * 1024) * 100:6.1f} "
          f"{av(b, 'SQ_LDS_IDX_ACTIVE') / (cyc * 256) * 100:6.1f} {av(b, 'SQ_LDS_BANK_CONFLICT') / max(av(b, 'SQ_LDS_IDX_ACTIVE'), 1) * 100:6.1f} "
          f"{av(a, 'SQ_WAIT_INST_ANY') / av(a, 'SQ_WAVE_CYCLES') * 100:6.1f} {av(b, 'SQ_INSTS_VALU') / waves:7.0f} {av(b, 'SQ_INSTS_LDS') / waves:6.0f} "
          f"{av(b, 'SQ_INSTS_VMEM_RD') / waves:6.0f} {av(b, 'SQ_INSTS_VMEM_WR') / waves:6.0f} {waves:7.0f}")
