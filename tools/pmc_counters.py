#!/usr/bin/env python3
"""Per-kernel averages of every counter in one or more rocprofv3 --pmc output directories (counter_collection.csv).
usage: pmc_counters.py <dir> [<dir> ...] [--match substring]"""
import collections, csv, glob, os, sys

match = None
dirs = []
it = iter(sys.argv[1:])
for a in it:
    if a == "--match":
        match = next(it)
    else:
        dirs.append(a)
for d in dirs:
    files = sorted(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    if not files:
        print(f"{d}: no counter_collection.csv")
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[-1])):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"== {d}")
    for k, cs in acc.items():
        if match and match not in k:
            continue
        print(f"  {k[:90]}  ({max(len(v) for v in cs.values())} dispatches)")
        for c, v in sorted(cs.items()):
            print(f"      {c:28s} {sum(v) / len(v):16.1f}")
