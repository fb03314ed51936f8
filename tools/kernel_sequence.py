#!/usr/bin/env python3
"""From a rocprofv3 kernel-trace (csv): the kernel sequence of ONE forward pass in the last quarter of the trace (from one
fbank_wav_kernel launch to the next), with durations -- where every copy / glue kernel sits."""
import csv, glob, sys

root = sys.argv[1]
f = glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True)[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "")))
rows.sort()
marks = [i for i, r in enumerate(rows) if "fbank_wav_kernel" in r[2]]
a, b = marks[-3], marks[-2]
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")[:90]
t0 = rows[a][0]
for s, e, n, st in rows[a:b]:
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f} us  s{st:>3}  {short(n)}")
