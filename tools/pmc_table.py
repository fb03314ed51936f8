#!/usr/bin/env python3
"""Per-kernel table from rocprofv3 --pmc passes over `bench.py --no-graph` (counter_collection.csv + kernel_trace.csv):
average duration, MFMA-pipe and VALU utilisation, FETCH_SIZE / WRITE_SIZE per launch and the memory-side GB/s they
imply.  usage: pmc_table.py <dir with sq pass> <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass>"""
import csv, glob, sys, collections, os


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


def main():
    sq, dur = load(sys.argv[1])
    fe, _ = load(sys.argv[2])
    wr, _ = load(sys.argv[3])
    names = sorted(dur, key=lambda k: -sum(dur[k]))
    print(f"{'kernel':60s} {'calls':>5s} {'avg us':>8s} {'MFMA %':>7s} {'VALU %':>7s} {'fetch MB':>9s} {'write MB':>9s} {'GB/s':>7s}")
    for k in names[:int(sys.argv[4]) if len(sys.argv) > 4 else 16]:
        n = len(dur[k])
        us = sum(dur[k]) / n
        avg = lambda a, c: (sum(a[k][c]) / len(a[k][c])) if a[k].get(c) else float("nan")
        gui = avg(sq, "GRBM_GUI_ACTIVE") / 8.0                       # summed over 8 XCDs -> cycles
        mfma = avg(sq, "SQ_VALU_MFMA_BUSY_CYCLES") / (gui * 1024) * 100 if gui == gui else float("nan")
        valu = avg(sq, "SQ_ACTIVE_INST_VALU") * 4 / (gui * 1024) * 100 if gui == gui else float("nan")
        f_mb, w_mb = avg(fe, "FETCH_SIZE") / 1024, avg(wr, "WRITE_SIZE") / 1024
        print(f"{k[:60]:60s} {n:5d} {us:8.1f} {mfma:7.1f} {valu:7.1f} {f_mb:9.1f} {w_mb:9.1f} {(f_mb + w_mb) / us * 1e3:7.0f}")


if __name__ == "__main__":
    main()
