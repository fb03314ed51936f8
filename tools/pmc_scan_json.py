#!/usr/bin/env python3
"""pmc_scan_rows.json (the record bench.py's `roofline.traffic` reads) from the FETCH_SIZE / WRITE_SIZE passes that
tools/profile_round.sh collects on tools/pmc_scan.py.  usage: pmc_scan_json.py <prof dir>  (prints the JSON)"""
import collections, csv, glob, json, os, sys

KERNEL = "scan_rows_fwd_kernel"


FULL_NAME = [KERNEL]


def counter(d, name):
    f = sorted(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1]
    rows = [r for r in csv.DictReader(open(f)) if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == name]
    vals = [float(r["Counter_Value"]) for r in rows]
    FULL_NAME[0] = rows[0]["Kernel_Name"]
    return sum(vals) / len(vals), len(vals)


def main():
    out = sys.argv[1]
    runs = []
    for b in (64, 32):
        fetch_kb, n = counter(f"{out}/pmc_fetch_b{b}", "FETCH_SIZE")
        write_kb, _ = counter(f"{out}/pmc_write_b{b}", "WRITE_SIZE")
        e, nstate, l = 512, 16, 1000
        fetch = fetch_kb * 1024 * 2            # MI355X_MICROARCH.md, HBM: 16-B-per-lane streaming reads are tallied at half their bytes
        write = write_kb * 1024
        runs.append({
            "workload": {"batch": b, "seqlen": l, "dim": e, "dstate": nstate, "directions": 2, "dtype": "bf16", "in_kernel_dt_proj": True},
            "dispatches_averaged": n, "FETCH_SIZE_KB": round(fetch_kb, 2), "WRITE_SIZE_KB": round(write_kb, 2),
            "fetch_bytes_corrected": int(fetch), "write_bytes": int(write), "traffic_bytes_per_launch": int(fetch + write),
            "algorithmic_bytes_per_launch": (4 * e + 2 * nstate) * 2 * b * l * 2,
            "expected_reads_bytes": {"u_both_directions": 2 * b * l * e * 2, "z_read_by_each_direction": 2 * b * l * e * 2,
                                     "x_dbl_rows": b * l * 96 * 2},
        })
    print(json.dumps({
        "kernel": FULL_NAME[0],
        "command": "tools/profile_round.sh: rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 tools/pmc_scan.py ; second pass --pmc WRITE_SIZE ; B=64 and B=32",
        "correction": "FETCH_SIZE x 1024 x 2: every global read of this kernel is a 16-byte-per-lane buffer_load_dwordx4 streaming read, the pattern for which "
                      "MI355X_MICROARCH.md (HBM) says FETCH_SIZE tallies each 128-B request as 64 B; WRITE_SIZE x 1024 is exact for its 16-B-per-lane... "
                      "(2-byte element) stores as checked against the output bytes.",
        "runs": runs}, indent=1))


if __name__ == "__main__":
    main()
