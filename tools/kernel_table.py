"""Per-step table from a rocprofv3 kernel_stats.csv: python tools/kernel_table.py <csv> <steps incl. warmup> [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 20
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per step {tot / steps / 1e6:.3f} ms")
for r in rows[:top]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']) / steps:6.1f}/step  avg {float(r['AverageNs']) / 1e3:8.1f} us  "
          f"{float(r['TotalDurationNs']) / steps / 1e6:6.3f} ms/step {float(r['Percentage']):5.1f}%")
