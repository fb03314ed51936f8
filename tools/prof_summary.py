#!/usr/bin/env python3
"""Per-kernel summary (calls, average ns, share) from a rocprofv3 rocpd database or kernel_stats.csv directory."""
import glob, sqlite3, sys


def main():
    root = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    skip = sys.argv[3:] or ["naive_conv"]
    db = glob.glob(f"{root}/**/*_results.db", recursive=True)[0]
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, count(*), avg(end-start), sum(end-start) from kernels group by name order by 4 desc"))
    rows = [r for r in rows if not any(s in r[0] for s in skip)]
    tot = sum(r[3] for r in rows)
    print(f"total kernel time (excluding {skip}): {tot/1e6:.2f} ms")
    for r in rows[:top]:
        print(f"{r[0][:100]:100s} {r[1]:6d} {r[2]/1e3:9.1f} us {100*r[3]/tot:5.1f}%")


if __name__ == "__main__":
    main()
