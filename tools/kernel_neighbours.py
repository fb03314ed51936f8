#!/usr/bin/env python3
"""From a rocprofv3 kernel-trace database: for every launch of kernels matching PATTERN in the last quarter of the trace,
the names of the two kernels before and after it (the timeline context of layout-glue kernels)."""
import glob, sqlite3, sys, collections

root, pat = sys.argv[1], sys.argv[2]
db = glob.glob(f"{root}/**/*_results.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end from kernels order by start"))
rows = rows[len(rows) * 3 // 4:]
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")[:70]
ctx = collections.Counter()
for i, (n, s, e) in enumerate(rows):
    if pat in n and 2 <= i < len(rows) - 2:
        ctx[(short(rows[i - 2][0]), short(rows[i - 1][0]), f"<{(e - s) / 1e3:.0f}us>"[:0] + "*", short(rows[i + 1][0]), short(rows[i + 2][0]))] += 1
for k, v in ctx.most_common(20):
    print(v, "\n    " + "\n    ".join(k))
