#!/usr/bin/env python3
"""Per-kernel table (total, calls, avg / min / max) from a rocprofv3 kernel-trace csv."""
import csv, sys
from collections import defaultdict
rows = defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tot = sum(sum(v) for v in rows.values())
print(f"total kernel time {tot / 1e3:.2f} ms over {sum(len(v) for v in rows.values())} dispatches")
print(f"{'total_ms':>10} {'pct':>6} {'calls':>6} {'avg_us':>10} {'min_us':>10} {'max_us':>10}  name")
for n, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print(f"{sum(v) / 1e3:10.2f} {100 * sum(v) / tot:6.1f} {len(v):6d} {sum(v) / len(v):10.1f} {min(v):10.1f} {max(v):10.1f}  {n[:120]}")
