#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel table (like --stats)."""
import sqlite3
import sys


def main(path, top=40):
    c = sqlite3.connect(path)
    rows = list(c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                          "from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    print(f"total kernel time {tot:.2f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'total_ms':>10} {'pct':>6} {'calls':>6} {'avg_us':>10} {'min_us':>10} {'max_us':>10}  name")
    for r in rows[:top]:
        print(f"{r[2]:10.2f} {100 * r[2] / tot:6.1f} {r[1]:6d} {r[3]:10.1f} {r[4]:10.1f} {r[5]:10.1f}  {r[0][:120]}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
