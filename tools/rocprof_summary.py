#!/usr/bin/env python3
"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd database output) as a small text table.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/r01_x_kernel_stats.txt
"""
import sqlite3
import sys


def main(path, top=25):
    c = sqlite3.connect(path)
    cur = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
    rows = cur.fetchall()
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print(f"# {'calls':>7} {'total_ms':>14} {'avg_us':>12} {'pct':>7}  kernel")
    for name, calls, total, avg, pct in rows[:top]:
        if len(name) > 150:
            name = name[:147] + "..."
        print(f"  {calls:7d} {total / 1e3:14.2f} {avg:12.2f} {pct:7.2f}  {name}")


if __name__ == "__main__":
    main(sys.argv[1])
