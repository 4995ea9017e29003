#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database (one --pmc pass per database).

    python tools/pmc_summary.py gpurun_out/pmc3/p3_results.db [substring-of-kernel-name ...]
"""
import collections
import sqlite3
import sys


def main(path, filters):
    c = sqlite3.connect(path)
    rows = c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                     "group by kernel_name, counter_name").fetchall()
    agg = collections.defaultdict(dict)
    for k, n, v, cnt in rows:
        if filters and not any(f in k for f in filters):
            continue
        agg[k][n] = (v / cnt, cnt)
    for k, d in sorted(agg.items()):
        name = k if len(k) < 110 else k[:107] + "..."
        print(name)
        for n, (avg, cnt) in sorted(d.items()):
            print(f"    {n:28s} avg/dispatch {avg:16.1f}   ({cnt} dispatches)")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
