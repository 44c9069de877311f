#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd SQLite) kernel trace: per-kernel count / avg / min / max / total.
usage: tools/rocpd_summary.py <results.db> [> profiles/<name>.txt]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name[:110]


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), avg(duration), min(duration), max(duration), sum(duration) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[5] for r in rows) or 1
    print("# rocprofv3 --kernel-trace summary of %s" % path)
    print("# %-110s %8s %10s %10s %10s %12s %6s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "total_us", "%"))
    for n, cnt, avg, mn, mx, sm in rows:
        print("%-112s %8d %10.2f %10.2f %10.2f %12.1f %6.2f" % (short(n), cnt, avg / 1e3, mn / 1e3, mx / 1e3, sm / 1e3, 100.0 * sm / tot))


if __name__ == "__main__":
    main(sys.argv[1])
