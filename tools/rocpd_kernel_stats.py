#!/usr/bin/env python3
"""Kernel statistics (the table `rocprofv3 --stats` prints as *_kernel_stats.csv) from the rocpd SQLite file that
rocprofv3 writes by default: name, calls, total / average / min / max duration in ns, percentage.
usage: rocpd_kernel_stats.py <results.db> [out.csv]"""
import csv
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in con.execute("pragma table_info(rocpd_kernel_dispatch)")]
    sym_cols = [r[1] for r in con.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "display_name" if "display_name" in sym_cols else "kernel_name"
    start, end = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
    rows = con.execute("select s.%s, count(*), sum(d.%s - d.%s), min(d.%s - d.%s), max(d.%s - d.%s) from rocpd_kernel_dispatch d "
                       "join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.%s order by 3 desc" %
                       (name_col, end, start, end, start, end, start, name_col)).fetchall()
    total = float(sum(r[2] for r in rows)) or 1.0
    out = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
    out.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, calls, tot, mn, mx in rows:
        out.writerow([name, calls, tot, "%.1f" % (tot / calls), "%.2f" % (100.0 * tot / total), mn, mx])


if __name__ == "__main__":
    main()
