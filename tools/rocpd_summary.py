#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.x rocpd sqlite) result DB: per-kernel stats and mean PMC values.
Usage: python tools/rocpd_summary.py <results.db> [...]   (prints a text table per DB)"""
import sqlite3
import sys


def main():
    for path in sys.argv[1:]:
        con = sqlite3.connect(path)
        print("== %s" % path)
        try:
            rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
            print("%-72s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
            for r in rows:
                print("%-72s %8d %14.0f %12.0f %7.2f" % (r[0][:72], r[1], r[2], r[3], r[4]))
        except sqlite3.Error as e:
            print("no top_kernels view:", e)
        try:
            rows = con.execute(
                "select kernel_name, counter_name, avg(value), count(*), max(grid_size), max(workgroup_size), "
                "max(vgpr_count), max(sgpr_count), max(lds_block_size), max(scratch_size) "
                "from counters_collection group by kernel_name, counter_name").fetchall()
            if rows:
                print("%-60s %-22s %18s %6s  grid/wg/vgpr/sgpr/lds/scratch" % ("kernel", "counter", "mean per dispatch", "n"))
                for r in rows:
                    print("%-60s %-22s %18.1f %6d  %s/%s/%s/%s/%s/%s" % (r[0][:60], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9]))
        except sqlite3.Error:
            pass
        con.close()


if __name__ == "__main__":
    main()
